#!/bin/bash
# round 5, call 11: profiles of the tree (tools/refresh_profiles.sh: kernel trace of the bench command without warm-up launches, PMC traffic, SQ counters,
# bench lines of every workload incl. default_chain and the default line) + the executed-instruction count of config2's chain at 16 and 8 frames per visit
set -u
export RIP_ROUND=5
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
out=gpurun_out/r5b11; mkdir -p $out
for f in 16 8; do RIP_CHAIN_FRAMES=$f python tools/collect_pmc_sq.py $out/sq$f config2 default_chain > /dev/null 2>&1; echo "# RIP_CHAIN_FRAMES=$f"; cat $out/sq$f/pmc_sq_summary.txt; done 2>&1 | tee $out/two_point.txt
for m in 0 1; do RIP_FP_CONTRACT=$m python bench.py --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/fc=$m /"; done | tee $out/fc.txt
ls gpurun_out/refresh
