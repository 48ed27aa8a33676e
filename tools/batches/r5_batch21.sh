#!/bin/bash
# round 5, call 21: frames per item visit of the Lab chain beyond 16 (RIP_CHAIN_FRAMES 16 / 24 / 32 / 64), alternating; default bench wall time
set -u
out=gpurun_out/r5b21; mkdir -p $out
for i in 1 2 3; do for fr in 16 24 32 64; do RIP_CHAIN_FRAMES=$fr python bench.py --workload config2 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/frames=$fr /"; done; done 2>&1 | tee $out/frames.log
s=$(date +%s); python bench.py 2>/dev/null | tail -1 > $out/bench_default.json; e=$(date +%s); echo "default bench.py wall time $((e-s)) s" | tee $out/bench_time.txt; python tools/bench_summary.py < $out/bench_default.json
