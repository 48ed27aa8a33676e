#!/bin/bash
# round 5, call 25: non-temporal stores of the chain when the remap reads the image back (RIP_CHAIN_NT -1 new default / 1 rounds 3-4 / 0 never), alternating
set -u
out=gpurun_out/r5b25; mkdir -p $out
run() { python bench.py --workload $1 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2 3 4; do for nt in -1 1 0; do RIP_CHAIN_NT=$nt run config2 | sed "s/^/chain_nt=$nt /"; done; done 2>&1 | tee $out/nt.log
for nt in -1 1; do RIP_CHAIN_NT=$nt RIP_REMAP_FUSED=0 run config5 | sed "s/^/two kernels chain_nt=$nt /"; done 2>&1 | tee -a $out/nt.log
