#!/bin/bash
# round 5, call 36: ring remap run-ahead: stages 2 / 3 x frames per visit 2 / 3 / 4 / 5, alternating on one box
set -u
out=gpurun_out/r5b36; mkdir -p $out
run() { python bench.py --workload config2 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2 3; do for st in 3 2; do for fr in 4 3 2 5; do RIP_REMAP_STAGES=$st RIP_REMAP_FRAMES=$fr run | sed "s/^/stages=$st frames=$fr /"; done; done; done 2>&1 | tee $out/sweep.log
