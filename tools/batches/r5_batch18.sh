#!/bin/bash
# round 5, call 18: launch-shape sweep of the remap (after the gather change) and the statistics pre-pass, config2
set -u
out=gpurun_out/r5b18; mkdir -p $out
run() { python bench.py --workload config2 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for pc in 4 5 6 7; do for fr in 2 4 6 8; do for st in 2 3 4; do RIP_REMAP_PER_CU=$pc RIP_REMAP_FRAMES=$fr RIP_REMAP_STAGES=$st run | sed "s/^/per_cu=$pc frames=$fr stages=$st /"; done; done; done 2>&1 | tee $out/remap_sweep.log
for sb in 1024 2048 4096 8192; do RIP_STATS_BLOCKS=$sb run | sed "s/^/stats_blocks=$sb /"; done 2>&1 | tee $out/stats_sweep.log
