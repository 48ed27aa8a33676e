#!/bin/bash
# round 5, call 17: launch-shape sweep of the fused chain after the restructure (frames per item visit x persistent workgroups), default_chain and chain
set -u
out=gpurun_out/r5b17; mkdir -p $out
for wl in default_chain chain; do
  for fr in 1 2 4 8 16; do
    for bl in 1024 2048 4096 8192; do
      RIP_CHAIN_FRAMES=$fr RIP_CHAIN_BLOCKS=$bl python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/frames=$fr blocks=$bl /"
    done
  done
done 2>&1 | tee $out/sweep.log
