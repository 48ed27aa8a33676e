#!/bin/bash
# round 5, call 9: can the VALU-bound chain share CUs with the memory-bound remap?  Chain with 33.7 KB of tables (plain 4 KB inverse-gamma table + clamps)
# and 72 VGPRs: three chain workgroups (101 KB, 432 registers per SIMD lane) leave room for two remap workgroups per CU.  run_batch's overlap mode
# (remap(g) beside stats(g+1) + chain(g+1)) with 2 / 4 / 8 frame groups.
set -u
out=gpurun_out/r5b9; mkdir -p $out
for i in 1 2; do
for v in base7 small small72; do
  for g in 0 2 4 8; do
    RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so RIP_OVERLAP_GROUPS=$g python bench.py --workload config2 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/$v groups=$g /"
  done
done
done 2>&1 | tee $out/overlap.log
