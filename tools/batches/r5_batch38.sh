#!/bin/bash
# round 5, call 38: does the ring remap want more resident workgroups?  two ring stages (16 KB) with 6 / 8 / 9 workgroups per CU against three stages with 6
set -u
out=gpurun_out/r5b38; mkdir -p $out
run() { python bench.py --workload config2 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2; do
  RIP_REMAP_STAGES=3 RIP_REMAP_PER_CU=6 run | sed "s/^/stages=3 per_cu=6 /"
  for pc in 6 8 9; do RIP_REMAP_STAGES=2 RIP_REMAP_PER_CU=$pc run | sed "s/^/stages=2 per_cu=$pc /"; done
done 2>&1 | tee $out/ab.log
