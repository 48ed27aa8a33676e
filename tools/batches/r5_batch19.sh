#!/bin/bash
# round 5, call 19: RIP_REMAP_PER_CU 6 / 7 / 8 / 10, alternating (the single sweep of call 18 is inside the box's noise)
set -u
out=gpurun_out/r5b19; mkdir -p $out
run() { python bench.py --workload config2 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2 3 4; do for pc in 6 7 8 10; do RIP_REMAP_PER_CU=$pc run | sed "s/^/per_cu=$pc /"; done; done 2>&1 | tee $out/per_cu.log
