#!/bin/bash
# round 5, call 28: frames per item visit of the Lab chain between 8 and 16 (whole rounds of resident workgroups?), alternating
set -u
out=gpurun_out/r5b28; mkdir -p $out
for i in 1 2 3; do for fr in 16 14 13 12 11 10; do RIP_CHAIN_FRAMES=$fr python bench.py --workload chain --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/frames=$fr /"; done; done 2>&1 | tee $out/frames.log
