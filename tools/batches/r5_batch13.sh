#!/bin/bash
# round 5, call 13: upper bound of what a 1 B/px remap plan could buy (timing-only builds: RIP_EXP_PLAN_BYTES = 16 real / 4 / 0) + fetch counters
set -u
out=gpurun_out/r5b13; mkdir -p $out
for i in 1 2 3; do python tools/ab_chain.py run --workload config2 plan16 plan4 plan0; done 2>&1 | tee $out/ab.log
for f in 8; do for v in plan16 plan4; do RIP_REMAP_FRAMES=$f RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so python bench.py --workload config2 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/$v frames_per_visit=$f /"; done; done | tee -a $out/ab.log
for v in plan16 plan4; do RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so python tools/collect_pmc_any.py $out/$v config2 FETCH_SIZE 2>&1 | grep remap_ring | sed "s/^/$v /"; done | tee $out/fetch.log
