#!/bin/bash
# round 5, call 37: ring stages sized to the plan's largest rectangle (whole waves of LDS-DMA) instead of PRE * 4 KiB: more workgroups per CU. parity + A/B
set -u
out=gpurun_out/r5b37; mkdir -p $out
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py tests/test_determinism_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
run() { python bench.py --workload $1 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2 3; do
  RIP_REMAP_EXACT_STAGE=0 run config2 | sed "s/^/exact=0 per_cu=6 /"
  for pc in 6 7 8 10; do RIP_REMAP_EXACT_STAGE=1 RIP_REMAP_PER_CU=$pc run config2 | sed "s/^/exact=1 per_cu=$pc /"; done
done 2>&1 | tee $out/ab.log
python -c "
from raw_image_pipeline_amd import RawImagePipeline, synth
p = RawImagePipeline(False, '', '', '', device=0); synth.configure_full_chain(p, 2448, 2048, 'grey_world'); print(p.debug_plan_info(2048, 2448))" 2>/dev/null | tail -1 | tee $out/plan.log
