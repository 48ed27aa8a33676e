#!/bin/bash
# round 5, call 39: ring remap with two stage sizes in one LDS budget (RIP_REMAP_SMALL_STAGE chunks; 0 = one size): parity + A/B
set -u
out=gpurun_out/r5b39; mkdir -p $out
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py tests/test_determinism_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
run() { python bench.py --workload $1 --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2 3; do
  RIP_REMAP_SMALL_STAGE=0 RIP_REMAP_PER_CU=6 run config2 | sed "s/^/small=0 per_cu=6 /"
  for ss in 384 320 448; do RIP_REMAP_SMALL_STAGE=$ss run config2 | sed "s/^/small=$ss /"; done
done 2>&1 | tee $out/ab.log
