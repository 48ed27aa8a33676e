#!/bin/bash
# round 5, call 34: the ring kernels' counted waits with the stores counted (vmcnt is one in-order queue): parity, determinism, A/B
set -u
out=gpurun_out/r5b34; mkdir -p $out
RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/cnt1.so timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py tests/test_determinism_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload config2,config5 cnt0 cnt1; done 2>&1 | tee $out/ab.log
