#!/bin/bash
# round 5, call 4: the per-L' records of the Lab round trip through the texture path instead of LDS (RIP_EXP_YF_TA)
set -u
out=gpurun_out/r5b4; mkdir -p $out
RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/yfta.so timeout 600 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "vignetting or config2 or full_chain" 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload chain,config2 base5 yfta; done 2>&1 | tee $out/ab.log
