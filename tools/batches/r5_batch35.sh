#!/bin/bash
# round 5, call 35: soak of the final tree: 1 200 fuzzed configurations, the determinism stress, fuzz under the contracted model
set -u
out=gpurun_out/r5b35; mkdir -p $out
RIP_FUZZ_CASES=1200 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -q -x -m gpu 2>&1 | tail -2 | tee $out/fuzz.log
RIP_FP_CONTRACT=0 timeout 900 python tools/probes/determinism_stress.py 2>&1 | tail -4 | tee $out/determinism.log
