#!/bin/bash
# round 5, call 16: statistics pre-pass with selector merges and no ballot per row pair: parity + A/B
set -u
out=gpurun_out/r5b16; mkdir -p $out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_determinism_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload config2 sold snew; done 2>&1 | tee $out/ab.log
