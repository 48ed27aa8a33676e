#!/bin/bash
# round 5, call 23: the fused chain with a frame's stores issued after the next frame's window loads (vmcnt is one in-order queue): parity + A/B
set -u
out=gpurun_out/r5b23; mkdir -p $out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_fp_contraction_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload chain,config2,default_chain,config3 dold dnew; done 2>&1 | tee $out/ab.log
