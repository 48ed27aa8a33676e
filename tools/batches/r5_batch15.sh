#!/bin/bash
# round 5, call 15: remap gather with one v_perm_b32 + three v_dot4 per tap row and perm-merged output bytes (114 -> 91 VALU per lane and frame): parity + A/B
set -u
out=gpurun_out/r5b15; mkdir -p $out
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload config2,config5 rold rnew; done 2>&1 | tee $out/ab.log
