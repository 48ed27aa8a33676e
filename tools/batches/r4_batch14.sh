#!/bin/bash
# round 4, call 14: the memory-rate chain variants with a wave's contiguous run written as 16-byte stores through LDS
# (RIP_CHAIN_WIDE_STORES) against the 12-byte lane stores; parity tests with the switch on
for v in 0 1 0 1; do
  echo "== RIP_CHAIN_WIDE_STORES=$v"
  RIP_CHAIN_WIDE_STORES=$v python tools/perf_explore.py 2448x2048:256: 2448x2048:256:f 2448x2048:256:fwcg 3840x2160:256: 1920x1200:256:wcg 2>&1 | grep -v Loading | grep -i "chain\|ms"
done
RIP_CHAIN_WIDE_STORES=1 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_golden.py -q -x -m gpu 2>&1 | tail -3
