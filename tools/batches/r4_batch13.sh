#!/bin/bash
# round 4, call 13: chain(g + 1) beside remap(g) on two streams when the chain leaves LDS for the remap's workgroups
# (RIP_CHAIN_LDS_PAD: 2 instead of 3 chain workgroups per CU)
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
run X=1
run RIP_CHAIN_LDS_PAD=2048
for g in 2 4 8; do
  run RIP_OVERLAP_GROUPS=$g
  run RIP_OVERLAP_GROUPS=$g RIP_CHAIN_LDS_PAD=2048
  run RIP_OVERLAP_GROUPS=$g RIP_CHAIN_LDS_PAD=2048 RIP_REMAP_PER_CU=2
done
run RIP_OVERLAP_GROUPS=4 RIP_CHAIN_LDS_PAD=30000 RIP_REMAP_PER_CU=4
run X=1
