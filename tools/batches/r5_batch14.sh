#!/bin/bash
# round 5, call 14: where the remap's 2.0 ms are -- timing-only builds (wrong pixels) that drop one part each:
# RIP_EXP_REMAP bit 0 no gather arithmetic / LDS reads, bit 1 no store, bit 2 no source loads (the LDS-DMA reads an out-of-range offset)
set -u
out=gpurun_out/r5b14; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload config2 x0 x1 x2 x4 x3 x5 x6; done 2>&1 | tee $out/ab.log
