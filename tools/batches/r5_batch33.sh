#!/bin/bash
# round 5, call 33: where the 2.6 ms of the chain-inside-remap kernel are (timing only): bit 0 no gather reads, 1 no demosaic, 2 no stores, 3 no Bayer loads
set -u
out=gpurun_out/r5b33; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload config5 g0 g1 g2 g4 g8 g12 g3; done 2>&1 | tee $out/ab.log
