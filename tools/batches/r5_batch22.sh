#!/bin/bash
# round 5, call 22: the fused chain with its stores and / or its window loads dropped (timing only): how much of its time is memory?
set -u
out=gpurun_out/r5b22; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload chain,default_chain c0 c1 c2 c3; done 2>&1 | tee $out/ab.log
