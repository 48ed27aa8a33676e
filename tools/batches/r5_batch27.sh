#!/bin/bash
# round 5, call 27: cache-policy bits of the fused chain's window loads (aux 1 sc0, 2 nt, 16 sc1)
set -u
out=gpurun_out/r5b27; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload config2,default_chain l_base l_nt l_sc1 l_sc0; done 2>&1 | tee $out/ab.log
