#!/bin/bash
# round 5, call 24: cache-policy bits of the fused chain's stores (aux: 1 sc0, 2 nt, 16 sc1): streaming (chain, default_chain) and re-read (config2) destinations
set -u
out=gpurun_out/r5b24; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload chain,default_chain,config2 a_base a_nt3 a_nt18 a_nt19 a_nt0; done 2>&1 | tee $out/ab.log
