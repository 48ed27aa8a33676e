#!/bin/bash
# round 5, call 26: cache-policy bits of the remap's output stores (aux 2 nt, 16 sc1, 18 both), config2 and config5, alternating
set -u
out=gpurun_out/r5b26; mkdir -p $out
for i in 1 2 3; do python tools/ab_chain.py run --workload config2,config5 r_base r_nt r_sc1 r_ntsc1; done 2>&1 | tee $out/ab.log
