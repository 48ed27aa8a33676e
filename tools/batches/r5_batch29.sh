#!/bin/bash
# round 5, call 29: timing-only: the window's left / right dwords from the neighbour lanes (DPP wave_shr / wave_shl) instead of 8 of the 12 loads
set -u
out=gpurun_out/r5b29; mkdir -p $out
for i in 1 2 3; do python tools/ab_chain.py run --workload chain,default_chain,config5 w_base w_dpp; done 2>&1 | tee $out/ab.log
