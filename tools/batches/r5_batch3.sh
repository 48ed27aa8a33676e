#!/bin/bash
# round 5, call 3: per-table cost of LDS bank conflicts in the fused chain: flat (one mid-table entry) against control (same extra instruction, real addresses)
set -u
out=gpurun_out/r5b3; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload chain r5a c_lin f_lin c_cbx f_cbx c_cby f_cby c_yf f_yf c_invg f_invg c_all f_all; done 2>&1 | tee $out/ab.log
