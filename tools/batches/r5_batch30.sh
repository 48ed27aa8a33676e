#!/bin/bash
# round 5, call 30: timing-only: the chain-inside-remap kernel of config 5 without its gather's LDS reads (upper bound of any gather / LDS layout work)
set -u
out=gpurun_out/r5b30; mkdir -p $out
for i in 1 2 3; do python tools/ab_chain.py run --workload config5 f_base f_nogather; done 2>&1 | tee $out/ab.log
