#!/bin/bash
# round 5, call 40: what about the chain's stores costs 13 %?  s1 no stores, s4 every store into one 64 KB window (same instructions, nothing reaches HBM); nt and ordinary stores
set -u
out=gpurun_out/r5b40; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload chain s0 s1 s4; for v in s0 s1 s4; do RIP_CHAIN_NT=0 RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so python bench.py --workload chain --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/ordinary stores $v /"; done; done 2>&1 | tee $out/ab.log
