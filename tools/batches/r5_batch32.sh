#!/bin/bash
# round 5, call 32: the four-byte LDS colour image: row padding 0 / 8 / 24 / 40 bytes, 6 waves per SIMD, ring stages 2 / 3, workgroups per CU
set -u
out=gpurun_out/r5b32; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload config5 p8 p0 p24 p40; done 2>&1 | tee $out/pad.log
run() { RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$1.so python bench.py --workload config5 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; }
for i in 1 2; do for v in p8 w6; do for st in 2 3; do for pc in 5 6 7; do RIP_REMAP_STAGES=$st RIP_REMAP_PER_CU=$pc run $v | sed "s/^/$v stages=$st per_cu=$pc /"; done; done; done; done 2>&1 | tee $out/occ.log
