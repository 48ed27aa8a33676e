#!/bin/bash
# round 5, call 5: a wave of the fused chain as a compact 2-D tile (64 px x 8 rows) instead of 256 px x 2 rows: fewer distinct table entries per LDS instruction?
set -u
out=gpurun_out/r5b5; mkdir -p $out
RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/tile16.so timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2; do python tools/ab_chain.py run --workload chain,config2 base6 tile8 tile16 tile32; done 2>&1 | tee $out/ab.log
for k in flat uniform; do RIP_BENCH_FRAMES=$k python tools/ab_chain.py run --workload chain base6 tile16; done 2>&1 | tee $out/ab_kinds.log
