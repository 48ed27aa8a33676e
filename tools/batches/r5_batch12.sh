#!/bin/bash
# round 5, call 12: remap: a wave's four tile rows four apart instead of consecutive (LDS bank spread) -- parity, timing, LDS counters
set -u
out=gpurun_out/r5b12; mkdir -p $out
RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/rstride4.so timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "undistortion or full_chain or config5 or config2 or remap" 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload config2 rbase rstride4; done 2>&1 | tee $out/ab.log
for v in rbase rstride4; do RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so python tools/collect_pmc_any.py $out/$v config2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_LDS_CMD_FIFO_FULL 2>&1 | grep remap_ring | sed "s/^/$v /"; done | tee $out/lds.log
