#!/bin/bash
# round 5, call 31: chain inside the remap's tiles with a four-byte-per-pixel LDS colour image (tap row = one aligned 8-byte read, 5 instead of 6 VALU per tap row): parity + A/B
set -u
out=gpurun_out/r5b31; mkdir -p $out
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py tests/test_fp_contraction_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $out/tests.log
for i in 1 2 3; do python tools/ab_chain.py run --workload config5 fx_old fx_new; done 2>&1 | tee $out/ab.log
for v in fx_old fx_new; do RIP_LIBRARY=$PWD/raw_image_pipeline_amd/variants/$v.so python tools/collect_pmc_any.py $out/$v config5 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE 2>&1 | grep remap_bayer | sed "s/^/$v /"; done | tee $out/lds.log
