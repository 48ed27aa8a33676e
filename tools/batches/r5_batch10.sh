#!/bin/bash
# round 5, call 10: LDS unaligned-read probe; LDS counters of the remap kernels (config2, config5)
set -u
out=gpurun_out/r5b10; mkdir -p $out
tools/probes/bin/lds_unaligned_probe 2>&1 | tee $out/lds_unaligned.txt
python tools/collect_pmc_any.py $out config2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL 2>&1 | tee $out/lds_config2.log
python tools/collect_pmc_any.py $out config5 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL 2>&1 | tee $out/lds_config5.log
