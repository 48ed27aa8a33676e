#!/bin/bash
# round 5, call 2: which table's LDS bank conflicts cost the fused chain how much (timing-only builds, RIP_EXP_FLAT) + LDS counters of the tree
set -u
out=gpurun_out/r5b2; mkdir -p $out
for i in 1 2; do python tools/ab_chain.py run --workload chain flat_none flat_lin flat_cbx flat_cby flat_yf flat_invg flat_all; done 2>&1 | tee $out/ab.log
python tools/collect_pmc_any.py $out chain SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL 2>&1 | tee $out/lds.log
python tools/collect_pmc_any.py $out chain SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES 2>&1 | tee -a $out/lds.log
