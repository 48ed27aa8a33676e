set -x
mkdir -p gpurun_out/r3
python tools/perf_explore.py 2304x2048:256:d 2432x2048:256:d 2448x2048:256:d 2560x2048:256:d 3840x2048:128:d 3840x2160:128:d 2448x2048:256:wcg 2560x2048:256:wcg 2448x2048:256:d > gpurun_out/r3/widths.log 2>&1
for c in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES"; do
  python tools/collect_pmc_any.py gpurun_out/r3/pmc chain $c >> gpurun_out/r3/pmc_chain.log 2>&1
done
