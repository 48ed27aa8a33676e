#!/bin/bash
# round 5, call 20: whole GPU suite + smoke + profiles of the final tree (tools/refresh_profiles.sh) + the default bench line, timed
set -u
out=gpurun_out/r5b20; mkdir -p $out
timeout 2400 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
export RIP_ROUND=5
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
for f in 16 8; do RIP_CHAIN_FRAMES=$f python tools/collect_pmc_sq.py $out/sq$f config2 > /dev/null 2>&1; echo "# RIP_CHAIN_FRAMES=$f"; cat $out/sq$f/pmc_sq_summary.txt; done > $out/two_point.txt 2>&1
python tools/collect_pmc_any.py $out config2 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL > $out/lds_config2.log 2>&1
s=$(date +%s); python bench.py 2>/dev/null | tail -1 > $out/bench_default.json; e=$(date +%s); echo "default bench.py wall time $((e-s)) s" | tee $out/bench_time.txt; python tools/bench_summary.py < $out/bench_default.json
