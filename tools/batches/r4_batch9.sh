#!/bin/bash
# round 4, call 9: the final tree -- whole GPU suite, fuzz soak, determinism stress, smoke, then tools/refresh_profiles.sh
set -u
out=gpurun_out/r4b9; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log
tail -3 $out/tests.log
RIP_FUZZ_CASES=1200 timeout 900 python -m pytest tests/test_fuzz_gpu.py -q -x -m gpu > $out/fuzz.log 2>&1; echo "fuzz exit $?" >> $out/fuzz.log
tail -2 $out/fuzz.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
RIP_ROUND=4 timeout 1500 bash tools/refresh_profiles.sh > $out/refresh.log 2>&1; tail -3 $out/refresh.log
cat gpurun_out/refresh/bench_default.json | python tools/bench_summary.py
