#!/bin/bash
# round 5, call 41: last check of the committed tree on a fresh box -- whole GPU suite, smoke, the default bench line
set -u
out=gpurun_out/r5b41; mkdir -p $out
timeout 2400 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
s=$(date +%s); python bench.py 2>/dev/null | tail -1 > $out/bench_default.json; e=$(date +%s); echo "default bench.py wall time $((e-s)) s" | tee $out/bench_time.txt; python tools/bench_summary.py < $out/bench_default.json
