#!/bin/bash
# round 4, call 12: whole GPU suite + fuzz soak + determinism + smoke on the tree with the latency / host-path changes
set -u
out=gpurun_out/r4b12; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log
tail -3 $out/tests.log
RIP_FUZZ_CASES=1200 timeout 900 python -m pytest tests/test_fuzz_gpu.py -q -x -m gpu > $out/fuzz.log 2>&1; echo "fuzz exit $?" >> $out/fuzz.log
tail -2 $out/fuzz.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 600 python bench.py 2>/dev/null | tail -1 > $out/bench_default.json; python tools/bench_summary.py < $out/bench_default.json
for wl in config3 config5; do python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; done
