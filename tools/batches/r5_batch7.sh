#!/bin/bash
# round 5, call 7: whole suite on the tree (contracted-model kernels, gamma look-up folded into the conversion), default_chain A/B, default bench line
set -u
out=gpurun_out/r5b7; mkdir -p $out
timeout 1800 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
for i in 1 2; do python tools/ab_chain.py run --workload default_chain r5a r5c; done 2>&1 | tee $out/ab.log
for m in 0 1; do RIP_FP_CONTRACT=$m python bench.py --workload default_chain --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/fc=$m /"; done | tee -a $out/ab.log
timeout 900 python bench.py 2>/dev/null | tail -1 > $out/bench_default.json; python tools/bench_summary.py < $out/bench_default.json
python -c "
import json; j=json.load(open('$out/bench_default.json')); print(json.dumps(j['roofline'].get('memory_rate_variant'), indent=1))"
