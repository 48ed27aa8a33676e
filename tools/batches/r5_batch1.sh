#!/bin/bash
# round 5, call 1: parity of the restructured fused chain (selectors for pattern parity + flip, PLAIN body) and A/B against round 4's library
set -u
out=gpurun_out/r5b1; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log
tail -3 $out/tests.log
for i in 1 2; do python tools/ab_chain.py run --workload chain,config2 r4base r5a; done 2>&1 | tee $out/ab.log
for wl in config3 config5; do for n in r4base r5a; do RIP_LIBRARY=raw_image_pipeline_amd/variants/$n.so python bench.py --workload $wl --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py; done; done 2>&1 | tee $out/others.log
