#!/bin/bash
# round 5, call 6: the contracted-model kernels (rip_set_fp_contraction(1)) against oracle model 1; whole suite; timing of both models
set -u
out=gpurun_out/r5b6; mkdir -p $out
timeout 900 python -m pytest tests/test_fp_contraction_gpu.py -q -x -m gpu 2>&1 | tail -15 | tee $out/fc_tests.log
timeout 1500 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -3 $out/tests.log
for m in 0 1 0 1; do for wl in chain config2; do RIP_FP_CONTRACT=$m python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | sed "s/^/fc=$m /"; done; done | tee $out/fc_bench.log
