#!/bin/bash
# round 5, call 8: new tests (config 3 sequence, N > 1 end-to-end leg, rig error path) + whole suite
set -u
out=gpurun_out/r5b8; mkdir -p $out
timeout 2400 python -m pytest tests -q -x -m gpu > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -15 $out/tests.log
