#!/bin/bash
# round 4, call 8: the mono8 chain (flip 180 + gamma table) folded into the one-channel ring remap, against the two-kernel path
set -u
out=gpurun_out/r4b8; mkdir -p $out
python -m pytest tests/test_parity_gpu.py -q -x -m gpu -k "mono or gathers" > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log
tail -3 $out/tests.log
PATH_SURVEY_ONLY=mono8 python tools/path_survey.py 64 2>&1 | grep frames/s | sed 's/^/folded:      /' | tee -a $out/survey.log
PATH_SURVEY_ONLY=mono8 RIP_REMAP_FUSED=0 python tools/path_survey.py 64 2>&1 | grep frames/s | sed 's/^/two kernels: /' | tee -a $out/survey.log
