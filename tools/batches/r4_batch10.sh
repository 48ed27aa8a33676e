#!/bin/bash
# round 4, call 10: where a single resident frame of the reference's example configuration spends its time (kernel trace:
# run time of every launch of a call)
set -u
out=$PWD/gpurun_out/r4b10; mkdir -p $out
python -m pytest tests -q -x -m gpu -k "ccc or config3 or temporal or fuzz or random" > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -2 $out/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o r -- python $GRAFT_REPO_ROOT/tools/probes/single_frame_timeline.py run > $out/run.log 2>&1
DB=$(find $out/kt -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/probes/single_frame_timeline.py report $DB | tee $out/timeline.txt
rm -rf $out/kt
cd $GRAFT_REPO_ROOT
PATH_SURVEY_ONLY="reference example" python tools/path_survey.py 64 2>&1 | grep frames/s | tee $out/survey.log
PATH_SURVEY_ONLY="ONE resident" python tools/path_survey.py 64 2>&1 | grep frames/s | tee -a $out/survey.log
python tools/latency_probe.py > $out/latency.log 2>&1; tail -8 $out/latency.log
