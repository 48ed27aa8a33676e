set -u
out=$PWD/gpurun_out/r4b11; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for d in 3 4; do
rocprofv3 --kernel-trace --memory-copy-trace -d $out/kt$d -o r -- python $GRAFT_REPO_ROOT/tools/probes/ring_trace.py run $d > $out/run$d.log 2>&1
DB=$(find $out/kt$d -name '*.db' | head -1)
echo "== depth $d"; python $GRAFT_REPO_ROOT/tools/probes/ring_trace.py report $DB 2>&1 | tail -45
rm -rf $out/kt$d
done
