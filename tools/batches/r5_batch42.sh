#!/bin/bash
# round 5, call 42: what do the two modes of the remap's duration (1.82 / 1.99 ms per 256 frames) follow?
set -u
out=gpurun_out/r5b42; mkdir -p $out
export RIP_TRACE_ALLOC=1
P=tools/probes/remap_modes_probe.py
{
for i in 1 2 3 4 5 6; do echo "## process $i, one trial"; python $P --trials 1 2>&1 | grep -v "^rip alloc [0-9]\{1,7\} bytes"; done
echo "## one process, five trials, fresh handle + fresh output each"; python $P --trials 5 2>&1 | grep -v "^rip alloc [0-9]\{1,7\} bytes"
echo "## one process, six trials, one handle, output shifted by a pad"; python $P --trials 6 --keep-handle --pad-mib 0,1,2.5,7,33,0 2>&1 | grep -v "^rip alloc [0-9]\{1,7\} bytes"
which setarch && for i in 1 2 3 4; do echo "## no ASLR, process $i"; setarch x86_64 -R python $P --trials 1 2>&1 | grep -v "^rip alloc [0-9]\{1,7\} bytes"; done
} > $out/modes.txt 2>&1
grep -c trial $out/modes.txt; grep "trial" $out/modes.txt | cut -c1-200
