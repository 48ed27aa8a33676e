#!/bin/bash
# round 5, call 44: the remap's three duration levels -- do they follow the offset of the output inside one fixed allocation
# (address interleave against the intermediate image) or the allocation itself (physical placement)?
set -u
out=gpurun_out/r5b44; mkdir -p $out
P=tools/probes/remap_modes_probe.py
{
for i in 1 2; do
echo "## process $i, one handle, one arena: offsets 0 0 1 2 3 4 8 16 32 64 0.0625 0.25 0.5 0 MiB"; python $P --keep-handle --trials 14 --modes o0,o0,o1,o2,o3,o4,o8,o16,o32,o64,o0.0625,o0.25,o0.5,o0 2>&1 | grep "^trial"
done
echo "## one handle, fresh output each trial (n x 10)"; python $P --keep-handle --trials 10 --modes n 2>&1 | grep "^trial"
} > $out/modes.txt 2>&1
cut -c1-18,66-220 $out/modes.txt
