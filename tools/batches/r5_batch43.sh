#!/bin/bash
# round 5, call 43: a live pad allocation made statistics + chain 3-6 % faster in call 42 -- which part of it matters?
set -u
out=gpurun_out/r5b43; mkdir -p $out
P=tools/probes/remap_modes_probe.py
{
echo "## one handle: n,b1,n,b1,f1,a1,g1,g32,n,b33,n"; python $P --keep-handle --trials 11 --modes n,b1,n,b1,f1,a1,g1,g32,n,b33,n 2>&1 | grep "^trial"
echo "## one handle: n,n,n,b1,b1,b1,n,n"; python $P --keep-handle --trials 8 --modes n,n,n,b1,b1,b1,n,n 2>&1 | grep "^trial"
echo "## fresh handle each: n,b1,n,b1,n,b1"; python $P --trials 6 --modes n,b1,n,b1,n,b1 2>&1 | grep "^trial"
} > $out/modes.txt 2>&1
cut -c1-40,95-220 $out/modes.txt
