#!/bin/bash
# round 5, call 46: the remap's duration levels against the memory-side counters, launch by launch in one process
set -u
out=gpurun_out/r5b46; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\bTCC_[A-Za-z0-9_]*\|\bMALL[A-Za-z0-9_]*\|\b[A-Z_]*HBM[A-Za-z0-9_]*\|\bTCP_T[A-Za-z0-9_]*" | sort -u > /root/repo/$out/mem_counters.txt
cd /root/repo
wc -l $out/mem_counters.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  echo "#### $c"; python tools/probes/remap_modes_pmc.py $out 8 $c 2>&1 | grep -A 9 "## remap_ring"
done | tee $out/levels.txt
