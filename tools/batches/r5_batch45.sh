#!/bin/bash
# round 5, call 45: do the remap's duration levels follow address-translation misses?  TLB counters + durations of the same process, six processes
set -u
out=gpurun_out/r5b45; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "\b[A-Z0-9_]*\(UTCL\|TLB\|XNACK\)[A-Za-z0-9_]*" | sort -u > /root/repo/$out/tlb_counters.txt
cd /root/repo
wc -l $out/tlb_counters.txt; head -60 $out/tlb_counters.txt | tr '\n' ' '
for i in 1 2 3 4 5 6; do python tools/collect_pmc_any.py $out/p$i config2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum 2>&1 | grep -i "remap_ring\|chain_fast"; done | tee $out/utcl1.txt
