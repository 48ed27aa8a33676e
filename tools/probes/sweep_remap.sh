#!/bin/bash
# Sweep of the remap launch tunables on config2 (remap ms per 256-frame step); run on the GPU box.
cd $GRAFT_REPO_ROOT
for pc in 4 5 6 7; do for fr in 2 4 8; do for st in 2 3 4; do
echo -n "per_cu=$pc frames=$fr stages=$st: "; RIP_REMAP_PER_CU=$pc RIP_REMAP_FRAMES=$fr RIP_REMAP_STAGES=$st python bench.py --no-cpu-baseline --no-hbm-probe --no-pmc --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms_per_step']['remap'])"
done; done; done
