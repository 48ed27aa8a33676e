run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python tools/bench_summary.py | cut -c1-200; }
run X=1
for b in 3072 6144 8192; do run RIP_CHAIN_BLOCKS=$b; done
for b in 1024 4096 8192; do run RIP_STATS_BLOCKS=$b; done
for f in 8 32; do run RIP_CHAIN_FRAMES=$f; done
for s in 2 4; do run RIP_REMAP_STAGES=$s; done
for c in 4 5 8; do run RIP_REMAP_PER_CU=$c; done
for f in 3 6; do run RIP_REMAP_FRAMES=$f; done
run X=1
