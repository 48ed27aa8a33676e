cd $GRAFT_REPO_ROOT
for fr in 8 16 32 64; do for bl in 1536 2048 4096; do
echo -n "frames=$fr blocks=$bl: "; RIP_CHAIN_FRAMES=$fr RIP_CHAIN_BLOCKS=$bl python bench.py --no-cpu-baseline --no-hbm-probe --no-pmc --workload chain --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['roofline']['kernel_ms_per_step']['chain'])"
done; done
