set -x
mkdir -p gpurun_out/r3
for g in 2 4; do
  RIP_OVERLAP_MODE=2 RIP_OVERLAP_GROUPS=$g python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovlm2_$g.log 2>&1
done
RIP_OVERLAP_GROUPS=1 python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovlm2_1.log 2>&1
python tools/ab_chain.py run --workload chain,config5 base invg_l1 nt1 nt2 nt3 base > gpurun_out/r3/ab_a.log 2>&1
