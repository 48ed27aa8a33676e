set -x
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests1.log 2>&1
for g in 1 2 4 8 16; do
  RIP_OVERLAP_GROUPS=$g python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovl_$g.log 2>&1
done
for g in 1 4; do
  RIP_OVERLAP_GROUPS=$g python bench.py --workload config5 --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovl5_$g.log 2>&1
done
