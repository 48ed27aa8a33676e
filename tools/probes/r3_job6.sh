set -x
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests3.log 2>&1
python bench.py --steps 10 > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err
python tools/latency_probe.py > gpurun_out/r3/latency2.log 2>&1
for g in 1 2 4 8; do
  RIP_OVERLAP_GROUPS=$g python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovl_$g.log 2>&1
done
for g in 2 4; do
  RIP_OVERLAP_MODE=2 RIP_OVERLAP_GROUPS=$g python bench.py --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py > gpurun_out/r3/ovlm2_$g.log 2>&1
done
