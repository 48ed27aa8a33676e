set -x
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests2.log 2>&1
python tools/latency_probe.py > gpurun_out/r3/latency1.log 2>&1
for w in config2 config3 config5 chain; do python bench.py --workload $w --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py; done > gpurun_out/r3/bench1.log 2>&1
