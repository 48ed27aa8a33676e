set -x
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/gpu_tests4.log 2>&1
for w in config2 chain config3 config5; do python bench.py --workload $w --steps 20 --no-cpu-baseline --no-pmc --no-hbm-probe 2>/dev/null | python tools/bench_summary.py; done > gpurun_out/r3/bench2.log 2>&1
python tools/latency_probe.py > gpurun_out/r3/latency3.log 2>&1
