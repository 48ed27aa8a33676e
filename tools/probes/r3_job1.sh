set -x
mkdir -p gpurun_out/r3
python tools/perf_explore.py 2304x2048:256:d 2432x2048:256:d 2448x2048:256:d 2560x2048:256:d 3840x2048:128:d 2448x2048:256:wcg 2560x2048:256:wcg 2448x2048:256:fwcgv > gpurun_out/r3/widths.log 2>&1
python bench.py --steps 10 --no-cpu-baseline > gpurun_out/r3/bench0.json 2> gpurun_out/r3/bench0.err
python tools/latency_probe.py > gpurun_out/r3/latency0.log 2>&1
