mkdir -p gpurun_out/r3
for f in 1 4 8 16 32 64; do
  echo "RIP_CHAIN_FRAMES=$f"
  RIP_CHAIN_FRAMES=$f python tools/perf_explore.py 2448x2048:256:d 3840x2160:256:d 2448x2048:256:wcg 2>/dev/null | grep -v Loading
done > gpurun_out/r3/frame_major.log 2>&1
for b in 1024 4096 8192; do
  echo "RIP_CHAIN_BLOCKS=$b frames 16"
  RIP_CHAIN_BLOCKS=$b RIP_CHAIN_FRAMES=16 python tools/perf_explore.py 2448x2048:256:d 3840x2160:256:d 2>/dev/null | grep -v Loading
done >> gpurun_out/r3/frame_major.log 2>&1
python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -m gpu -x -q >> gpurun_out/r3/frame_major.log 2>&1
