#!/bin/bash
# round 4, call 11: the host path after the in-flight limit and the split copies: ring / rig tests, latency probe and the
# four-camera rig with deep copies for 0 / 3 / 7 copy threads
python -m pytest tests/test_parity_gpu.py tests/test_frontend.py tests/test_determinism_gpu.py tests/test_abi.py -q -x -m gpu -k "ring or submit or rig or frontend or determinism or deeper" 2>&1 | tail -4
for t in 0 3 7; do
echo "== RIP_COPY_THREADS=$t"
RIP_COPY_THREADS=$t python tools/latency_probe.py 2>&1 | grep "copied out"
RIP_COPY_THREADS=$t python -m pytest tests/test_frontend.py -q -x -m gpu -k rig -s 2>&1 | grep "camera rig"
done
