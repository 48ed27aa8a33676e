python -m pytest tests/test_parity_gpu.py tests/test_frontend.py tests/test_determinism_gpu.py -q -x -m gpu -k "ring or submit or rig or frontend or determinism or deeper" 2>&1 | tail -4
python tools/latency_probe.py 2>&1 | grep -v Loading | tail -9
