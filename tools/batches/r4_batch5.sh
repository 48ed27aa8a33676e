#!/bin/bash
# round 4, GPU batch 5: whole -m gpu suite, a fuzz soak and the long determinism stress on the final kernels
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b5
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > "$OUT/pytest.log"
RIP_FUZZ_CASES=1200 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -5 > "$OUT/fuzz.log"
timeout 600 python tools/probes/determinism_stress.py 150 128 > "$OUT/stress.log" 2>&1
cat "$OUT/pytest.log" "$OUT/fuzz.log" "$OUT/stress.log"
