#!/bin/bash
# round 4, GPU batch 7: launch-shape sweep of the chain-inside-remap kernel (config5; config-2-like stage sets via path_survey)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b7
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "inside_the_remap" 2>&1 | tail -3 > "$OUT/pytest.log"
run() { echo "$1 $(env $1 python bench.py --workload config5 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], json.dumps(j["roofline"]["kernel_ms_per_step"]))')" >> "$OUT/sweep.log"; }
run "RIP_REMAP_FUSED=1"
for fr in 2 3 6 8 12 16; do run "RIP_REMAP_FRAMES=$fr"; done
for pc in 3 4 5 7; do run "RIP_REMAP_PER_CU=$pc"; done
for st in 2 4; do run "RIP_REMAP_STAGES=$st"; done
run "RIP_REMAP_FUSED=0"
cat "$OUT/pytest.log" "$OUT/sweep.log"
