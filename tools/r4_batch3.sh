#!/bin/bash
# round 4, GPU batch 3: remap with frame groups folded into blockIdx.x (plan words shared through one XCD's L2)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b3
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_determinism_gpu.py -m gpu -x -q 2>&1 | tail -5 > "$OUT/pytest.log"
run() { echo "$1 $(env $1 python bench.py --workload $2 --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], json.dumps(j["roofline"]["kernel_ms_per_step"]))')" >> "$OUT/fold.log"; }
for rep in 1 2; do
  for f in 1 2 4 8; do run "RIP_REMAP_FOLD=$f" config2; done
done
for f in 1 2 4; do run "RIP_REMAP_FOLD=$f" config5; done
for fr in 2 3; do for f in 2 4; do run "RIP_REMAP_FOLD=$f RIP_REMAP_FRAMES=$fr" config2; done; done
cd /tmp && export TMPDIR=/tmp
for f in 2 4; do RIP_REMAP_FOLD=$f python "$ROOT/tools/collect_pmc.py" "$OUT/pmc_fold$f" config2 > "$OUT/pmc_fold$f.log" 2>&1; done
rm -rf "$OUT"/pmc_fold*/pmc_*_fetch "$OUT"/pmc_fold*/pmc_*_write "$OUT"/pmc_fold*/pmc_*_rdsplit
cat "$OUT/pytest.log" "$OUT/fold.log"; grep remap_ring "$OUT"/pmc_fold*/pmc_summary.txt
