#!/bin/bash
# round 4, GPU batch 2: the whole -m gpu suite on the micro-changed build; the fused chain on flat / scene / iid frames (what the
# LDS bank conflicts cost); one PC-sampling attempt on the chain (bounded by timeout)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > "$OUT/pytest.log"
for k in flat scene uniform; do
  for wl in chain config2; do
    echo "$k $wl $(RIP_BENCH_FRAMES=$k python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], json.dumps(j["roofline"]["kernel_ms_per_step"]))')" >> "$OUT/frames_kind.log"
  done
done
cd /tmp && export TMPDIR=/tmp
for k in flat uniform; do RIP_BENCH_FRAMES=$k python "$ROOT/tools/collect_pmc_sq.py" "$OUT/sq_$k" chain > "$OUT/sq_$k.log" 2>&1; done
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --kernel-trace --output-format csv -d "$OUT/pcs" -o p -- python "$ROOT/bench.py" --workload chain --steps 3 --warmup 1 --batch 64 --no-cpu-baseline --no-hbm-probe --no-pmc > "$OUT/pcs.log" 2>&1
echo "pc sampling rc=$?" >> "$OUT/pcs.log"
ls -la "$OUT/pcs" >> "$OUT/pcs.log" 2>&1
find "$OUT/pcs" -name "*.csv" -size +20M -delete
rm -rf "$OUT"/sq_*/sq_*
cat "$OUT/pytest.log" "$OUT/frames_kind.log"; tail -5 "$OUT/pcs.log"
