#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace summary of the bench command, PMC traffic passes, the
# bench lines of every workload.  Results land in gpurun_out/refresh/ and are copied to profiles/ by hand.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-hbm-probe --no-pmc"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o r -- $CMD > "$OUT/kt_bench.log" 2>&1
DB=$(find "$OUT/kt" -name '*.db' | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" "round ${RIP_ROUND:-5}: $CMD (config2, default 256 frames/step, 2448x2048 bayer_rggb8 full chain)" --skip-first 2 > "$OUT/kernel_stats.txt" 2>&1
python "$ROOT/tools/collect_pmc.py" "$OUT/pmc" config2 chain default_chain config5 config3 > "$OUT/pmc.log" 2>&1
for wl in config2 chain default_chain config3 config5; do
  python "$ROOT/bench.py" --workload $wl --steps 20 --no-cpu-baseline --no-hbm-probe --no-pmc 2>/dev/null | tail -1 > "$OUT/bench_$wl.json"
done
python "$ROOT/bench.py" 2>/dev/null | tail -1 > "$OUT/bench_default.json"
python "$ROOT/tools/collect_pmc_sq.py" "$OUT/sq" config2 chain default_chain config3 config5 > "$OUT/sq.log" 2>&1
"$ROOT/tools/probes/bin/valu_probe2" > "$OUT/valu_probe2.txt" 2>&1
"$ROOT/tools/probes/bin/valu_probe3" > "$OUT/valu_probe3.txt" 2>&1
cp "$OUT/sq/pmc_sq_summary.txt" "$OUT/pmc_sq_summary.txt"
rm -rf "$OUT/kt" "$OUT"/pmc/pmc_*_fetch "$OUT"/pmc/pmc_*_write "$OUT"/sq/sq_*
ls -la "$OUT" "$OUT/pmc"
