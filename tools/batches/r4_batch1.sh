#!/bin/bash
# round 4, GPU batch 1: parity of the chain micro-changes, A/B against the previous build, default bench line with the
# hand-written HBM probes, two-point VALU count of the chain, plan share of the remap's fetch
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b1
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_probe_gpu.py tests/test_determinism_gpu.py tests/test_parity_gpu.py tests/test_golden.py -m gpu -x -q -s 2>&1 | tail -15 > "$OUT/pytest.log"
python tools/ab_chain.py run --workload chain,config2 r4base r4micro r4base r4micro > "$OUT/ab.log" 2>&1
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
for f in 16 8; do RIP_CHAIN_FRAMES=$f python "$ROOT/tools/collect_pmc_sq.py" "$OUT/sq_f$f" chain > "$OUT/sq_f$f.log" 2>&1; done
for f in 4 8 16; do RIP_REMAP_FRAMES=$f python "$ROOT/tools/collect_pmc.py" "$OUT/pmc_rf$f" config2 > "$OUT/pmc_rf$f.log" 2>&1; done
rm -rf "$OUT"/sq_f*/sq_* "$OUT"/pmc_rf*/pmc_*_fetch "$OUT"/pmc_rf*/pmc_*_write
cat "$OUT/pytest.log" "$OUT/ab.log"; tail -c 1500 "$OUT/bench_default.json"
