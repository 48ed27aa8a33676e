#!/bin/bash
# round 4, GPU batch 4: wide-expand probe; which PC-sampling configurations this box offers, and one bounded attempt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r4b4
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
python -m pytest tests/test_probe_gpu.py -m gpu -x -q -s 2>&1 | tail -6 > "$OUT/pytest.log"
python - > "$OUT/probe_1g.log" 2>&1 <<'PY'
from raw_image_pipeline_amd import RawImagePipeline
p = RawImagePipeline(False, "", "", "", device=0)
for rep in range(2):
    print(", ".join("%s %.0f" % (k, p.hbm_probe(k, 1 << 30, 10)) for k in p.PROBE_KINDS))
PY
cd /tmp && export TMPDIR=/tmp
(timeout 120 rocprofv3 -L 2>&1 | grep -i -B2 -A12 "pc.sampl\|PC_SAMPL\|stochastic\|host_trap" | head -80) > "$OUT/pcs_list.log" 2>&1
CMD="python $ROOT/bench.py --workload chain --steps 3 --warmup 1 --batch 64 --no-cpu-baseline --no-hbm-probe --no-pmc"
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --kernel-trace --output-format csv -d "$OUT/pcs" -o p -- $CMD > "$OUT/pcs_stoch.log" 2>&1
echo "stochastic rc=$?" >> "$OUT/pcs_stoch.log"
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 100 --kernel-trace --output-format csv -d "$OUT/pcs2" -o p -- $CMD > "$OUT/pcs_host.log" 2>&1
echo "host_trap rc=$?" >> "$OUT/pcs_host.log"
ls -laR "$OUT/pcs" "$OUT/pcs2" > "$OUT/pcs_ls.log" 2>&1
find "$OUT" -name "*.csv" -size +30M -delete
cat "$OUT/pytest.log" "$OUT/probe_1g.log"; tail -3 "$OUT/pcs_stoch.log" "$OUT/pcs_host.log"; head -30 "$OUT/pcs_list.log"
