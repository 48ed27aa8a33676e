#!/bin/bash
# per-kernel average durations of one bench workload (rocprofv3 --kernel-trace --stats); usage: kernel_times.sh <workload> [env...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
WL=${1:-config3}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$WL
rocprofv3 --kernel-trace --stats -d /tmp/kt_$WL -o p --output-format csv -- python $ROOT/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-pmc > /dev/null 2>&1
f=$(find /tmp/kt_$WL -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Name"])
    print("%-34s calls=%-4s avg_us=%9.1f total_us=%10.1f %5s%%" % (m.group(1) if m else r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
PY
