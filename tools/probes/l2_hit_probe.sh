#!/bin/bash
# L2 hit / miss counters of the hot kernels (config2 bench).  usage (GPU box): l2_hit_probe.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum --kernel-trace --output-format csv -d /tmp/l2 -o p -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-probe --no-pmc > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections, re
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if m: d[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = lambda v: sorted(v)[len(v)//2] if v else 0.0
for k, c in d.items():
    h, m_, rd, wr = (med(c[x]) for x in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_READ_sum", "TCC_WRITE_sum"))
    print("%-22s hit %.3e miss %.3e (hit rate %.1f%%)  read req %.3e write req %.3e" % (k, h, m_, 100*h/max(h+m_,1), rd, wr))
PY
