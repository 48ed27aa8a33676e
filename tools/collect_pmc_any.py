#!/usr/bin/env python3
"""Median per-launch value of arbitrary rocprofv3 PMC counters for the kernels of one bench.py workload.
usage (GPU box): collect_pmc_any.py <out_dir> <workload> COUNTER [COUNTER ...]     (<= 8 SQ counters per call)"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out_dir, wl, counters = os.path.abspath(sys.argv[1]), sys.argv[2], sys.argv[3:]
    os.makedirs(out_dir, exist_ok=True)
    d = os.path.join(out_dir, "pmc_%s_%s" % (wl, counters[0]))
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc", "--workload", wl]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            if m:
                per.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
    # the same run's kernel trace: median duration per kernel (counter collection serialises the launches but does not change
    # where the buffers landed -- tools/probes/remap_modes_probe.py)
    dur = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
            if m:
                dur.setdefault(m.group(1), []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for k, c in sorted(per.items()):
        print("%-8s %-24s %8.4f ms  " % (wl, k, med(dur.get(k, []))) + "  ".join("%s=%.4g" % (n, med(c.get(n, []))) for n in counters), flush=True)


if __name__ == "__main__":
    main()
