#!/usr/bin/env python3
"""Per-launch duration and PMC counters of the remap in ONE process that re-allocates the output tensor before every trial
(tools/probes/remap_modes_probe.py --keep-handle --modes n): do the duration levels (1.82 / 1.90 / 2.00 ms) show in a counter?
usage (GPU box): remap_modes_pmc.py <out_dir> <trials> COUNTER [COUNTER ...]"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out_dir, trials, counters = os.path.abspath(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
    d = os.path.join(out_dir, "pmc_" + counters[0])
    os.makedirs(d, exist_ok=True)
    steps = 3
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
           os.path.join(ROOT, "tools", "probes", "remap_modes_probe.py"), "--keep-handle", "--trials", str(trials), "--modes", "n", "--steps", str(steps)]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    dur, name = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            name[row["Dispatch_Id"]] = row["Kernel_Name"]
    val = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            val.setdefault(row["Dispatch_Id"], {})[row["Counter_Name"]] = float(row["Counter_Value"])
    for kernel in ("remap_ring_kernel", "chain_fast_kernel"):
        ids = sorted((i for i in dur if re.search(kernel, name[i])), key=int)
        per = steps + 2  # launches per trial: two warm-up steps + the timed ones
        print("## %s: %d launches, %d per trial" % (kernel, len(ids), per))
        for t in range(0, len(ids), per):
            grp = ids[t:t + per][2:]
            if not grp:
                continue
            ms = sorted(dur[i] for i in grp)[len(grp) // 2]
            cs = {c: sorted(val.get(i, {}).get(c, 0.0) for i in grp)[len(grp) // 2] for c in counters}
            print("trial %2d  %.4f ms  " % (t // per, ms) + "  ".join("%s=%.5g" % (c, cs[c]) for c in counters), flush=True)


if __name__ == "__main__":
    main()
