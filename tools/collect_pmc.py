#!/usr/bin/env python3
"""Runs on the GPU box: HBM traffic of every kernel class of `bench.py` from rocprofv3 PMC counters, in
separate passes (FETCH_SIZE; WRITE_SIZE), as MI355X_MICROARCH.md prescribes.  Writes
<out>/pmc_traffic.json ({workload: {class: bytes per launch}}) and <out>/pmc_summary.txt.

Corrections (MI355X_MICROARCH.md, HBM section, calibrated on our own access patterns -- see the summary):
  * FETCH_SIZE / WRITE_SIZE are reported in KiB.
  * FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) streaming loads: x2 for kernels whose
    reads are dwordx4 (remap_tiled staging); x1 for the 4 B/lane window loads of stats/chain (calibrated:
    the cold stats pass reads 1.12x its 1 B/px).
usage: collect_pmc.py <out_dir> [workload ...]"""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS = {"stats": "stats_", "chain": "chain_", "remap": "remap_", "ccc": "ccc_"}


def run_pass(out_dir, name, counters, workload):
    d = os.path.join(out_dir, "pmc_%s_%s" % (workload, name))
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc", "--workload", workload]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    out_dir = os.path.abspath(sys.argv[1])
    workloads = sys.argv[2:] or ["config2"]
    os.makedirs(out_dir, exist_ok=True)
    result, lines = {}, []
    for wl in workloads:
        per = {}
        for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                               ("rdsplit", ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"])):
            for row in run_pass(out_dir, name, counters, wl):
                kname = row["Kernel_Name"]
                cls = next((c for c, pat in CLASS.items() if pat in kname), None)
                if cls is None:
                    continue
                m = re.search(r"(\w+_kernel)", kname)
                short = m.group(1) if m else kname[:40]  # e.g. remap_tiled_kernel
                per.setdefault(cls, {}).setdefault(short, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        result[wl] = {}
        med = lambda v: sorted(v)[len(v) // 2] if v else 0.0  # the first launches are warm-up: use the median
        for cls, kernels in sorted(per.items()):
            total = 0.0
            for short, d in sorted(kernels.items()):
                fetch, write = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
                # FETCH_SIZE = TCC_EA0_RDREQ x 64 B, but every read request of these kernels is 128 B wide (the size-split
                # counters below; tools/probes/fetch_calib_probe.hip reads 2147 MB and reports 1074 MB for 4 B/lane and
                # 16 B/lane loads alike): x2, the gfx950 correction of MI355X_MICROARCH.md's HBM section
                factor = 2.0
                fb, wb = med(fetch) * 1024 * factor, med(write) * 1024
                exact = 32 * med(d.get("TCC_EA0_RDREQ_32B_sum", [])) + 64 * med(d.get("TCC_EA0_RDREQ_64B_sum", [])) + \
                    128 * med(d.get("TCC_EA0_RDREQ_128B_sum", []))
                total += fb + wb
                lines.append("%-8s %-6s %-22s launches=%-3d FETCH_SIZE(med)=%.0f KiB x%.0f -> %.1f MB (32/64/128-B request counters: %.1f MB)   "
                             "WRITE_SIZE(med)=%.0f KiB -> %.1f MB"
                             % (wl, cls, short, len(fetch), med(fetch), factor, fb / 1e6, exact / 1e6, med(write), wb / 1e6))
            result[wl][cls] = int(total)
            lines.append("%-8s %-6s total %.1f MB per step (one launch of each kernel of the class)" % (wl, cls, total / 1e6))
    sys.path.insert(0, ROOT)
    import bench  # the default --batch of the command that was profiled
    sys.argv = sys.argv[:1]
    result["frames_per_launch"] = bench.parse_args().batch
    with open(os.path.join(out_dir, "pmc_traffic.json"), "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
    with open(os.path.join(out_dir, "pmc_summary.txt"), "w") as f:
        f.write("# HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes) of\n"
                "# `bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-hbm-probe --workload <wl>` (%d frames per launch)\n" % result["frames_per_launch"])
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
