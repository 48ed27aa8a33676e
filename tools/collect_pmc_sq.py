#!/usr/bin/env python3
"""Instruction-issue counters of the hot kernels (SQ block, one pass + GRBM_GUI_ACTIVE) for
`bench.py --workload <wl>`; writes <out_dir>/pmc_sq_summary.txt.
usage (on a GPU box): collect_pmc_sq.py <out_dir> [workload ...]

GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: kernel cycles = GRBM_GUI_ACTIVE / 8 (matches the kernel-trace
duration at ~2.1 GHz).  "VALU/SIMD/cyc" = SQ_INSTS_VALU (wave instructions, all waves) / 1024 SIMDs / kernel cycles;
multiplied by the kernel's average issue cost (2 cycles for full-rate ops, 4 for the rest: tools/isa_cycles.py)
it is the fraction of VALU issue slots in use."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
      "SQ_WAVES", "GRBM_GUI_ACTIVE"]


def main():
    out_dir = os.path.abspath(sys.argv[1])
    workloads = sys.argv[2:] or ["config2"]
    os.makedirs(out_dir, exist_ok=True)
    lines = ["# rocprofv3 --pmc " + " ".join(SQ) + " --kernel-trace -- bench.py --steps 4 --warmup 1 --workload <wl>",
             "# per launch (median over launches); quad-cycle counters are summed over all CUs by rocprofv3",
             "%-8s %-22s %-8s %-13s %-10s %-10s %-11s %-12s %-12s %-10s" % ("workload", "kernel", "launches", "VALU/SIMD/cyc", "act/wave%", "wait/wave%",
                                                                             "stall/wave%", "VALU inst/wv", "LDS inst/wv", "kernel cyc")]
    for wl in workloads:
        d = os.path.join(out_dir, "sq_%s" % wl)
        cmd = ["rocprofv3", "--pmc"] + SQ + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc", "--workload", wl]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r"(\w+_kernel)", row["Kernel_Name"])
                if not m:
                    continue
                per.setdefault(m.group(1), {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
        for k, c in sorted(per.items(), key=lambda kv: -med(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
            g = med(c.get("GRBM_GUI_ACTIVE", []))
            wc = med(c.get("SQ_WAVE_CYCLES", [])) or 1.0
            waves = med(c.get("SQ_WAVES", [])) or 1.0
            lines.append("%-8s %-22s %-8d %-13.3f %-10.1f %-10.1f %-11.1f %-12.0f %-12.0f %-10.0f" % (
                wl, k, len(c.get("GRBM_GUI_ACTIVE", [])), med(c.get("SQ_INSTS_VALU", [])) / 1024.0 / ((g or 8.0) / 8.0),
                100.0 * med(c.get("SQ_ACTIVE_INST_ANY", [])) / wc, 100.0 * med(c.get("SQ_WAIT_ANY", [])) / wc,
                100.0 * med(c.get("SQ_WAIT_INST_ANY", [])) / wc, med(c.get("SQ_INSTS_VALU", [])) / waves, med(c.get("SQ_INSTS_LDS", [])) / waves, g / 8.0))
    with open(os.path.join(out_dir, "pmc_sq_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
