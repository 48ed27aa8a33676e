#!/usr/bin/env python3
"""A/B harness for kernel experiments.
  build side (no GPU):  tools/ab_chain.py build name1=-DFLAG1,-DFLAG2 name2=...   -> raw_image_pipeline_amd/variants/<name>.so
  GPU side:             tools/ab_chain.py run [--workload chain] [names...]         -> one line per variant
Only rip_chain.hip (and whatever else reads the flags) differs between variants; every variant is a complete library,
selected at run time through RIP_LIBRARY (raw_image_pipeline_amd/pipeline.py)."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "raw_image_pipeline_amd", "variants")


def main():
    sys.path.insert(0, ROOT)
    if sys.argv[1] == "build":
        from raw_image_pipeline_amd import build as b
        os.makedirs(VDIR, exist_ok=True)
        for spec in sys.argv[2:]:
            name, _, flags = spec.partition("=")
            out = b.build(force=True, out=os.path.join(VDIR, name + ".so"), extra_flags=[f for f in flags.split(",") if f], tag="_" + name)
            shutil.rmtree(os.path.join(ROOT, "raw_image_pipeline_amd", "build_" + name), ignore_errors=True)
            print("built", out)
        return
    args = sys.argv[2:]
    workloads = ["chain"]
    if args and args[0] == "--workload":
        workloads = args[1].split(",")
        args = args[2:]
    names = args or sorted(f[:-3] for f in os.listdir(VDIR) if f.endswith(".so"))
    for wl in workloads:
        for n in names:
            env = dict(os.environ, RIP_LIBRARY=os.path.join(VDIR, n + ".so"))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-hbm-probe", "--no-pmc", "--workload", wl, "--steps", "10"],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(wl, n, "FAILED", r.stdout[-400:])
                continue
            j = json.loads(line[-1])
            print("%-8s %-24s %9.1f frames/s  %s" % (wl, n, j["value"], json.dumps(j["roofline"]["kernel_ms_per_step"])), flush=True)


if __name__ == "__main__":
    main()
