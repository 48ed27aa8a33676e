#!/usr/bin/env python3
"""A/B of one environment tunable on bench.py workloads.
usage: ab_env.py NAME v1,v2,... workload1,workload2,... [repeats]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, values, workloads = sys.argv[1], sys.argv[2].split(","), sys.argv[3].split(",")
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 2
for wl in workloads:
    for rep in range(repeats):
        for v in values:
            env = dict(os.environ)
            env[name] = v
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-hbm-probe",
                                "--no-pmc", "--workload", wl], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300)
            for l in r.stdout.splitlines():
                if l.startswith("{"):
                    j = json.loads(l)
                    print("%-8s %s=%-6s %9.0f frames/s  %.3f ms/step  %s" % (wl, name, v, j["value"], j["ms_per_step"], j["roofline"].get("kernel_ms_per_step")), flush=True)
