#!/usr/bin/env python3
"""Four-camera rig, pipelined host path, three published images per frame, with RIP_DEBUG_RING: throughput and the
distribution of the per-frame download copy times (45 MB per frame; the link moves 15 MB in 0.28-0.30 ms)."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from raw_image_pipeline_amd import synth
    from raw_image_pipeline_amd.frontend import CameraRig
    ncam = int(sys.argv[2])
    W, H = 2448, 2048
    params = [{"flip/enabled": True, "flip/angle": 180, "white_balance/enabled": True, "white_balance/method": "gray_world",
               "gamma_correction/enabled": True, "vignetting_correction/enabled": True} for _ in range(ncam)]
    rig = CameraRig(params, n_devices=1)
    frames = [synth.gen_frame(W, H, "bayer_rggb8", seed=c, kind="scene") for c in range(ncam)]
    enc = ["bayer_rggb8"] * ncam
    for _ in range(3):
        rig.on_images(frames, enc, mode="pipelined", copy=False)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        rig.on_images(frames, enc, mode="pipelined", copy=False)
    dt = time.perf_counter() - t0
    print("RIG %d cameras pipelined, views: %.0f frames/s (%.3f ms per frame)" % (ncam, n * ncam / dt, dt / (n * ncam) * 1e3))
    sys.exit(0)
for ncam in (1, 2, 4):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(ncam)], env=dict(os.environ, RIP_DEBUG_RING="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    copies = [float(m) for m in re.findall(r"of which the copies ([0-9.]+)", r.stdout)]
    copies = sorted(copies[len(copies) // 4:])
    print([l for l in r.stdout.splitlines() if l.startswith("RIG")])
    if copies:
        q = lambda f: copies[min(len(copies) - 1, int(f * len(copies)))]
        print("   download copies per frame, ms: min %.3f median %.3f p90 %.3f max %.3f (n = %d)" % (copies[0], q(0.5), q(0.9), copies[-1], len(copies)))
