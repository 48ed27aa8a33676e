#!/usr/bin/env python3
"""Host-path throughput of a camera rig on one GPU (full-size frames, taps as the ROS node publishes them):
sequential vs one thread + HIP stream per camera.  usage: rig_probe.py [n_cameras]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from raw_image_pipeline_amd import synth
from raw_image_pipeline_amd.frontend import CameraRig

ncam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 2448, 2048
params = [{"flip/enabled": True, "flip/angle": 180, "white_balance/enabled": True, "white_balance/method": "gray_world",
           "gamma_correction/enabled": True, "vignetting_correction/enabled": True} for _ in range(ncam)]
rig = CameraRig(params, n_devices=1)
frames = [synth.gen_frame(W, H, "bayer_rggb8", seed=c, kind="scene") for c in range(ncam)]
enc = ["bayer_rggb8"] * ncam
for mode in (False, True, False, True):
    rig.on_images(frames, enc, parallel=mode)
    t0 = time.perf_counter()
    for _ in range(5):
        rig.on_images(frames, enc, parallel=mode)
    dt = time.perf_counter() - t0
    print("%d cameras %dx%d, %s: %.0f frames/s (%.2f ms per frame)" % (ncam, W, H, "threaded" if mode else "sequential", 5 * ncam / dt, dt / (5 * ncam) * 1e3))
