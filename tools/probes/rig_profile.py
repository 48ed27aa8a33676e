#!/usr/bin/env python3
"""cProfile of the four-camera rig's pipelined trigger with deep copies (three published images per camera)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from raw_image_pipeline_amd import synth  # noqa: E402
from raw_image_pipeline_amd.frontend import CameraRig  # noqa: E402

ncam, W, H = 4, 2448, 2048
params = [{"flip/enabled": True, "flip/angle": 180, "white_balance/enabled": True, "white_balance/method": "gray_world",
           "gamma_correction/enabled": True, "vignetting_correction/enabled": True} for _ in range(ncam)]
rig = CameraRig(params, n_devices=1)
frames = [synth.gen_frame(W, H, "bayer_rggb8", seed=c, kind="scene") for c in range(ncam)]
enc = ["bayer_rggb8"] * ncam
for copy in (True, False):
    for _ in range(3):
        rig.on_images(frames, enc, mode="pipelined", copy=copy)
    n = 20
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        rig.on_images(frames, enc, mode="pipelined", copy=copy)
    pr.disable()
    dt = time.perf_counter() - t0
    print("copy=%s: %.0f frames/s, %.3f ms per trigger of %d cameras" % (copy, n * ncam / dt, dt / n * 1e3, ncam))
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
