#!/usr/bin/env python3
"""Cost of a calibration change (fisheye maps + remap plan) with the plan compiled on the device (default) and on the host
(RIP_PLAN_ON_HOST=1): wall time of set_undistortion_balance() + the first frame after it, 2448x2048 and 3840x2160."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from raw_image_pipeline_amd import RawImagePipeline, synth
    for w, h in ((2448, 2048), (3840, 2160)):
        pipe = RawImagePipeline(False, "", "", "", device=0)
        synth.configure_full_chain(pipe, w, h)
        frame = synth.gen_frame(w, h, "bayer_rggb8", seed=1)
        pipe.process(frame, "bayer_rggb8")
        base = []
        for _ in range(5):
            t0 = time.perf_counter(); pipe.process(frame, "bayer_rggb8"); base.append(time.perf_counter() - t0)
        ts = []
        for i in range(5):
            t0 = time.perf_counter()
            pipe.set_undistortion_balance(0.1 * (i + 1))
            pipe.process(frame, "bayer_rggb8")
            ts.append(time.perf_counter() - t0)
        print("%s %dx%d: frame %.2f ms; balance change + frame %.2f ms (min %.2f)" % (sys.argv[1], w, h, 1e3 * sorted(base)[2], 1e3 * sorted(ts)[2], 1e3 * min(ts)))
else:
    for mode, env in (("device plan", {}), ("host plan  ", {"RIP_PLAN_ON_HOST": "1"})):
        subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=dict(os.environ, **env), stderr=subprocess.DEVNULL)
