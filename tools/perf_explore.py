#!/usr/bin/env python3
"""Development probe: per-kernel-class times (HIP events) for arbitrary geometry / batch / stage sets.
usage: perf_explore.py WxH[:batch[:stages]] ...   stages = letters from f(lip) w(b grey) c g v h u"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth


def run(spec):
    parts = spec.split(":")
    w, h = [int(v) for v in parts[0].split("x")]
    batch = int(parts[1]) if len(parts) > 1 else 16
    st = parts[2] if len(parts) > 2 else "fwcgvu"
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_stream(torch.cuda.current_stream())
    synth.configure_full_chain(pipe, w, h)
    pipe.set_flip("f" in st)
    pipe.set_white_balance("w" in st or "p" in st)
    if "p" in st:
        pipe.set_white_balance_method("pca")
    pipe.set_color_calibration("c" in st)
    pipe.set_gamma_correction("g" in st)
    pipe.set_vignetting_correction("v" in st)
    pipe.set_color_enhancer("h" in st)
    pipe.set_color_enhancer_saturation_gain(1.2)
    pipe.set_undistortion("u" in st)
    base = synth.gen_frame(w, h, "bayer_rggb8", seed=1, kind="uniform")
    frames = torch.from_numpy(np.stack([np.roll(base, 2 * i, axis=0) for i in range(batch)])).cuda()
    out = pipe.apply_device(frames, "bayer_rggb8")
    for _ in range(2):
        pipe.apply_device(frames, "bayer_rggb8", out=out)
    torch.cuda.synchronize()
    steps = 8
    pipe.profile_begin(4 * steps + 4)
    for _ in range(steps):
        pipe.apply_device(frames, "bayer_rggb8", out=out)
    prof = pipe.profile_end()
    px = w * h * batch
    msg = []
    for k, (ms, n) in prof.items():
        if n:
            t = ms / n
            msg.append("%s %.3f ms (%.2f us/frame, %.1f Gpx/s)" % (k, t, t * 1e3 / batch, px / t / 1e6))
    print("%-28s %s" % (spec, " | ".join(msg)), flush=True)


for s in sys.argv[1:]:
    run(s)
