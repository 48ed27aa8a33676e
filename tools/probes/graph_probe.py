#!/usr/bin/env python3
"""Would a HIP graph shorten the device-resident single-frame call?  The launches of one apply_device() (memset, statistics,
finalisation, fused chain, remap, border patch) are captured through torch's graph capture (the handle's stream is torch's
capturing stream) and replayed; wall time per call incl. synchronisation, config2 at 2448x2048, and a sweep of the chain /
statistics grid caps for the one-frame case."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

W, H = 2448, 2048
frame = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
pipe = RawImagePipeline(False, "", "", "", device=0)
synth.configure_full_chain(pipe, W, H)
dev = torch.from_numpy(frame[None]).cuda()
o = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")

def timed(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return np.median(ts), ts.min()

s = torch.cuda.Stream()
with torch.cuda.stream(s):
    pipe.set_stream(s)
    eager = lambda: pipe.apply_device(dev, "bayer_rggb8", out=o)
    print("eager on a side stream: median %.1f us, min %.1f" % timed(eager))
    ref = o.clone()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        pipe.apply_device(dev, "bayer_rggb8", out=o)
    o.zero_()
    g.replay()
    torch.cuda.synchronize()
    print("graph replay equals eager:", bool(torch.equal(o, ref)))
    print("graph replay: median %.1f us, min %.1f" % timed(g.replay))
    for name, vals in (("chain_blocks", (768, 1024, 1536, 2048, 4096)), ("stats_blocks", (512, 1024, 2048, 4096))):
        for v in vals:
            pipe.set_tunable(name, v)
            print("eager %s=%d: median %.1f us, min %.1f" % ((name, v) + timed(eager)))
        pipe.set_tunable(name, 0)
