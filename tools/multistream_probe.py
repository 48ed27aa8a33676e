#!/usr/bin/env python3
"""Probe: does running K camera streams (one handle + one HIP stream each) concurrently on one GPU beat
one stream with the same total number of frames?  usage: multistream_probe.py total_frames K [K...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

W, H = 2448, 2048
total = int(sys.argv[1])
base = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
frames_all = torch.from_numpy(np.stack([np.roll(base, 2 * i, axis=0) for i in range(total)])).cuda()
for K in [int(v) for v in sys.argv[2:]]:
    per = total // K
    pipes, streams, ins, outs = [], [], [], []
    for k in range(K):
        p = RawImagePipeline(False, "", "", "", device=0)
        s = torch.cuda.Stream()
        p.set_stream(s)
        synth.configure_full_chain(p, W, H)
        pipes.append(p)
        streams.append(s)
        ins.append(frames_all[k * per:(k + 1) * per])
        outs.append(torch.empty((per, H, W, 3), dtype=torch.uint8, device="cuda"))

    def step():
        for k in range(K):
            pipes[k].apply_device(ins[k], "bayer_rggb8", out=outs[k])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 10
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("streams=%d frames/stream=%d  %.1f frames/s  (%.2f us/frame)" % (K, per, steps * per * K / el, el / (steps * per * K) * 1e6), flush=True)
    del pipes, outs
