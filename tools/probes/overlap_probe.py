#!/usr/bin/env python3
"""Probe: does the issue-bound fused chain (stream A) overlap with HBM-bound work (stream B: colour-input copy + remap) when
both are enqueued on separate HIP streams?  Prints the time of A alone, B alone and both together for a few grid caps."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

W, H, N = 2448, 2048, int(os.environ.get("OVL_N", "128"))
b0 = torch.from_numpy(synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")).cuda()
c0 = torch.from_numpy(synth.gen_scene_bgr(W, H, seed=2)).cuda()
bayer = torch.stack([torch.roll(b0, 2 * i, 0) for i in range(N)])
bgr = torch.stack([torch.roll(c0, 3 * i, 0) for i in range(N)])


def make(kind):
    p = RawImagePipeline(False, "", "", "", device=0)
    s = torch.cuda.Stream()
    p.set_stream(s)
    synth.configure_full_chain(p, W, H)
    if kind == "chain":
        p.set_white_balance(False)
        p.set_undistortion(False)
    else:
        for f in (p.set_flip, p.set_white_balance, p.set_color_calibration, p.set_gamma_correction, p.set_vignetting_correction):
            f(False)
    return p, s


def run(jobs, steps=10):
    for _ in range(2):
        for p, x, enc, o in jobs:
            p.apply_device(x, enc, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for p, x, enc, o in jobs:
            p.apply_device(x, enc, out=o)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print("inputs ready", flush=True)
pa, sa = make("chain")
pb, sb = make("remap")
oa = torch.empty((N, H, W, 3), dtype=torch.uint8, device="cuda")
ob = torch.empty((N, H, W, 3), dtype=torch.uint8, device="cuda")
A = (pa, bayer, "bayer_rggb8", oa)
B = (pb, bgr, "bgr8", ob)
print("pipes ready", flush=True)
for blocks in (os.environ.get("RIP_CHAIN_BLOCKS", "default"),):
    ta, tb, tab = run([A]), run([B]), run([A, B])
    print("chain alone %.3f ms, copy+remap alone %.3f ms, both streams %.3f ms (sum %.3f): overlap saves %.1f %%"
          % (ta, tb, tab, ta + tb, 100 * (1 - tab / (ta + tb))), flush=True)
