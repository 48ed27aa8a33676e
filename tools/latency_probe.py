#!/usr/bin/env python3
"""Single-frame latency of the drop-in call (host numpy in/out, as the reference's apply()/process()) and of the
device-resident call, config2 full chain at 2448x2048.  usage: latency_probe.py [reps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H = 2448, 2048
frame = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
pipe = RawImagePipeline(False, "", "", "", device=0)
synth.configure_full_chain(pipe, W, H)
for _ in range(3):
    out = pipe.process(frame, "bayer_rggb8")
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    out = pipe.process(frame, "bayer_rggb8")
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("process() host->host 1 frame: median %.3f ms  min %.3f  p90 %.3f   (%.0f frames/s)" % (np.median(ts), ts.min(), np.percentile(ts, 90), 1e3 / np.median(ts)))
# pipelined host -> host: rip_submit / rip_collect with `depth` frames in flight (upload f+1 | kernels f | download f-1)
def pipelined(p, src, depth, n=200, copy=False):
    p.set_ring_depth(depth)
    tickets = []
    for _ in range(depth):  # warm the slots
        tickets.append(p.submit(src, "bayer_rggb8"))
    while tickets:
        p.collect(tickets.pop(0), copy=copy)
    t0 = time.perf_counter()
    for i in range(n):
        if len(tickets) == depth:
            p.collect(tickets.pop(0), copy=copy)
        tickets.append(p.submit(src, "bayer_rggb8"))
    while tickets:
        p.collect(tickets.pop(0), copy=copy)
    return (time.perf_counter() - t0) / n * 1e3

import ctypes
lib = pipe._lib
lib.rip_host_alloc.restype = ctypes.c_void_p
lib.rip_host_alloc.argtypes = [ctypes.c_size_t]
ptr = lib.rip_host_alloc(frame.nbytes)
pinned = np.frombuffer((ctypes.c_uint8 * frame.nbytes).from_address(ptr), np.uint8).reshape(frame.shape)
pinned[...] = frame
for depth in (1, 2, 3, 4):
    a = pipelined(pipe, frame, depth)
    b = pipelined(pipe, pinned, depth)
    print("submit/collect depth %d, pinned result view: pageable input %.3f ms/frame (%.0f frames/s) | pinned input %.3f ms/frame (%.0f frames/s)"
          % (depth, a, 1e3 / a, b, 1e3 / b))
c = pipelined(pipe, pinned, 3, copy=True)
print("submit/collect depth 3, pinned input, result copied out of the pinned buffer: %.3f ms/frame (%.0f frames/s)" % (c, 1e3 / c))
pipe.set_ring_depth(3)
dev = torch.from_numpy(frame[None]).cuda()
o = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
pipe.set_stream(torch.cuda.current_stream())
for _ in range(3):
    pipe.apply_device(dev, "bayer_rggb8", out=o)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    pipe.apply_device(dev, "bayer_rggb8", out=o)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("apply_device() resident 1 frame: median %.3f ms  min %.3f  p90 %.3f" % (np.median(ts), ts.min(), np.percentile(ts, 90)))
pipe.profile_begin(64)
for _ in range(10):
    pipe.apply_device(dev, "bayer_rggb8", out=o)
torch.cuda.synchronize()
print("kernel classes over 10 single-frame calls (ms total, launches):", pipe.profile_end())
