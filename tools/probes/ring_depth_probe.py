#!/usr/bin/env python3
"""rip_submit / rip_collect throughput against the ring depth (config2 frame, pinned input, pinned result view), each depth
on a FRESH handle and in two orders: is a deeper ring slower, or is it one slow slot?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from raw_image_pipeline_amd import RawImagePipeline, synth  # noqa: E402
from raw_image_pipeline_amd.pipeline import host_alloc  # noqa: E402

W, H = 2448, 2048
frame = host_alloc((H, W))
frame[...] = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")


def run(depth, n=300):
    p = RawImagePipeline(False, "", "", "", device=0)
    synth.configure_full_chain(p, W, H, "grey_world")
    p.set_ring_depth(depth)
    tickets = []
    per_slot = []
    for _ in range(2 * depth):
        tickets.append(p.submit(frame, "bayer_rggb8"))
        if len(tickets) == depth:
            p.collect(tickets.pop(0), copy=False)
    t0 = time.perf_counter()
    last = t0
    for i in range(n):
        if len(tickets) == depth:
            p.collect(tickets.pop(0), copy=False)
            now = time.perf_counter()
            per_slot.append(now - last)
            last = now
        tickets.append(p.submit(frame, "bayer_rggb8"))
    while tickets:
        p.collect(tickets.pop(0), copy=False)
    dt = (time.perf_counter() - t0) / n * 1e3
    ps = np.array(per_slot[depth:]) * 1e3
    by_slot = [float(np.median(ps[k::depth])) for k in range(depth)] if depth > 1 else [float(np.median(ps))]
    print("depth %d: %.3f ms/frame (%.0f frames/s); median collect-to-collect by position in the ring: %s" % (depth, dt, 1e3 / dt, " ".join("%.3f" % v for v in by_slot)), flush=True)


for order in ((1, 2, 3, 4, 5, 6, 8), (8, 6, 5, 4, 3, 2, 1)):
    for d in order:
        run(d)
