#!/usr/bin/env python3
"""Stress for rare races (LDS-DMA ring, frame groups, atomics): the same resident batch is processed many times and
every result must equal the first one bit for bit; then a few frames are checked against the single-frame call.
usage: determinism_stress.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W, H = 2448, 2048
pipe = RawImagePipeline(False, "", "", "", device=0)
pipe.set_stream(torch.cuda.current_stream())
synth.configure_full_chain(pipe, W, H)
base = [synth.gen_frame(W, H, "bayer_rggb8", seed=i, kind="scene", tint=(0.6 + 0.05 * i, 1.0, 0.55)) for i in range(5)]
frames = torch.from_numpy(np.stack([base[i % 5] for i in range(batch)])).cuda()
first = pipe.apply_device(frames, "bayer_rggb8").clone()
out = torch.empty_like(first)
bad = 0
for s in range(steps):
    pipe.apply_device(frames, "bayer_rggb8", out=out)
    torch.cuda.synchronize()
    if not torch.equal(out, first):
        bad += 1
        d = (out != first).flatten(1).any(dim=1).nonzero().flatten().tolist()
        print("step %d: frames %s differ" % (s, d[:8]))
singles = [pipe.process(b, "bayer_rggb8") for b in base]
ok = all(np.array_equal(first[i].cpu().numpy(), singles[i % 5]) for i in (0, 1, 2, 3, 4, batch // 2, batch - 1))
print("steps %d, batch %d: %d steps differed from the first; batch == single-frame results: %s" % (steps, batch, bad, ok))
sys.exit(1 if bad or not ok else 0)
