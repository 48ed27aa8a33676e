#!/usr/bin/env python3
"""Probe: the VALU-bound fused chain and the HBM-bound remap on disjoint sets of CUs (hipExtStreamCreateWithCUMask), run
concurrently.  Job A = fused chain (Bayer in), job B = colour-input copy + remap (the only way to run the remap alone through
the API: B carries 6 B/px more traffic than the remap of the real pipeline).  Prints A and B alone on all CUs, alone on their
share, and together."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth

hip = C.CDLL("libamdhip64.so")
W, H, N = 2448, 2048, int(os.environ.get("OVL_N", "128"))
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (C.c_uint32 * ((NCU + 31) // 32))()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), words)
    assert rc == 0, rc
    return s.value  # raw handle: the Python mirror then skips its ordering against torch's current (legacy null) stream, which
                    # would serialise the two blocking streams


b0 = torch.from_numpy(synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")).cuda()
c0 = torch.from_numpy(synth.gen_scene_bgr(W, H, seed=2)).cuda()
bayer = torch.stack([torch.roll(b0, 2 * i, 0) for i in range(N)])
bgr = torch.stack([torch.roll(c0, 3 * i, 0) for i in range(N)])
oa = torch.empty((N, H, W, 3), dtype=torch.uint8, device="cuda")
ob = torch.empty((N, H, W, 3), dtype=torch.uint8, device="cuda")


def make(kind, stream):
    p = RawImagePipeline(False, "", "", "", device=0)
    p.set_stream(stream)
    synth.configure_full_chain(p, W, H)
    if kind == "chain":
        p.set_white_balance(False)
        p.set_undistortion(False)
    else:
        for f in (p.set_flip, p.set_white_balance, p.set_color_calibration, p.set_gamma_correction, p.set_vignetting_correction):
            f(False)
    return p


def run(jobs, steps=8):
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


ta = run([(make("chain", masked_stream(range(NCU))), bayer, "bayer_rggb8", oa)])
tb = run([(make("remap", masked_stream(range(NCU))), bgr, "bgr8", ob)])
both = run([(make("chain", masked_stream(range(NCU))), bayer, "bayer_rggb8", oa), (make("remap", masked_stream(range(NCU))), bgr, "bgr8", ob)])
print("both on all CUs, two streams: %.3f ms" % both, flush=True)
print("%d CUs, %d frames: chain alone %.3f ms, copy+remap alone %.3f ms (sum %.3f)" % (NCU, N, ta, tb, ta + tb), flush=True)
for layout in ("contiguous", "interleaved"):
    for frac in (0.5, 0.625, 0.75):
        na = int(NCU * frac) // 8 * 8
        if layout == "contiguous":
            bits_a, bits_b = list(range(na)), list(range(na, NCU))
        else:  # every 8 consecutive bits: the first 8 * frac to A
            k = int(8 * frac)
            bits_a = [i for i in range(NCU) if i % 8 < k]
            bits_b = [i for i in range(NCU) if i % 8 >= k]
        A = (make("chain", masked_stream(bits_a)), bayer, "bayer_rggb8", oa)
        B = (make("remap", masked_stream(bits_b)), bgr, "bgr8", ob)
        a1, b1, ab = run([A]), run([B]), run([A, B])
        print("%-11s chain on %3d CUs %.3f ms, copy+remap on %3d CUs %.3f ms, together %.3f ms (%.1f %% below the all-CU sum)"
              % (layout, len(bits_a), a1, len(bits_b), b1, ab, 100 * (1 - ab / (ta + tb))), flush=True)
