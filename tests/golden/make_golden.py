#!/usr/bin/env python3
"""Generates tests/golden/vectors.npz: small seeded inputs and the CPU oracle's outputs for them.

The reference ships no golden vectors and cannot run here (OpenCV is absent), so these vectors freeze
the ORACLE (parity unpinned): they detect silent changes of the restatement and let the GPU suite
compare against fixed data.  Regenerate with `python tests/golden/make_golden.py` only when the
oracle is deliberately changed, and say why in the commit."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from helpers import cfg, oracle_run  # noqa: E402
from raw_image_pipeline_amd import synth  # noqa: E402

W, H = 48, 32


def cases():
    """name -> (config dict, encoding, frame)"""
    cam = synth.camera_model(W, H)
    out = {}
    for pat in sorted(synth.PATTERNS):
        out["debayer_" + pat] = (cfg(), pat, synth.gen_frame(W, H, pat, seed=1, kind="uniform"))
    f = synth.gen_frame(W, H, "bayer_rggb8", seed=2, kind="scene")
    for a in (90, 180, 270):
        out["flip_%d" % a] = (cfg(flip=True, flip_angle=a), "bayer_rggb8", f)
    out["gamma_0.8"] = (cfg(gamma=True, gamma_k=0.8), "bayer_rggb8", f)
    out["grey_world"] = (cfg(wb=True, wb_method="grey_world"), "bayer_rggb8", f)
    out["pca"] = (cfg(wb=True, wb_method="pca"), "bayer_rggb8", f)
    out["color_calibration"] = (cfg(cc=True), "bayer_rggb8", f)
    out["vignetting"] = (cfg(vig=True), "bayer_rggb8", f)
    out["color_enhancer"] = (cfg(ce=True, ce_sat=1.2, ce_hue=1.1, ce_val=0.9), "bayer_rggb8", f)
    out["undistortion"] = (cfg(undistort=True, cam=cam), "bayer_rggb8", f)
    out["full_chain"] = (cfg(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, vig=True, ce=True,
                             ce_sat=1.2, undistort=True, cam=cam), "bayer_gbrg8", synth.gen_frame(W, H, "bayer_gbrg8", seed=3))
    return out


def ccc_sequence():
    filt, bias = synth.ccc_model()
    occ = O.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    c = cfg(wb=True, wb_method="ccc", wb_temporal=True)
    frames, outs = [], []
    for i in range(3):
        fr = synth.gen_frame(96, 72, "bayer_rggb8", seed=40 + i, kind="scene", tint=(0.6 + 0.1 * i, 1.0, 0.5))
        out, _ = oracle_run(O, c, fr, "bayer_rggb8", ccc=occ)
        frames.append(fr)
        outs.append(out)
    return c, np.stack(frames), np.stack(outs)


def main():
    data = {}
    for name, (c, enc, frame) in cases().items():
        out, _ = oracle_run(O, c, frame, enc)
        data["in__" + name] = frame
        data["out__" + name] = out
    _, fr, outs = ccc_sequence()
    data["in__ccc_sequence"] = fr
    data["out__ccc_sequence"] = outs
    np.savez_compressed(os.path.join(HERE, "vectors.npz"), **data)
    print("wrote %d arrays" % len(data))


if __name__ == "__main__":
    main()
