#!/usr/bin/env python3
"""Throughput of every path a user of the reference can reach, not only the benchmark's: flips by quarter turns, colour and
mono input, the four white-balance methods, the enhancer, the reference's own example configuration.  One line per path:
frames/s of a resident batch and the per-kernel-class times, so that a path that is several times slower than the tuned one
stands out.  usage (GPU box): path_survey.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from raw_image_pipeline_amd import RawImagePipeline, synth  # noqa: E402

BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ONLY = os.environ.get("PATH_SURVEY_ONLY", "")  # substring filter of the path names


def measure(name, pipe, frames, enc, steps=6):
    if ONLY and ONLY not in name:
        return
    pipe.set_stream(torch.cuda.current_stream())
    out = pipe.apply_device(frames, enc)
    pipe.apply_device(frames, enc, out=out)
    torch.cuda.synchronize()
    pipe.profile_begin(8 * steps + 8)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.apply_device(frames, enc, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = pipe.profile_end()
    cls = " ".join("%s %.3f" % (k, ms / n * (n / steps)) for k, (ms, n) in prof.items() if n)
    px = frames.shape[1] * frames.shape[2]
    print("%-46s %9.0f frames/s  %7.1f us/frame  %6.2f Gpx/s  | ms per step: %s" % (name, frames.shape[0] / dt, dt / frames.shape[0] * 1e6,
                                                                                 frames.shape[0] * px / dt / 1e9, cls), flush=True)


def full(w, h, wb="grey_world"):
    p = RawImagePipeline(False, "", "", "", device=0)
    synth.configure_full_chain(p, w, h, wb)
    return p


def main():
    w, h = 2448, 2048
    base = synth.gen_frame(w, h, "bayer_rggb8", seed=1, kind="scene")
    bayer = torch.from_numpy(np.stack([np.roll(base, 2 * i, axis=0) for i in range(BATCH)])).cuda()
    for angle in (180, 0, 90, 270):
        p = full(w, h)
        p.set_undistortion(False)
        p.set_flip(angle != 0)
        p.set_flip_angle(angle)
        measure("bayer full chain, no remap, flip %d" % angle, p, bayer, "bayer_rggb8")
    for wb in ("grey_world", "pca", "simple"):
        p = full(w, h, wb)
        measure("bayer full chain + remap, wb %s" % wb, p, bayer, "bayer_rggb8")
    p = full(w, h)
    filt, bias = synth.ccc_model()
    p.set_ccc_model(filt, bias)
    p.set_white_balance_method("ccc")
    measure("bayer full chain + remap, wb ccc", p, bayer, "bayer_rggb8")
    p = full(w, h)
    p.set_color_enhancer(True)
    p.set_color_enhancer_saturation_gain(1.2)
    measure("bayer full chain + enhancer + remap", p, bayer, "bayer_rggb8")
    for pat in ("bayer_bggr8", "bayer_gbrg8", "bayer_grbg8"):
        measure("bayer full chain + remap, %s" % pat, full(w, h), bayer, pat)
    bgr = torch.from_numpy(np.stack([synth.gen_scene_bgr(w, h, seed=2)] * BATCH)).cuda()
    for enc in ("bgr8", "rgb8"):
        measure("colour input %s, full chain + remap" % enc, full(w, h), bgr, enc)
    for enc in ("bgr8", "rgb8"):  # the memory-rate stage set (grey-world gains, colour matrix, gamma): chain inside the remap's tiles
        p = full(w, h)
        p.set_vignetting_correction(False)
        measure("colour input %s, no vignetting, + remap" % enc, p, bgr, enc)
    p = RawImagePipeline(False, "", "", "", device=0)
    p.set_flip(True)
    p.set_flip_angle(180)
    p.set_gamma_correction(True)
    p.set_undistortion(True)
    synth.load_camera(p, synth.camera_model(w, h))
    measure("mono8 input, flip + gamma + remap", p, bayer, "mono8")
    # the reference's example configuration (pipeline_params_example.yaml: debayer + ccc + undistortion, fov_scale 0.8) on its
    # 1.6 MP sensor (alphasense_calib_1.6mp_example.yaml: 1440 x 1080, the camera model of synth.camera_model)
    w2, h2 = 1440, 1080
    b2 = synth.gen_frame(w2, h2, "bayer_gbrg8", seed=3, kind="scene")
    f2 = torch.from_numpy(np.stack([np.roll(b2, 2 * i, axis=0) for i in range(BATCH)])).cuda()
    p = RawImagePipeline(False, "", "", "", device=0)
    p.set_ccc_model(filt, bias)
    p.set_white_balance(True)
    p.set_white_balance_method("ccc")
    p.set_gamma_correction(False)
    p.set_undistortion(True)
    synth.load_camera(p, synth.camera_model(w2, h2))
    p.set_undistortion_fov_scale(0.8)
    measure("reference example config, 1440x1080 ccc + remap", p, f2, "bayer_gbrg8")
    one = f2[:1].contiguous()
    measure("  the same, ONE resident frame per call", p, one, "bayer_gbrg8", steps=50)


if __name__ == "__main__":
    main()
