"""Seeded random configurations of the whole path, HIP pipeline vs CPU oracle, bit for bit: every stage toggle,
all white-balance methods, random colour matrices / gains / camera crops, all flips and Bayer patterns, sizes on and
off the fast paths (widths not a multiple of 4, odd heights), single frames with taps and resident batches."""
import numpy as np
import pytest

from raw_image_pipeline_amd import synth
from helpers import assert_images_equal, cfg, configure, oracle_run

import os

pytestmark = pytest.mark.gpu
N_BAYER = int(os.environ.get("RIP_FUZZ_CASES", "60"))   # RIP_FUZZ_CASES=2000 for a soak run
N_COLOUR = max(1, N_BAYER * 2 // 5)

PATTERNS = ["bayer_rggb8", "bayer_bggr8", "bayer_gbrg8", "bayer_grbg8"]


def random_case(rng):
    fast = rng.random() < 0.7
    if fast:
        w, h = int(rng.integers(8, 80)) * 4, int(rng.integers(8, 70)) * 2
    else:
        w, h = int(rng.integers(9, 200)), int(rng.integers(9, 150))
    angle = int(rng.choice([0, 90, 180, 270]))
    flip = bool(rng.random() < 0.6)
    ow, oh = (h, w) if (flip and angle in (90, 270)) else (w, h)
    wb_method = str(rng.choice(["grey_world", "pca", "simple"]))
    c = cfg(
        flip=flip, flip_angle=angle,
        wb=bool(rng.random() < 0.7), wb_method=wb_method, wb_bright=float(rng.uniform(0.3, 1.0)), wb_percentile=float(rng.uniform(1, 40)),
        cc=bool(rng.random() < 0.6), cc_matrix=list((np.eye(3) + rng.normal(0, 0.35, (3, 3))).ravel()),
        cc_bias=tuple(float(v) for v in (rng.normal(0, 6, 3) if rng.random() < 0.4 else np.zeros(3))),
        gamma=bool(rng.random() < 0.6), gamma_k=float(rng.uniform(0.5, 2.4)),
        vig=bool(rng.random() < 0.6), vig_params=(float(rng.uniform(0.5, 3.0)), float(rng.uniform(1e-4, 3e-3)), float(rng.uniform(0, 8e-6))),
        ce=bool(rng.random() < 0.5), ce_hue=float(rng.uniform(0.8, 1.2)), ce_sat=float(rng.uniform(0.5, 2.0)), ce_val=float(rng.uniform(0.8, 1.3)),
        undistort=bool(rng.random() < 0.6), cam=synth.camera_model(ow, oh), balance=float(rng.uniform(0, 1)), fov_scale=float(rng.uniform(0.6, 1.6)),
    )
    kind = str(rng.choice(["scene", "uniform"]))
    return w, h, str(rng.choice(PATTERNS)), kind, c


@pytest.mark.parametrize("seed", range(N_BAYER))
def test_random_configuration(gpu_pipe, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    w, h, pattern, kind, c = random_case(rng)
    configure(gpu_pipe, c)
    frame = synth.gen_frame(w, h, pattern, seed=seed, kind=kind, tint=(float(rng.uniform(0.5, 1)), 1.0, float(rng.uniform(0.5, 1))))
    what = "seed %d: %dx%d %s %s" % (seed, w, h, pattern, {k: v for k, v in c.items() if k not in ("cam", "cc_matrix")})
    got = gpu_pipe.process(frame, pattern)
    ref, enc, t_deb, t_col = oracle_run(oracle, c, frame, pattern, taps=True)
    assert_images_equal(got, ref, what)
    assert_images_equal(gpu_pipe.get_dist_debayered_image(), t_deb.reshape(gpu_pipe.get_dist_debayered_image().shape), what + " debayered tap")
    assert_images_equal(gpu_pipe.get_dist_color_image(), t_col.reshape(gpu_pipe.get_dist_color_image().shape), what + " colour tap")
    if seed % 3 == 0:  # the same configuration as a resident batch (frames innermost in the kernels)
        import torch
        n = int(rng.integers(2, 7))
        frames = np.stack([synth.gen_frame(w, h, pattern, seed=100 * seed + i, kind=kind) for i in range(n)])
        out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), pattern)
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        for i in range(n):
            ref, _ = oracle_run(oracle, c, frames[i], pattern)
            assert_images_equal(out[i], ref, what + " batch frame %d/%d" % (i, n))


@pytest.mark.parametrize("seed", range(N_COLOUR))
def test_random_configuration_on_colour_and_mono_input(gpu_pipe, oracle, seed):
    rng = np.random.default_rng(9000 + seed)
    w, h, _, _, c = random_case(rng)
    encoding = str(rng.choice(["bgr8", "rgb8", "mono8"]))
    if encoding == "mono8":
        c.update(vig=False)  # cvtColor(BGR2Lab) asserts on one channel (tested in test_error_behaviour)
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        img = synth.gen_scene_bgr(w, h, seed=seed) if rng.random() < 0.5 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    configure(gpu_pipe, c)
    what = "seed %d: %dx%d %s %s" % (seed, w, h, encoding, {k: v for k, v in c.items() if k not in ("cam", "cc_matrix")})
    got = gpu_pipe.process(img, encoding)
    ref, _ = oracle_run(oracle, c, img, encoding)
    assert_images_equal(got, ref.reshape(got.shape), what)
    if seed % 2 == 0:  # resident batch, no tap: the paths that skip the chain (remap straight from the caller's frames; mono8 flip + table in the gather)
        import torch
        n = int(rng.integers(2, 6))
        frames = np.stack([img] + [rng.integers(0, 256, img.shape, dtype=np.uint8) for _ in range(n - 1)])
        out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), encoding)
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        for i in range(n):
            ref, _ = oracle_run(oracle, c, frames[i], encoding)
            assert_images_equal(out[i].reshape(ref.shape), ref, what + " batch frame %d/%d" % (i, n))


@pytest.mark.parametrize("seed", range(max(1, N_BAYER // 3)))
def test_random_ccc_sequences(gpu_pipe, oracle, seed):
    """Convolutional colour constancy on random sizes (720x540 takes the 2x2 area path of the resize), flips,
    thresholds and Kalman models; a short sequence per case so the filter state carries from frame to frame,
    every other case as one resident batch."""
    import torch
    rng = np.random.default_rng(12000 + seed)
    if rng.random() < 0.3:
        w, h = 720, 540
    else:
        w, h = int(rng.integers(24, 120)) * 4, int(rng.integers(20, 100)) * 2
    pattern = str(rng.choice(PATTERNS))
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    occ = oracle.CCC(filt, bias)
    kal = (0.0, 1.0) if rng.random() < 0.3 else (1.0, float(rng.uniform(1, 20)))
    occ.set_kalman_model(*kal)
    gpu_pipe.set_ccc_kalman_model(*kal)
    angle = int(rng.choice([0, 90, 180, 270]))
    c = cfg(wb=True, wb_method="ccc", wb_bright=float(rng.uniform(0.6, 1.0)), wb_dark=float(rng.uniform(0.0, 0.3)),
            wb_temporal=bool(rng.random() < 0.7), flip=bool(rng.random() < 0.5), flip_angle=angle,
            gamma=bool(rng.random() < 0.5), gamma_k=0.9, ce=bool(rng.random() < 0.5), ce_sat=1.2)
    configure(gpu_pipe, c)
    gpu_pipe.reset_white_balance_temporal_consistency()
    occ.reset()
    n = int(rng.integers(2, 6))
    frames = [synth.gen_frame(w, h, pattern, seed=31 * seed + i, kind="scene", tint=(0.6 + 0.08 * i, 1.0, 0.5 + 0.05 * i)) for i in range(n)]
    what = "ccc seed %d: %dx%d %s kalman %s %s" % (seed, w, h, pattern, kal, {k: c[k] for k in ("wb_bright", "wb_dark", "wb_temporal", "flip", "flip_angle")})
    if seed % 2:
        out = gpu_pipe.apply_device(torch.from_numpy(np.stack(frames)).cuda(), pattern)
        torch.cuda.synchronize()
        got = list(out.cpu().numpy())
    else:
        got = [gpu_pipe.process(f, pattern) for f in frames]
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], pattern, ccc=occ)
        assert_images_equal(got[i], ref, what + " frame %d/%d" % (i, n))
