"""The contracted floating-point model on the GPU: rip_set_fp_contraction(1) selects the kernels compiled with
RIP_FP_CONTRACT=1 (colour matrix, pca map, HSV inverse as fused multiply-adds; the vignetting mask with its two contracted
sums) and they must equal the oracle's contraction model 1 bit for bit, exactly as the default kernels equal model 0.
Reference expressions: color_calibration.cpp:93-103, white_balance.cpp:122-127, color_enhancer.cpp:38-47,
vignetting_correction.cpp:42-43; why the model exists: README.md:191-201 (the reference's Jetson / aarch64 deployment, where
GCC contracts a*b + c by default)."""
import numpy as np
import pytest

from helpers import assert_images_equal, cfg, configure, oracle_run
from raw_image_pipeline_amd import synth

pytestmark = pytest.mark.gpu


def run_mode(pipe, O, c, frame, encoding, mode, what):
    configure(pipe, c)
    pipe.set_fp_contraction(mode)
    got = pipe.process(frame, encoding)
    with O.fp_contraction(mode):
        ref, _ = oracle_run(O, c, frame, encoding)
    assert_images_equal(got, ref, "%s, contraction model %d" % (what, mode), 0)
    return got, ref


def full_chain_cfg(w, h, **kw):
    base = dict(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, gamma_k=0.8, vig=True,
                undistort=True, cam=synth.camera_model(w, h))
    base.update(kw)
    return cfg(**base)


def test_colour_matrix_both_models_and_they_differ(gpu_pipe, oracle):
    """A matrix with full 24-bit mantissas on every byte triple of a 2^18-value sample: the two models must each match
    their oracle, and somewhere they must differ (else the test would not tell the kernels apart)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)
    m = [1.0113, -0.2377, 0.1049, -0.3151, 1.4127, -0.1633, 0.0871, -0.6419, 1.5237]
    c = cfg(cc=True, cc_matrix=m, cc_bias=(0.37, -1.21, 2.6))
    got0, ref0 = run_mode(gpu_pipe, oracle, c, img, "bgr8", 0, "colour matrix")
    got1, ref1 = run_mode(gpu_pipe, oracle, c, img, "bgr8", 1, "colour matrix")
    d = np.abs(got0.astype(np.int16) - got1.astype(np.int16))
    assert d.max() == 1 and 0 < (d > 0).sum() < d.size // 1000, "models differ on %d values, max %d" % ((d > 0).sum(), d.max())
    # Bayer input takes the 4x2-item kernel, bgr8 the 12-byte-lane one: both sets were compiled twice
    frame = synth.gen_frame(256, 192, "bayer_grbg8", seed=3, kind="uniform")
    run_mode(gpu_pipe, oracle, cfg(cc=True, cc_matrix=m), frame, "bayer_grbg8", 1, "colour matrix on bayer_grbg8")


@pytest.mark.parametrize("encoding", ["bayer_rggb8", "bgr8"])
def test_pca_map_contracted(gpu_pipe, oracle, encoding):
    w, h = 320, 240
    frame = synth.gen_frame(w, h, encoding, seed=12, kind="scene") if encoding != "bgr8" else synth.gen_scene_bgr(w, h, seed=12)
    run_mode(gpu_pipe, oracle, cfg(wb=True, wb_method="pca"), frame, encoding, 1, "pca white balance")


@pytest.mark.parametrize("gains", [(1.0, 1.2, 1.0), (1.3, 0.7, 1.1), (2.0, 3.0, 0.5)])
def test_colour_enhancer_contracted(gpu_pipe, oracle, gains):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    img[:4] = img[:4, :, :1]  # grey rows: s == 0
    run_mode(gpu_pipe, oracle, cfg(ce=True, ce_hue=gains[0], ce_sat=gains[1], ce_val=gains[2]), img, "bgr8", 1, "hsv gains %s" % (gains,))


@pytest.mark.parametrize("size", [(64, 48), (37, 29), (640, 480)])
def test_vignetting_mask_and_stage_contracted(gpu_pipe, oracle, size):
    """The mask plane is built on the host under the same model (k = fma(r^2, a2, r^4 a4)); switching the model rebuilds it."""
    w, h = size
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=4, kind="uniform")
    c = cfg(vig=True, gamma=True, flip=True, flip_angle=180)
    run_mode(gpu_pipe, oracle, c, frame, "bayer_rggb8", 1, "vignetting %s" % (size,))
    with oracle.fp_contraction(1):
        ref_mask = oracle.vignetting_mask(h, w, *c["vig_params"])
    assert np.array_equal(gpu_pipe.get_vignetting_mask(h, w), ref_mask)
    run_mode(gpu_pipe, oracle, c, frame, "bayer_rggb8", 0, "vignetting %s, back to the default model" % (size,))
    assert np.array_equal(gpu_pipe.get_vignetting_mask(h, w), oracle.vignetting_mask(h, w, *c["vig_params"]))


@pytest.mark.parametrize("pattern,angle", [("bayer_rggb8", 180), ("bayer_bggr8", 0), ("bayer_gbrg8", 90)])
def test_full_chain_contracted(gpu_pipe, oracle, pattern, angle):
    w, h = 320, 240
    frame = synth.gen_frame(w, h, pattern, seed=21, kind="scene")
    cam = synth.camera_model(w, h) if angle != 90 else synth.camera_model(h, w)
    c = full_chain_cfg(w, h, flip_angle=angle, ce=True, ce_sat=1.2, cam=cam)
    run_mode(gpu_pipe, oracle, c, frame, pattern, 1, "full chain %s flip %d" % (pattern, angle))


def test_chain_inside_the_remap_tiles_contracted(gpu_pipe, oracle):
    """debayer + gains + matrix + gamma + undistortion, no tap: ONE kernel (rip_fused.hip), compiled under both models."""
    import torch
    w, h, n = 448, 272, 5
    c = full_chain_cfg(w, h, vig=False)
    configure(gpu_pipe, c)
    gpu_pipe.set_fp_contraction(1)
    frames = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=40 + i, kind="scene") for i in range(n)])
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_rggb8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    with oracle.fp_contraction(1):
        for i in range(n):
            ref, _ = oracle_run(oracle, c, frames[i], "bayer_rggb8")
            assert_images_equal(out[i], ref, "frame %d" % i)


def test_config2_full_size_contracted(gpu_pipe, oracle):
    w, h = 2448, 2048
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=0, kind="scene")
    run_mode(gpu_pipe, oracle, full_chain_cfg(w, h), frame, "bayer_rggb8", 1, "config 2 at 2448x2048")


def test_only_models_0_and_1_exist(gpu_pipe):
    with pytest.raises(ValueError):  # std::invalid_argument
        gpu_pipe.set_fp_contraction(2)
    gpu_pipe.set_fp_contraction(0)
