"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs.  Bars (BASELINE.json north_star / BASELINE.md section 4): flip and LUT gamma bit-exact;
debayer / remap within +-1 LSB; the other stages +-1 LSB as declared by the build.  The oracle is
an integer/IEEE restatement, so every test below in fact demands tol = 0."""
import numpy as np
import pytest

from helpers import assert_images_equal, cfg, configure, oracle_run
from raw_image_pipeline_amd import RipAssertError, RipError, synth

pytestmark = pytest.mark.gpu

PATTERNS = ["bayer_rggb8", "bayer_grbg8", "bayer_gbrg8", "bayer_bggr8"]
TOL_EXACT = 0        # flip, gamma LUT: bit-exact (north_star)
TOL_INTERP = 0       # debayer, remap: north_star allows 1 LSB; the integer restatement is exact
TOL_DECLARED = 0     # wb gains, colour matrix, hsv, vignetting: build declares 1 LSB; measured 0


def run_both(pipe, O, c, frame, encoding, tol, ccc=None, what=""):
    configure(pipe, c)
    got = pipe.process(frame, encoding)
    ref, enc = oracle_run(O, c, frame, encoding, ccc=ccc)
    assert pipe.last_encoding == enc
    assert_images_equal(got, ref, what, tol)
    return got


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("size", [(64, 48), (132, 36), (37, 29), (6, 4), (3, 3), (5, 8)])
@pytest.mark.parametrize("kind", ["uniform", "scene"])
def test_debayer(gpu_pipe, oracle, pattern, size, kind):
    w, h = size
    frame = synth.gen_frame(w, h, pattern, seed=w * 1000 + h, kind=kind)
    run_both(gpu_pipe, oracle, cfg(), frame, pattern, TOL_INTERP, what="debayer %s %s" % (pattern, size))


@pytest.mark.parametrize("angle", [0, 90, 180, 270, 45])
@pytest.mark.parametrize("size", [(64, 48), (37, 29)])
def test_flip_after_debayer(gpu_pipe, oracle, angle, size):
    w, h = size
    frame = synth.gen_frame(w, h, "bayer_gbrg8", seed=angle, kind="uniform")
    got = run_both(gpu_pipe, oracle, cfg(flip=True, flip_angle=angle), frame, "bayer_gbrg8", TOL_EXACT, what="flip %d" % angle)
    assert got.shape[:2] == ((w, h) if angle in (90, 270) else (h, w))


@pytest.mark.parametrize("encoding,cn", [("bgr8", 3), ("rgb8", 3), ("mono8", 1)])
@pytest.mark.parametrize("angle", [90, 180, 270])
def test_flip_colour_and_mono_inputs(gpu_pipe, oracle, encoding, cn, angle):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (30, 44, cn) if cn == 3 else (30, 44), dtype=np.uint8)
    run_both(gpu_pipe, oracle, cfg(flip=True, flip_angle=angle, gamma=True, gamma_k=0.9), img, encoding, TOL_EXACT,
             what="%s flip %d" % (encoding, angle))
    if encoding == "rgb8":
        assert gpu_pipe.last_encoding == "rgb8"  # debayer.cpp:72-73: the CPU path keeps the string


def test_flip_twice_is_identity(gpu_pipe):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    configure(gpu_pipe, cfg(flip=True, flip_angle=180))
    once = gpu_pipe.process(img, "bgr8")
    assert np.array_equal(gpu_pipe.process(once, "bgr8"), img)
    configure(gpu_pipe, cfg(flip=True, flip_angle=90))
    r90 = gpu_pipe.process(img, "bgr8")
    configure(gpu_pipe, cfg(flip=True, flip_angle=270))
    assert np.array_equal(gpu_pipe.process(r90, "bgr8"), img)


@pytest.mark.parametrize("k", [0.8, 0.9, 1.0, 2.2])
def test_gamma(gpu_pipe, oracle, k):
    frame = synth.gen_frame(64, 48, "bayer_rggb8", seed=11, kind="uniform")
    run_both(gpu_pipe, oracle, cfg(gamma=True, gamma_k=k), frame, "bayer_rggb8", TOL_EXACT, what="gamma %g" % k)


def test_gamma_default_method_equals_custom(gpu_pipe, oracle):
    frame = synth.gen_frame(640, 480, "bayer_rggb8", seed=0, kind="scene")  # BASELINE config 1
    a = run_both(gpu_pipe, oracle, cfg(gamma=True, gamma_k=0.8, gamma_method="default"), frame, "bayer_rggb8", TOL_EXACT)
    b = run_both(gpu_pipe, oracle, cfg(gamma=True, gamma_k=0.8, gamma_method="custom"), frame, "bayer_rggb8", TOL_EXACT)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("method", ["grey_world", "gray_world", "pca"])
@pytest.mark.parametrize("kind,size", [("scene", (128, 96)), ("uniform", (64, 48)), ("scene", (51, 33))])
def test_white_balance_statistics_methods(gpu_pipe, oracle, method, kind, size):
    frame = synth.gen_frame(size[0], size[1], "bayer_rggb8", seed=5, kind=kind)
    run_both(gpu_pipe, oracle, cfg(wb=True, wb_method=method, wb_bright=0.8), frame, "bayer_rggb8", TOL_DECLARED,
             what="wb %s %s" % (method, kind))


@pytest.mark.parametrize("percentile", [1.0, 10.0, 20.0, 45.0])
@pytest.mark.parametrize("kind,size", [("scene", (128, 96)), ("uniform", (64, 48)), ("scene", (51, 33))])
def test_simple_white_balance(gpu_pipe, oracle, percentile, kind, size):
    frame = synth.gen_frame(size[0], size[1], "bayer_gbrg8", seed=7, kind=kind)
    run_both(gpu_pipe, oracle, cfg(wb=True, wb_method="simple", wb_percentile=percentile), frame, "bayer_gbrg8", TOL_DECLARED,
             what="simple wb p=%g %s" % (percentile, kind))


def test_grey_world_gains_match_oracle(gpu_pipe, oracle):
    frame = synth.gen_frame(128, 96, "bayer_grbg8", seed=9, kind="scene")
    configure(gpu_pipe, cfg(wb=True, wb_method="grey_world", wb_bright=0.9))
    gpu_pipe.process(frame, "bayer_grbg8")
    info = gpu_pipe.get_white_balance_info(1)[0]
    _, sums, ig = oracle.wb_grayworld(oracle.debayer(frame, "bayer_grbg8"), 0.9, return_stats=True)
    assert [int(v) for v in info[3:6]] == ig


def test_color_calibration(gpu_pipe, oracle):
    frame = synth.gen_frame(64, 48, "bayer_rggb8", seed=2, kind="uniform")
    run_both(gpu_pipe, oracle, cfg(cc=True), frame, "bayer_rggb8", TOL_DECLARED, what="cc example matrix")
    run_both(gpu_pipe, oracle, cfg(cc=True, cc_matrix=[0.5, 0.25, 0.25, -0.5, 1.5, 0.1, 0.3, 0.3, 0.41], cc_bias=(3.5, -7.25, 10.0)),
             frame, "bayer_rggb8", TOL_DECLARED, what="cc with bias")
    got = run_both(gpu_pipe, oracle, cfg(cc=True, cc_matrix=[1, 0, 0, 0, 1, 0, 0, 0, 1]), frame, "bayer_rggb8", TOL_EXACT)
    assert np.array_equal(got, oracle.debayer(frame, "bayer_rggb8"))  # identity matrix is a no-op


@pytest.mark.parametrize("size", [(64, 48), (48, 64), (40, 40), (37, 29)])
@pytest.mark.parametrize("gamma", [False, True])
def test_vignetting(gpu_pipe, oracle, size, gamma):
    frame = synth.gen_frame(size[0], size[1], "bayer_rggb8", seed=4, kind="uniform")
    run_both(gpu_pipe, oracle, cfg(vig=True, gamma=gamma, flip=True, flip_angle=180), frame, "bayer_rggb8", TOL_DECLARED,
             what="vignetting %s gamma=%s" % (size, gamma))


def test_vignetting_dark_and_saturated_pixels(gpu_pipe, oracle):
    # exercises both branches of abToXZ (dark pixels take the linear segment) and the clamps
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    img[:16] //= 16
    img[16:24] = 0
    img[24:32] = 255
    img[32:40, :, 0] = 255
    img[32:40, :, 1:] //= 32
    run_both(gpu_pipe, oracle, cfg(vig=True, vig_params=(2.5, 2e-3, 5e-6)), img, "bgr8", TOL_DECLARED, what="vignetting extremes")


@pytest.mark.parametrize("gains", [(1.0, 1.2, 1.0), (1.0, 1.5, 1.0), (1.3, 0.7, 1.1), (1.0, 1.0, 1.0), (2.0, 3.0, 0.5)])
def test_color_enhancer(gpu_pipe, oracle, gains):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    img[:4] = img[:4, :, :1]  # grey rows: s == 0 branch
    run_both(gpu_pipe, oracle, cfg(ce=True, ce_hue=gains[0], ce_sat=gains[1], ce_val=gains[2]), img, "bgr8", TOL_DECLARED,
             what="hsv gains %s" % (gains,))


@pytest.mark.parametrize("size", [(64, 48), (96, 80), (50, 38)])
@pytest.mark.parametrize("balance,fov", [(0.0, 1.0), (0.5, 1.2), (1.0, 0.8)])
def test_undistortion(gpu_pipe, oracle, size, balance, fov):
    w, h = size
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=6, kind="uniform")
    c = cfg(undistort=True, cam=synth.camera_model(w, h), balance=balance, fov_scale=fov)
    run_both(gpu_pipe, oracle, c, frame, "bayer_rggb8", TOL_INTERP, what="remap %s b=%g f=%g" % (size, balance, fov))


def test_undistortion_mono_and_map_size_differs_from_image(gpu_pipe, oracle):
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    c = cfg(undistort=True, cam=synth.camera_model(64, 48))  # maps are 48x64, image is 40x56
    got = run_both(gpu_pipe, oracle, c, img, "mono8", TOL_INTERP, what="remap mono")
    assert got.shape == (48, 64)


@pytest.mark.parametrize("fov,stages,ring", [(0.6, 2, 1), (0.6, 4, 1), (1.0, 2, 1), (1.0, 3, 1), (1.0, 4, 1), (2.0, 2, 1),
                                               (2.0, 3, 1), (3.6, 2, 1), (1.0, 2, 0), (2.0, 2, 0)])
def test_undistortion_batch_through_the_lds_ring(gpu_pipe, oracle, monkeypatch, fov, stages, ring):
    """The tiled remap streams the frames of a batch through an LDS ring (LDS-DMA, `stages` buffers).  fov_scale
    widens the source rectangle of a tile: 1 chunk per lane (0.6), 2 (1.0), 4 (2.0), then the unpipelined fallback kernel (3.6)."""
    import torch
    gpu_pipe.set_tunable("remap_stages", stages)
    gpu_pipe.set_tunable("remap_ring", ring)
    w, h, n = 448, 272, 7
    c = cfg(undistort=True, cam=synth.camera_model(w, h), fov_scale=fov)
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, "bayer_grbg8", seed=300 + i, kind="uniform") for i in range(n)])
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_grbg8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], "bayer_grbg8")
        assert_images_equal(out[i], ref, "ring frame %d (fov %g, %d stages)" % (i, fov, stages))


@pytest.mark.parametrize("size,n", [((448, 272), 7), ((1000, 752), 13), ((2448, 2048), 4), ((640, 482), 1)])
def test_the_deals_of_tiles_and_chunks_to_the_xcds_do_not_change_a_byte(gpu_pipe, oracle, size, n):
    """Round 6: the remap's tiles and the chain's chunks are dealt to the XCDs round-robin in runs (RIP_REMAP_DEAL /
    RIP_CHAIN_DEAL) instead of one contiguous range each.  The deal decides WHICH workgroup takes a tile / chunk, never what
    it computes: every run length -- contiguous (0), the defaults, run lengths that leave ragged last runs and XCD shares
    without work (tile / chunk counts that are no multiples of 8 x run) -- gives the image of the default deal, and that one
    equals the oracle's (full chain of config 2: statistics, fused chain with the Lab round trip, ring remap)."""
    import torch
    w, h = size
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, gamma_k=0.8, vig=True, undistort=True,
            cam=synth.camera_model(w, h))
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=900 + i, kind="scene") for i in range(n)])
    dev = torch.from_numpy(frames).cuda()
    base = gpu_pipe.apply_device(dev, "bayer_rggb8").cpu().numpy()
    ref, _ = oracle_run(oracle, c, frames[n - 1], "bayer_rggb8")
    assert_images_equal(base[n - 1], ref, "default deals vs oracle %s" % (size,))
    for chain_deal, remap_deal, frames_per_visit in [(0, 0, 0), (1, 1, 1), (2, 2, 2), (5, 3, 5), (7, 1, 3), (64, 9, 0), (3, 0, 4), (0, 1, 16)]:
        gpu_pipe.set_tunable("chain_deal", chain_deal)
        gpu_pipe.set_tunable("remap_deal", remap_deal)
        gpu_pipe.set_tunable("remap_frames", frames_per_visit)
        got = gpu_pipe.apply_device(dev, "bayer_rggb8").cpu().numpy()
        assert np.array_equal(got, base), "chain_deal %d remap_deal %d frames %d changes the image at %s" % (chain_deal, remap_deal, frames_per_visit, size)


@pytest.mark.parametrize("size,balance,fov", [((2448, 2048), 0.0, 1.0), ((1920, 1200), 1.0, 0.8), ((450, 270), 0.5, 3.6), ((131, 97), 1.0, 0.6)])
def test_remap_plan_compiled_on_the_device_equals_the_host_plan(rip_lib, oracle, monkeypatch, size, balance, fov):
    """The remap plan (tile rectangles, 4-byte plan words, border pixels) is compiled on the device, where the maps are
    (rip_maps.hip remap_plan_kernel); RIP_PLAN_ON_HOST=1 keeps rip_host.cpp's compiler.  Both handles must produce the same
    image -- geometries with many border pixels (balance 1: the rectified image reaches beyond the source), rectangles too
    wide for the ring (fov 3.6), a partial last tile column and row -- and the oracle's, and hand out the same maps."""
    from raw_image_pipeline_amd import RawImagePipeline
    w, h = size
    frame = synth.gen_frame(w, h, "bayer_gbrg8", seed=77, kind="uniform")
    c = cfg(undistort=True, cam=synth.camera_model(w, h), balance=balance, fov_scale=fov)
    dev = RawImagePipeline(False, "", "", "", device=0)
    configure(dev, c)
    got_dev = dev.process(frame, "bayer_gbrg8")
    monkeypatch.setenv("RIP_PLAN_ON_HOST", "1")
    host = RawImagePipeline(False, "", "", "", device=0)
    configure(host, c)
    got_host = host.process(frame, "bayer_gbrg8")
    assert_images_equal(got_dev, got_host, "device plan vs host plan %s" % (size,))
    # same plan: tile grid, number of border pixels, largest source rectangle (a device plan that lists interior pixels as
    # border pixels gives the same image, only much slower)
    info_d, info_h = dev.debug_plan_info(h, w), host.debug_plan_info(h, w)
    assert info_d["on_device"] == 1 and info_h["on_device"] == 0
    for key in ("tiles_x", "tiles_y", "border_pixels", "max_lds_bytes", "max_rect_w", "max_rect_h"):
        assert info_d[key] == info_h[key], (key, info_d, info_h)
    mx_d, my_d = dev.get_undistortion_maps()
    mx_h, my_h = host.get_undistortion_maps()
    assert np.array_equal(mx_d, mx_h) and np.array_equal(my_d, my_h)
    if w * h <= 1920 * 1200:
        ref, _ = oracle_run(oracle, c, frame, "bayer_gbrg8")
        assert_images_equal(got_dev, ref, "device plan vs oracle %s" % (size,))


@pytest.mark.parametrize("angle", [90, 270])
@pytest.mark.parametrize("size,pattern", [((160, 120), "bayer_rggb8"), ((152, 100), "bayer_grbg8"), ((264, 130), "bayer_bggr8")])
def test_full_chain_with_quarter_turn_flips(gpu_pipe, oracle, angle, size, pattern):
    """90 / 270 degree flips of Bayer input take the rotated fast kernel (6-byte column pieces); sizes with and
    without whole 4-group x 64-pair workgroup tiles, taps included, then a batch on the device."""
    import torch
    w, h = size
    cam = synth.camera_model(h, w)  # the image is h wide after the flip
    c = full_chain_cfg(w, h, flip_angle=angle, cam=cam, ce=True, ce_sat=1.2)
    configure(gpu_pipe, c)
    frame = synth.gen_frame(w, h, pattern, seed=77, kind="scene")
    got = gpu_pipe.process(frame, pattern)
    ref, enc, t_deb, t_col = oracle_run(oracle, c, frame, pattern, taps=True)
    assert got.shape == (w, h, 3)
    assert_images_equal(got, ref, "final")
    assert_images_equal(gpu_pipe.get_dist_debayered_image(), t_deb.reshape(w, h, 3), "debayered tap")
    assert_images_equal(gpu_pipe.get_dist_color_image(), t_col.reshape(w, h, 3), "colour tap")
    frames = np.stack([synth.gen_frame(w, h, pattern, seed=500 + i, kind="uniform") for i in range(3)])
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), pattern)
    torch.cuda.synchronize()
    for i in range(3):
        ref, _ = oracle_run(oracle, c, frames[i], pattern)
        assert_images_equal(out[i].cpu().numpy(), ref, "batch frame %d" % i)


def full_chain_cfg(w, h, **kw):
    base = dict(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, gamma_k=0.8, vig=True,
                undistort=True, cam=synth.camera_model(w, h))
    base.update(kw)
    return cfg(**base)


@pytest.mark.parametrize("pattern", PATTERNS)
def test_full_chain_config2_small(gpu_pipe, oracle, pattern):
    w, h = 320, 240
    frame = synth.gen_frame(w, h, pattern, seed=21, kind="scene")
    run_both(gpu_pipe, oracle, full_chain_cfg(w, h), frame, pattern, 0, what="full chain %s" % pattern)


def test_full_chain_with_enhancer_and_taps(gpu_pipe, oracle):
    w, h = 160, 120
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=22, kind="scene")
    c = full_chain_cfg(w, h, ce=True, ce_sat=1.2)
    configure(gpu_pipe, c)
    got = gpu_pipe.process(frame, "bayer_rggb8")
    ref, enc, t_deb, t_col = oracle_run(oracle, c, frame, "bayer_rggb8", taps=True)
    assert_images_equal(got, ref, "final")
    assert_images_equal(gpu_pipe.get_dist_debayered_image(), t_deb.reshape(h, w, 3), "debayered tap")
    assert_images_equal(gpu_pipe.get_dist_color_image(), t_col.reshape(h, w, 3), "colour tap")
    assert_images_equal(gpu_pipe.get_processed_image(), ref, "processed tap")
    assert gpu_pipe.get_rect_mask().size == 0


@pytest.mark.parametrize("encoding", ["bgr8", "rgb8"])
@pytest.mark.parametrize("wb", ["grey_world", "pca"])
@pytest.mark.parametrize("size", [(160, 120), (150, 111)])  # vectorised colour path / generic path
def test_full_chain_on_colour_input(gpu_pipe, oracle, encoding, wb, size):
    """The reference's Python demo feeds bgr8 images (apply_pipeline.py:49-53)."""
    w, h = size
    img = synth.gen_scene_bgr(w, h, seed=31)
    cam = synth.camera_model(w, h)
    c = full_chain_cfg(w, h, wb_method=wb, ce=True, ce_sat=1.3, cam=cam)
    run_both(gpu_pipe, oracle, c, img, encoding, 0, what="colour input %s %s %s" % (encoding, wb, size))
    c2 = full_chain_cfg(w, h, wb_method=wb, flip=False, undistort=False, vig=False)
    run_both(gpu_pipe, oracle, c2, img, encoding, 0, what="colour input no-vignette %s %s %s" % (encoding, wb, size))


@pytest.mark.parametrize("wb", ["grey_world", "pca", "simple", "none"])
def test_batch_with_many_frames_per_workgroup(gpu_pipe, oracle, monkeypatch, wb):
    """Frames are the innermost loop of the chain and remap kernels.  Small grids (environment caps) make every
    workgroup walk several frames of the batch, as it does at the benchmark's size: each frame must still equal
    the oracle's single-frame result."""
    import torch
    gpu_pipe.set_tunable("chain_blocks", 8)
    gpu_pipe.set_tunable("stats_blocks", 8)
    gpu_pipe.set_tunable("remap_per_cu", 1)
    w, h, n = 448, 272, 9
    c = full_chain_cfg(w, h, wb=(wb != "none"), wb_method=wb if wb != "none" else "grey_world", ce=True, ce_sat=1.1)
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=900 + i, kind="scene", tint=(0.6 + 0.03 * i, 1.0, 0.5 + 0.02 * i))
                       for i in range(n)])
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_rggb8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], "bayer_rggb8")
        assert_images_equal(out[i], ref, "frame %d of %d (%s)" % (i, n, wb))


def test_statistics_records_are_handed_back_clean(gpu_pipe, oracle):
    """The grey-world / pca statistics kernels finish their frames themselves: the workgroup that ends a frame last writes
    its gains and zeroes the frame's record, and the library skips the memset and the finalisation launch for as long as
    it knows the records are clean (rip_api.cpp run_batch).  One handle, batches of changing length, method, input kind and
    geometry back to back (SimpleWB in between keeps its own finalisation kernel, colour input takes another statistics
    kernel, a longer batch re-allocates the records): every frame of every batch must equal the oracle, i.e. no sum of an
    earlier batch may survive and no ticket counter may be left half-way."""
    import torch
    steps = [("grey_world", 3, (128, 96), "bayer_rggb8"), ("grey_world", 3, (128, 96), "bayer_rggb8"), ("pca", 2, (128, 96), "bayer_rggb8"),
             ("simple", 2, (128, 96), "bayer_rggb8"), ("grey_world", 5, (132, 36), "bayer_gbrg8"), ("pca", 1, (132, 36), "bayer_gbrg8"),
             ("grey_world", 4, (64, 48), "bgr8"), ("pca", 4, (37, 29), "bayer_grbg8"), ("grey_world", 17, (128, 96), "bayer_rggb8"),
             ("grey_world", 1, (128, 96), "bayer_rggb8"), ("pca", 9, (128, 96), "bayer_rggb8"), ("grey_world", 2, (640, 480), "bayer_rggb8")]
    for k, (method, n, (w, h), enc) in enumerate(steps):
        c = cfg(wb=True, wb_method=method, wb_bright=0.85, cc=True, gamma=True)
        configure(gpu_pipe, c)
        if enc == "bgr8":
            frames = np.stack([synth.gen_scene_bgr(w, h, seed=70 * k + i, tint=(0.6 + 0.05 * i, 1.0, 0.5)) for i in range(n)])
        else:
            frames = np.stack([synth.gen_frame(w, h, enc, seed=70 * k + i, kind="scene", tint=(0.6 + 0.05 * i, 1.0, 0.5)) for i in range(n)])
        out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), enc)
        torch.cuda.synchronize()
        out = out.cpu().numpy()
        for i in range(n):
            ref, _ = oracle_run(oracle, c, frames[i], enc)
            assert_images_equal(out[i], ref, "step %d (%s, %d frames of %dx%d %s), frame %d" % (k, method, n, w, h, enc, i))


def test_very_long_batch_is_sliced(gpu_pipe, oracle):
    """More frames than a grid dimension holds: the batch goes through in slices of 16384; 40 000 tiny frames, a few
    of them checked against the oracle, all of them against the frame pattern they repeat."""
    import torch
    w, h, n = 16, 8, 40000
    c = cfg(wb=True, wb_method="grey_world", gamma=True, gamma_k=0.9, flip=True, flip_angle=180)
    configure(gpu_pipe, c)
    base = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=600 + i, kind="uniform") for i in range(7)])
    frames = torch.from_numpy(base).cuda()[torch.arange(n, device="cuda") % 7]
    out = gpu_pipe.apply_device(frames.contiguous(), "bayer_rggb8")
    torch.cuda.synchronize()
    refs = [oracle_run(oracle, c, base[i], "bayer_rggb8")[0] for i in range(7)]
    out7 = out.reshape(-1, h, w, 3)
    for i in (0, 1, 6, 16383, 16384, 16385, 32768, n - 1):
        assert_images_equal(out7[i].cpu().numpy(), refs[i % 7], "frame %d of %d" % (i, n))
    expect = torch.from_numpy(np.stack(refs)).cuda()[torch.arange(n, device="cuda") % 7]
    assert torch.equal(out7, expect)


def test_full_size_batch_equals_single_frames(gpu_pipe):
    """Size-independent property at BASELINE's size: a resident batch gives, frame by frame, what the
    single-frame call gives (which test_full_chain_full_size_2448x2048 checks against the oracle)."""
    import torch
    w, h, n = 2448, 2048, 6
    configure(gpu_pipe, full_chain_cfg(w, h))
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=40 + i, kind="scene", tint=(0.65 + 0.05 * i, 1.0, 0.55)) for i in range(3)]
    singles = [gpu_pipe.process(f, "bayer_rggb8") for f in frames]
    batch = torch.from_numpy(np.stack([frames[i % 3] for i in range(n)])).cuda()
    out = gpu_pipe.apply_device(batch, "bayer_rggb8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], singles[i % 3]), "batch frame %d differs from the single-frame result" % i


def test_full_chain_full_size_2448x2048(gpu_pipe, oracle):
    """BASELINE configs[1] at its real size, against the oracle (a few seconds of CPU)."""
    w, h = 2448, 2048
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=1000, kind="scene")
    run_both(gpu_pipe, oracle, full_chain_cfg(w, h), frame, "bayer_rggb8", 0, what="full chain 2448x2048")


def test_device_batch_equals_per_frame_and_respects_pitch(gpu_pipe, oracle):
    import torch
    w, h, n = 128, 96, 5
    c = full_chain_cfg(w, h)
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, "bayer_rggb8", seed=1000 + i, kind="scene", tint=(0.7 + 0.02 * i, 1.0, 0.55))
                       for i in range(n)])
    dev = torch.from_numpy(frames).cuda()
    out = gpu_pipe.apply_device(dev, "bayer_rggb8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], "bayer_rggb8")
        assert_images_equal(out[i], ref, "batch frame %d" % i)
    # padded rows (pitch > width): view into a wider buffer
    wide = torch.zeros((n, h, w + 32), dtype=torch.uint8, device="cuda")
    wide[:, :, :w] = dev
    out2 = gpu_pipe.apply_device(wide[:, :, :w], "bayer_rggb8")
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), out)


def test_ccc_white_balance_sequence_config3(gpu_pipe, oracle):
    """BASELINE configs[2] scaled down: gbrg8, ccc with temporal consistency, HSV enhancer."""
    w, h, n = 384, 240, 6
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    # One stream on both sides: first the pipeline's degenerate Kalman model (H = 0, what the
    # one-argument constructor leaves behind), then -- after resetWhiteBalanceTemporalConsistency,
    # which only re-arms first_frame_ and keeps the covariance -- loadModel's H = I, R = 10 I.
    occ = oracle.CCC(filt, bias)
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=True, ce=True, ce_sat=1.2)
    configure(gpu_pipe, c)
    seen = []
    for kal in [(0.0, 1.0), (1.0, 10.0)]:
        occ.set_kalman_model(*kal)
        occ.reset()
        gpu_pipe.set_ccc_kalman_model(*kal)
        gpu_pipe.reset_white_balance_temporal_consistency()
        for i in range(n):
            tint = (0.70 + 0.10 * i / (n - 1), 1.0, 0.55)
            frame = synth.gen_frame(w, h, "bayer_gbrg8", seed=2000 + i, kind="scene", tint=tint)
            got = gpu_pipe.process(frame, "bayer_gbrg8")
            ref, _ = oracle_run(oracle, c, frame, "bayer_gbrg8", ccc=occ)
            assert_images_equal(got, ref, "ccc frame %d kalman %s" % (i, kal), TOL_DECLARED)
            seen.append(tuple(int(v) for v in gpu_pipe.get_white_balance_info(1)[0][6:8]))
    assert len(set(seen[:n])) == 1, "H = 0: the estimate must stay frozen at the first frame's argmax"
    assert len(set(seen[n:])) > 1, "H = I: the estimate must follow the drifting tint"


def test_config1_640x480_debayer_and_default_gamma(gpu_pipe, oracle):
    """BASELINE configs[0]: one 640x480 bayer_rggb8 frame, debayer + gamma (method `default`, the same LUT on the
    CPU path) only -- the plumbing case, through the same C-ABI call a reference caller makes."""
    frame = synth.gen_frame(640, 480, "bayer_rggb8", seed=0, kind="scene")
    c = cfg(gamma=True, gamma_method="default", gamma_k=0.8)
    got = run_both(gpu_pipe, oracle, c, frame, "bayer_rggb8", TOL_EXACT, what="config 1")
    assert got.shape == (480, 640, 3) and gpu_pipe.last_encoding == "bgr8"


def test_config3_full_size_1920x1200_ccc_batch(gpu_pipe, oracle):
    """BASELINE configs[2] at its own size: 1920x1200 gbrg8, ccc with the Kalman filter following a drifting tint,
    HSV enhancer; four frames as ONE resident batch (the estimator's sequential Kalman step runs on the device)."""
    import torch
    w, h, n = 1920, 1200, 4
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    occ = oracle.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=True, ce=True, ce_sat=1.2)
    configure(gpu_pipe, c)
    gpu_pipe.reset_white_balance_temporal_consistency()
    frames = np.stack([synth.gen_frame(w, h, "bayer_gbrg8", seed=3000 + i, kind="scene", tint=(0.70 + 0.03 * i, 1.0, 0.55))
                       for i in range(n)])
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_gbrg8")
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], "bayer_gbrg8", ccc=occ)
        assert_images_equal(out[i], ref, "config3 frame %d" % i, TOL_DECLARED)


@pytest.mark.parametrize("size,pattern,flip", [((384, 240), "bayer_gbrg8", 0), ((720, 540), "bayer_rggb8", 180), ((250, 190), "bayer_bggr8", 90),
                                               ((1000, 700), "bayer_grbg8", 180), ((500, 334), "bayer_gbrg8", 180), ((612, 512), "bayer_rggb8", 180)])
def test_ccc_lds_histogram_path(gpu_pipe, oracle, monkeypatch, size, pattern, flip):
    """Batches take the estimator whose histogram is accumulated in LDS (two workgroups per frame, no global atomics, no
    memset, 16-bit counters with wrap repair); small batches the atomic one.  Forced on here for a short batch: every frame equals the oracle, and equals
    what the atomic path gives (bilinear and exact-2x area resize, flipped sampling, generic geometry)."""
    import torch
    w, h = size
    n = 5
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    # temporal consistency off: the filter's covariance survives resetWhiteBalanceTemporalConsistency, so two passes over the
    # same frames would legitimately differ; the estimator itself (histogram -> argmax -> gains) is what is compared
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=False, flip=flip != 0, flip_angle=flip)
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, pattern, seed=7000 + i, kind="scene", tint=(0.62 + 0.05 * i, 1.0, 0.5)) for i in range(n)])
    # last frame: a flat colour, so that (nearly) all 97 200 samples land in ONE bin and its 16-bit LDS counter wraps
    flat = np.empty((h, w), np.uint8)
    for i in range(2):
        for j in range(2):
            flat[i::2, j::2] = (90, 200, 120)[synth.PATTERNS[pattern][i][j]]  # B, G, R
    frames[-1] = flat
    dev = torch.from_numpy(frames).cuda()
    outs = {}
    for mode, lds_min in (("lds", "1"), ("atomic", "1000000")):
        gpu_pipe.set_tunable("ccc_lds_hist_min", int(lds_min))
        outs[mode] = gpu_pipe.apply_device(dev, pattern).cpu().numpy()
        outs[mode + "_uv"] = gpu_pipe.get_white_balance_info(n)[:, 6:8].copy()
    assert np.array_equal(outs["lds"], outs["atomic"]) and np.array_equal(outs["lds_uv"], outs["atomic_uv"])
    assert len({tuple(uv) for uv in outs["lds_uv"]}) > 1, "the drifting tint must move the estimate"
    occ = oracle.CCC(filt, bias)
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], pattern, ccc=occ)
        assert_images_equal(outs["lds"][i], ref, "ccc lds histogram frame %d" % i, TOL_DECLARED)


def test_ccc_histogram_is_handed_back_zeroed_across_path_switches(gpu_pipe, oracle):
    """Small batches accumulate the log-chroma histogram with global atomics into counters the row transforms hand back
    zeroed (one memset for a stream of single frames instead of one per frame), the per-frame argmax is folded into the
    finalisation kernel, and up to four frames run their 256-point transforms on 64 one-wave workgroups per frame.  Everything
    that writes those counters in between must be noticed: a batch on the LDS-histogram path (leaves real counts), SimpleWB
    (its own histograms live in the same buffer), a larger batch (the buffer is reallocated), a change of image size.  Every
    frame equals the oracle's (calculateHistogramFeature / computeResponse, convolutional_color_constancy.cpp:210-298)."""
    import torch
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    occ = oracle.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=False)
    c_simple = cfg(wb=True, wb_method="simple", wb_percentile=10.0)
    seed = [8800]

    def frames_of(n, w, h):
        seed[0] += n
        return np.stack([synth.gen_frame(w, h, "bayer_grbg8", seed=seed[0] + i, kind="scene", tint=(0.55 + 0.04 * ((seed[0] + i) % 9), 1.0, 0.5)) for i in range(n)])

    def check(n, w=384, h=240, resident=True):
        fr = frames_of(n, w, h)
        if resident:
            out = gpu_pipe.apply_device(torch.from_numpy(fr).cuda(), "bayer_grbg8").cpu().numpy()
        else:
            out = np.stack([gpu_pipe.process(f, "bayer_grbg8") for f in fr])
        for i in range(n):
            ref, _ = oracle_run(oracle, c, fr[i], "bayer_grbg8", ccc=occ)
            assert_images_equal(out[i], ref, "ccc, batch of %d (%dx%d), frame %d" % (n, w, h, i), TOL_DECLARED)

    configure(gpu_pipe, c)
    check(1)                    # first use: memset
    check(1)                    # counters handed back zeroed: no memset
    check(3, resident=False)    # three host frames, one after the other
    check(4)                    # transforms on one-wave workgroups, argmax in the finalisation (<= 8 frames)
    check(7)                    # 16-column transforms, argmax still folded
    check(9)                    # own argmax launch
    check(16)                   # LDS-histogram path: the counters stay dirty (and the buffer grew)
    check(1)                    # must clear again
    check(2)
    configure(gpu_pipe, c_simple)   # SimpleWB writes its 3 x 256 bins into the same buffer
    fr = frames_of(2, 384, 240)
    out = gpu_pipe.apply_device(torch.from_numpy(fr).cuda(), "bayer_grbg8").cpu().numpy()
    for i in range(2):
        ref, _ = oracle_run(oracle, c_simple, fr[i], "bayer_grbg8")
        assert_images_equal(out[i], ref, "simple wb between ccc frames, frame %d" % i, TOL_DECLARED)
    configure(gpu_pipe, c)
    check(1)
    check(1, w=720, h=540)      # another geometry (exact 2 x area resize)
    check(2, w=250, h=190, resident=False)


def test_config5_full_size_3840x2160_debayer_undistort(gpu_pipe, oracle):
    """BASELINE configs[4] at its own size: 3840x2160 rggb8, debayer + fisheye undistortion."""
    w, h = 3840, 2160
    c = cfg(undistort=True, cam=synth.camera_model(w, h))
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=5, kind="scene")
    run_both(gpu_pipe, oracle, c, frame, "bayer_rggb8", TOL_INTERP, what="config5 3840x2160")


@pytest.mark.parametrize("pattern", PATTERNS)
@pytest.mark.parametrize("size,angle", [((64, 48), 0), ((37, 29), 180), ((50, 38), 90), ((6, 4), 270), ((3, 3), 0)])
def test_16bit_bayer_extension(gpu_pipe, oracle, pattern, size, angle):
    """rip_set_debayer_16bit: uint16 Bayer -> bgr16 through the C-ABI against the oracle's 16-bit demosaic + flip
    (bit-exact), through the host path and as a resident batch."""
    import torch
    w, h = size
    rng = np.random.default_rng(w * 100 + h + angle)
    frame = rng.integers(0, 65536, (h, w)).astype(np.uint16)
    enc = pattern.replace("8", "16")
    configure(gpu_pipe, cfg(flip=angle != 0, flip_angle=angle))
    gpu_pipe.set_debayer_16bit(True)
    got = gpu_pipe.process(frame, enc)
    ref = oracle.debayer16(frame, enc)
    if angle:
        ref = oracle.flip(ref.view(np.uint8).reshape(h, w, 6), angle)
        ref = np.ascontiguousarray(ref).view(np.uint16).reshape(ref.shape[0], ref.shape[1], 3)
    assert got.dtype == np.uint16 and gpu_pipe.last_encoding == "bgr16"
    assert got.shape == ref.shape and np.array_equal(got, ref)
    batch = torch.from_numpy(np.stack([frame, frame[::-1].copy()]).view(np.uint8).reshape(2, h, w * 2)).cuda()
    # resident batch: uint8 view of the uint16 rows (pitch in bytes), [n, rows, cols * 2] in, [n, R, C * 6] bytes out
    out = torch.empty((2,) + (ref.shape[0], ref.shape[1] * 6), dtype=torch.uint8, device="cuda")
    gpu_pipe._call("rip_apply_device", __import__("ctypes").c_void_p(batch.data_ptr()), __import__("ctypes").c_size_t(w * 2),
                   __import__("ctypes").c_size_t(h * w * 2), 2, h, w, 1, enc.encode(), __import__("ctypes").c_void_p(out.data_ptr()),
                   __import__("ctypes").c_size_t(0), __import__("ctypes").c_size_t(0), None, None)
    torch.cuda.synchronize()
    out0 = out[0].cpu().numpy().view(np.uint16).reshape(ref.shape)
    assert np.array_equal(out0, ref)
    gpu_pipe.set_debayer_16bit(False)
    with pytest.raises(ValueError, match="valid pattern but is not supported"):
        gpu_pipe.process(frame, enc)


def test_device_atan_is_correctly_rounded(gpu_pipe):
    """The map builder's double-double atan (rip_maps.hip) over the range fisheye maps reach (r = tan of the incidence
    angle: 0 .. a few), dense near 0 and 1 and at the table nodes k/16.  It agrees with libm's atan except where libm's is
    not the correctly rounded result (glibc 2.35: about 0.1 % of the arguments are off by one ulp); every disagreement is
    settled by an 80-digit evaluation, and the device value must be the nearest double each time.  A one-ulp difference
    in theta moves a map coordinate by 2^-53 relative, i.e. changes the float map with probability 2^-29 per such pixel:
    test_device_maps_equal_host_and_oracle_maps checks the maps themselves."""
    import os
    import sys
    from decimal import Decimal as D
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        import gen_atan_table as gen
    rng = np.random.default_rng(5)
    r = np.concatenate([rng.uniform(0, 1, 400000), rng.uniform(1, 6, 300000), rng.uniform(0, 1e-3, 50000), 1 + rng.uniform(-1e-6, 1e-6, 50000),
                        np.arange(0, 97) / 16.0, np.arange(1, 97) / 16.0 + 2.0 ** -40, 10.0 ** rng.uniform(-12, 6, 100000),
                        [0.0, 1.0, 1e300]])
    got = gpu_pipe.debug_atan(r)
    want = np.arctan(r)
    ulps = np.abs(got.view(np.int64) - want.view(np.int64))
    assert ulps.max() <= 1
    bad = np.flatnonzero(ulps)
    assert bad.size <= 0.005 * r.size, bad.size
    half_pi = gen.atan_dec(1) * 2
    for i in bad[:: max(1, bad.size // 200)]:
        x = D(float(r[i]))
        true = gen.atan_dec(x) if x <= 1 else half_pi - gen.atan_dec(1 / x)
        assert abs(D(float(got[i])) - true) < abs(D(float(want[i])) - true), (float(r[i]), float(got[i]), float(want[i]))


@pytest.mark.parametrize("size", [(2448, 2048), (1920, 1200), (3840, 2160), (131, 97)])
def test_device_maps_equal_host_and_oracle_maps(gpu_pipe, rip_lib, oracle, monkeypatch, size):
    """undistortion.cpp:212-220 on the device (one thread per map row, FP64, double-double atan) against the host builder
    and the oracle's independent one, at the three BASELINE sizes and an odd one, two cameras each: every map float identical."""
    from raw_image_pipeline_amd import RawImagePipeline
    w, h = size
    for balance, fov in ((0.0, 1.0), (0.6, 1.3)):
        cam = synth.camera_model(w, h)
        synth.load_camera(gpu_pipe, cam)
        gpu_pipe.set_undistortion_balance(balance)
        gpu_pipe.set_undistortion_fov_scale(fov)
        mx, my = gpu_pipe.get_undistortion_maps()
        host = RawImagePipeline(False, "", "", "", device=-1)
        synth.load_camera(host, cam)
        host.set_undistortion_balance(balance)
        host.set_undistortion_fov_scale(fov)
        hx, hy = host.get_undistortion_maps()
        assert np.array_equal(mx.view(np.uint32), hx.view(np.uint32)) and np.array_equal(my.view(np.uint32), hy.view(np.uint32))
        c = cfg(undistort=True, cam=cam, balance=balance, fov_scale=fov)
        from helpers import oracle_maps
        ox, oy = oracle_maps(oracle, c)
        assert np.array_equal(mx.view(np.uint32), ox.view(np.uint32)) and np.array_equal(my.view(np.uint32), oy.view(np.uint32))


def test_error_behaviour(gpu_pipe):
    frame = synth.gen_frame(64, 48)
    configure(gpu_pipe, cfg())
    with pytest.raises(ValueError, match="valid pattern but is not supported"):
        gpu_pipe.process(frame, "bayer_rggb16")  # debayer.cpp:76-78
    configure(gpu_pipe, cfg(wb=True, wb_method="magic"))
    with pytest.raises(ValueError, match="not supported"):
        gpu_pipe.process(frame, "bayer_rggb8")  # white_balance.hpp:82-84
    configure(gpu_pipe, cfg(vig=True))
    with pytest.raises(RipAssertError):
        gpu_pipe.process(frame, "mono8")  # cvtColor(BGR2Lab) on one channel
    configure(gpu_pipe, cfg(wb=True, wb_method="grey_world", cc=True, ce=True))
    out = gpu_pipe.process(frame, "mono8")  # 3-channel-only stages are skipped silently
    assert np.array_equal(out, frame)


# ---- debug stage dumps (setDebug, raw_image_pipeline.hpp:143-186) ------------------------------------
@pytest.mark.parametrize("flip_angle", [180, 90])
def test_debug_stage_dumps_full_chain(rip_lib, oracle, monkeypatch, tmp_path, flip_angle):
    """With setDebug(true) every apply() leaves /tmp/0N_<module>.png (here RIP_DEBUG_DIR): the image after each of the eight
    modules, min-max normalised.  Each must equal the normalised oracle output of the chain cut after that module; the
    returned image is unaffected."""
    from raw_image_pipeline_amd import RawImagePipeline
    from helpers import DUMP_NAMES, normalize_minmax, prefix_cfg, read_png
    monkeypatch.setenv("RIP_DEBUG_DIR", str(tmp_path))
    w, h = 320, 240
    cam = synth.camera_model(h, w) if flip_angle == 90 else synth.camera_model(w, h)  # the remap sees the rotated image
    c = full_chain_cfg(w, h, flip_angle=flip_angle, ce=True, ce_sat=1.2, cam=cam)
    pipe = RawImagePipeline(False, device=0)
    configure(pipe, c)
    frame = synth.gen_frame(w, h, "bayer_grbg8", seed=77, kind="scene")
    pipe.set_debug(True)
    got = pipe.process(frame, "bayer_grbg8")
    ref, _ = oracle_run(oracle, c, frame, "bayer_grbg8")
    assert_images_equal(got, ref, "final image with debug on")
    for k, name in enumerate(DUMP_NAMES):
        want, _ = oracle_run(oracle, prefix_cfg(c, k), frame, "bayer_grbg8")
        assert_images_equal(read_png(str(tmp_path / (name + ".png"))), normalize_minmax(want), "dump %s" % name)
    # debug off again: nothing is written
    for name in DUMP_NAMES:
        (tmp_path / (name + ".png")).unlink()
    pipe.set_debug(False)
    pipe.process(frame, "bayer_grbg8")
    assert not list(tmp_path.iterdir())


def test_debug_stage_dumps_do_not_advance_the_ccc_filter_and_handle_mono(rip_lib, oracle, monkeypatch, tmp_path):
    """The dumps re-run prefixes of the chain with the gains of the real pass: a ccc stream with temporal consistency gives the
    same frames with debug on as the oracle's single pass per frame (no extra Kalman updates), and dump 02 shows those gains.
    A mono8 frame (only the gamma LUT applies) is dumped as grey PNGs."""
    from raw_image_pipeline_amd import RawImagePipeline
    from helpers import DUMP_NAMES, normalize_minmax, prefix_cfg, read_png
    monkeypatch.setenv("RIP_DEBUG_DIR", str(tmp_path))
    w, h, n = 384, 240, 4
    filt, bias = synth.ccc_model()
    pipe = RawImagePipeline(False, device=0)
    pipe.set_ccc_model(filt, bias)
    pipe.set_ccc_kalman_model(1.0, 10.0)
    occ = oracle.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=True, gamma=True, gamma_k=0.9)
    configure(pipe, c)
    pipe.reset_white_balance_temporal_consistency()
    pipe.set_debug(True)
    for i in range(n):
        frame = synth.gen_frame(w, h, "bayer_gbrg8", seed=4100 + i, kind="scene", tint=(0.70 + 0.04 * i, 1.0, 0.55))
        got = pipe.process(frame, "bayer_gbrg8")
        ref, _ = oracle_run(oracle, c, frame, "bayer_gbrg8", ccc=occ)
        assert_images_equal(got, ref, "ccc frame %d with debug on" % i, TOL_DECLARED)
        # dump 07 is the final image, dump 04 too (nothing enabled after gamma), 01 == 00 (flip disabled)
        assert np.array_equal(read_png(str(tmp_path / "07_undistortion.png")), normalize_minmax(got))
        assert np.array_equal(read_png(str(tmp_path / "04_gamma_correction.png")), normalize_minmax(got))
        assert np.array_equal(read_png(str(tmp_path / "01_flip.png")), read_png(str(tmp_path / "00_debayer.png")))
        deb, _ = oracle_run(oracle, prefix_cfg(c, 1), frame, "bayer_gbrg8")
        assert np.array_equal(read_png(str(tmp_path / "00_debayer.png")), normalize_minmax(deb))
    mono = synth.gen_frame(w, h, "bayer_gbrg8", seed=5, kind="scene")  # any single-channel image
    c1 = cfg(gamma=True, gamma_k=0.8)
    configure(pipe, c1)
    got = pipe.process(mono, "mono8")
    ref, _ = oracle_run(oracle, c1, mono, "mono8")
    assert_images_equal(got.reshape(ref.shape), ref, "mono8 with debug on")
    for k, name in enumerate(DUMP_NAMES):
        img = read_png(str(tmp_path / (name + ".png")))
        assert img.ndim == 2
        assert np.array_equal(img, normalize_minmax(ref if k >= 4 else mono).reshape(h, w)), name


@pytest.mark.gpu
def test_submit_to_downloads_into_the_callers_page_locked_arrays(rip_lib, oracle):
    """rip_submit_to: the caller names the destination of the result and of the taps -- page-locked arrays the downloads are
    written into directly; collect() and the image getters return those very arrays (the deep copies the reference's
    process() / getters hand out, raw_image_pipeline.cpp:182-236, without a copy).  Same pixels as the oracle; frames with
    and without destinations interleave; a pageable, a too small or an unkept-tap destination is refused before anything is
    enqueued."""
    from raw_image_pipeline_amd import OutputPool, RawImagePipeline, TAP_COLOR, TAP_DEBAYERED, TAP_PROCESSED
    from raw_image_pipeline_amd.pipeline import host_alloc
    w, h = 640, 480
    pipe = RawImagePipeline(False, "", "", "", device=0)
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="gray_world", cc=True, gamma=True, gamma_k=0.8, undistort=True, cam=synth.camera_model(w, h))
    configure(pipe, c)
    pipe.set_taps(TAP_PROCESSED | TAP_COLOR | TAP_DEBAYERED)
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=9500 + i, kind="scene") for i in range(6)]
    refs = [oracle_run(oracle, c, f, "bayer_rggb8", taps=True) for f in frames]
    pool = OutputPool(limit=12, pinned=True)
    pending = []
    for i, f in enumerate(frames):
        if i % 3 == 2:   # a frame without destinations in between: the handle's own buffers
            pending.append((i, pipe.submit(f, "bayer_rggb8"), None))
        else:
            dst = dict(out=pool.take((h, w, 3)), tap_debayered=pool.take((h, w, 3)), tap_color=pool.take((h, w, 3)) if i % 2 else None)
            pending.append((i, pipe.submit(f, "bayer_rggb8", **{k: v for k, v in dst.items() if v is not None}), dst))
        if len(pending) == 2:
            j, t, dst = pending.pop(0)
            got = pipe.collect(t)
            assert_images_equal(got, refs[j][0], "frame %d result" % j)
            assert_images_equal(pipe.get_dist_debayered_image(), refs[j][2].reshape(h, w, 3), "frame %d debayered tap" % j)
            assert_images_equal(pipe.get_dist_color_image(), refs[j][3].reshape(h, w, 3), "frame %d colour tap" % j)
            if dst is not None:
                assert got is dst["out"] and pipe.get_dist_debayered_image() is dst["tap_debayered"]
                assert (pipe.get_dist_color_image() is dst["tap_color"]) == (dst["tap_color"] is not None)
    for j, t, dst in pending:
        assert_images_equal(pipe.collect(t), refs[j][0], "tail frame %d" % j)
    # a synchronous frame afterwards: the getters belong to it again
    assert_images_equal(pipe.process(frames[0], "bayer_rggb8"), refs[0][0], "process() after the ring")
    assert_images_equal(pipe.get_dist_color_image(), refs[0][3].reshape(h, w, 3), "tap of process()")
    # refusals: nothing is enqueued, the ring stays usable
    with pytest.raises(ValueError, match="page-locked"):
        pipe.submit(frames[0], "bayer_rggb8", out=np.empty((h, w, 3), np.uint8))
    with pytest.raises(RipError, match="too small"):
        pipe.submit(frames[0], "bayer_rggb8", out=host_alloc((h // 2, w, 3)))
    pipe.set_taps(TAP_PROCESSED)
    with pytest.raises(ValueError, match="not kept"):
        pipe.submit(frames[0], "bayer_rggb8", out=host_alloc((h, w, 3)), tap_color=host_alloc((h, w, 3)))
    assert_images_equal(pipe.collect(pipe.submit(frames[1], "bayer_rggb8", out=host_alloc((h, w, 3)))), refs[1][0], "after the refusals")


@pytest.mark.gpu
def test_a_deeper_ring_is_not_slower(rip_lib):
    """More than three host frames enqueued on one device -- a ring deeper than the default, or several handles -- made every
    other 15 MB download take 1.2 ms or more instead of 0.3 (the runtime's handling of SDMA copies behind cross-stream waits),
    and a ring of six ran at 0.76 ms per frame instead of 0.30.  rip_submit therefore lets at most three frames be in flight
    per device (RIP_RING_INFLIGHT); with that a deeper ring costs nothing.  Generous margin: 0.76 against 0.30 is the effect."""
    import time
    from raw_image_pipeline_amd import RawImagePipeline
    from raw_image_pipeline_amd.pipeline import host_alloc
    w, h = 2448, 2048
    frame = host_alloc((h, w))
    frame[...] = synth.gen_frame(w, h, "bayer_rggb8", seed=1, kind="scene")

    def ms_per_frame(depth, n=120):
        pipe = RawImagePipeline(False, "", "", "", device=0)
        synth.configure_full_chain(pipe, w, h, "grey_world")
        pipe.set_ring_depth(depth)
        best = 1e9
        for _ in range(3):
            tickets = []
            for _ in range(2 * depth):
                if len(tickets) == depth:
                    pipe.collect(tickets.pop(0), copy=False)
                tickets.append(pipe.submit(frame, "bayer_rggb8"))
            t0 = time.perf_counter()
            for _ in range(n):
                if len(tickets) == depth:
                    pipe.collect(tickets.pop(0), copy=False)
                tickets.append(pipe.submit(frame, "bayer_rggb8"))
            while tickets:
                pipe.collect(tickets.pop(0), copy=False)
            best = min(best, (time.perf_counter() - t0) / n * 1e3)
        return best

    t3, t6 = ms_per_frame(3), ms_per_frame(6)
    assert t6 <= 1.5 * t3, "ring of 6: %.3f ms per frame, ring of 3: %.3f" % (t6, t3)


@pytest.mark.gpu
def test_three_threads_stream_through_their_own_handles_under_the_in_flight_limit(rip_lib):
    """The device-wide in-flight limit is shared by every handle of the process: three threads, each streaming 40 frames
    through its own handle with a ring of four (so the limit is hit constantly and a thread regularly waits for another
    thread's frame), deep copies through the shared copy pool.  No deadlock, and every result equals the synchronous
    process() of the same frame on a fourth handle."""
    import threading
    from raw_image_pipeline_amd import RawImagePipeline
    w, h, n = 1008, 502, 40
    frames = [synth.gen_frame(w, h, "bayer_gbrg8", seed=9100 + i, kind="scene") for i in range(6)]
    ref_pipe = RawImagePipeline(False, "", "", "", device=0)
    synth.configure_full_chain(ref_pipe, w, h, "grey_world")
    want = [ref_pipe.process(f, "bayer_gbrg8") for f in frames]
    errors = []

    def stream(k):
        try:
            pipe = RawImagePipeline(False, "", "", "", device=0)
            synth.configure_full_chain(pipe, w, h, "grey_world")
            pipe.set_ring_depth(4)
            tickets = []
            for i in range(n):
                if len(tickets) == 4:
                    j, t = tickets.pop(0)
                    got = pipe.collect(t, copy=(i % 2 == 0))
                    if not np.array_equal(got, want[j]):
                        errors.append("thread %d frame %d differs" % (k, j))
                j = (i + k) % len(frames)
                tickets.append((j, pipe.submit(frames[j], "bayer_gbrg8")))
            for j, t in tickets:
                if not np.array_equal(pipe.collect(t), want[j]):
                    errors.append("thread %d tail frame %d differs" % (k, j))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append("thread %d: %r" % (k, e))

    threads = [threading.Thread(target=stream, args=(k,), daemon=True) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a streaming thread did not finish: deadlock under the in-flight limit"
    assert not errors, errors[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 3, 6])
def test_submit_collect_stream_of_32_frames_ccc_temporal(rip_lib, oracle, depth):
    """The asynchronous host path (rip_submit / rip_collect, what a streaming caller like raw_image_pipeline_ros.cpp:219-288
    would use): 32 frames of a ccc + temporal-consistency stream with up to `depth` frames in flight.  Every collected frame
    equals the oracle's, which sees the frames strictly one after the other: the Kalman state advances in submission order
    whatever overlaps.  Also: the ring refuses a frame too many, tickets can be collected out of order, the taps follow the
    collected frame, and the zero-copy view holds the same pixels as the copy."""
    from raw_image_pipeline_amd import RawImagePipeline
    w, h, n = 384, 240, 32
    filt, bias = synth.ccc_model()
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_ccc_model(filt, bias)
    pipe.set_ccc_kalman_model(1.0, 10.0)
    occ = oracle.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    cam = synth.camera_model(w, h)
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=True, cc=True, gamma=True,
            gamma_k=0.8, vig=True, undistort=True, cam=cam)
    configure(pipe, c)
    pipe.reset_white_balance_temporal_consistency()
    pipe.set_ring_depth(depth)
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=700 + i, kind="scene", tint=(0.55 + 0.012 * i, 1.0, 0.62 - 0.008 * i)) for i in range(n)]
    refs = [oracle_run(oracle, c, f, "bayer_rggb8", ccc=occ, taps=True) for f in frames]
    # at most depth - 1 frames in flight while a collected view is still being read (depth 1: strictly one after the other)
    in_flight = max(1, depth - 1)
    tickets, got = [], {}
    for i, f in enumerate(frames):
        if len(tickets) == in_flight:
            # collect the NEWEST first now and then: any outstanding ticket may be collected
            j, t = tickets.pop(-1 if (i % 5 == 0 and in_flight > 1) else 0)
            view = pipe.collect(t, copy=False)
            if depth > 1:  # the view and the taps of the collected frame survive the next submit
                tickets.append((i, pipe.submit(f, "bayer_rggb8")))
            got[j] = view.copy()
            assert_images_equal(pipe.get_dist_color_image(), refs[j][3].reshape(h, w, 3), "tap of collected frame %d" % j)
            if depth > 1:
                continue
        tickets.append((i, pipe.submit(f, "bayer_rggb8")))
    for j, t in tickets:
        got[j] = pipe.collect(t)
    # a full ring refuses one frame more and stays as it is
    extra = [pipe.submit(frames[0], "bayer_rggb8") for _ in range(depth)]
    with pytest.raises(RipError, match="in flight"):
        pipe.submit(frames[0], "bayer_rggb8")
    for t in extra:
        pipe.collect(t)
    with pytest.raises(ValueError):
        pipe.collect(10 ** 9)
    for i in range(n):
        assert_images_equal(got[i], refs[i][0], "submit/collect frame %d (depth %d)" % (i, depth))
    assert pipe.last_encoding == "bgr8"


@pytest.mark.gpu
def test_submit_reads_a_pageable_5_MB_frame_before_it_returns(rip_lib):
    """rip.h promises that a pageable frame is read before rip_submit returns.  A 2448x2048 frame (5 MB: far
    above any staging threshold of the runtime's asynchronous 2-D copy) is overwritten the moment submit() returns, three
    frames in flight from ONE reused buffer; every collected frame -- and the taps that travel with it -- must equal the
    synchronous process() of the original frame.  A frame in pinned memory (host_alloc) takes the copy-free branch."""
    from raw_image_pipeline_amd import RawImagePipeline
    from raw_image_pipeline_amd.pipeline import host_alloc
    w, h = 2448, 2048
    pipe = RawImagePipeline(False, "", "", "", device=0)
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="gray_world", wb_bright=0.8, cc=True, gamma=True, gamma_k=0.8, vig=True,
            undistort=True, cam=synth.camera_model(w, h))
    configure(pipe, c)
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=900 + i, kind="scene", tint=(0.6 + 0.1 * i, 1.0, 0.7)) for i in range(3)]
    want, want_deb, want_col = [], [], []
    for f in frames:
        want.append(pipe.process(f, "bayer_rggb8"))
        want_deb.append(pipe.get_dist_debayered_image())
        want_col.append(pipe.get_dist_color_image())
    assert not np.array_equal(want[0], want[1])
    # without rip_set_tap_download the taps of a collected frame stay on the device: the view getter falls back to a copy
    t = pipe.submit(frames[0], "bayer_rggb8")
    assert_images_equal(pipe.collect(t), want[0], "submit without tap download")
    assert pipe.get_dist_color_image(copy=False).flags.writeable  # a fresh copy, not a view of pinned memory
    assert_images_equal(pipe.get_dist_color_image(copy=False), want_col[0], "tap read from the device after collect")
    pipe.set_tap_download(3)  # TAP_DEBAYERED | TAP_COLOR: from here on they travel with the result
    buf = np.empty((h, w), np.uint8)
    tickets = []
    for i, f in enumerate(frames):
        buf[...] = f
        tickets.append(pipe.submit(buf, "bayer_rggb8"))
        buf[...] = 0xA5 ^ i  # the caller's buffer is the caller's again
    for i, t in enumerate(tickets):
        got = pipe.collect(t, copy=False)
        assert_images_equal(got, want[i], "pageable submit, frame %d" % i)
        deb = pipe.get_dist_debayered_image(copy=False)
        col = pipe.get_dist_color_image(copy=False)
        assert not deb.flags.writeable and not col.flags.writeable  # views of the slot's pinned buffers
        assert_images_equal(deb, want_deb[i], "debayered tap of frame %d" % i)
        assert_images_equal(col, want_col[i], "colour tap of frame %d" % i)
        assert_images_equal(pipe.get_processed_image(), want[i], "processed tap (copy) of frame %d" % i)
    # a pitched pageable frame (row stride > width) goes through the row-by-row staging copy
    wide = np.zeros((h, w + 64), np.uint8)
    wide[:, :w] = frames[1]
    t = pipe.submit(wide[:, :w], "bayer_rggb8")
    wide[...] = 7
    assert_images_equal(pipe.collect(t), want[1], "pitched pageable submit")
    # pinned memory: uploaded from where it lies
    pinned = host_alloc((h, w))
    pinned[...] = frames[2]
    t = pipe.submit(pinned, "bayer_rggb8")
    assert_images_equal(pipe.collect(t), want[2], "pinned submit")
    # frames of the synchronous path have no host view: the getter falls back to the device read
    pipe.process(frames[0], "bayer_rggb8")
    assert_images_equal(pipe.get_dist_color_image(copy=False), want_col[0], "tap of a process() frame through the view getter")


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 48), (132, 36), (640, 480), (1008, 502), (2448, 2048)])
@pytest.mark.parametrize("angle,gamma,undistort", [(0, False, False), (180, True, False), (0, True, True), (180, False, True), (180, True, True)])
def test_mono8_fast_path_flip_gamma_undistortion(gpu_pipe, oracle, size, angle, gamma, undistort):
    """mono8 frames (the reference passes single-channel images through flip, the gamma LUT and undistortion; every colour
    module is skipped or asserts: debayer.cpp:45-79, flip.cpp:45-62, gamma_correction.cpp, undistortion.cpp:240-249) on the
    dword-per-lane chain kernel and the one-channel LDS-DMA ring remap: single frames with taps, then a resident batch."""
    import torch
    w, h = size
    c = cfg(flip=angle != 0, flip_angle=angle, gamma=gamma, gamma_k=0.8, undistort=undistort, cam=synth.camera_model(w, h), balance=0.2, fov_scale=1.1)
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=4200 + i, kind="scene" if i else "uniform") for i in range(3)]  # any one-channel image
    singles = []
    for i, f in enumerate(frames):
        got = run_both(gpu_pipe, oracle, c, f, "mono8", TOL_INTERP, what="mono8 %s flip %d gamma %s undistort %s frame %d" % (size, angle, gamma, undistort, i))
        singles.append(got)
        ref_taps = oracle_run(oracle, c, f, "mono8", taps=True)
        assert_images_equal(gpu_pipe.get_processed_image().reshape(got.shape), got, "processed tap")
        if undistort:  # the pre-undistortion image (dist_image_): flipped and gamma-corrected, not remapped
            assert_images_equal(gpu_pipe.get_dist_color_image().reshape(h, w), np.asarray(ref_taps[3])[:h * w].reshape(h, w), "mono colour tap")
    dev = torch.from_numpy(np.stack([frames[i % 3] for i in range(7)])).cuda()
    out = gpu_pipe.apply_device(dev, "mono8").cpu().numpy()
    for i in range(7):
        assert_images_equal(out[i].reshape(singles[i % 3].shape), singles[i % 3], "mono8 batch frame %d" % i)


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(64, 48), (256, 130), (644, 482), (1008, 502), (2448, 2048)])
@pytest.mark.parametrize("pattern,wb,cc,gamma,angle", [("bayer_rggb8", None, False, False, 0), ("bayer_bggr8", "gray_world", True, True, 0),
                                                       ("bayer_gbrg8", "pca", False, True, 180), ("bayer_grbg8", "simple", True, False, 0),
                                                       ("bayer_rggb8", "ccc", True, True, 180), ("bayer_grbg8", None, False, False, 180)])
def test_chain_inside_the_remap_tiles_equals_the_two_kernel_path(gpu_pipe, oracle, size, pattern, wb, cc, gamma, angle):
    """Memory-rate stage sets with no tap requested run debayer + gains + colour matrix + gamma INSIDE the remap's tiles
    (rip_fused.hip: the Bayer bytes under a tile's source rectangle through the LDS-DMA ring, demosaiced LDS -> LDS, gathered
    from there; undistortion.cpp:240-249 after debayer.cpp:45-79 / white_balance.cpp / color_calibration.cpp:93-103 /
    gamma_correction.cpp).  A resident batch must equal the two-kernel path (remap_fused = 0) and the oracle bit for bit,
    border pixels included; with the colour tap requested, or a Bayer pitch that is no multiple of 16, the fused path must
    step aside."""
    import torch
    w, h = size
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    c = cfg(flip=angle != 0, flip_angle=angle, wb=wb is not None, wb_method=wb or "gray_world", wb_bright=0.8, wb_dark=0.2, wb_temporal=False, cc=cc,
            cc_bias=(1.5, -2.0, 0.5) if cc and wb == "simple" else (0.0, 0.0, 0.0),
            gamma=gamma, gamma_k=0.8, undistort=True, cam=synth.camera_model(w, h), balance=0.3, fov_scale=1.15)
    configure(gpu_pipe, c)
    frames = np.stack([synth.gen_frame(w, h, pattern, seed=5100 + i, kind="uniform" if i == 1 else "scene", tint=(0.6 + 0.07 * i, 1.0, 0.55)) for i in range(5)])
    dev = torch.from_numpy(frames).cuda()
    outs = {}
    for fused in (1, 0):
        gpu_pipe.set_tunable("remap_fused", fused)
        gpu_pipe.profile_begin(64)
        outs[fused] = gpu_pipe.apply_device(dev, pattern).cpu().numpy()
        prof = gpu_pipe.profile_end()
        # the fused path launches no chain kernel at all; it needs a 16-byte-aligned Bayer pitch (644 is not: two kernels)
        assert (prof["chain"][1] == 0) == (fused == 1 and w % 16 == 0), (fused, prof)
    assert np.array_equal(outs[1], outs[0]), "fused and two-kernel results differ on %d values" % int((outs[1] != outs[0]).sum())
    occ = oracle.CCC(filt, bias) if wb == "ccc" else None
    if occ is not None:
        occ.set_kalman_model(1.0, 10.0)
    for i in (0, 1, 4):
        ref, _ = oracle_run(oracle, c, frames[i], pattern, ccc=occ)
        assert_images_equal(outs[1][i], ref, "fused chain + remap, frame %d" % i)
    # a requested tap needs the intermediate image: two kernels again, same result
    gpu_pipe.set_tunable("remap_fused", 1)
    tap = torch.empty((5, h, w, 3), dtype=torch.uint8, device="cuda")
    gpu_pipe.profile_begin(64)
    out_t = gpu_pipe.apply_device(dev, pattern, tap_color=tap).cpu().numpy()
    assert gpu_pipe.profile_end()["chain"][1] > 0
    assert np.array_equal(out_t, outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", ["bgr8", "mono8", "rgb8"])
@pytest.mark.parametrize("size", [(64, 48), (132, 36), (640, 480)])
def test_undistortion_alone_gathers_from_the_input_frames(gpu_pipe, oracle, encoding, size):
    """bgr8 / mono8 frames with nothing enabled but the undistortion (the reference's Python demo on an already debayered
    image): the chain would be a copy, so the remap reads the caller's frames directly (no chain launch).  rgb8 still needs
    its channel swap (debayer.cpp:72-73) and keeps the chain.  Resident batch and host frame against the oracle."""
    import torch
    w, h = size
    rng = np.random.default_rng(77)
    shape = (3, h, w) if encoding == "mono8" else (3, h, w, 3)
    frames = rng.integers(0, 256, shape, dtype=np.uint8)
    c = cfg(undistort=True, cam=synth.camera_model(w, h), balance=0.4, fov_scale=0.9)
    configure(gpu_pipe, c)
    gpu_pipe.profile_begin(16)
    out = gpu_pipe.apply_device(torch.from_numpy(frames).cuda(), encoding).cpu().numpy()
    prof = gpu_pipe.profile_end()
    assert (prof["chain"][1] == 0) == (encoding != "rgb8"), prof
    for i in range(3):
        ref, _ = oracle_run(oracle, c, frames[i], encoding)
        assert_images_equal(out[i].reshape(ref.shape), ref, "%s direct remap frame %d" % (encoding, i))
    run_both(gpu_pipe, oracle, c, frames[0], encoding, TOL_INTERP, what="%s undistortion only, host frame" % encoding)


@pytest.mark.gpu
@pytest.mark.parametrize("encoding,angle,gamma", [("mono8", 180, True), ("mono8", 0, True), ("mono8", 180, False), ("bayer_grbg8", 180, True), ("bayer_rggb8", 0, False)])
@pytest.mark.parametrize("w,pitch", [(1000, 1008), (644, 656), (2448, 2464)])
def test_chain_skipping_paths_on_pitched_resident_frames(gpu_pipe, oracle, encoding, angle, gamma, w, pitch):
    """The paths that run the chain inside the gather (mono8: flip + gamma table in the one-channel ring remap; Bayer: the
    chain inside the remap's tiles) stage source rectangles from 16-byte-aligned columns of the caller's frames: resident
    frames whose row pitch is a multiple of 16 while their width is not (1000 / 1008, 644 / 656) or with padding behind
    every row (2448 / 2464) must give the oracle's pixels -- the rectangle's phase inside its first 16-byte chunk, the mirrored
    addressing under the 180-degree flip and the frame's readable extent all depend on the pitch."""
    import torch
    h = 120 if w != 2448 else 200
    c = cfg(flip=angle != 0, flip_angle=angle, gamma=gamma, gamma_k=0.8, undistort=True, cam=synth.camera_model(w, h), balance=0.3, fov_scale=1.1)
    configure(gpu_pipe, c)
    n = 5
    frames = [synth.gen_frame(w, h, "bayer_rggb8", seed=9700 + i, kind="scene" if i % 2 else "uniform") for i in range(n)]
    padded = torch.full((n, h, pitch), 0xA5, dtype=torch.uint8, device="cuda")
    view = padded[:, :, :w]
    view.copy_(torch.from_numpy(np.stack(frames)).cuda())
    gpu_pipe.profile_begin(16)
    out = gpu_pipe.apply_device(view, encoding).cpu().numpy()
    prof = gpu_pipe.profile_end()
    assert prof["chain"][1] == 0, "a 16-byte-aligned pitch must take the chain-skipping path: %s" % (prof,)
    for i in range(n):
        ref, _ = oracle_run(oracle, c, frames[i], encoding)
        assert_images_equal(out[i].reshape(ref.shape), ref, "%s pitched %d/%d frame %d" % (encoding, w, pitch, i))


@pytest.mark.gpu
def test_ccc_small_batches_in_frame_groups_and_16_bit_submit_to(rip_lib, oracle):
    """Two corners of this round's host / latency changes.  (1) The ccc estimator's zero-hand-back of its histogram counters is
    tracked for unsplit batches only: with the batch cut into frame groups on two streams (overlap_groups) every group clears
    its own counters -- small ccc batches with undistortion, groups 2 and 3, against the oracle.  (2) rip_submit_to with a
    16-bit Bayer frame (the opt-in extension): the result lands in the caller's page-locked uint16 array."""
    import torch
    from raw_image_pipeline_amd import RawImagePipeline
    from raw_image_pipeline_amd.pipeline import host_alloc
    w, h = 384, 240
    filt, bias = synth.ccc_model()
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_ccc_model(filt, bias)
    pipe.set_ccc_kalman_model(1.0, 10.0)
    occ = oracle.CCC(filt, bias)
    occ.set_kalman_model(1.0, 10.0)
    c = cfg(wb=True, wb_method="ccc", wb_bright=0.8, wb_dark=0.2, wb_temporal=False, gamma=True, gamma_k=0.8, vig=True, undistort=True,
            cam=synth.camera_model(w, h))
    configure(pipe, c)
    seed = 9900
    for groups, n in ((2, 5), (3, 7), (1, 2), (2, 4), (1, 1)):
        pipe.set_tunable("overlap_groups", groups)
        frames = np.stack([synth.gen_frame(w, h, "bayer_bggr8", seed=seed + i, kind="scene", tint=(0.6 + 0.03 * i, 1.0, 0.55)) for i in range(n)])
        seed += n
        out = pipe.apply_device(torch.from_numpy(frames).cuda(), "bayer_bggr8").cpu().numpy()
        for i in range(n):
            ref, _ = oracle_run(oracle, c, frames[i], "bayer_bggr8", ccc=occ)
            assert_images_equal(out[i], ref, "ccc, %d frames in %d groups, frame %d" % (n, groups, i), TOL_DECLARED)
    # (2)
    p16 = RawImagePipeline(False, "", "", "", device=0)
    configure(p16, cfg())
    p16.set_debayer_16bit(True)
    f16 = (synth.gen_frame(w, h, "bayer_rggb8", seed=5, kind="scene").astype(np.uint16) * 257)
    want = p16.process(f16, "bayer_rggb16")
    dst = host_alloc((h, w, 3), np.uint16)
    got = p16.collect(p16.submit(f16, "bayer_rggb16", out=dst))
    assert got is dst and got.dtype == np.uint16 and np.array_equal(got, want)
