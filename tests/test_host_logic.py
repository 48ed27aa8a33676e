"""Host-side logic of the product library (parameter handling, YAML loaders, constant tables, fisheye
maps, remap plan) through a RIP_DEVICE_NONE handle -- and its agreement with the oracle's independent
implementation of the same host computations."""
import os

import numpy as np
import pytest

from raw_image_pipeline_amd import RawImagePipeline, RipIOError, synth

CFG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs")


def test_four_argument_constructor_defaults(host_pipe):
    p = host_pipe  # empty paths: the values of the reference's example files (raw_image_pipeline.cpp:23-40)
    assert p.is_debayer_enabled() and not p.is_flip_enabled() and p.is_white_balance_enabled()
    assert not p.is_color_calibration_enabled() and not p.is_gamma_correction_enabled()
    assert not p.is_vignetting_correction_enabled() and not p.is_color_enhancer_enabled() and p.is_undistortion_enabled()
    m = p.get_color_calibration_matrix()
    assert m.shape == (3, 3) and abs(m[0, 0] - 2.4276948) < 1e-6 and abs(m[2, 2] - 2.099912) < 1e-6
    assert p.get_color_calibration_bias().ravel().tolist() == [0, 0, 0, 0]
    # no camera calibration was loaded
    assert p.get_dist_distortion_model() == "none" and p.get_rect_distortion_model() == "none"
    assert (p.get_dist_image_width(), p.get_dist_image_height()) == (320, 240)


def test_one_argument_constructor_loads_example_calibration(rip_lib):
    p = RawImagePipeline(False, device=-1)  # RawImagePipeline(bool use_gpu), raw_image_pipeline.cpp:16-21
    assert (p.get_dist_image_width(), p.get_dist_image_height()) == (720, 540)
    assert p.get_dist_distortion_model() == "equidistant"
    assert p.get_rect_distortion_model() == "none"  # undistortion enabled -> the published image is rectified
    p.set_undistortion(False)
    assert p.get_rect_distortion_model() == "equidistant"
    K = p.get_dist_camera_matrix()
    assert abs(K[0, 0] - 347.548139773951) < 1e-9 and abs(K[1, 2] - 271.368057185649) < 1e-9
    assert p.get_rect_distortion_coefficients().ravel().tolist() == [0, 0, 0, 0]
    assert np.array_equal(p.get_rect_rectification_matrix(), np.eye(3))
    assert np.array_equal(p.get_rect_projection_matrix()[:, :3], p.get_rect_camera_matrix())


def test_load_params_full_file(host_pipe):
    p = host_pipe
    p.load_params(os.path.join(CFG, "params_full.yaml"))
    assert p.is_flip_enabled() and p.is_white_balance_enabled() and p.is_color_calibration_enabled()
    assert p.is_gamma_correction_enabled() and p.is_vignetting_correction_enabled() and p.is_color_enhancer_enabled()
    assert p.is_undistortion_enabled()
    assert p.get_table(8)[128] == round(255 * (128 / 255) ** 0.9)  # gamma k = 0.9 reached the LUT builder


def test_load_params_sparse_file_takes_loader_defaults(host_pipe):
    p = host_pipe
    p.set_gamma_correction(True)
    p.load_params(os.path.join(CFG, "params_sparse.yaml"))
    # raw_image_pipeline.cpp:54-160 defaults: debayer on, everything else off
    assert p.is_debayer_enabled() and not p.is_flip_enabled() and not p.is_white_balance_enabled()
    assert not p.is_gamma_correction_enabled() and not p.is_undistortion_enabled()
    assert not p.is_color_enhancer_enabled()  # the reference reads `run_color_enhancer`, not `enabled` (:137)
    assert p.query_output(48, 64, 1, "bayer_rggb8")[:2] == (48, 64)  # flip disabled although angle = 90


def test_missing_files_are_soft_failures(host_pipe):
    p = host_pipe
    p.load_params("/nonexistent/params.yaml")               # "Warning: parameters file doesn't exist"
    p.load_color_calibration("/nonexistent/color.yaml")     # calibration_available_ = false
    p.load_camera_calibration("/nonexistent/calib.yaml")    # falls back to 320x240 / identity / "none"
    assert (p.get_dist_image_width(), p.get_dist_image_height()) == (320, 240)
    assert p.get_dist_distortion_model() == "none"


def test_malformed_yaml_raises(host_pipe):
    with pytest.raises(RipIOError):
        host_pipe.load_camera_calibration(os.path.join(CFG, "malformed.yaml"))


def test_camera_calibration_file_and_rect_parameters(host_pipe, oracle):
    p = host_pipe
    p.load_camera_calibration(os.path.join(CFG, "calib_64x48.yaml"))
    p.set_undistortion_balance(0.5)
    p.set_undistortion_fov_scale(1.2)
    assert (p.get_dist_image_width(), p.get_dist_image_height()) == (64, 48)
    assert p.get_dist_distortion_model() == "equidistant"
    K, D = p.get_dist_camera_matrix(), p.get_dist_distortion_coefficients().ravel()
    assert abs(K[0, 2] - 31.63584) < 1e-12 and abs(D[3] - 0.0026955514) < 1e-15
    newK = oracle.fisheye_new_camera_matrix(K, D, (64, 48), np.eye(3), 0.5, None, 1.2)
    assert np.array_equal(p.get_rect_camera_matrix(), newK)
    mx, my = p.get_undistortion_maps()
    omx, omy = oracle.fisheye_maps(K, D, np.eye(3), newK, (64, 48))
    assert np.array_equal(mx, omx) and np.array_equal(my, omy)
    # setNewImageSize rescales the rect intrinsics but the maps keep the dist size (undistortion.cpp:216)
    p.set_undistortion_new_image_size(32, 24)
    assert (p.get_rect_image_width(), p.get_rect_image_height()) == (32, 24)
    assert p.get_undistortion_maps()[0].shape == (48, 64)
    assert np.allclose(p.get_rect_camera_matrix()[0, 0], newK[0, 0] * 0.5)


def test_color_calibration_file(host_pipe):
    p = host_pipe
    p.load_color_calibration(os.path.join(CFG, "color_calib.yaml"))
    assert p.get_color_calibration_matrix().ravel().tolist() == [1.5, -0.25, 0.0, 0.125, 1.0, -0.125, 0.0, -0.5, 1.75]
    assert p.get_color_calibration_bias().ravel().tolist() == [1.0, -2.0, 3.5, 0.0]
    p.set_color_calibration_matrix([0.1] * 9)  # stored as Matx33f (color_calibration.cpp:79)
    assert p.get_color_calibration_matrix()[0, 0] == float(np.float32(0.1))
    with pytest.raises(ValueError):
        p.set_color_calibration_matrix([1.0] * 8)


def test_tables_agree_with_oracle(host_pipe, oracle):
    for i, name in enumerate(["srgb_gamma", "cbrt", "lab_to_yf", "inv_gamma", "fwd_coeffs", "inv_coeffs", "sdiv", "hdiv180"]):
        assert np.array_equal(host_pipe.get_table(i), oracle.table(name)), name
    for k in (0.5, 0.8, 1.0, 1.7):
        host_pipe.set_gamma_correction_k(k)
        assert np.array_equal(host_pipe.get_table(8), oracle.gamma_lut(k))


def test_query_output_and_encoding_rules(host_pipe):
    p = host_pipe
    p.set_undistortion(False)
    assert p.query_output(48, 64, 1, "bayer_gbrg8") == (48, 64, 3, "bgr8")
    assert p.query_output(48, 64, 3, "rgb8") == (48, 64, 3, "rgb8")     # string kept on the CPU path
    assert p.query_output(48, 64, 1, "mono8") == (48, 64, 1, "mono8")
    p.set_flip(True)
    p.set_flip_angle(270)
    assert p.query_output(48, 64, 1, "bayer_gbrg8") == (64, 48, 3, "bgr8")
    p.set_flip_angle(33)                                                # unknown angles do nothing
    assert p.query_output(48, 64, 1, "bayer_gbrg8")[:2] == (48, 64)
    with pytest.raises(ValueError, match="valid pattern but is not supported"):
        p.query_output(48, 64, 1, "bayer_bggr16")   # including the name the reference's missing comma lets through
    with pytest.raises(ValueError, match="valid pattern but is not supported"):
        p.query_output(48, 64, 1, "bayer_rggb16")
    p.set_white_balance(True)
    p.set_white_balance_method("learned")
    with pytest.raises(ValueError, match="not implemented"):
        p.query_output(48, 64, 1, "bayer_gbrg8")
    p.set_white_balance_method("nonsense")
    with pytest.raises(ValueError, match="not supported"):
        p.query_output(48, 64, 1, "bayer_gbrg8")
    assert p.query_output(48, 64, 1, "mono8")[3] == "mono8"             # wb is skipped on one channel
    # undistortion: output takes the map (dist) size whatever the input size
    p.set_white_balance(False)
    p.set_flip(False)
    synth.load_camera(p, synth.camera_model(64, 48))
    p.set_undistortion(True)
    assert p.query_output(40, 56, 1, "mono8")[:2] == (48, 64)
    p.set_undistortion_distortion_model("none")
    assert p.query_output(40, 56, 1, "mono8")[:2] == (40, 56)           # model "none": no remap


def test_remap_plan_words_reproduce_the_quantised_map(host_pipe, oracle):
    """The compiled plan must encode exactly cv::remap's quantisation of the float map."""
    import ctypes as C
    p = host_pipe
    cam = synth.camera_model(128, 96)
    synth.load_camera(p, cam)
    mx, my = p.get_undistortion_maps()
    sxq = np.rint(mx.astype(np.float64) * 32).astype(np.int64)
    syq = np.rint(my.astype(np.float64) * 32).astype(np.int64)
    ix, iy = sxq >> 5, syq >> 5
    inside = (ix >= 0) & (ix < 127) & (iy >= 0) & (iy < 95)
    assert inside.mean() > 0.5
    # remapping through the oracle with the float maps equals remapping with the re-built quantised coordinates
    img = np.random.default_rng(0).integers(0, 256, (96, 128, 3), dtype=np.uint8)
    a = oracle.remap(img, mx, my)
    b = oracle.remap(img, (sxq / 32.0).astype(np.float32), (syq / 32.0).astype(np.float32))
    assert np.array_equal(a, b)


REF_WB = "/root/reference/raw_image_pipeline_white_balance"


@pytest.mark.skipif(not os.path.exists(REF_WB + "/model/default.bin"), reason="reference tree not mounted (GPU box)")
def test_reference_ccc_model_file_loads_and_balances_the_reference_sample_image(oracle):
    """The reference's own model file (default.bin: int w, int h, float filter[w*h], float bias[w*h],
    convolutional_color_constancy.cpp:116-130) goes through rip_load_ccc_model, and the oracle's estimator with that
    model pulls the reference's sample image (data/alphasense.png, a greenish raw-looking frame) towards grey.
    Runs only where the reference tree is mounted; nothing is copied from it."""
    from PIL import Image
    raw = np.fromfile(REF_WB + "/model/default.bin", dtype=np.uint8)
    w, h = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
    assert (w, h) == (256, 256) and raw.size == 8 + 2 * 4 * w * h
    filt = np.frombuffer(raw[8:8 + 4 * w * h].tobytes(), dtype=np.float32).reshape(h, w)
    bias = np.frombuffer(raw[8 + 4 * w * h:].tobytes(), dtype=np.float32).reshape(h, w)
    assert np.isfinite(filt).all() and np.isfinite(bias).all() and filt.std() > 0
    p = RawImagePipeline(False, device=-1)
    p.load_ccc_model(REF_WB + "/model/default.bin")          # host-side parse + spectrum build
    with pytest.raises(Exception):
        p.load_ccc_model(REF_WB + "/model/missing.bin")
    img = np.asarray(Image.open(REF_WB + "/data/alphasense.png").convert("RGB"))[..., ::-1].copy()  # BGR
    ccc = oracle.CCC(filt, bias)
    out, info, gains = ccc.balance(img)
    assert all(np.isfinite(g) and 0.2 < g < 5.0 for g in gains), gains
    spread = lambda im: np.ptp(im.reshape(-1, 3).mean(axis=0))
    assert spread(out) < spread(img), (spread(img), spread(out), gains)  # channel means move together


def test_yaml_flow_maps_and_any_key_order(tmp_path):
    """yaml-cpp accepts flow maps and any key order; so does the subset reader (the fixtures under tests/golden/configs
    use both on purpose)."""
    f = tmp_path / "cc.yaml"
    f.write_text("bias: {rows: 3, cols: 1,\n       data: [0.5, -1.5, 2.0]}   # continues on the next line\n"
                 "matrix: {data: [1, 0, 0, 0, 2, 0, 0, 0, 3], 'cols': 3, \"rows\": 3}\n")
    p = RawImagePipeline(False, device=-1)
    p.load_color_calibration(str(f))
    assert np.allclose(p.get_color_calibration_matrix() if hasattr(p, "get_color_calibration_matrix") else np.diag([1, 2, 3]), np.diag([1, 2, 3]))
    bad = tmp_path / "bad.yaml"
    bad.write_text("matrix: {rows: 3, cols: 3, data: [1, 0, 0\n")
    with pytest.raises(RipIOError):
        p.load_color_calibration(str(bad))


@pytest.mark.parametrize("rows,cols", [(2048, 2448), (2160, 3840), (1200, 1920), (480, 640), (97, 131), (64, 64), (3, 5)])
def test_vignetting_mask_plane_equals_the_reference_formula_bit_for_bit(host_pipe, oracle, rows, cols):
    """The product's mask plane (host, quadrant-mirrored) against the oracle's literal restatement of
    precomputeVignettingMask (vignetting_correction.cpp:32-63: sqrt, pow(r, 2), pow(r, 4), /max, *scale, +1):
    every float identical, including the sizes where the algebraically equal s*a2 + s^2*a4 differs (2448x2048:
    ~830 floats, 3840x2160: L values flip)."""
    for scale, a2, a4 in ((1.5, 1e-3, 1e-6), (0.7, 3.3e-4, -2.0e-7), (1.0, 0.0, 0.0)):
        host_pipe.set_vignetting_correction_parameters(scale, a2, a4)
        got = host_pipe.get_vignetting_mask(rows, cols)
        want = oracle.vignetting_mask(rows, cols, scale, a2, a4)
        assert got.shape == want.shape == (rows, cols)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_load_params_recreates_the_modules_like_the_reference(host_pipe):
    """RawImagePipeline::loadParams (raw_image_pipeline.cpp:44-165) builds every module anew when the file exists: a colour
    or camera calibration loaded before is gone afterwards (the constructors reload them after loadParams, :27-29); a
    missing params file changes nothing."""
    p = host_pipe
    p.load_color_calibration(os.path.join(CFG, "color_calib.yaml"))
    p.load_camera_calibration(os.path.join(CFG, "calib_64x48.yaml"))
    m_before = p.get_color_calibration_matrix().copy()
    assert not np.array_equal(m_before, np.eye(3))
    p.set_undistortion(True)
    assert p.query_output(40, 56, 1, "mono8")[:2] == (48, 64)      # remap to the calibration's size
    p.load_params("/nonexistent/params.yaml")                       # soft failure: nothing is re-created
    assert np.array_equal(p.get_color_calibration_matrix(), m_before)
    assert p.query_output(40, 56, 1, "mono8")[:2] == (48, 64)
    p.load_params(os.path.join(CFG, "params_full.yaml"))
    assert np.array_equal(p.get_color_calibration_matrix(), np.eye(3))
    assert p.is_undistortion_enabled() and p.query_output(40, 56, 3, "bgr8")[:2] == (40, 56)    # no calibration any more


def test_16bit_bayer_extension_is_opt_in(host_pipe):
    """bayer_*16 raise exactly like the reference (debayer.cpp:76-78) unless rip_set_debayer_16bit(1) opts into the extension;
    with it the frame takes debayer + flip only and comes out as bgr16; any 8-bit stage left enabled is an error."""
    p = host_pipe
    p.set_white_balance(False)
    p.set_undistortion(False)
    with pytest.raises(ValueError, match="valid pattern but is not supported"):
        p.query_output(48, 64, 1, "bayer_gbrg16")
    p.set_debayer_16bit(True)
    assert p.query_output(48, 64, 1, "bayer_gbrg16") == (48, 64, 3, "bgr16")
    p.set_flip(True)
    p.set_flip_angle(90)
    assert p.query_output(48, 64, 1, "bayer_rggb16") == (64, 48, 3, "bgr16")
    p.set_gamma_correction(True)
    with pytest.raises(Exception, match="8-bit stages"):
        p.query_output(48, 64, 1, "bayer_rggb16")
    assert p.query_output(48, 64, 1, "bayer_rggb8") == (64, 48, 3, "bgr8")  # 8-bit frames are unaffected


def test_debug_dump_normalisation_and_png_writer(host_pipe, tmp_path):
    """saveDebugImage (raw_image_pipeline.hpp:179-186): cv::normalize(0, 255, NORM_MINMAX) over all channels, then a PNG that
    a standard decoder reads back (RGB order in the file, BGR in memory)."""
    from helpers import normalize_minmax, read_png
    rng = np.random.default_rng(11)
    bgr = rng.integers(17, 204, (37, 53, 3), dtype=np.uint8)
    bgr[0, 0] = (17, 99, 203)  # both extremes present
    path = str(tmp_path / "dump.png")
    host_pipe.debug_write_png(path, bgr, normalize=False)
    assert np.array_equal(read_png(path), bgr)
    host_pipe.debug_write_png(path, bgr, normalize=True)
    got = read_png(path)
    assert np.array_equal(got, normalize_minmax(bgr)) and got.min() == 0 and got.max() == 255
    # known answers: 17 -> 0, 203 -> 255, 110 -> round(93 * (255 / 186)) = round(127.5) -> half to even = 128
    lut = normalize_minmax(np.array([[17, 110, 203]], np.uint8))
    assert lut.tolist() == [[0, 128, 255]]
    # a flat image has no range: scale 0, everything becomes 0
    flat = np.full((9, 5), 77, np.uint8)
    host_pipe.debug_write_png(path, flat, normalize=True)
    assert np.array_equal(read_png(path), np.zeros_like(flat))
    # grey, more than one stored deflate block (> 65535 bytes) and a row length that is not a multiple of anything
    grey = rng.integers(0, 256, (301, 333), dtype=np.uint8)
    host_pipe.debug_write_png(path, grey, normalize=False)
    assert np.array_equal(read_png(path), grey)
    with pytest.raises(RipIOError):
        host_pipe.debug_write_png(str(tmp_path / "no_such_dir" / "x.png"), grey)


def test_bench_without_a_gpu_exits_loudly_and_prints_no_line():
    """bench.py on a host without an MI355X (this container): non-zero exit, a message, no JSON line -- whatever --gpus says.
    (The product path has no CPU execution; a silent number from anything else would be worse than none.)"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gpus in ("1", "2"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "MI355X" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_output_pool_recycles_only_arrays_nobody_references():
    """OutputPool (the destination arrays of the front end's deep copies): an array is handed out again only when the pool's
    own list is its last reference -- a caller's variable, a message holding it, a VIEW of it (its base) all keep it out."""
    from raw_image_pipeline_amd import OutputPool
    pool = OutputPool(limit=3)
    a = pool.take((4, 5, 3))
    b = pool.take((4, 5, 3))
    assert b is not a and a.shape == (4, 5, 3) and a.dtype == np.uint8
    a_id = id(a)
    held = {"image": b}          # e.g. a published message
    del b
    view = a[::2]                # a view keeps its base alive
    del a
    c = pool.take((4, 5, 3))
    assert not np.shares_memory(c, view) and not np.shares_memory(c, held["image"])
    del view
    d = pool.take((4, 5, 3))     # the first array is free again
    assert id(d) == a_id
    other = pool.take((2, 2), np.uint16)
    assert other.dtype == np.uint16 and other.shape == (2, 2)
    many = [pool.take((4, 5, 3)) for _ in range(6)]   # more than the limit in use at once: plain allocations, never shared
    assert len({id(m) for m in many}) == 6
    assert len(pool._arrays[((4, 5, 3), np.dtype(np.uint8).str)]) <= 3


def test_pinned_output_pool_counts_views_too(monkeypatch):
    """a page-locked pool array does not own its memory, and numpy collapses the base of a view of a
    plain non-owning array onto the hidden buffer object -- the pool then saw no reference and recycled memory a caller's
    slice still looked at (and rip_submit_to would DMA the next frame into it).  host_alloc now builds its array directly on the
    buffer object: views of it reference it.  rip_host_alloc is replaced by ordinary memory here (no GPU); the wrapping code is the product's."""
    import ctypes
    from raw_image_pipeline_amd import OutputPool, pipeline
    monkeypatch.setattr(pipeline, "host_alloc",
                        lambda shape, dtype=np.uint8: pipeline._wrap_pinned((ctypes.c_uint8 * (int(np.prod(shape)) * np.dtype(dtype).itemsize))(), shape, dtype))
    pool = OutputPool(limit=4, pinned=True)
    for make_view in (lambda a: a[1:3], lambda a: a.reshape(-1), lambda a: a[..., ::-1], lambda a: np.asarray(a)[0], lambda a: a.view(np.ndarray)):
        a = pool.take((4, 5, 3))
        assert not a.flags.owndata and not isinstance(a.base, np.ndarray)
        a[...] = 7
        view = make_view(a)
        del a
        b = pool.take((4, 5, 3))
        assert not np.shares_memory(b, view), "the pool recycled an array whose view is still alive"
        b[...] = 9
        assert (view == 7).all()
        del view, b
    # once nothing references them the same arrays come back (no growth)
    n = len(pool._arrays[((4, 5, 3), np.dtype(np.uint8).str)])
    c = pool.take((4, 5, 3))
    assert len(pool._arrays[((4, 5, 3), np.dtype(np.uint8).str)]) == n
    del c
