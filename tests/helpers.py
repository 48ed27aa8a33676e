"""Shared test helpers: one stage configuration applied both to the HIP pipeline (through the public
setters, as a reference caller would) and to the CPU oracle."""
import numpy as np

from raw_image_pipeline_amd import synth

DEFAULTS = dict(
    flip=False, flip_angle=0,
    wb=False, wb_method="grey_world", wb_bright=0.8, wb_dark=0.2, wb_temporal=False, wb_percentile=10.0,
    cc=False, cc_matrix=None, cc_bias=(0.0, 0.0, 0.0),
    gamma=False, gamma_k=0.8, gamma_method="custom",
    vig=False, vig_params=(1.5, 1e-3, 1e-6),
    ce=False, ce_hue=1.0, ce_sat=1.0, ce_val=1.0,   # as passed to the public (cross-wired) setters
    undistort=False, cam=None, balance=0.0, fov_scale=1.0,
)


def cfg(**kw):
    c = dict(DEFAULTS)
    c.update(kw)
    if c["cc_matrix"] is None:
        c["cc_matrix"] = list(synth.COLOR_MATRIX)
    return c


def configure(pipe, c):
    pipe.set_debayer(True)
    pipe.set_flip(c["flip"])
    pipe.set_flip_angle(c["flip_angle"])
    pipe.set_white_balance(c["wb"])
    pipe.set_white_balance_method(c["wb_method"])
    pipe.set_white_balance_saturation_threshold(c["wb_bright"], c["wb_dark"])
    pipe.set_white_balance_percentile(c["wb_percentile"])
    pipe.set_white_balance_temporal_consistency(c["wb_temporal"])
    pipe.set_color_calibration(c["cc"])
    pipe.set_color_calibration_matrix(c["cc_matrix"])
    pipe.set_color_calibration_bias(list(c["cc_bias"]))
    pipe.set_gamma_correction(c["gamma"])
    pipe.set_gamma_correction_method(c["gamma_method"])
    pipe.set_gamma_correction_k(c["gamma_k"])
    pipe.set_vignetting_correction(c["vig"])
    pipe.set_vignetting_correction_parameters(*c["vig_params"])
    pipe.set_color_enhancer(c["ce"])
    pipe.set_color_enhancer_hue_gain(c["ce_hue"])
    pipe.set_color_enhancer_saturation_gain(c["ce_sat"])
    pipe.set_color_enhancer_value_gain(c["ce_val"])
    pipe.set_undistortion(c["undistort"])
    if c["cam"] is not None:
        synth.load_camera(pipe, c["cam"])
        pipe.set_undistortion_balance(c["balance"])
        pipe.set_undistortion_fov_scale(c["fov_scale"])


def oracle_maps(O, c):
    cam = c["cam"]
    size = (cam["width"], cam["height"])
    newK = O.fisheye_new_camera_matrix(cam["K"], cam["D"], size, cam["R"], c["balance"], None, c["fov_scale"])
    return O.fisheye_maps(cam["K"], cam["D"], cam["R"], newK, size)


def oracle_params(O, c, keep):
    p = O.Params()
    p.flip_enabled, p.flip_angle = int(c["flip"]), int(c["flip_angle"])
    p.wb_enabled = int(c["wb"])
    p.wb_method = O.WB_METHODS.get(c["wb_method"], -1)
    p.wb_bright_thr, p.wb_dark_thr = c["wb_bright"], c["wb_dark"]
    p.wb_percentile = c["wb_percentile"]
    p.wb_temporal_consistency = int(c["wb_temporal"])
    p.cc_enabled, p.cc_available = int(c["cc"]), 1
    m32 = np.asarray(c["cc_matrix"], np.float64)
    for i in range(9):
        p.cc_matrix[i] = float(m32[i])
    for i in range(3):
        p.cc_bias[i] = float(c["cc_bias"][i])
    p.gamma_enabled, p.gamma_k = int(c["gamma"]), c["gamma_k"]
    p.vig_enabled = int(c["vig"])
    p.vig_scale, p.vig_a2, p.vig_a4 = c["vig_params"]
    p.ce_enabled = int(c["ce"])
    # public setters are cross-wired (color_enhancer.cpp:23-33): hue param -> V, value param -> H
    p.ce_h_gain, p.ce_s_gain, p.ce_v_gain = c["ce_val"], c["ce_sat"], c["ce_hue"]
    p.und_enabled = int(bool(c["undistort"] and c["cam"] is not None))
    if p.und_enabled:
        mx, my = oracle_maps(O, c)
        keep.extend([mx, my])
        p.map_x, p.map_y = mx.ctypes.data, my.ctypes.data
        p.map_rows, p.map_cols = mx.shape
    return p


def oracle_run(O, c, frame, encoding, ccc=None, taps=False):
    keep = []
    prm = oracle_params(O, c, keep)
    return O.pipeline(prm, frame, encoding, ccc=ccc, taps=taps)


def assert_images_equal(got, ref, what="", tol=0):
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, got.shape, ref.shape)
    d = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= tol, "%s: max |diff| = %d on %d of %d values" % (what, d.max(), int((d > tol).sum()), d.size)


# ---- debug stage dumps (raw_image_pipeline.hpp:143-186) ---------------------------------------------
DUMP_NAMES = ("00_debayer", "01_flip", "02_white_balancing", "03_color_calibration", "04_gamma_correction",
              "05_vignetting_correction", "06_color_enhancer", "07_undistortion")
# configuration keys of the modules that run AFTER dump k was taken
_LATER = (("flip", "wb", "cc", "gamma", "vig", "ce", "undistort"), ("wb", "cc", "gamma", "vig", "ce", "undistort"),
          ("cc", "gamma", "vig", "ce", "undistort"), ("gamma", "vig", "ce", "undistort"), ("vig", "ce", "undistort"),
          ("ce", "undistort"), ("undistort",), ())


def prefix_cfg(c, k):
    """The configuration whose output is the image after module k of configuration c."""
    s = dict(c)
    for key in _LATER[k]:
        s[key] = False
    return s


def normalize_minmax(img):
    """cv::normalize(src, dst, 0, 255, NORM_MINMAX) for 8-bit data, restated: scale and shift in double, applied in float
    (multiply, then add), rounded half to even, saturated."""
    a = np.asarray(img, np.uint8)
    smin, smax = float(a.min()), float(a.max())
    scale = 255.0 * (1.0 / (smax - smin) if smax - smin > np.finfo(np.float64).eps else 0.0)
    shift = 0.0 - smin * scale
    r = a.astype(np.float32) * np.float32(scale)
    r = r + np.float32(shift)
    return np.clip(np.rint(r), 0, 255).astype(np.uint8)


def read_png(path):
    """8-bit grey / RGB PNG -> H x W or H x W x 3 **BGR** array.  Pillow decodes when it is installed (an independent
    decoder: it checks the chunk CRCs and the zlib stream); otherwise a minimal reader for non-interlaced files."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            im.load()
            a = np.asarray(im)
        assert a.dtype == np.uint8
        return a if a.ndim == 2 else a[:, :, ::-1].copy()
    except ImportError:
        pass
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(typ + body) & 0xFFFFFFFF)
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and ctype in (0, 2) and interlace == 0
    cn = 3 if ctype == 2 else 1
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w * cn + 1)
    assert not rows[:, 0].any(), "only filter type 0 is handled"
    a = rows[:, 1:].reshape(h, w, cn)
    return a[:, :, 0].copy() if cn == 1 else a[:, :, ::-1].copy()
