"""ctypes binding of the CPU oracle (oracle/rip_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by raw_image_pipeline_amd.  PARITY UNPINNED (see rip_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RIP_ORACLE_ASAN=1: the sanitizer build of the oracle (`make -C oracle asan`, gcc's ASan + UBSan; the process needs
# LD_PRELOAD of GCC's shared ASan runtime -- tools/run_asan.sh sets both)
_ASAN = os.environ.get("RIP_ORACLE_ASAN", "") == "1"
_SO = os.path.join(_HERE, "_build", "librip_oracle_asan.so" if _ASAN else "librip_oracle.so")

BAYER = {"bayer_rggb8": 0, "bayer_grbg8": 1, "bayer_gbrg8": 2, "bayer_bggr8": 3}
WB_METHODS = {"simple": 0, "grey_world": 1, "gray_world": 1, "learned": 2, "ccc": 3, "pca": 4}


def build(force=False):
    """Compile the oracle with gcc (seconds).  Safe to call repeatedly."""
    src = os.path.join(_HERE, "rip_oracle.c")
    hdr = os.path.join(_HERE, "rip_oracle.h")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-B"] + (["asan"] if _ASAN else []), check=True, capture_output=True)
    return _SO


class Params(C.Structure):
    _fields_ = [
        ("flip_enabled", C.c_int), ("flip_angle", C.c_int),
        ("wb_enabled", C.c_int), ("wb_method", C.c_int),
        ("wb_bright_thr", C.c_double), ("wb_dark_thr", C.c_double),
        ("wb_percentile", C.c_double),
        ("wb_temporal_consistency", C.c_int),
        ("cc_enabled", C.c_int), ("cc_available", C.c_int),
        ("cc_matrix", C.c_double * 9), ("cc_bias", C.c_double * 3),
        ("gamma_enabled", C.c_int), ("gamma_k", C.c_double),
        ("vig_enabled", C.c_int),
        ("vig_scale", C.c_double), ("vig_a2", C.c_double), ("vig_a4", C.c_double),
        ("vig_mask", C.c_void_p),
        ("ce_enabled", C.c_int),
        ("ce_h_gain", C.c_double), ("ce_s_gain", C.c_double), ("ce_v_gain", C.c_double),
        ("und_enabled", C.c_int),
        ("map_x", C.c_void_p), ("map_y", C.c_void_p),
        ("map_rows", C.c_int), ("map_cols", C.c_int),
        ("reference_schedule", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ripo_ccc_create.restype = C.c_void_p
        _lib.ripo_ab_to_xz.restype = C.c_int
    return _lib


class fp_contraction:
    """Context manager: the oracle's floating-point contraction model (rip_oracle.c: 0 none -- the default, what the HIP
    kernels implement --, 1 fused as GCC / Clang contract the float stages on FMA targets, 2 the other association of the
    colour matrix's dot product).  Process-wide: not for concurrent use."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = lib().ripo_get_fp_contraction()
        lib().ripo_set_fp_contraction(self.mode)
        return self

    def __exit__(self, *exc):
        lib().ripo_set_fp_contraction(self.prev)
        return False


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def _f64(a, n):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    assert a.size == n, (a.size, n)
    return a, a.ctypes.data_as(C.c_void_p)


def debayer(bayer, encoding):
    bayer, pb = _u8(bayer)
    rows, cols = bayer.shape
    out = np.empty((rows, cols, 3), np.uint8)
    lib().ripo_debayer_bilinear(pb, rows, cols, BAYER[encoding], out.ctypes.data_as(C.c_void_p))
    return out


def debayer16(bayer, encoding):
    """Extension: 16-bit Bayer frame (uint16, HxW) -> BGR uint16 with the 8-bit path's formulas."""
    a = np.ascontiguousarray(bayer, np.uint16)
    rows, cols = a.shape
    out = np.empty((rows, cols, 3), np.uint16)
    lib().ripo_debayer_bilinear16(a.ctypes.data_as(C.c_void_p), rows, cols, BAYER[encoding.replace("16", "8")], out.ctypes.data_as(C.c_void_p))
    return out


def swap_rb(img):
    img, p = _u8(img)
    out = np.empty_like(img)
    lib().ripo_swap_rb(p, C.c_size_t(img.shape[0] * img.shape[1]), out.ctypes.data_as(C.c_void_p))
    return out


def flip(img, angle):
    img, p = _u8(img)
    rows, cols = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty(rows * cols * cn, np.uint8)
    orows, ocols = C.c_int(), C.c_int()
    lib().ripo_flip(p, rows, cols, cn, int(angle), out.ctypes.data_as(C.c_void_p), C.byref(orows), C.byref(ocols))
    shape = (orows.value, ocols.value) if img.ndim == 2 else (orows.value, ocols.value, cn)
    return out.reshape(shape)


def wb_grayworld(img, thr, return_stats=False):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    sums = (C.c_uint64 * 3)()
    ig = (C.c_int * 3)()
    lib().ripo_wb_grayworld(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.shape[0] * img.shape[1]),
                            C.c_double(thr), sums, ig)
    if return_stats:
        return img, list(sums), list(ig)
    return img


def wb_simple(img, percentile, return_coeffs=False):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    ab = (C.c_float * 6)()
    lib().ripo_wb_simple(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.shape[0] * img.shape[1]), C.c_double(percentile), ab)
    if return_coeffs:
        return img, list(ab)
    return img


def wb_pca(img, return_coeffs=False):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    co = (C.c_float * 4)()
    lib().ripo_wb_pca(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.shape[0] * img.shape[1]), co)
    if return_coeffs:
        return img, list(co)
    return img


class CCC:
    """ConvolutionalColorConstancyWB restatement; holds the Kalman state of one stream."""

    def __init__(self, filt, bias):
        filt, pf = _f32(filt)
        bias, pb = _f32(bias)
        assert filt.shape == (256, 256) and bias.shape == (256, 256)
        self._h = C.c_void_p(lib().ripo_ccc_create(256, 256, pf, pb))
        assert self._h
        self._keep = (filt, bias)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ripo_ccc_destroy(self._h)
            self._h = None

    def set_thresholds(self, bright, dark):
        lib().ripo_ccc_set_thresholds(self._h, C.c_float(bright), C.c_float(dark))

    def set_temporal_consistency(self, on):
        lib().ripo_ccc_set_temporal_consistency(self._h, int(bool(on)))

    def set_kalman_model(self, h, r):
        lib().ripo_ccc_set_kalman_model(self._h, C.c_float(h), C.c_float(r))

    def reset(self):
        lib().ripo_ccc_reset(self._h)

    def balance(self, img):
        img = np.array(img, dtype=np.uint8, order="C", copy=True)
        info = (C.c_int * 4)()
        gains = (C.c_float * 3)()
        lib().ripo_ccc_balance(self._h, img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], info, gains)
        return img, list(info), list(gains)

    def histogram(self, small):
        small, p = _u8(small)
        hist = np.empty((256, 256), np.float32)
        lib().ripo_ccc_histogram(self._h, p, small.shape[0], small.shape[1], hist.ctypes.data_as(C.c_void_p))
        return hist

    def response(self, hist):
        hist, p = _f32(hist)
        out = np.empty((256, 256), np.float32)
        lib().ripo_ccc_response(self._h, p, out.ctypes.data_as(C.c_void_p))
        return out

    def response_direct(self, hist):
        hist, p = _f32(hist)
        out = np.empty((256, 256), np.float64)
        lib().ripo_ccc_response_direct(self._h, p, out.ctypes.data_as(C.c_void_p))
        return out


def ccc_gains_from_uv(x, y):
    g = (C.c_float * 3)()
    lib().ripo_ccc_gains_from_uv(int(x), int(y), g)
    return list(g)


def fft256_twiddles():
    re = np.empty(128, np.float32)
    im = np.empty(128, np.float32)
    lib().ripo_fft256_twiddles(re.ctypes.data_as(C.c_void_p), im.ctypes.data_as(C.c_void_p))
    return re, im


def resize_linear(img, drows, dcols):
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((drows, dcols, cn), np.uint8)
    lib().ripo_resize_linear_8u(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(C.c_void_p), drows, dcols)
    return out if img.ndim == 3 else out[:, :, 0]


def color_matrix(img, m, bias):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    m, pm = _f64(m, 9)
    bias, pb = _f64(bias, 3)
    lib().ripo_color_matrix(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.shape[0] * img.shape[1]), pm, pb)
    return img


def gamma_lut(k):
    lut = np.empty(256, np.uint8)
    lib().ripo_gamma_lut(C.c_double(k), lut.ctypes.data_as(C.c_void_p))
    return lut


def apply_lut(img, lut):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    lut, pl = _u8(lut)
    lib().ripo_apply_lut(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.size), pl)
    return img


def vignetting_mask(rows, cols, scale, a2, a4):
    mask = np.empty((rows, cols), np.float32)
    lib().ripo_vignetting_mask(rows, cols, C.c_double(scale), C.c_double(a2), C.c_double(a4),
                               mask.ctypes.data_as(C.c_void_p))
    return mask


def vignetting(img, mask):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    mask, pm = _f32(mask)
    lib().ripo_vignetting(img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], pm)
    return img


def _px_op(name, img):
    img, p = _u8(img)
    out = np.empty_like(img)
    getattr(lib(), name)(p, C.c_size_t(img.size // 3), out.ctypes.data_as(C.c_void_p))
    return out


def bgr2lab(img):
    return _px_op("ripo_bgr2lab", img)


def lab2bgr(img):
    return _px_op("ripo_lab2bgr", img)


def bgr2hsv(img):
    return _px_op("ripo_bgr2hsv", img)


def hsv2bgr(img):
    return _px_op("ripo_hsv2bgr", img)


def color_enhance(img, h_gain, s_gain, v_gain):
    img = np.array(img, dtype=np.uint8, order="C", copy=True)
    lib().ripo_color_enhance(img.ctypes.data_as(C.c_void_p), C.c_size_t(img.shape[0] * img.shape[1]),
                             C.c_double(h_gain), C.c_double(s_gain), C.c_double(v_gain))
    return img


TABLE_IDS = {"srgb_gamma": 0, "cbrt": 1, "lab_to_yf": 2, "inv_gamma": 3, "fwd_coeffs": 4, "inv_coeffs": 5,
             "sdiv": 6, "hdiv180": 7}


def table(name):
    buf = np.empty(4096, np.int32)
    n = lib().ripo_table(TABLE_IDS[name], buf.ctypes.data_as(C.c_void_p), 4096)
    return buf[:n].copy()


def ab_to_xz(i):
    return lib().ripo_ab_to_xz(int(i))


def fisheye_new_camera_matrix(K, D, size, R, balance, new_size, fov_scale):
    K, pk = _f64(K, 9)
    D, pd = _f64(D, 4)
    R, pr = _f64(R, 9)
    out = np.empty(9, np.float64)
    nw, nh = (new_size if new_size else (0, 0))
    lib().ripo_fisheye_new_camera_matrix(pk, pd, int(size[0]), int(size[1]), pr, C.c_double(balance), int(nw),
                                         int(nh), C.c_double(fov_scale), out.ctypes.data_as(C.c_void_p))
    return out.reshape(3, 3)


def fisheye_maps(K, D, R, P, size):
    K, pk = _f64(K, 9)
    D, pd = _f64(D, 4)
    R, pr = _f64(R, 9)
    P, pp = _f64(np.asarray(P, dtype=np.float64).reshape(3, -1)[:, :3], 9)
    w, h = int(size[0]), int(size[1])
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    lib().ripo_fisheye_maps(pk, pd, pr, pp, w, h, mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p))
    return mx, my


def remap(img, map_x, map_y):
    img, p = _u8(img)
    map_x, px = _f32(map_x)
    map_y, py = _f32(map_y)
    cn = 1 if img.ndim == 2 else img.shape[2]
    drows, dcols = map_x.shape
    out = np.empty((drows, dcols) if img.ndim == 2 else (drows, dcols, cn), np.uint8)
    lib().ripo_remap_linear(p, img.shape[0], img.shape[1], cn, px, py, drows, dcols, out.ctypes.data_as(C.c_void_p))
    return out


def pipeline(params, image, encoding, ccc=None, taps=False):
    """Whole chain on one frame.  params: oracle.Params.  Returns (out, encoding_out[, taps])."""
    image, p = _u8(image)
    rows, cols = image.shape[:2]
    cn = 1 if image.ndim == 2 else image.shape[2]
    out = np.empty(rows * cols * 3, np.uint8)
    orows, ocols, ocn = C.c_int(), C.c_int(), C.c_int()
    enc = C.create_string_buffer(32)
    t0 = np.empty(rows * cols * 3, np.uint8) if taps else None
    t1 = np.empty(rows * cols * 3, np.uint8) if taps else None
    rc = lib().ripo_pipeline(C.byref(params), ccc._h if ccc is not None else None, p, rows, cols, cn,
                             encoding.encode(), out.ctypes.data_as(C.c_void_p), C.byref(orows), C.byref(ocols),
                             C.byref(ocn), enc, t0.ctypes.data_as(C.c_void_p) if taps else None,
                             t1.ctypes.data_as(C.c_void_p) if taps else None)
    if rc == -1:
        raise ValueError("Encoding [%s] is a valid pattern but is not supported!" % encoding)
    if rc != 0:
        raise RuntimeError("oracle pipeline failed: %d" % rc)
    n = orows.value * ocols.value * ocn.value
    shape = (orows.value, ocols.value) if ocn.value == 1 else (orows.value, ocols.value, ocn.value)
    res = out[:n].reshape(shape).copy()
    if taps:
        return res, enc.value.decode(), t0, t1
    return res, enc.value.decode()
