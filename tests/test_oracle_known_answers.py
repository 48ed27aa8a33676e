"""Known-answer tests that pin the CPU oracle (PARITY UNPINNED otherwise: the reference has no tests
or golden vectors and OpenCV is not installed here).  Every expected value below is derived by hand
from the reference text or from the published OpenCV 4.2 algorithm -- never from running the oracle."""
import math

import numpy as np
import pytest

from raw_image_pipeline_amd import synth


# ---- debayer (debayer.cpp:45-79 -> cv::demosaicing bilinear + RGB2BGR) ---------------------------------
def hand_debayer(bayer, pattern):
    """Straight per-pixel transcription of the bilinear rule, independent of the oracle's loop."""
    cell = synth.PATTERNS[pattern]
    h, w = bayer.shape
    s = bayer.astype(np.int32)
    out = np.zeros((h, w, 3), np.int32)
    for y in range(1, h - 1):
        for x in range(1, w - 1):
            c = cell[y & 1][x & 1]
            if c == 1:
                hc = cell[y & 1][(x + 1) & 1]
                out[y, x, 1] = s[y, x]
                out[y, x, hc] = (s[y, x - 1] + s[y, x + 1] + 1) >> 1
                out[y, x, 2 - hc] = (s[y - 1, x] + s[y + 1, x] + 1) >> 1
            else:
                out[y, x, c] = s[y, x]
                out[y, x, 1] = (s[y, x - 1] + s[y, x + 1] + s[y - 1, x] + s[y + 1, x] + 2) >> 2
                out[y, x, 2 - c] = (s[y - 1, x - 1] + s[y - 1, x + 1] + s[y + 1, x - 1] + s[y + 1, x + 1] + 2) >> 2
    out[:, 0] = out[:, 1]
    out[:, -1] = out[:, -2]
    out[0] = out[1]
    out[-1] = out[-2]
    return out.astype(bayer.dtype)


@pytest.mark.parametrize("pattern", sorted(synth.PATTERNS))
def test_debayer_matches_hand_transcription(oracle, pattern):
    bayer = np.random.default_rng(1).integers(0, 256, (12, 14), dtype=np.uint8)
    assert np.array_equal(oracle.debayer(bayer, pattern), hand_debayer(bayer, pattern))


@pytest.mark.parametrize("pattern", sorted(synth.PATTERNS))
def test_debayer16_matches_hand_transcription(oracle, pattern):
    """The 16-bit extension (the reference rejects bayer_*16): same rule on uint16, incl. values whose sums need 18 bits."""
    bayer = np.random.default_rng(3).integers(0, 65536, (12, 14)).astype(np.uint16)
    bayer[4:6, 4:8] = 65535
    got = oracle.debayer16(bayer, pattern.replace("8", "16"))
    assert got.dtype == np.uint16 and np.array_equal(got, hand_debayer(bayer, pattern))
    # 8-bit data in 16-bit containers goes through unchanged: the rule does not depend on the sample width
    b8 = np.random.default_rng(4).integers(0, 256, (10, 12), dtype=np.uint8)
    assert np.array_equal(oracle.debayer16(b8.astype(np.uint16), pattern.replace("8", "16")), oracle.debayer(b8, pattern))


def test_debayer_constant_colour_planes(oracle):
    # a mosaic of a constant colour must demosaic to that colour everywhere (all taps equal)
    bgr = np.empty((10, 12, 3), np.uint8)
    bgr[:] = (30, 120, 200)
    for pattern in synth.PATTERNS:
        out = oracle.debayer(synth.mosaic(bgr, pattern), pattern)
        assert (out == np.array([30, 120, 200], np.uint8)).all(), pattern


def test_debayer_equals_reference_call_sequence(oracle):
    # the reference demosaics with the R/B-exchanged OpenCV code and then swaps R and B back
    # (debayer.cpp:48-52); both routes must agree
    bayer = np.random.default_rng(2).integers(0, 256, (16, 20), dtype=np.uint8)
    names = ["bayer_rggb8", "bayer_grbg8", "bayer_gbrg8", "bayer_bggr8"]
    for i, name in enumerate(names):
        assert np.array_equal(oracle.debayer(bayer, name), oracle.swap_rb(oracle.debayer(bayer, names[3 - i])))


# ---- flip (flip.cpp:37-58) ---------------------------------------------------------------------------------
def test_flip_permutations(oracle):
    img = np.arange(4 * 5 * 3, dtype=np.uint8).reshape(4, 5, 3)
    assert np.array_equal(oracle.flip(img, 180), img[::-1, ::-1])
    assert np.array_equal(oracle.flip(img, 90), np.rot90(img, k=-1))   # transpose + mirror x = clockwise
    assert np.array_equal(oracle.flip(img, 270), np.rot90(img, k=1))   # transpose + mirror y = counter-clockwise
    assert np.array_equal(oracle.flip(img, 0), img)
    assert np.array_equal(oracle.flip(img, 45), img)                   # any other angle: nothing
    assert np.array_equal(oracle.flip(oracle.flip(img, 90), 270), img)


# ---- gamma (gamma_correction.cpp:35-43) ------------------------------------------------------------------
def test_gamma_lut_known_entries(oracle):
    # values computed by restating the five lines of GammaCorrectionModule::init()
    idx = [1, 2, 16, 64, 128, 200, 254]
    assert list(oracle.gamma_lut(0.8)[idx]) == [3, 5, 28, 84, 147, 210, 254]
    assert list(oracle.gamma_lut(0.9)[idx]) == [2, 3, 21, 73, 137, 205, 254]
    assert np.array_equal(oracle.gamma_lut(1.0), np.arange(256, dtype=np.uint8))
    lut = oracle.gamma_lut(0.8)
    assert lut[0] == 0 and lut[255] == 255 and (np.diff(lut.astype(int)) >= 0).all()


def test_gamma_lut_literal_formula(oracle):
    for k in (0.45, 0.8, 2.2):
        exp = []
        for i in range(256):
            f = np.float32(i / 255.0)
            f = np.float32(math.pow(float(f), k))
            v = float(f) * 255.0
            exp.append(min(255, max(0, int(np.rint(v)))))
        assert list(oracle.gamma_lut(k)) == exp


# ---- grey world (xphoto GrayworldWB) -------------------------------------------------------------------------
def test_grayworld_hand_example(oracle):
    # two unsaturated pixels and one saturated one that must be skipped
    img = np.array([[[100, 50, 200], [60, 30, 120], [0, 255, 255]]], np.uint8)
    # thr 0.9 -> thresh255 = 230 (cvRound(229.5) = 230, half to even);  pixel 3: (255-0)*255 > 230*255 -> skipped
    out, sums, ig = oracle.wb_grayworld(img, 0.9, return_stats=True)
    assert sums == [160, 80, 320]
    # gains = max/sum = (2, 4, 1) / 4 -> Q8 (128, 256, 64)
    assert ig == [128, 256, 64]
    assert out.tolist() == [[[50, 50, 50], [30, 30, 30], [0, 255, 63]]]


def test_grayworld_uniform_grey_is_identity(oracle):
    img = np.full((6, 7, 3), 90, np.uint8)
    assert np.array_equal(oracle.wb_grayworld(img, 0.8), img)


# ---- simple WB (xphoto SimpleWB) ------------------------------------------------------------------------------
def test_simple_wb_percentile_stretch(oracle):
    # one channel plane 0..199 uniformly (each value equally often): p = 10 % cuts at 20 and 180
    vals = np.repeat(np.arange(200, dtype=np.uint8), 5)
    img = np.stack([vals, vals, vals], axis=-1).reshape(25, 40, 3)
    out, ab = oracle.wb_simple(img, 10.0, return_coeffs=True)
    # Hand run of balanceWhiteSimple's tree (16 x 16 bins; the second level of first-level bin 0 aliases
    # hist[0..15], so those entries hold 80 + 5 / 40 + 5 / 5 counts):
    #   level 0: low  85 < 100 -> skip bin 0 (min 15.5), 170 >= 100 stop;  high: 995, 990, 985, 940 > 900 -> max 191.5
    #   level 1: low  90, 95 < 100 -> min 17.5;  high: 935 ... 905 > 900 (7 steps) -> max 184.5
    # (an alias-free tree would give 19.5 / 179.5)
    for c in range(3):
        a, b = ab[2 * c], ab[2 * c + 1]
        assert abs(-b / a - 17.5) < 1e-3 and abs((255 - b) / a - 184.5) < 1e-3
    assert out.min() == 0 and out.max() == 255
    mid = img[..., 0] == 100
    assert (out[..., 0][mid] == int(np.rint(np.float32(100) * np.float32(ab[0]) + np.float32(ab[1])))).all()


# ---- colour matrix (color_calibration.cpp:91-104) -------------------------------------------------------------
def test_color_matrix_rounding_and_saturation(oracle):
    img = np.array([[[10, 20, 30], [255, 255, 255], [1, 1, 1]]], np.uint8)
    m = [1, 0, 0, 0, 0.5, 0, 0.25, 0.25, 0.25]
    out = oracle.color_matrix(img, m, [0.5, 0.0, -100.0])
    # B: 10 + .5 = 10.5 -> 10 (half to even); G: 10; R: 15 - 100 -> 0
    assert out[0, 0].tolist() == [10, 10, 0]
    # B: 255.5 -> 256 -> 255; G: 127.5 -> 128 (even); R: 191.25 - 100 = 91
    assert out[0, 1].tolist() == [255, 128, 91]
    assert out[0, 2].tolist() == [2, 0, 0]  # 1.5 -> 2 (even); 0.5 -> 0 (even)
    ident = oracle.color_matrix(img, [1, 0, 0, 0, 1, 0, 0, 0, 1], [0, 0, 0])
    assert np.array_equal(ident, img)


# ---- vignetting mask (vignetting_correction.cpp:32-63) --------------------------------------------------------
def test_vignetting_mask_known_values(oracle):
    m = oracle.vignetting_mask(2048, 2448, 1.5, 1e-3, 1e-6)
    assert m.shape == (2048, 2448)
    assert m[1024, 1224] == np.float32(1.0)      # r = 0
    assert m[0, 0] == np.float32(2.5)            # the farthest pixel defines the maximum: 1 + scale
    # mid left edge: r^2 = 1224^2
    r2 = 1224.0 ** 2
    rmax2 = 1224.0 ** 2 + 1024.0 ** 2
    k = lambda s: s * 1e-3 + s * s * 1e-6
    assert abs(float(m[1024, 0]) - (1 + 1.5 * k(r2) / k(rmax2))) < 1e-6
    # the double argument swap of the reference cancels: geometry follows (row, col)
    small = oracle.vignetting_mask(4, 8, 1.0, 1.0, 0.0)
    assert small[2, 4] == 1.0 and small[0, 0] == 2.0 and small[2, 0] > small[0, 4]


def test_vignetting_identity_mask_is_lab_roundtrip(oracle):
    img = np.random.default_rng(3).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    ones = np.ones((8, 8), np.float32)
    assert np.array_equal(oracle.vignetting(img, ones), oracle.lab2bgr(oracle.bgr2lab(img)))


# ---- 8-bit Lab / HSV (OpenCV color_lab.cpp / color_hsv.cpp) ---------------------------------------------------
def test_lab_known_points(oracle):
    px = np.array([[[0, 0, 0], [255, 255, 255], [128, 128, 128]]], np.uint8)
    lab = oracle.bgr2lab(px)
    assert lab[0, 0].tolist() == [0, 128, 128]
    assert lab[0, 1].tolist() == [255, 128, 128]
    assert lab[0, 2, 1] == 128 and lab[0, 2, 2] == 128 and abs(int(lab[0, 2, 0]) - 137) <= 1  # L*(0.2159) = 53.6 -> 136.7
    assert np.array_equal(oracle.lab2bgr(lab)[0, :2], px[0, :2])
    # pure sRGB red: L* = 53.24, a* = 80.09, b* = 67.20 (textbook values)
    red = oracle.bgr2lab(np.array([[[0, 0, 255]]], np.uint8))[0, 0]
    assert abs(int(red[0]) - round(53.24 * 2.55)) <= 1 and abs(int(red[1]) - 208) <= 1 and abs(int(red[2]) - 195) <= 1


def test_lab_and_hsv_of_the_primaries_equal_the_values_opencv_is_quoted_with(oracle):
    """The one pin from OUTSIDE this repository that needs no OpenCV at run time: what `cv2.cvtColor` returns for the uint8
    primaries is quoted all over OpenCV's own tutorials and Q&A sites (py_colorspaces: "green = np.uint8([[[0,255,0]]]) ...
    [[[ 60 255 255]]]"; COLOR_BGR2LAB of red / green / blue: [136 208 195], [224 42 211], [82 207 20]).  Exact, not +- 1:
    the 8-bit paths are integer table code (RGB2Lab_b, RGB2HSV_b), so a restatement either has these bytes or is wrong."""
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0]]], np.uint8)  # BGR: red, green, blue
    assert oracle.bgr2lab(px)[0].tolist() == [[136, 208, 195], [224, 42, 211], [82, 207, 20]]
    assert oracle.bgr2hsv(px)[0].tolist() == [[0, 255, 255], [60, 255, 255], [120, 255, 255]]


def test_lab_tables_closed_forms(oracle):
    g = oracle.table("srgb_gamma")
    assert g[0] == 0 and g[255] == 2040 and g[10] == round(2040 * (10 / 255) / 12.92)
    assert abs(int(g[128]) - 2040 * ((128 / 255 + 0.055) / 1.055) ** 2.4) <= 1
    cb = oracle.table("cbrt")
    assert cb[0] == round(32768 * 16 / 116) and cb[2040] == 32768
    yf = oracle.table("lab_to_yf")
    assert yf[2 * 255] == 16384 and yf[2 * 255 + 1] == 16384 and yf[1] == round(16384 * 16 / 116)
    assert oracle.table("fwd_coeffs").reshape(3, 3).sum(axis=1).tolist() == [4096, 4096, 4096]
    # abToXZ_b: linear below 6/29, cube above, continuous at the switch
    assert oracle.ab_to_xz(16384) == 16384 and oracle.ab_to_xz(0) == -290 and oracle.ab_to_xz(3390) == 145
    assert oracle.ab_to_xz(3391) == (3391 * 3391 // 16384) * 3391 // 16384
    assert oracle.ab_to_xz(-8145) == -(8145 * 108 // 841) - 290  # C truncation toward zero


def test_hsv_known_points(oracle):
    px = np.array([[[0, 0, 255], [0, 255, 0], [255, 0, 0], [255, 255, 255], [0, 0, 0], [50, 100, 200]]], np.uint8)
    hsv = oracle.bgr2hsv(px)[0]
    assert hsv[0].tolist() == [0, 255, 255]      # red
    assert hsv[1].tolist() == [60, 255, 255]     # green: 120 deg / 2
    assert hsv[2].tolist() == [120, 255, 255]    # blue
    assert hsv[3].tolist() == [0, 0, 255] and hsv[4].tolist() == [0, 0, 0]
    # (b,g,r) = (50,100,200): v=200, diff=150, s = 255*150/200 = 191.25 -> 191, h = 60*(g-b)/diff /2 = 10
    assert hsv[5].tolist() == [10, 191, 200]
    assert np.array_equal(oracle.hsv2bgr(oracle.bgr2hsv(px[:, :5])), px[:, :5])


def test_color_enhancer_identity_gain_is_hsv_roundtrip(oracle):
    img = np.random.default_rng(4).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    assert np.array_equal(oracle.color_enhance(img, 1.0, 1.0, 1.0), oracle.hsv2bgr(oracle.bgr2hsv(img)))
    grey = np.full((2, 2, 3), 77, np.uint8)
    assert np.array_equal(oracle.color_enhance(grey, 1.0, 3.0, 1.0), grey)  # s == 0 stays grey


# ---- remap (cv::remap INTER_LINEAR, BORDER_CONSTANT 0) -----------------------------------------------------------
def test_remap_identity_and_weights(oracle):
    img = np.random.default_rng(5).integers(0, 256, (6, 7, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:6, 0:7].astype(np.float32)
    assert np.array_equal(oracle.remap(img, xx, yy), img)
    # quarter-pixel offsets: weights from the 1/32 grid: fx = 8/32, fy = 16/32
    mx = np.full((1, 1), 2.25, np.float32)
    my = np.full((1, 1), 3.5, np.float32)
    p = img.astype(np.int64)
    exp = (p[3, 2] * 32 * 24 * 16 + p[3, 3] * 32 * 8 * 16 + p[4, 2] * 32 * 24 * 16 + p[4, 3] * 32 * 8 * 16 + (1 << 14)) >> 15
    assert oracle.remap(img, mx, my)[0, 0].tolist() == exp.tolist()
    # coordinates are quantised to 1/32 px with round-half-even: 2 + 1/64 -> 2.0 exactly (tie to even)
    assert np.array_equal(oracle.remap(img, np.full((1, 1), 2 + 1 / 64, np.float32), np.full((1, 1), 1.0, np.float32))[0, 0], img[1, 2])


def test_remap_border_constant(oracle):
    img = np.full((4, 4, 3), 200, np.uint8)
    def one(x, y):
        return oracle.remap(img, np.full((1, 1), x, np.float32), np.full((1, 1), y, np.float32))[0, 0, 0]
    assert one(-1.0, 1.0) == 0 and one(4.0, 1.0) == 0 and one(1.0, -1.0) == 0 and one(1.0, 4.0) == 0
    assert one(-0.5, 1.0) == 100          # half of the weight falls on the zero border
    assert one(3.5, 1.0) == 100 and one(3.0, 3.0) == 200 and one(3.5, 3.5) == 50
    assert one(float("nan"), 1.0) == 0 and one(1e30, 1.0) == 0


# ---- fisheye (cv::fisheye::initUndistortRectifyMap / estimateNewCameraMatrix...) ----------------------------------
def test_fisheye_maps_closed_form(oracle):
    K = [100.0, 0, 50.0, 0, 100.0, 40.0, 0, 0, 1]
    R = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    D = [0.0, 0.0, 0.0, 0.0]
    mx, my = oracle.fisheye_maps(K, D, R, K, (100, 80))
    # zero coefficients: theta_d = theta = atan(r): u = f * x * atan(r)/r + c
    j, i = 80, 10
    x, y = (j - 50.0) / 100.0, (i - 40.0) / 100.0
    r = math.hypot(x, y)
    assert abs(mx[i, j] - (100.0 * x * math.atan(r) / r + 50.0)) < 1e-4
    assert abs(my[i, j] - (100.0 * y * math.atan(r) / r + 40.0)) < 1e-4
    assert mx[40, 50] == 50.0 and my[40, 50] == 40.0  # principal point maps to itself


def test_fisheye_new_camera_matrix_properties(oracle):
    cam = synth.camera_model(640, 480)
    size = (640, 480)
    k_min = oracle.fisheye_new_camera_matrix(cam["K"], cam["D"], size, cam["R"], 1.0, None, 1.0)
    k_max = oracle.fisheye_new_camera_matrix(cam["K"], cam["D"], size, cam["R"], 0.0, None, 1.0)
    assert k_min[0, 0] < k_max[0, 0]                       # balance interpolates between fmin and fmax
    k_half = oracle.fisheye_new_camera_matrix(cam["K"], cam["D"], size, cam["R"], 0.0, None, 2.0)
    assert abs(k_half[0, 0] * 2.0 - k_max[0, 0]) < 1e-9    # fov_scale divides the focal length
    k_resized = oracle.fisheye_new_camera_matrix(cam["K"], cam["D"], size, cam["R"], 0.0, (320, 240), 1.0)
    assert np.allclose(k_resized[:2], k_max[:2] * 0.5)     # new_size rescales f and c
    assert k_max[2].tolist() == [0, 0, 1] and k_max[0, 1] == 0 and k_max[1, 0] == 0


# ---- ccc (convolutional_color_constancy.cpp) -------------------------------------------------------------------
def test_ccc_gain_formula(oracle):
    # gains = exp(k/64 - 1.421875) normalised by the minimum (computeGains :342-381)
    g = oracle.ccc_gains_from_uv(100, 120)
    lu, lv = 100 / 64 - 1.421875, 120 / 64 - 1.421875
    r, gg, b = math.exp(lu), 1.0, math.exp(lv)
    f = min(r, gg, b)
    assert np.allclose(g, [b / f, gg / f, r / f], rtol=1e-6)
    assert oracle.ccc_gains_from_uv(91, 91) == pytest.approx([1.0, 1.0, 1.0], abs=1e-6)  # uv0 = -91/64


def test_ccc_fft_response_matches_direct_convolution(oracle):
    filt, bias = synth.ccc_model()
    bias = bias + np.random.default_rng(6).normal(0, 1e-5, bias.shape).astype(np.float32)
    c = oracle.CCC(filt, bias)
    hist = np.zeros((256, 256), np.float32)
    hist[40, 200] = 0.5
    hist[41, 201] = 0.3
    hist[250, 3] = 0.2
    resp = c.response(hist) / 65536.0       # unscaled forward and inverse transforms
    direct = c.response_direct(hist)
    assert np.abs(resp - direct).max() < 1e-6 * max(1.0, np.abs(direct).max())
    assert np.unravel_index(resp.argmax(), resp.shape) == np.unravel_index(direct.argmax(), direct.shape)


def test_ccc_histogram_binning(oracle):
    filt, bias = synth.ccc_model()
    c = oracle.CCC(filt, bias)
    c.set_thresholds(0.9, 0.1)
    small = np.zeros((270, 360, 3), np.uint8)
    small[:] = (50, 100, 200)          # one colour everywhere: a single populated bin
    h = c.histogram(small)
    u = int(round((math.log(100) - math.log(200) + 1.421875) * 64))
    v = int(round((math.log(100) - math.log(50) + 1.421875) * 64))
    assert np.count_nonzero(h) == 1 and h[u, v] > 0.99     # hist.at(u, v): row u, column v
    small[:] = (1, 1, 1)               # darker than 255 * dark_thr: masked out
    assert np.count_nonzero(c.histogram(small)) == 0
    small[:] = (0, 100, 100)           # log(0) = -inf: skipped
    assert np.count_nonzero(c.histogram(small)) == 0


def test_ccc_kalman_models(oracle):
    filt, bias = synth.ccc_model()
    frames = [synth.gen_scene_bgr(180, 135, seed=i, tint=(0.6 + 0.08 * i, 1.0, 0.5)) for i in range(4)]
    free = oracle.CCC(filt, bias)
    raw = [free.balance(f)[1][:2] for f in frames]
    assert len(set(map(tuple, raw))) > 1, "the drifting tint must move the argmax"
    frozen = oracle.CCC(filt, bias)          # pipeline default: H = 0 (one-argument constructor)
    frozen.set_temporal_consistency(True)
    used = [frozen.balance(f)[1][2:] for f in frames]
    assert all(u == used[0] for u in used) and used[0] == raw[0]
    filt_kf = oracle.CCC(filt, bias)         # loadModel's filter: H = I, R = 10 I, Q = I, P0 = 0
    filt_kf.set_temporal_consistency(True)
    filt_kf.set_kalman_model(1.0, 10.0)
    x = None
    p = 0.0
    for i, f in enumerate(frames):
        info = filt_kf.balance(f)[1]
        z = info[:2]
        if i == 0:
            x = [float(z[0]), float(z[1])]
            assert info[2:] == z
            continue
        pp = p + 1.0
        k = pp / (pp + 10.0)
        x = [x[0] + k * (z[0] - x[0]), x[1] + k * (z[1] - x[1])]
        p = (1 - k) * pp
        assert info[2:] == [int(x[0]), int(x[1])]


# ---- resize used by ccc (cv::resize INTER_LINEAR, 8U) ------------------------------------------------------------
def test_resize_area_switch_and_constant(oracle):
    img = np.random.default_rng(7).integers(0, 256, (540, 720, 3), dtype=np.uint8)
    small = oracle.resize_linear(img, 270, 360)   # exact 2x: INTER_LINEAR silently becomes INTER_AREA
    p = img.astype(np.int32)
    exp = (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2
    assert np.array_equal(small, exp.astype(np.uint8))
    flat = np.full((300, 500, 3), 93, np.uint8)
    assert (oracle.resize_linear(flat, 270, 360) == 93).all()


# ---- whole chain ------------------------------------------------------------------------------------------------
def test_pipeline_all_stages_off_is_debayer_only(oracle):
    from helpers import cfg, oracle_run
    frame = synth.gen_frame(32, 24, "bayer_rggb8", seed=9, kind="uniform")
    out, enc = oracle_run(oracle, cfg(), frame, "bayer_rggb8")
    assert enc == "bgr8" and np.array_equal(out, oracle.debayer(frame, "bayer_rggb8"))
    bgr = np.random.default_rng(1).integers(0, 256, (8, 8, 3), dtype=np.uint8)
    out, enc = oracle_run(oracle, cfg(), bgr, "bgr8")
    assert enc == "bgr8" and np.array_equal(out, bgr)
    out, enc = oracle_run(oracle, cfg(), bgr, "rgb8")
    assert enc == "rgb8" and np.array_equal(out, bgr[..., ::-1])      # swapped, string kept (debayer.cpp:72-73)
    with pytest.raises(ValueError):
        oracle_run(oracle, cfg(), frame, "bayer_bggr16")


def test_pipeline_reference_schedule_gives_same_pixels(oracle):
    from helpers import cfg, oracle_params
    w, h = 96, 64
    frame = synth.gen_frame(w, h, "bayer_grbg8", seed=10, kind="scene")
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, vig=True, ce=True, ce_sat=1.2,
            undistort=True, cam=synth.camera_model(w, h))
    keep = []
    prm = oracle_params(oracle, c, keep)
    a, _ = oracle.pipeline(prm, frame, "bayer_grbg8")
    prm.reference_schedule = 1
    b, _ = oracle.pipeline(prm, frame, "bayer_grbg8")
    assert np.array_equal(a, b)


def test_lab_ab_never_saturate(oracle):
    """RGB2Lab_b saturates a and b to [0, 255]; over all 2^24 BGR inputs the unsaturated values stay inside
    [42, 226] and [20, 223], so the device kernel drops the clamp (rip_device.hpp apply_vignette) and folds
    the float-to-int bias into the next multiply-add.  L covers exactly [0, 255]."""
    g = oracle.table("srgb_gamma").astype(np.int64)
    cb = oracle.table("cbrt").astype(np.int64)
    C = oracle.table("fwd_coeffs").astype(np.int64)
    lo = np.array([1 << 30] * 3)
    hi = -lo
    v1, v2 = g[:, None], g[None, :]
    for v0 in g:
        fX = cb[(v0 * C[0] + v1 * C[1] + v2 * C[2] + 2048) >> 12]
        fY = cb[(v0 * C[3] + v1 * C[4] + v2 * C[5] + 2048) >> 12]
        fZ = cb[(v0 * C[6] + v1 * C[7] + v2 * C[8] + 2048) >> 12]
        L = (296 * fY - 1336935 + 16384) >> 15
        a = (500 * (fX - fY) + 128 * 32768 + 16384) >> 15
        b = (200 * (fY - fZ) + 128 * 32768 + 16384) >> 15
        lo = np.minimum(lo, [L.min(), a.min(), b.min()])
        hi = np.maximum(hi, [L.max(), a.max(), b.max()])
    assert (lo[0], hi[0]) == (0, 255)
    assert 0 < lo[1] and hi[1] < 255 and (lo[1], hi[1]) == (42, 226)
    assert 0 < lo[2] and hi[2] < 255 and (lo[2], hi[2]) == (20, 223)


def test_lab_inverse_x_and_y_fit_16_bits(oracle):
    """Lab2RGBinteger: with a in [42, 226] (see above) and any L, x = abToXZ_b[ify + adiv] stays inside int16 and so
    does y; z (b in [20, 223]) does not fit as it is -- the colour / rotating kernels (apply_vignette) feed (x, y) to
    v_dot2_i32_i16 and keep z on the 24-bit multiply, the fast kernel packs (x, z - 27500): see
    test_ranges_the_fast_lab_kernel_relies_on."""
    yf = oracle.table("lab_to_yf").reshape(256, 2)
    y, ify = yf[:, 0], yf[:, 1]
    assert 0 <= y.min() and y.max() <= 16384 and ify.max() <= 16384
    xs, zs = [], []
    for a in (42, 226):
        adiv = ((a * 5 * 53687 + 128) >> 13) - 128 * 16384 // 500
        xs += [oracle.ab_to_xz(int(ify.min()) + adiv), oracle.ab_to_xz(int(ify.max()) + adiv)]
    for b in (20, 223):
        bdiv = ((b * 41943 + 16) >> 9) - 128 * 16384 // 200 + 1
        zs += [oracle.ab_to_xz(int(ify.min()) - bdiv), oracle.ab_to_xz(int(ify.max()) - bdiv)]
    assert -32768 <= min(xs) and max(xs) <= 32767, xs   # abToXZ_b is monotone: the extremes are at the corners
    assert max(zs) > 32767                               # which is why z stays on the 24-bit multiply
    c = oracle.table("inv_coeffs")
    assert np.abs(c).max() <= 32767
    assert 2 * 32767 * int(np.abs(c).max()) + 65536 * int(np.abs(c).max()) + 8192 < 2 ** 31


# ---- independent-formula cross checks (float textbook colour science / scipy interpolation, not OpenCV) ------------
def _srgb_to_lab_float(bgr):
    rgb = bgr[..., ::-1].astype(np.float64) / 255.0
    lin = np.where(rgb <= 0.04045, rgb / 12.92, ((rgb + 0.055) / 1.055) ** 2.4)
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = lin @ m.T / np.array([0.950456, 1.0, 1.088754])
    f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
    L = np.where(xyz[..., 1] > 0.008856, 116.0 * f[..., 1] - 16.0, 903.3 * xyz[..., 1])
    return np.stack([L * 255.0 / 100.0, 500.0 * (f[..., 0] - f[..., 1]) + 128.0, 200.0 * (f[..., 1] - f[..., 2]) + 128.0], axis=-1)


def test_lab_8bit_tracks_the_float_cie_formulas(oracle):
    """RGB2Lab_b is a fixed-point evaluation of the CIE formulas OpenCV documents for cvtColor: every channel of the
    oracle's 8-bit result stays within 1.25 LSB (L) / 2 LSB (a, b: the cube-root table is coarse for dark colours) of
    the float evaluation over 16k random colours, 99.9 % within 1.15 LSB.  The 8-bit round trip is lossy by design (a, b are quantised to 1/255 of their range):
    half of the samples return exactly, 99 % within 6 LSB."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (128, 128, 3), dtype=np.uint8)
    lab = oracle.bgr2lab(img).astype(np.float64)
    ref = _srgb_to_lab_float(img)
    d = np.abs(lab - ref)
    assert d[..., 0].max() <= 1.25 and d[..., 1:].max() <= 2.0 and np.percentile(d, 99.9) <= 1.15, (d.max(axis=(0, 1)), np.percentile(d, 99.9))
    back = oracle.lab2bgr(oracle.bgr2lab(img)).astype(int)
    err = np.abs(back - img.astype(int))
    assert np.median(err) <= 1 and np.percentile(err, 99) <= 6 and (err == 0).mean() > 0.45


def test_hsv_hue_needs_no_saturation(oracle):
    """rip_device.hpp apply_hsv drops RGB2HSV_b's saturate_cast<uchar> of the hue and, for a hue gain of exactly 1, the
    fmod(h, 6) of the inverse: for every chroma `diff` and every numerator the three branches can produce ([-diff, diff],
    [diff, 3 diff], [3 diff, 5 diff]) the scaled, wrapped hue is within [0, 179], and float32(179) * (6.f / 180.f) < 6.
    The oracle's own RGB2HSV_b over all 2^24 colours agrees."""
    hdiv = oracle.table("hdiv180").astype(np.int64)
    assert len(hdiv) == 256 and hdiv[0] == 0
    lo, hi = 0, 0
    for diff in range(256):
        n = np.arange(-diff, 5 * diff + 1, dtype=np.int64)
        h = (n * hdiv[diff] + (1 << 11)) >> 12
        h = h + np.where(h < 0, 180, 0)
        lo, hi = min(lo, int(h.min())), max(hi, int(h.max()))
    assert (lo, hi) == (0, 179)
    assert np.float32(179) * (np.float32(6.0) / np.float32(180.0)) < np.float32(6.0)
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    assert int(oracle.bgr2hsv(img)[..., 0].max()) == 179


def test_hsv_8bit_tracks_the_float_formulas(oracle):
    import colorsys
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    hsv = oracle.bgr2hsv(img).reshape(-1, 3).astype(np.float64)
    px = img.reshape(-1, 3)
    for (b, g, r), (h8, s8, v8) in zip(px, hsv):
        h, s, v = colorsys.rgb_to_hsv(r / 255.0, g / 255.0, b / 255.0)
        dh = abs(h8 - h * 180.0)
        assert min(dh, 180.0 - dh) <= 1.0 + 90.0 / max(1, max(b, g, r) - min(b, g, r)) * 0.02 + 0.5 or s8 < 8  # hue is ill-conditioned near grey
        assert abs(s8 - s * 255.0) <= 1.0 and abs(v8 - v * 255.0) <= 0.5


def test_remap_tracks_scipy_bilinear(oracle):
    """cv::remap INTER_LINEAR = bilinear interpolation with the coordinates quantised to 1/32 px and the weights to
    Q15: within 1 LSB + the effect of the 1/64-px quantisation of scipy.ndimage.map_coordinates(order=1), zero outside."""
    from scipy import ndimage
    rng = np.random.default_rng(13)
    base = rng.integers(0, 256, (6, 8, 3)).astype(np.float64)
    img = np.clip(ndimage.zoom(base, (8, 8, 1), order=1), 0, 255).astype(np.uint8)  # smooth 48 x 64 image
    h, w = img.shape[:2]
    mx = rng.uniform(1, w - 2, (40, 50)).astype(np.float32)
    my = rng.uniform(1, h - 2, (40, 50)).astype(np.float32)
    got = oracle.remap(img, mx, my).astype(np.float64)
    for c in range(3):
        ref = ndimage.map_coordinates(img[..., c].astype(np.float64), [my.astype(np.float64), mx.astype(np.float64)], order=1, mode="constant")
        gy, gx = np.gradient(img[..., c].astype(np.float64))
        slack = 1.0 + (np.abs(gx).max() + np.abs(gy).max()) / 64.0
        assert np.abs(got[..., c] - ref).max() <= slack, (c, np.abs(got[..., c] - ref).max(), slack)


def test_lab_gamma_tables_do_not_depend_on_the_pow_implementation(oracle):
    """OpenCV builds sRGBGammaTab_b / sRGBInvGammaTab_b with its softfloat pow, the oracle with libm's.  Evaluated with
    50 digits, every entry is >= 13 (forward) / >= 5.5 (inverse, entry 3654 = 242.49992) float32 ulps of its value away
    from a rounding boundary, and the oracle's tables equal the exactly rounded ones: a pow that is good to a few ulps
    cannot produce different integers."""
    from decimal import Decimal as D, getcontext
    getcontext().prec = 50
    gamma = lambda x: x / D("12.92") if x <= D("0.04045") else ((x + D("0.055")) / D("1.055")) ** D("2.4")
    inv = lambda x: x * D("12.92") if x <= D("0.0031308") else D("1.055") * x ** (D(1) / D("2.4")) - D("0.055")
    cases = [("srgb_gamma", [D(2040) * gamma(D(i) / D(255)) for i in range(256)], 13.0),
             ("inv_gamma", [D(255) * inv(D(i) / D(4096)) for i in range(4096)], 5.0)]
    for name, exact, min_ulps in cases:
        tab = oracle.table(name)
        assert len(tab) == len(exact)
        worst = 1e9
        for i, v in enumerate(exact):
            assert int(v.to_integral_value(rounding="ROUND_HALF_EVEN")) == int(tab[i]), (name, i)
            if v > 0:
                worst = min(worst, float(abs((v % 1) - D("0.5"))) / float(np.spacing(np.float32(float(v)))))
        assert worst >= min_ulps, (name, worst)


# ---- exhaustive closed-form pins (every 8-bit colour; a few seconds of numpy each) -----------------------------------
def _all_colours():
    r, g = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for b in range(256):
        yield np.stack([np.full_like(r, b), g, r], -1)


def test_lab_8bit_against_the_float_cie_formulas_for_all_colours(oracle):
    """All 2^24 BGR colours through the oracle's RGB2Lab_b against the float CIE formulas OpenCV documents.  The
    fixed-point path (gamma table at 1/2040, cube-root table sampled at the same step, 15-bit descale) is NOT a 1-LSB
    approximation of the float formulas for dark colours -- the linear segment of f(t) is sampled too coarsely, L skips
    the codes 4, 13, 22 -- and that is OpenCV's documented 8-bit behaviour, not an oracle defect.  Measured and pinned:
    L within 1.55, a within 2.67, b within 1.70 LSB everywhere; 99.6 % of all channels within 1 LSB; every channel within
    1.5 LSB once L >= 60 (of 255)."""
    worst = np.zeros(3)
    over1 = np.zeros(3, np.int64)
    worst_bright = 0.0
    seen_L = np.zeros(256, bool)
    for img in _all_colours():
        lab = oracle.bgr2lab(img).astype(np.float64)
        d = np.abs(lab - _srgb_to_lab_float(img))
        worst = np.maximum(worst, d.reshape(-1, 3).max(0))
        over1 += (d.reshape(-1, 3) > 1.0).sum(0)
        bright = lab[..., 0] >= 60
        if bright.any():
            worst_bright = max(worst_bright, d[bright].max())
        seen_L[np.unique(lab[..., 0].astype(int))] = True
    assert worst[0] <= 1.55 and worst[1] <= 2.67 and worst[2] <= 1.70, worst
    assert over1.sum() / (3 * 2.0 ** 24) <= 0.004, over1
    assert worst_bright <= 1.5, worst_bright
    assert sorted(np.flatnonzero(~seen_L)) == [4, 13, 22]


def test_hsv_8bit_against_the_float_formulas_for_all_colours(oracle):
    """All 2^24 colours: RGB2HSV_b (12-bit reciprocal tables) stays within 0.64 (H, of 180), 0.53 (S) of the float
    formulas and V is exact; the 8-bit round trip BGR -> HSV -> BGR never moves a channel by more than 5 LSB (H keeps
    180 steps), 31 % of the colours return exactly."""
    mh = ms = mv = 0.0
    mrt, exact = 0, 0
    for img in _all_colours():
        b, g, r = [img[..., i].astype(np.float64) for i in range(3)]
        v = np.maximum(np.maximum(b, g), r)
        d = v - np.minimum(np.minimum(b, g), r)
        s = np.where(v > 0, d / np.where(v > 0, v, 1) * 255.0, 0.0)
        dd = np.where(d > 0, d, 1)
        h = np.where(v == r, (g - b) / dd, np.where(v == g, 2 + (b - r) / dd, 4 + (r - g) / dd)) * 30.0
        h = np.where(d > 0, h, 0.0)
        h = np.where(h < 0, h + 180.0, h)
        hsv = oracle.bgr2hsv(img)
        dh = np.abs(hsv[..., 0] - h)
        mh = max(mh, np.minimum(dh, 180 - dh).max())
        ms = max(ms, np.abs(hsv[..., 1] - s).max())
        mv = max(mv, np.abs(hsv[..., 2] - v).max())
        e = np.abs(oracle.hsv2bgr(hsv).astype(int) - img.astype(int)).max(-1)
        mrt = max(mrt, int(e.max()))
        exact += int((e == 0).sum())
    assert mh <= 0.641 and ms <= 0.53 and mv == 0, (mh, ms, mv)
    assert mrt <= 5 and exact / 2.0 ** 24 >= 0.30, (mrt, exact)


def test_remap_is_the_exactly_rounded_bilinear_of_the_quantised_coordinates(oracle):
    """cv::remap(INTER_LINEAR): once the map is quantised to 1/32 px the Q15 weights 32 (32 - fx)(32 - fy) ... are exact, so
    the result must be round-half-up of the exact bilinear interpolation: |oracle - float64 bilinear| <= 0.5, on random
    maps that also leave the image (BORDER_CONSTANT 0 taps)."""
    rng = np.random.default_rng(21)
    for (h, w) in ((37, 53), (64, 48)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        q = rng.integers(-3 * 32, (max(h, w) + 3) * 32, (2, 80, 90))
        mx, my = (q[0] / 32.0).astype(np.float32), (q[1] / 32.0).astype(np.float32)
        got = oracle.remap(img, mx, my).astype(np.float64)
        ix, iy = q[0] >> 5, q[1] >> 5
        fx, fy = (q[0] & 31) / 32.0, (q[1] & 31) / 32.0
        pad = np.zeros((h + 2, w + 2, 3))
        pad[1:-1, 1:-1] = img

        def tap(yy, xx):
            ok = (yy >= -1) & (yy <= h) & (xx >= -1) & (xx <= w)
            return np.where(ok[..., None], pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)], 0.0)
        ref = ((1 - fx) * (1 - fy))[..., None] * tap(iy, ix) + (fx * (1 - fy))[..., None] * tap(iy, ix + 1) + \
              ((1 - fx) * fy)[..., None] * tap(iy + 1, ix) + (fx * fy)[..., None] * tap(iy + 1, ix + 1)
        assert np.abs(got - ref).max() <= 0.5 + 1e-9
        assert np.array_equal(got, np.floor(ref + 0.5))


def test_ranges_the_fast_lab_kernel_relies_on(oracle):
    """rip_device.hpp vignette4: (1) every forward row of RGB2Lab_b sums to 4096, so the cube-root table index never exceeds
    2040 (tables of 2048 entries); (2) for every L' in [0, 255] and every (a, b) the forward transform can produce, x and
    z - 27500 fit int16 (operands of v_dot2_i32_i16) and the three descaled sums stay inside int32; (3) the re-scaled
    a / b formulas RN((25 fX + 1/8 - 25 fY) * 5 / 8192) + 128 and RN((25 fY - (25 fZ - 1/8)) / 4096) + 128 equal CV_DESCALE's
    round-half-up for every pair of table values."""
    fwd = oracle.table("fwd_coeffs").astype(np.int64).reshape(3, 3)
    assert fwd.sum(1).tolist() == [4096, 4096, 4096] and fwd.ravel().tolist() == [778, 1541, 1777, 296, 2929, 871, 3575, 448, 73]
    inv = oracle.table("inv_coeffs").astype(np.int64)
    assert inv.tolist() == [217, -836, 4715, -3773, 7684, 185, 12615, -6296, -2223]
    gmax = int(oracle.table("srgb_gamma").max())
    assert gmax == 2040 and ((fwd.clip(0) * gmax).sum(1) + 2048 >> 12).max() == 2040
    yf = oracle.table("lab_to_yf").astype(np.int64)
    y, ify = yf[0::2][:, None, None], yf[1::2][:, None, None]
    a, b = np.arange(42, 227)[None, :, None], np.arange(20, 224)[None, None, :]
    fx = ify + ((5 * a * 53687 + 128) >> 13) - 128 * 16384 // 500
    fz = ify - (((b * 41943 + 16) >> 9) - 128 * 16384 // 200 + 1)

    def ab_to_xz(i):
        lin = np.trunc(i * 108 / 841).astype(np.int64) - 16384 * 16 // 116 * 108 // 841
        return np.where(i <= 3390, lin, (i * i // 16384) * i // 16384)
    x, z = ab_to_xz(fx), ab_to_xz(fz)
    for i in (int(fx.min()), 0, 3390, 3391, int(fz.max())):
        assert int(ab_to_xz(np.array([i]))[0]) == oracle.ab_to_xz(i)
    assert -32768 <= x.min() and x.max() <= 32767
    assert -32768 <= z.min() - 27500 and z.max() - 27500 <= 32767, (z.min(), z.max())
    xb, yb, zb = np.broadcast_arrays(x, y, z)
    lo, hi = 1 << 40, -(1 << 40)
    for c in range(3):
        s = inv[c * 3] * xb + inv[c * 3 + 1] * yb + inv[c * 3 + 2] * zb + (1 << 13)
        assert np.abs(s).max() < 2 ** 31
        lo, hi = min(lo, int((s >> 14).min())), max(hi, int((s >> 14).max()))
    # (4) the descaled sums index sRGBInvGammaTab_b after a clamp to [0, 4095]; the fast kernel reads a table extended by its
    # saturated ends instead (VigTabs::invg, kInvgOff = 7256, kInvgExtN = 24012) and adds the table's LDS address
    # (< 160 KB) to the accumulators, whose sum is taken modulo 2^32 and shifted as an unsigned number: the index range,
    # and the address fits the 18 bits above the 14 fraction bits
    assert (lo, hi) == (-7253, 16753)
    assert 0 <= lo + 7256 and hi + 7256 < 24012
    assert hi + 7256 + 160 * 1024 < 2 ** 18
    cb = np.unique(oracle.table("cbrt")[:2048].astype(np.int64))
    fX, fY = cb[:, None], cb[None, :]
    for k, sh, mul in ((500, 15, 5.0 / 8192.0), (200, 15, 1.0 / 4096.0)):
        want = (k * (fX - fY) + 128 * 32768 + 16384) >> sh
        d = np.float32(25 * fX) + np.float32(0.125) - np.float32(25 * fY)  # exact: multiples of 1/8 below 2^20
        got = np.rint(d.astype(np.float64) * mul).astype(np.int64) + 128  # one rounding, as the FMA into 1.5 * 2^23 + 128 does
        assert np.array_equal(got, want), k
        assert not np.any(np.abs((d.astype(np.float64) * mul) % 1 - 0.5) < 1e-12)  # never a tie


def test_lab_inverse_offsets_as_one_multiply_add_each(oracle):
    """rip_device.hpp vignette_n: fx = ify + adiv(a) and fz = ify - bdiv(b) of Lab2RGBinteger as one 24-bit multiply-add
    each on top of one shared table word per L' (kBiasA, kBiasB, kFoldC; tools/lab_fold_search.py) -- emulated with the
    instructions' 24-bit operand and 32-bit wrap-around semantics for every (L', a, b), a and b over all of [0, 255]."""
    M = 1 << 32
    kA, kB16, bias_a, bias_b = 5 * 53687, 16 * 41943, -38465, -39212
    fold_c = (128 - 4194 * 8192 - 2 - kA * (0x400000 + bias_a)) % M
    assert fold_c == 0x40A41FD1
    ify = oracle.table("lab_to_yf").astype(np.int64)[1::2][:, None]
    word = (ify * 8192 + fold_c) % M
    s32 = lambda v: np.where(v % M >= M // 2, v % M - M, v % M)
    v = np.arange(256)[None, :]
    # the floats hold magic + 128 + bias + (a - 128): their low 24 bits are what v_mad_u32_u24 / v_mad_i32_i24 read
    magic = 12582912
    for bias in (bias_a, bias_b):
        f = np.float32(magic + 128 + bias) + (v - 128).astype(np.float32)
        assert np.all(f >= 2 ** 23) and np.all(f < 2 ** 24)
    va = (np.float32(magic + 128 + bias_a) + (v - 128).astype(np.float32)).view(np.uint32).astype(np.int64) & 0xFFFFFF
    vb = (np.float32(magic + 128 + bias_b) + (v - 128).astype(np.float32)).view(np.uint32).astype(np.int64) & 0xFFFFFF
    assert np.array_equal(va, 0x400000 + v + bias_a) and np.array_equal(vb, 0x400000 + v + bias_b) and vb.max() < 2 ** 23
    fx = s32(va * kA + word) >> 13
    fz = s32(word - vb * kB16) >> 13
    assert np.array_equal(fx, ify + ((5 * v * 53687 + 128) >> 13) - 128 * 16384 // 500)
    assert np.array_equal(fz, ify - (((v * 41943 + 16) >> 9) - 128 * 16384 // 200 + 1))


def test_lab_forward_indices_as_scaled_byte_offsets(oracle):
    """rip_device.hpp vignette_n: the cube-root table index (C . v + 2048) >> 12 leaves the fp32 arithmetic as a byte offset
    (4 i for the float table, 8 i for the float2 one): t = k (C . v) / 4096 + k / 2 - 1/2 + 2^-11 accumulated exactly,
    RN(t + 1.5 * 2^23) and an AND.  Emulated in float32 for every row of the forward matrix over a grid of table values
    that contains the extremes and every residue of the sums."""
    g = np.unique(oracle.table("srgb_gamma").astype(np.int64))
    C = oracle.table("fwd_coeffs").astype(np.int64).reshape(3, 3)
    rng = np.random.default_rng(5)
    v = np.concatenate([np.array([[0, 0, 0], [2040, 2040, 2040], [2040, 0, 0], [0, 2040, 0], [0, 0, 2040]]), g[rng.integers(0, len(g), (200000, 3))]])
    for k, init, mask in ((4, 1.5 + 2.0 ** -11, 0x1FFC), (8, 3.5 + 2.0 ** -10, 0x3FF8)):
        for r in range(3):
            acc = np.full(len(v), np.float32(init), np.float32)
            for j in range(3):  # v_fma_f32: exact products and sums here, so a float64 evaluation rounded once is the same
                t = acc.astype(np.float64) + v[:, j] * (np.float32(C[r, j]) * np.float32(k / 4096.0)).astype(np.float64)
                acc = t.astype(np.float32)
                assert np.array_equal(acc.astype(np.float64), t)  # every partial sum is exactly representable
            bits = (acc + np.float32(12582912.0)).view(np.uint32)
            want = (v @ C[r] + 2048) >> 12
            assert np.array_equal((bits & mask).astype(np.int64), k * want)


def test_four_tap_average_from_two_tap_averages():
    """rip_device.hpp debayer_row: (a + b + c + d + 2) >> 2 == lerp(lerp(a, b), lerp(c, d), ~((a ^ b) | (c ^ d))) with
    lerp(x, y, r) = (x + y + (r & 1)) >> 1 -- the only property of v_lerp_u8 the demosaic uses.  All 2^32 byte quadruples
    reduce to the 2^16 classes of (a + b, c + d) x parities; checked over every (a, b) pair against every (c, d) pair of a
    set that contains all sums and parities."""
    a, b = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    a, b = a.ravel()[:, None], b.ravel()[:, None]
    cd = np.array([(c, d) for c in (0, 1, 2, 3, 127, 128, 254, 255) for d in (0, 1, 2, 126, 129, 253, 254, 255)])
    c, d = cd[:, 0][None, :], cd[:, 1][None, :]
    lerp = lambda x, y, r: (x + y + (r & 1)) >> 1
    got = lerp(lerp(a, b, 1), lerp(c, d, 1), ~((a ^ b) | (c ^ d)))
    assert np.array_equal(got, (a + b + c + d + 2) >> 2)
