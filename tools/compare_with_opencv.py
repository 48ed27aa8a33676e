#!/usr/bin/env python3
"""Pins the CPU oracle (oracle/rip_oracle.c) to a real OpenCV, stage by stage -- whenever one is importable.

The reference's arithmetic lives in OpenCV 4.2 + opencv_contrib xphoto, which is not installed in the build
container or on the GPU box, so `oracle/` is a restatement written from the published algorithms ("parity unpinned").
This script is the way off that status: on any machine with `cv2` (ideally 4.2.x with contrib) run

    python tools/compare_with_opencv.py                 # report, exit code 1 on any difference beyond the stated bar
    python tools/compare_with_opencv.py --write         # also freeze OpenCV's outputs into tests/golden/opencv_vectors.npz

Every stage calls OpenCV exactly the way the reference's module does (file:line in the stage table below) on seeded
synthetic inputs (raw_image_pipeline_amd/synth.py), and compares with the matching `ripo_*` function.  The frozen vectors
are plain data (inputs are regenerated from seeds, outputs stored); tests/test_golden.py::test_opencv_vectors checks the
oracle against them on every run once the file exists.  Without cv2 the script says so and exits 0: it is never required.
Nothing here is imported by the product (raw_image_pipeline_amd/) or by bench.py."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "opencv_vectors.npz")

# north-star bars (BASELINE.json): bit-exact for flip and LUT gamma, +-1 LSB for interpolation; the build declares
# +-1 LSB for the float / fixed-point colour stages and expects 0 everywhere
BARS = {"debayer": 1, "rgb8_swap": 0, "flip": 0, "grey_world": 1, "simple": 1, "color_matrix": 1, "gamma": 0, "bgr2lab": 1, "lab2bgr": 1,
        "vignetting": 1, "bgr2hsv": 1, "hsv2bgr": 1, "color_enhancer": 1, "remap": 1, "resize_360x270": 1}


class Recorder:
    """Wraps the oracle module: every call is executed and remembered as (function name, arguments) so that the frozen
    vectors carry what is needed to re-run the oracle side without OpenCV (tests/test_golden.py)."""

    def __init__(self, O):
        self._o, self.last = O, None

    def __getattr__(self, name):
        fn = getattr(self._o, name)

        def call(*args):
            self.last = (name, [a.copy() if isinstance(a, np.ndarray) else a for a in args])
            return fn(*args)
        return call


def stages(cv2, O, synth):
    """Yields (name, reference call site, opencv_output, oracle_output)."""
    rng = np.random.default_rng(2024)
    scene = synth.gen_scene_bgr(192, 144, seed=7)
    noise = rng.integers(0, 256, (144, 192, 3), dtype=np.uint8)
    # debayer.cpp:45-79: cv::demosaicing with the ROS name -> OpenCV code table of the reference, then cvtColor(RGB2BGR)
    codes = {"bayer_bggr8": cv2.COLOR_BayerBG2BGR, "bayer_gbrg8": cv2.COLOR_BayerGB2BGR, "bayer_grbg8": cv2.COLOR_BayerGR2BGR,
             "bayer_rggb8": cv2.COLOR_BayerRG2BGR}
    for enc, code in codes.items():
        for kind in ("scene", "uniform"):
            for (w, h) in ((64, 48), (130, 98), (31, 17)):
                raw = synth.gen_frame(w, h, enc, seed=11, kind=kind)
                ref = cv2.cvtColor(cv2.demosaicing(raw, code), cv2.COLOR_RGB2BGR)
                yield "debayer", "debayer.cpp:48-67 %s %s %dx%d" % (enc, kind, w, h), ref, O.debayer(raw, enc)
    yield "rgb8_swap", "debayer.cpp:72-73", cv2.cvtColor(noise, cv2.COLOR_RGB2BGR), O.swap_rb(noise)
    # flip.cpp:37-58
    yield "flip", "flip.cpp:45-47 (180)", cv2.flip(noise, -1), O.flip(noise, 180)
    yield "flip", "flip.cpp:40-43 (90)", cv2.flip(cv2.transpose(noise), 1), O.flip(noise, 90)
    yield "flip", "flip.cpp:49-52 (270)", cv2.flip(cv2.transpose(noise), 0), O.flip(noise, 270)
    # white_balance.cpp:52-64 (xphoto)
    if hasattr(cv2, "xphoto"):
        for img, tag in ((scene, "scene"), (noise, "uniform")):
            for thr in (0.8, 0.98, 0.5):
                wb = cv2.xphoto.createGrayworldWB()
                wb.setSaturationThreshold(thr)
                yield "grey_world", "white_balance.cpp:59-64 thr %.2f %s" % (thr, tag), wb.balanceWhite(img), O.wb_grayworld(img.copy(), thr)
            for p in (10.0, 2.0, 0.5):
                wb = cv2.xphoto.createSimpleWB()
                wb.setP(p)
                yield "simple", "white_balance.cpp:52-57 p %.1f %s" % (p, tag), wb.balanceWhite(img), O.wb_simple(img.copy(), p)
    else:
        print("  (cv2.xphoto missing: grey_world / simple not compared -- install opencv-contrib)")
    # color_calibration.cpp:91-104: reshape -> convertTo(32F) -> gemm with M^T -> + bias -> convertTo(8U)
    M = np.asarray(synth.COLOR_MATRIX, np.float64).reshape(3, 3)
    for bias in ((0.0, 0.0, 0.0), (3.5, -2.0, 0.25)):
        for img, tag in ((scene, "scene"), (noise, "uniform")):
            flat = img.reshape(-1, 3).astype(np.float32)
            mixed = cv2.gemm(flat, np.ascontiguousarray(M.astype(np.float32).T), 1.0, None, 0.0)
            image_f = mixed.reshape(img.shape[0], img.shape[1], 3) + np.asarray(bias, np.float32)
            ref = np.clip(np.rint(image_f), 0, 255).astype(np.uint8)  # convertTo(CV_8UC3): saturate_cast<uchar> = cvRound + clamp
            yield "color_matrix", "color_calibration.cpp:93-103 bias %s %s" % (bias, tag), ref, O.color_matrix(img.copy(), M.ravel(), bias)
    # gamma_correction.cpp:35-60
    for k in (0.8, 0.9, 1.0, 2.2):
        lut = np.array([min(255, max(0, int(np.rint(np.float64(np.float32(np.float32(i / 255.0) ** np.float64(k))) * 255.0)))) for i in range(256)], np.uint8)
        yield "gamma", "gamma_correction.cpp:38-42,55 k %.1f" % k, cv2.LUT(noise, lut), O.apply_lut(noise.copy(), O.gamma_lut(k))
    # vignetting_correction.cpp:68-93 and its two conversions
    allc = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"), -1).reshape(-1, 1, 3).astype(np.uint8)
    for img, tag in ((scene, "scene"), (noise, "uniform"), (allc, "colour cube step 5")):
        lab = cv2.cvtColor(img, cv2.COLOR_BGR2Lab)
        yield "bgr2lab", "vignetting_correction.cpp:73 %s" % tag, lab, O.bgr2lab(img)
        yield "lab2bgr", "vignetting_correction.cpp:92 %s" % tag, cv2.cvtColor(lab, cv2.COLOR_Lab2BGR), O.lab2bgr(lab)
    for (s, a2, a4) in ((1.5, 1e-3, 1e-6), (0.5, 2e-4, 0.0)):
        h, w = scene.shape[:2]
        mask = O.vignetting_mask(h, w, s, a2, a4)
        lab = cv2.cvtColor(scene, cv2.COLOR_BGR2Lab)
        L = cv2.multiply(lab[..., 0].astype(np.float32), mask, dtype=cv2.CV_32F)
        lab[..., 0] = np.clip(np.rint(L), 0, 255).astype(np.uint8)
        yield "vignetting", "vignetting_correction.cpp:73-92 (%g, %g, %g)" % (s, a2, a4), cv2.cvtColor(lab, cv2.COLOR_Lab2BGR), O.vignetting(scene.copy(), mask)
    # color_enhancer.cpp:38-47
    for img, tag in ((scene, "scene"), (noise, "uniform"), (allc, "colour cube step 5")):
        hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
        yield "bgr2hsv", "color_enhancer.cpp:40 %s" % tag, hsv, O.bgr2hsv(img)
        yield "hsv2bgr", "color_enhancer.cpp:46 %s" % tag, cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR), O.hsv2bgr(hsv)
        for gains in ((1.0, 1.2, 1.0), (1.1, 0.8, 1.3)):
            en = cv2.multiply(hsv, np.array(gains + (0.0,), np.float64))  # cv::multiply(Mat, Scalar)
            yield "color_enhancer", "color_enhancer.cpp:40-46 gains %s %s" % (gains, tag), cv2.cvtColor(en, cv2.COLOR_HSV2BGR), O.color_enhance(img.copy(), *gains)
    # undistortion.cpp:197-220 (maps) and :242 (remap)
    for (w, h) in ((192, 144), (130, 98)):
        cam = synth.camera_model(w, h)
        K = np.asarray(cam["K"], np.float64).reshape(3, 3)
        D = np.asarray(cam["D"], np.float64).reshape(4, 1)
        R = np.asarray(cam["R"], np.float64).reshape(3, 3)
        for balance in (0.0, 0.5):
            newK = cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, (w, h), R, balance=balance, new_size=(w, h), fov_scale=1.0)
            mx, my = cv2.fisheye.initUndistortRectifyMap(K, D, R, newK, (w, h), cv2.CV_32FC1)
            oK = O.fisheye_new_camera_matrix(cam["K"], cam["D"], (w, h), cam["R"], balance, None, 1.0)
            omx, omy = O.fisheye_maps(cam["K"], cam["D"], cam["R"], oK, (w, h))
            yield "fisheye_newK", "undistortion.cpp:199-208 balance %.1f" % balance, newK, np.asarray(oK).reshape(3, 3)
            yield "fisheye_maps", "undistortion.cpp:212-220 balance %.1f" % balance, np.stack([mx, my]), np.stack([omx, omy])
            img = synth.gen_scene_bgr(w, h, seed=3)
            yield "remap", "undistortion.cpp:242 balance %.1f" % balance, cv2.remap(img, mx, my, cv2.INTER_LINEAR, borderMode=cv2.BORDER_CONSTANT, borderValue=0), O.remap(img, mx, my)
    # convolutional_color_constancy.cpp:97 cv::resize(image, small, Size(360, 270))
    for (w, h) in ((720, 540), (1440, 1080), (1000, 700), (384, 240)):
        img = synth.gen_scene_bgr(w, h, seed=9)
        yield "resize_360x270", "convolutional_color_constancy.cpp:97 from %dx%d" % (w, h), cv2.resize(img, (360, 270)), O.resize_linear(img, 270, 360)


def contraction_fingerprint(cv2, O, synth):
    """Which floating-point contraction model does THIS OpenCV build implement?  The float stages (colour matrix: cv::gemm's
    3-term dot product; colour enhancer: HSV2RGB's v * (1 - s * f)) are run over all 2^24 colours through OpenCV and through
    the oracle in its three models (oracle/rip_oracle.c: 0 none = what the HIP kernels implement, 1 GCC / Clang fma order, 2
    the other association); the model with zero mismatches is the build's.  tests/test_fp_contraction.py bounds the models
    against each other (<= 1 LSB, a few values per million), so whichever matches, the kernels are within that bound."""
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], -1).astype(np.uint8).reshape(4096, 4096, 3)
    M = np.asarray(synth.COLOR_MATRIX, np.float64).reshape(3, 3)
    flat = img.reshape(-1, 3).astype(np.float32)
    mixed = cv2.gemm(flat, np.ascontiguousarray(M.astype(np.float32).T), 1.0, None, 0.0)
    ref_cc = np.clip(np.rint(mixed), 0, 255).astype(np.uint8).reshape(img.shape)
    hsv = cv2.cvtColor(img, cv2.COLOR_BGR2HSV)
    ref_ce = cv2.cvtColor(cv2.multiply(hsv, np.array((1.0, 1.2, 1.0, 0.0), np.float64)), cv2.COLOR_HSV2BGR)
    out = {}
    for stage, ref, run in (("color_matrix", ref_cc, lambda: O.color_matrix(img, M.ravel(), (0.0, 0.0, 0.0))),
                            ("color_enhancer", ref_ce, lambda: O.color_enhance(img, 1.0, 1.2, 1.0))):
        mism = []
        for mode in (0, 1, 2):
            with O.fp_contraction(mode):
                got = run()
            mism.append(int((got != ref).sum()))
        best = int(np.argmin(mism))
        out[stage] = (best, mism)
        print("contraction fingerprint %-15s mismatching channel values per oracle model 0 / 1 / 2: %s -> this OpenCV build %s"
              % (stage, mism, ("matches model %d exactly" % best) if mism[best] == 0 else "matches NO model exactly (closest: %d)" % best))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="freeze OpenCV's outputs into tests/golden/opencv_vectors.npz")
    args = ap.parse_args()
    try:
        import cv2
    except ImportError:
        print("cv2 is not importable here: nothing compared (parity of oracle/ against OpenCV stays unpinned).")
        return 0
    import oracle as O
    from raw_image_pipeline_amd import synth
    O.build()
    O = Recorder(O)
    print("OpenCV %s (the reference pins 4.2.0 + contrib)" % cv2.__version__)
    worst = {}
    frozen = {}
    failed = False
    for i, (name, where, ref, got) in enumerate(stages(cv2, O, synth)):
        ref, got = np.asarray(ref), np.asarray(got)
        if ref.dtype == np.uint8:
            d = int(np.abs(ref.astype(np.int16) - got.astype(np.int16)).max()) if ref.shape == got.shape else 999
            frac = float((ref != got).mean()) if ref.shape == got.shape else 1.0
            bar = BARS[name]
        else:  # float results (camera matrix, maps): report the largest difference in ulps of float32 / relative
            d = float(np.abs(ref.astype(np.float64) - got.astype(np.float64)).max())
            frac = float((ref.astype(np.float32) != got.astype(np.float32)).mean())
            bar = 1e-3  # pixels; the kernels quantise map coordinates to 1/32 px
        ok = d <= bar
        failed |= not ok
        worst[name] = max(worst.get(name, 0), d)
        print("%-4s %-16s max |diff| %-8g differing %.4f%%   %s" % ("ok" if ok else "FAIL", name, d, 100 * frac, where))
        frozen["%03d_%s" % (i, name)] = ref
        frozen["%03d_%s_where" % (i, name)] = np.array(where)
        if ref.dtype == np.uint8 and O.last is not None:  # replay record: oracle function + arguments
            frozen["%03d_%s_fn" % (i, name)] = np.array(O.last[0])
            for k, a in enumerate(O.last[1]):
                frozen["%03d_%s_arg%d" % (i, name, k)] = np.asarray(a)
            frozen["%03d_%s_bar" % (i, name)] = np.array(bar)
    print("\nworst per stage:", worst)
    import oracle as plain_oracle
    for stage, (best, mism) in contraction_fingerprint(cv2, plain_oracle, synth).items():
        frozen["fp_contraction_%s_model" % stage] = np.array(best)
        frozen["fp_contraction_%s_mismatches" % stage] = np.array(mism)
    if args.write:
        np.savez_compressed(OUT, opencv_version=np.array(cv2.__version__), **frozen)
        print("wrote", OUT)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
