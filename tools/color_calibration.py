#!/usr/bin/env python3
"""Offline colour calibration: the numeric part of the reference's raw_image_pipeline_python/scripts/color_calibration.py
(SURVEY 8(f)-4), without its OpenCV / ArUco front end.  Fits the 3x3 matrix (and optionally a bias) that maps measured
patch colours to reference patch colours and writes the YAML file loadColorCalibration() / rip_load_color_calibration
read (color_calibration.cpp:52-72).

Patch colours come either from two N x 3 tables (.npy, or text with three numbers per line), or from two images that
show the same colour chart already cropped and rectified, sampled on a regular grid (--grid ROWSxCOLS, the central
half of every cell is averaged; needs Pillow).  Channels are B, G, R everywhere, as in the pipeline.

The fit is the reference's: scipy.optimize.least_squares over the 9 (+3) parameters, start 0.1 * I, residual
|| C x + 255 b - ref ||, selectable robust loss.  One difference, on purpose: the reference script stores b while the
C++ module adds the stored bias as is (image_f += bias, color_calibration.cpp:99-101); this tool stores 255 * b, the
number the pipeline applies (--reference-bias-units keeps the script's convention).

usage: color_calibration.py --input measured.npy --ref reference.npy [--compute-bias] [--loss soft_l1] -o out_dir
       color_calibration.py --input shot.png --ref chart.png --grid 4x6 -o out_dir
"""
import argparse
import os
import sys

import numpy as np


def load_patches(path, grid):
    if path.endswith(".npy"):
        a = np.load(path)
    elif path.lower().endswith((".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff")):
        if grid is None:
            raise SystemExit("--grid ROWSxCOLS is required when patch colours are sampled from images")
        from PIL import Image
        img = np.asarray(Image.open(path).convert("RGB"), dtype=np.float64)[..., ::-1]  # BGR
        rows, cols = grid
        h, w = img.shape[:2]
        a = []
        for r in range(rows):
            for c in range(cols):
                y0, y1 = int((r + 0.25) * h / rows), int((r + 0.75) * h / rows)
                x0, x1 = int((c + 0.25) * w / cols), int((c + 0.75) * w / cols)
                a.append(img[y0:max(y1, y0 + 1), x0:max(x1, x0 + 1)].reshape(-1, 3).mean(axis=0))
        a = np.array(a)
    else:
        a = np.loadtxt(path)
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise SystemExit("%s: expected N x 3 patch colours, got %s" % (path, a.shape))
    return a


def find_color_calibration(measured, reference, loss="linear", compute_bias=False):
    """Returns (matrix 3x3, bias 3 in pipeline units, scipy result).  measured, reference: N x 3, B G R."""
    from scipy.optimize import least_squares
    if measured.shape != reference.shape:
        raise ValueError("measured and reference patch tables differ in shape")
    x_in, x_ref = measured.T, reference.T

    def residual(x):
        out = x[:9].reshape(3, 3) @ x_in
        if compute_bias:
            out = out + 255.0 * x[9:].reshape(3, 1)
        return np.array([np.linalg.norm(out - x_ref)])

    x0 = np.zeros(12)
    x0[:9] = 0.1 * np.eye(3).ravel()
    sol = least_squares(residual, x0, loss=loss)
    matrix = sol.x[:9].reshape(3, 3).astype(np.float32)
    bias = (255.0 * sol.x[9:] if compute_bias else np.zeros(3)).astype(np.float32)
    return matrix, bias, sol


def apply_color_calibration(matrix, bias, bgr):
    """What the pipeline's colour stage computes (before rounding): C x + bias per pixel."""
    return np.asarray(bgr, dtype=np.float64) @ np.asarray(matrix, np.float64).T + np.asarray(bias, np.float64)


def write_yaml(path, matrix, bias):
    fmt = lambda v: "[" + ", ".join(repr(float(x)) for x in np.asarray(v).ravel()) + "]"
    with open(path, "w") as f:
        f.write("# colour calibration fitted by tools/color_calibration.py (rows: B, G, R out; columns: B, G, R in)\n")
        f.write("matrix:\n  rows: 3\n  cols: 3\n  data: %s\n" % fmt(matrix))
        f.write("bias:\n  rows: 3\n  cols: 1\n  data: %s\n" % fmt(bias))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-i", "--input", required=True, help="measured patch colours (.npy / text / image)")
    ap.add_argument("-r", "--ref", required=True, help="reference patch colours (.npy / text / image)")
    ap.add_argument("--grid", help="ROWSxCOLS of the colour chart when sampling images, e.g. 4x6")
    ap.add_argument("-o", "--output-path", default=".")
    ap.add_argument("-p", "--prefix", default="")
    ap.add_argument("--loss", default="linear", choices=["linear", "soft_l1", "huber", "cauchy", "arctan"])
    ap.add_argument("--compute-bias", action="store_true")
    ap.add_argument("--reference-bias-units", action="store_true", help="store b instead of 255 * b, as the reference script does")
    args = ap.parse_args(argv)
    grid = tuple(int(v) for v in args.grid.lower().split("x")) if args.grid else None
    measured, reference = load_patches(args.input, grid), load_patches(args.ref, grid)
    matrix, bias, sol = find_color_calibration(measured, reference, args.loss, args.compute_bias)
    err = np.abs(apply_color_calibration(matrix, bias, measured) - reference)
    print("fit over %d patches: cost %.4g, mean |error| %.2f, max %.2f (8-bit units)" % (len(measured), sol.cost, err.mean(), err.max()))
    out = os.path.join(args.output_path, args.prefix + "color_calibration.yaml")
    write_yaml(out, matrix, bias / 255.0 if args.reference_bias_units else bias)
    print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
