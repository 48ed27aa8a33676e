#!/usr/bin/env python3
"""Counterpart of the reference's raw_image_pipeline_python/scripts/apply_pipeline.py on synthetic data:
construct the pipeline from three YAML files, print the calibration it holds, run process() (returns a
copy) and apply() (re-seats the input) on a colour image, and write both results as PNG when Pillow is
available (the reference uses cv2 / rospkg and its bundled alphasense.png, which are not shipped here).

usage: python examples/apply_pipeline.py [--device 0] [--out-dir .]
"""
import argparse
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from raw_image_pipeline_amd import RawImagePipeline, synth  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out-dir", default=".")
    args = ap.parse_args()

    w, h = 720, 540  # the size of the reference's example calibration
    img = synth.gen_scene_bgr(w, h, seed=7)
    tmp = tempfile.mkdtemp(prefix="rip_demo_")
    calib_file = os.path.join(tmp, "calib.yaml")
    color_calib_file = os.path.join(tmp, "color_calib.yaml")
    param_file = os.path.join(tmp, "params.yaml")
    with open(calib_file, "w") as f:
        f.write(synth.calibration_yaml(synth.camera_model(w, h)))
    with open(color_calib_file, "w") as f:
        f.write("matrix:\n  rows: 3\n  cols: 3\n  data: %s\nbias:\n  rows: 3\n  cols: 1\n  data: [0.0, 0.0, 0.0]\n" % synth.COLOR_MATRIX)
    with open(param_file, "w") as f:
        f.write("debayer:\n  enabled: true\n  encoding: \"auto\"\nflip:\n  enabled: false\n  angle: 0\n"
                "white_balance:\n  enabled: true\n  method: \"grey_world\"\n  saturation_bright_thr: 0.8\n  saturation_dark_thr: 0.2\n"
                "color_calibration:\n  enabled: false\ngamma_correction:\n  enabled: false\n  method: \"custom\"\n  k: 0.8\n"
                "vignetting_correction:\n  enabled: false\n  scale: 1.5\n  a2: 1e-3\n  a4: 1e-6\n"
                "color_enhancer:\n  hue_gain: 1.0\n  saturation_gain: 1.5\n  value_gain: 1.0\n"
                "undistortion:\n  enabled: true\n  balance: 0.0\n  fov_scale: 0.8\n")

    proc = RawImagePipeline(False, param_file, calib_file, color_calib_file, device=args.device)

    # camera-info getters of both geometries (what the ROS node copies into sensor_msgs/CameraInfo)
    fields = ("image_height", "image_width", "distortion_model", "camera_matrix", "distortion_coefficients", "rectification_matrix",
              "projection_matrix")
    for title, side in (("distorted input (dist)", "dist"), ("after undistortion (rect)", "rect")):
        print("%s:" % title)
        for name in fields:
            value = getattr(proc, "get_%s_%s" % (side, name))()
            print("  %-28s %s" % (side + "_" + name, np.array2string(np.asarray(value), precision=4, suppress_small=True)
                                   if not isinstance(value, (str, int)) else value))

    img2 = proc.process(img, "bgr8")   # returns a new image, `img` stays as it was
    before = img.copy()
    out = proc.apply(img, "bgr8")      # in place: `img` now holds the result
    assert np.array_equal(out, img2) and np.array_equal(img, out) and not np.array_equal(before, img)
    try:
        from PIL import Image
        Image.fromarray(img[..., ::-1]).save(os.path.join(args.out_dir, "output_apply.png"))
        Image.fromarray(img2[..., ::-1]).save(os.path.join(args.out_dir, "output_process.png"))
        print("wrote output_apply.png / output_process.png")
    except ImportError:
        np.save(os.path.join(args.out_dir, "output_process.npy"), img2)
        print("Pillow not available: wrote output_process.npy")


if __name__ == "__main__":
    main()
