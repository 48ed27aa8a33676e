"""The reference's OWN configuration files through this repository's loaders (CPU; skipped where the reference tree is not
mounted, i.e. on the GPU box; the files are read in place, nothing is copied).

A user of the reference starts it as `RawImagePipeline(use_gpu, params_path, calibration_path, color_calibration_path)`
(raw_image_pipeline.hpp:40-41, .cpp:27-40) with the YAML files of raw_image_pipeline/config/.  A drop-in has to accept
exactly those files: this test parses them independently with PyYAML and checks every value the reference's getters expose
(hpp:109-137), plus the rectified camera matrix against the oracle's restatement of
cv::fisheye::estimateNewCameraMatrixForUndistortRectify (undistortion.cpp:199-220) fed with the file's numbers."""
import os

import numpy as np
import pytest
import yaml

from raw_image_pipeline_amd import RawImagePipeline

CFG = "/root/reference/raw_image_pipeline/config"
PARAMS = CFG + "/pipeline_params_example.yaml"
COLOR = CFG + "/alphasense_color_calib_example.yaml"
CALIBS = [CFG + "/alphasense_calib_1.6mp_example.yaml", CFG + "/alphasense_calib_example.yaml"]

pytestmark = pytest.mark.skipif(not os.path.exists(PARAMS), reason="reference tree not mounted (GPU box)")


def _yaml(path):
    with open(path) as f:
        return yaml.safe_load(f)


@pytest.mark.parametrize("calib_path", CALIBS)
def test_four_argument_constructor_with_the_reference_config_files(oracle, calib_path):
    prm, cal, col = _yaml(PARAMS), _yaml(calib_path), _yaml(COLOR)
    p = RawImagePipeline(False, PARAMS, calib_path, COLOR, device=-1)  # parameter handling needs no GPU
    # enable flags (loadParams, .cpp:56-160)
    assert p.is_debayer_enabled() == bool(prm["debayer"]["enabled"])
    assert p.is_flip_enabled() == bool(prm["flip"]["enabled"])
    assert p.is_white_balance_enabled() == bool(prm["white_balance"]["enabled"])
    assert p.is_color_calibration_enabled() == bool(prm["color_calibration"]["enabled"])
    assert p.is_gamma_correction_enabled() == bool(prm["gamma_correction"]["enabled"])
    assert p.is_vignetting_correction_enabled() == bool(prm["vignetting_correction"]["enabled"])
    assert p.is_undistortion_enabled() == bool(prm["undistortion"]["enabled"])
    # the example file has no color_enhancer/enabled key: utils::get falls back to the module default (off)
    assert p.is_color_enhancer_enabled() == bool(prm["color_enhancer"].get("enabled", False))
    # colour calibration (color_calibration.cpp:26-60): row-major 3x3 + bias
    assert np.array_equal(p.get_color_calibration_matrix(), np.asarray(col["matrix"]["data"], np.float32).astype(np.float64).reshape(3, 3))
    # getColorCalibrationBias() wraps a cv::Scalar: four rows, the fourth zero (color_calibration.cpp:121-123)
    assert np.array_equal(np.asarray(p.get_color_calibration_bias()).ravel(), np.asarray(list(col["bias"]["data"]) + [0.0], np.float32).astype(np.float64))
    # distorted camera (undistortion.cpp:150-176)
    w, h = int(cal["image_width"]), int(cal["image_height"])
    K = np.asarray(cal["camera_matrix"]["data"], np.float64).reshape(3, 3)
    D = np.asarray(cal["distortion_coefficients"]["data"], np.float64)
    R = np.asarray(cal["rectification_matrix"]["data"], np.float64).reshape(3, 3)
    P = np.asarray(cal["projection_matrix"]["data"], np.float64).reshape(3, 4)
    assert (p.get_dist_image_width(), p.get_dist_image_height()) == (w, h)
    assert p.get_dist_distortion_model() == cal["distortion_model"] == "equidistant"
    assert np.array_equal(p.get_dist_camera_matrix(), K)
    assert np.array_equal(np.asarray(p.get_dist_distortion_coefficients()).ravel(), D)
    assert np.array_equal(p.get_dist_rectification_matrix(), R)
    assert np.array_equal(p.get_dist_projection_matrix(), P)
    # rectified camera: new K from the file's balance / fov_scale, zero distortion, the model-string quirk of the reference
    und = prm["undistortion"]
    newK = oracle.fisheye_new_camera_matrix(K, D, (w, h), R, float(und["balance"]), None, float(und["fov_scale"]))
    assert np.array_equal(p.get_rect_camera_matrix(), newK)
    assert np.array_equal(p.get_rect_projection_matrix()[:, :3], newK)
    assert (p.get_rect_image_width(), p.get_rect_image_height()) == (w, h)
    assert not np.asarray(p.get_rect_distortion_coefficients()).any()
    assert p.get_rect_distortion_model() == "none"
    # the maps the first frame would be remapped with equal the oracle's initUndistortRectifyMap restatement
    mx, my = p.get_undistortion_maps()
    ox, oy = oracle.fisheye_maps(K, D, R, newK, (w, h))
    assert np.array_equal(mx, ox) and np.array_equal(my, oy)


def test_load_params_alone_keeps_the_compiled_in_calibration_defaults():
    """The ROS node's path (raw_image_pipeline_ros.cpp:52 -> one-argument constructor, then setters) and the Python
    binding's load_params (raw_image_pipeline_python.cpp:25): the params file alone, no camera / colour calibration."""
    p = RawImagePipeline(False, device=-1)
    p.load_params(PARAMS)
    prm = _yaml(PARAMS)
    assert p.is_white_balance_enabled() == bool(prm["white_balance"]["enabled"])
    assert p.is_undistortion_enabled() == bool(prm["undistortion"]["enabled"])
    assert p.get_dist_image_width() == 0 or p.get_dist_distortion_model() in ("none", "equidistant")
