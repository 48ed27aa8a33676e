"""tools/color_calibration.py: the fit recovers a known colour transform and its YAML goes through the
library's loader into the stage that applies it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import color_calibration as cc  # noqa: E402

from raw_image_pipeline_amd import RawImagePipeline  # noqa: E402


def test_fit_recovers_a_known_transform_and_the_loader_reads_the_file(tmp_path, oracle):
    rng = np.random.default_rng(5)
    true_m = np.array([[1.6, -0.3, 0.05], [-0.1, 1.2, -0.08], [0.02, -0.4, 1.7]])
    true_b = np.array([3.0, -2.0, 5.0])
    measured = rng.uniform(20, 200, (24, 3))
    reference = measured @ true_m.T + true_b + rng.normal(0, 0.2, (24, 3))
    np.save(tmp_path / "in.npy", measured)
    np.savetxt(tmp_path / "ref.txt", reference)
    assert cc.main(["-i", str(tmp_path / "in.npy"), "-r", str(tmp_path / "ref.txt"), "--compute-bias", "-o", str(tmp_path)]) == 0
    p = RawImagePipeline(False, device=-1)
    p.load_color_calibration(str(tmp_path / "color_calibration.yaml"))
    m = np.asarray(p.get_color_calibration_matrix()).reshape(3, 3)
    b = np.asarray(p.get_color_calibration_bias()).ravel()[:3]
    assert np.abs(m - true_m).max() < 0.02 and np.abs(b - true_b).max() < 1.5, (m, b)
    # the oracle's colour stage with the fitted file maps the measured patches onto the reference ones
    img = np.clip(np.rint(measured), 0, 255).astype(np.uint8).reshape(4, 6, 3)
    out = oracle.color_matrix(img, m.ravel(), b) if hasattr(oracle, "color_matrix") else None
    if out is not None:
        want = np.clip(np.rint(img.astype(np.float64).reshape(-1, 3) @ true_m.T + true_b), 0, 255).reshape(4, 6, 3)
        assert np.abs(out.astype(int) - want).max() <= 2


def test_without_bias_matches_plain_least_squares(tmp_path):
    rng = np.random.default_rng(6)
    measured = rng.uniform(10, 240, (30, 3))
    reference = measured @ np.array([[1.1, 0.1, 0.0], [0.0, 0.9, 0.1], [0.05, 0.0, 1.3]]).T
    m, b, _ = cc.find_color_calibration(measured, reference)
    lsq = np.linalg.lstsq(measured, reference, rcond=None)[0].T
    assert np.abs(m - lsq).max() < 1e-3 and not b.any()
