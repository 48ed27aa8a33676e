"""How much can the reference's float stages move with the FMA contraction of the OpenCV build it links?

The oracle (and the HIP kernels) implement the uncontracted sequence: every product and every sum rounded.  GCC and Clang
contract `a*b + c` into one fma wherever the target has one (every aarch64 build -- the reference's Jetson target --, the
AVX2 / FMA3 dispatch variants on x86-64), so a given OpenCV binary may round differently.  This file measures the
difference between the oracle's three contraction models (oracle/rip_oracle.c: 0 none, 1 GCC / Clang order, 2 the other
association) stage by stage, exhaustively where the domain allows, and asserts the bound PARITY.md and BASELINE.md
section 4 state: at most 1 LSB at the output of the stage that contains the float expression.  The counts are printed
(pytest -s) and are the numbers quoted in DESIGN.md.  CPU only."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from helpers import oracle_run  # noqa: E402
import make_golden  # noqa: E402
from raw_image_pipeline_amd import synth  # noqa: E402


def all_colours():
    """All 2^24 BGR triples as a 4096 x 4096 image."""
    v = np.arange(1 << 24, dtype=np.uint32)
    img = np.empty((1 << 24, 3), np.uint8)
    img[:, 0] = v & 0xFF
    img[:, 1] = (v >> 8) & 0xFF
    img[:, 2] = (v >> 16) & 0xFF
    return img.reshape(4096, 4096, 3)


def diff_stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int((d > 0).sum()), int(d.max()), d.size


@pytest.fixture(scope="module")
def colours():
    return all_colours()


MATRICES = {
    "example (config/alphasense_color_calib_example.yaml)": synth.COLOR_MATRIX,
    "identity": [1, 0, 0, 0, 1, 0, 0, 0, 1],
    "seeded random": list(np.random.default_rng(5).uniform(-0.6, 1.9, 9)),
}


@pytest.mark.parametrize("name", sorted(MATRICES))
def test_colour_matrix_all_colours_within_one_lsb(oracle, colours, name):
    """color_calibration.cpp:93-103 -> cv::gemm's 3-term dot product in float.  All 2^24 inputs, both contracted forms."""
    m = MATRICES[name]
    base = oracle.color_matrix(colours, m, [0.0, 0.0, 0.0])
    for mode in (1, 2):
        with oracle.fp_contraction(mode):
            alt = oracle.color_matrix(colours, m, [0.0, 0.0, 0.0])
        n, mx, total = diff_stats(base, alt)
        print("colour matrix [%s], contraction model %d: %d of %d channel values differ (%.4f %%), max |diff| = %d"
              % (name, mode, n, total, 100.0 * n / total, mx))
        assert mx <= 1
        if name == "identity":
            assert n == 0  # exact products: nothing to round


@pytest.mark.parametrize("gains", [(1.0, 1.2, 1.0), (1.1, 0.9, 1.05), (1.0, 1.0, 1.0)])
def test_colour_enhancer_all_colours_within_one_lsb(oracle, colours, gains):
    """color_enhancer.cpp:40-46 -> HSV2RGB_native's v * (1 - s * f) terms.  All 2^24 inputs."""
    base = oracle.color_enhance(colours, *gains)
    with oracle.fp_contraction(1):
        alt = oracle.color_enhance(colours, *gains)
    n, mx, total = diff_stats(base, alt)
    print("colour enhancer gains %s, contracted HSV inverse: %d of %d channel values differ (%.4f %%), max |diff| = %d"
          % (gains, n, total, 100.0 * n / total, mx))
    assert mx <= 1


def test_pca_map_within_one_lsb(oracle):
    """white_balance.cpp:117-118 -> cv::addWeighted (v_fma chain).  The quadratic map of every byte value for the coefficient
    pairs of forty seeded scenes (the map is per channel value: 256 inputs per coefficient pair)."""
    worst, differing, total = 0, 0, 0
    for seed in range(40):
        frame = synth.gen_scene_bgr(96, 64, seed, tint=(0.55 + 0.01 * seed, 1.0, 0.45 + 0.01 * seed))
        base = oracle.wb_pca(frame)
        with oracle.fp_contraction(1):
            alt = oracle.wb_pca(frame)
        n, mx, tot = diff_stats(base, alt)
        worst, differing, total = max(worst, mx), differing + n, total + tot
    print("pca map, contracted addWeighted: %d of %d channel values differ, max |diff| = %d" % (differing, total, worst))
    assert worst <= 1


@pytest.mark.parametrize("size", [(2448, 2048), (1920, 1200), (3840, 2160), (640, 480)])
def test_vignetting_mask_plane_barely_moves(oracle, size):
    """vignetting_correction.cpp:42-43 is double arithmetic in the reference's own translation unit; contraction moves an
    intermediate by at most an ulp of a double, which survives the rounding to float only when the double sits next to a
    float rounding boundary.  Counted at every BASELINE size; a changed mask float changes L only where L * mask sits on a
    rounding boundary as well (checked for every L)."""
    w, h = size
    base = oracle.vignetting_mask(h, w, 1.5, 1e-3, 1e-6)
    with oracle.fp_contraction(1):
        alt = oracle.vignetting_mask(h, w, 1.5, 1e-3, 1e-6)
    changed = np.flatnonzero(base.ravel() != alt.ravel())
    flips = 0
    if changed.size:
        L = np.arange(256, dtype=np.float32)[:, None]
        a = np.rint(L * base.ravel()[changed][None, :])
        b = np.rint(L * alt.ravel()[changed][None, :])
        flips = int((a != b).sum())
        assert np.abs(a - b).max() <= 1
    print("vignetting mask %dx%d: %d of %d floats differ under contraction; (L, pixel) pairs whose corrected L differs: %d"
          % (w, h, changed.size, base.size, flips))
    assert changed.size <= base.size // 1000


def test_golden_frames_stage_outputs_within_one_lsb_and_end_to_end_report(oracle):
    """The committed golden configurations under the contracted model: a stage that contains a float expression moves by at
    most 1 LSB at its own output (single-stage cases); for the full chain -- where later stages (gamma LUT slope, Lab round
    trip, bilinear remap) see the moved value -- the differing pixels are counted and bounded."""
    cases = make_golden.cases()
    single = ["color_calibration", "color_enhancer", "pca", "vignetting"]
    for name in single + ["full_chain"]:
        c, enc, frame = cases[name]
        base, _ = oracle_run(oracle, c, frame, enc)
        for mode in (1, 2):
            with oracle.fp_contraction(mode):
                alt, _ = oracle_run(oracle, c, frame, enc)
            n, mx, total = diff_stats(base, alt)
            print("golden %-18s contraction model %d: %d of %d channel values differ, max |diff| = %d" % (name, mode, n, total, mx))
            if name in single:
                assert mx <= 1
            else:
                assert mx <= 4 and n <= total // 20
    # one BASELINE-sized frame (config 2: 2448x2048 full chain) for a count at scale
    from helpers import cfg
    w, h = 2448, 2048
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="grey_world", cc=True, gamma=True, gamma_k=0.8, vig=True, undistort=True,
            cam=synth.camera_model(w, h))
    frame = synth.gen_frame(w, h, "bayer_rggb8", seed=0, kind="scene")
    base, _ = oracle_run(oracle, c, frame, "bayer_rggb8")
    for mode in (1, 2):
        with oracle.fp_contraction(mode):
            alt, _ = oracle_run(oracle, c, frame, "bayer_rggb8")
        n, mx, total = diff_stats(base, alt)
        print("config 2 frame (2448x2048 full chain) contraction model %d: %d of %d channel values differ, max |diff| = %d" % (mode, n, total, mx))
        assert mx <= 4 and n <= total // 10000
    # integer-only stages cannot move at all
    for name in ("debayer_bayer_rggb8", "flip_180", "gamma_0.8", "grey_world", "undistortion"):
        c, enc, frame = cases[name]
        base, _ = oracle_run(oracle, c, frame, enc)
        with oracle.fp_contraction(1):
            alt, _ = oracle_run(oracle, c, frame, enc)
        assert np.array_equal(base, alt), name
