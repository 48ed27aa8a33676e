"""Committed vectors (tests/golden/vectors.npz, made by tests/golden/make_golden.py): the oracle must
keep reproducing them (CPU), and the HIP path must match them (GPU)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from helpers import assert_images_equal, cfg, configure, oracle_run  # noqa: E402
import make_golden  # noqa: E402
from raw_image_pipeline_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def vectors():
    return np.load(os.path.join(HERE, "golden", "vectors.npz"))


CASES = make_golden.cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(oracle, vectors, name):
    c, enc, frame = CASES[name]
    assert np.array_equal(frame, vectors["in__" + name]), "seeded input changed"
    out, _ = oracle_run(oracle, c, frame, enc)
    assert_images_equal(out, vectors["out__" + name], name)


def test_oracle_reproduces_golden_ccc_sequence(oracle, vectors):
    _, frames, outs = make_golden.ccc_sequence()
    assert np.array_equal(frames, vectors["in__ccc_sequence"])
    assert np.array_equal(outs, vectors["out__ccc_sequence"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden(gpu_pipe, vectors, name):
    c, enc, _ = CASES[name]
    configure(gpu_pipe, c)
    got = gpu_pipe.process(vectors["in__" + name], enc)
    assert_images_equal(got, vectors["out__" + name], name)


@pytest.mark.gpu
def test_gpu_matches_golden_ccc_sequence(gpu_pipe, vectors):
    filt, bias = synth.ccc_model()
    gpu_pipe.set_ccc_model(filt, bias)
    gpu_pipe.set_ccc_kalman_model(1.0, 10.0)
    configure(gpu_pipe, cfg(wb=True, wb_method="ccc", wb_temporal=True))
    gpu_pipe.reset_white_balance_temporal_consistency()
    for i, frame in enumerate(vectors["in__ccc_sequence"]):
        assert_images_equal(gpu_pipe.process(frame, "bayer_rggb8"), vectors["out__ccc_sequence"][i], "ccc frame %d" % i)


OPENCV_VECTORS = os.path.join(HERE, "golden", "opencv_vectors.npz")


@pytest.mark.skipif(not os.path.exists(OPENCV_VECTORS), reason="tests/golden/opencv_vectors.npz has not been generated yet: "
                    "run tools/compare_with_opencv.py --write on a machine with OpenCV (parity of oracle/ vs OpenCV is unpinned until then)")
def test_opencv_vectors(oracle):
    """Outputs of a real OpenCV, frozen by tools/compare_with_opencv.py together with the oracle call that must
    reproduce each of them (function name + arguments): the pin of oracle/ to the reference's arithmetic library."""
    v = np.load(OPENCV_VECTORS)
    records = sorted(k[:-3] for k in v.files if k.endswith("_fn"))
    assert records, "no replayable record in opencv_vectors.npz"
    for rec in records:
        fn = getattr(oracle, str(v[rec + "_fn"]))
        args, k = [], 0
        while rec + "_arg%d" % k in v.files:
            a = v[rec + "_arg%d" % k]
            args.append(a.item() if a.ndim == 0 else a)
            k += 1
        got = np.asarray(fn(*args))
        ref = v[rec]
        assert got.shape == ref.shape, rec
        d = np.abs(got.astype(np.int16) - ref.astype(np.int16)).max()
        assert d <= int(v[rec + "_bar"]), "%s (%s): max |oracle - OpenCV %s| = %d" % (rec, v[rec + "_where"], v["opencv_version"], d)
