"""The header-only C++ facade (include/raw_image_pipeline/raw_image_pipeline.hpp) compiles as C++14
against the C-ABI library and behaves like the reference class for a caller."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cpp")


# "stand-in": the facade's own raw_image_pipeline::Mat (-DRIP_NO_OPENCV); "cv": the cv::Mat branch a ROS / pybind11 workspace
# takes, compiled against tests/cpp/fake_opencv (OpenCV itself is not installable here)
BRANCHES = {"stand-in": ["-DRIP_NO_OPENCV"], "cv": ["-I", os.path.join(ROOT, "tests", "cpp", "fake_opencv")]}


def build_facade_test(tmp_path, rip_lib, branch="stand-in"):
    exe = str(tmp_path / ("facade_test_" + branch.replace("-", "_")))
    libdir = os.path.join(ROOT, "raw_image_pipeline_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Werror"] + BRANCHES[branch] + ["-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-l:librip_hip.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run_env(device):
    env = dict(os.environ)
    env["RIP_DEVICE"] = str(device)
    # the standalone binary uses ROCm's HIP runtime (no torch in the process)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return env


@pytest.mark.parametrize("branch", sorted(BRANCHES))
def test_facade_compiles_and_handles_parameters(tmp_path, rip_lib, branch):
    exe = build_facade_test(tmp_path, rip_lib, branch)
    r = subprocess.run([exe, "host"], capture_output=True, text=True, env=run_env(-1))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "facade host OK" in r.stdout and "no CPU execution path" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("branch", sorted(BRANCHES))
def test_facade_apply_matches_oracle(tmp_path, rip_lib, oracle, branch):
    from helpers import cfg, oracle_run
    exe = build_facade_test(tmp_path, rip_lib, branch)
    w, h = 64, 48
    out_path = str(tmp_path / "out.bin")
    r = subprocess.run([exe, "gpu", str(w), str(h), out_path], capture_output=True, text=True, env=run_env(0))
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out_path, np.uint8).reshape(h, w, 3)
    s = 12345
    vals = []
    for _ in range(w * h):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals.append(s >> 24)
    frame = np.array(vals, np.uint8).reshape(h, w)
    c = cfg(flip=True, flip_angle=180, wb=True, wb_method="gray_world", wb_bright=0.8, cc=True,
            cc_matrix=[1.5, -0.25, 0.0, 0.125, 1.0, -0.125, 0.0, -0.5, 1.75], cc_bias=(1.0, -2.0, 3.5), gamma=True, gamma_k=0.8, vig=True)
    ref, _ = oracle_run(oracle, c, frame, "bayer_rggb8")
    assert np.array_equal(got, ref)
