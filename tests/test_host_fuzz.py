"""Seeded mutation fuzz of the four file loaders of the host layer (rip_load_params, rip_load_camera_calibration,
rip_load_color_calibration, rip_load_ccc_model: the YAML-subset reader and the binary model reader of rip_host.cpp) through a
RIP_DEVICE_NONE handle.  Contract: a malformed file makes the call fail with a RipError / ValueError (the reference: YAML::Exception /
cv::Exception, raw_image_pipeline.cpp:44-52, undistortion.cpp:157-173, color_calibration.cpp:52-70) or load with defaults --
it never crashes the process, never reads or writes out of bounds, and the handle stays usable afterwards.  The same file runs
under AddressSanitizer + UBSan through tools/run_asan.sh ("sanitizers")."""
import os
import random
import struct

import numpy as np
import pytest

from raw_image_pipeline_amd import RawImagePipeline
from raw_image_pipeline_amd.pipeline import RipError

REFUSED = (RipError, ValueError)  # RIP_ERR_IO / RIP_ERR_ASSERT / RIP_ERR_DEVICE and RIP_ERR_INVALID_ARGUMENT (std::invalid_argument)

PARAMS = """# pipeline parameters
debayer:
  enabled: true
  encoding: "auto"
flip:
  enabled: true
  angle: 180
white_balance:
  enabled: true
  method: "grey_world"   # simple, grey_world, learned, ccc, pca
  clipping_percentile: 20
  saturation_bright_thr: 0.8
  saturation_dark_thr: 0.2
  temporal_consistency: false
color_calibration:
  enabled: true
gamma_correction:
  enabled: true
  method: "custom"
  k: 0.8
vignetting_correction:
  enabled: true
  scale: 1.5
  a2: 1e-3
  a4: 1e-6
color_enhancer:
  run_color_enhancer: false
  hue_gain: 1.0
  saturation_gain: 1.2
  value_gain: 1.0
undistortion:
  enabled: true
  balance: 0.0
  fov_scale: 0.8
"""

CAMERA = """image_width: 720
image_height: 540
camera_name: cam0
camera_matrix:
  rows: 3
  cols: 3
  data: [347.5, 0.0, 342.4, 0.0, 347.4, 271.3, 0.0, 0.0, 1.0]
distortion_model: equidistant
distortion_coefficients:
  rows: 1
  cols: 4
  data: [-0.0396, -0.0036, 0.0039,
         -0.0017]
rectification_matrix: {rows: 3, cols: 3, data: [1, 0, 0, 0, 1, 0, 0, 0, 1]}
projection_matrix:
  rows: 3
  cols: 4
  data: [347.5, 0.0, 342.4, 0.0, 0.0, 347.4, 271.3, 0.0, 0.0, 0.0, 1.0, 0.0]
"""

COLOR = """matrix:
  rows: 3
  cols: 3
  data: [1.5, -0.3, -0.2, -0.25, 1.4, -0.15, -0.1, -0.4, 1.5]
bias:
  rows: 3
  cols: 1
  data: [0.5, -1.0, 2.0]
"""

NASTY_TOKENS = [b"{", b"}", b"[", b"]", b":", b": ", b",", b"#", b"'", b'"', b"-", b"- ", b"\t", b"\n", b"\r\n", b"\x00", b"\xff\xfe", b"\xc3\x28",
                b"1e999", b"-1e999", b"nan", b"inf", b"-inf", b"0x7fffffff", b"99999999999999999999999999", b"-2147483649", b"4294967296",
                b"1e-400", b".", b"..", b"---", b"...", b"~", b"null", b"true", b"!!binary", b"&a", b"*a", b"|", b">", b"%YAML 1.2", b" " * 100,
                b"data: [", b"data: {", b"{a: {a: {a: {a: [1, 2]}}}}", b"a: [1, [2, 3]]"]


def mutate(rng, doc):
    """One of: truncation, byte flips, token insertion, line duplication / deletion / re-indentation, value replacement."""
    b = bytearray(doc)
    kind = rng.randrange(9)
    if kind == 0 and b:
        del b[rng.randrange(len(b)):]
    elif kind == 1 and b:
        for _ in range(rng.randrange(1, 8)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif kind == 2:
        for _ in range(rng.randrange(1, 4)):
            at = rng.randrange(len(b) + 1)
            b[at:at] = rng.choice(NASTY_TOKENS)
    elif kind == 3 and b:
        lines = bytes(b).split(b"\n")
        i = rng.randrange(len(lines))
        lines[i:i] = [lines[i]] * rng.randrange(1, 4)
        b = bytearray(b"\n".join(lines))
    elif kind == 4 and b:
        lines = bytes(b).split(b"\n")
        del lines[rng.randrange(len(lines))]
        b = bytearray(b"\n".join(lines))
    elif kind == 5 and b:
        lines = bytes(b).split(b"\n")
        i = rng.randrange(len(lines))
        lines[i] = b" " * rng.randrange(0, 12) + lines[i].lstrip()
        b = bytearray(b"\n".join(lines))
    elif kind == 6 and b:
        lines = bytes(b).split(b"\n")
        i = rng.randrange(len(lines))
        if b":" in lines[i]:
            key = lines[i].split(b":")[0]
            lines[i] = key + b": " + rng.choice(NASTY_TOKENS)
        b = bytearray(b"\n".join(lines))
    elif kind == 7 and b:
        a, z = sorted((rng.randrange(len(b)), rng.randrange(len(b))))
        b[a:z] = bytes(b[a:z]) * rng.randrange(2, 4)
    else:
        rng.shuffle(lines := bytes(b).split(b"\n"))
        b = bytearray(b"\n".join(lines))
    return bytes(b)


LOADERS = [("load_params", PARAMS), ("load_camera_calibration", CAMERA), ("load_color_calibration", COLOR)]


@pytest.mark.parametrize("loader,doc", LOADERS, ids=[n for n, _ in LOADERS])
def test_yaml_loaders_survive_mutations(rip_lib, tmp_path, loader, doc):
    rng = random.Random(20260930 + len(doc))
    pipe = RawImagePipeline(False, device=-1)
    good = tmp_path / "good.yaml"
    good.write_bytes(doc.encode())
    getattr(pipe, loader)(str(good))
    path = tmp_path / "fuzz.yaml"
    failures = loads = 0
    for _ in range(600):
        data = doc.encode()
        for _ in range(rng.randrange(1, 4)):
            data = mutate(rng, data)
        path.write_bytes(data)
        try:
            getattr(pipe, loader)(str(path))
            loads += 1
        except REFUSED:
            failures += 1
    assert failures > 20 and loads > 20, (failures, loads)  # both outcomes are exercised
    getattr(pipe, loader)(str(good))  # the handle is still usable
    if loader == "load_camera_calibration":
        assert pipe.get_dist_image_width() == 720 and pipe.get_dist_image_height() == 540


def test_yaml_pathological_documents(rip_lib, tmp_path):
    """Sizes and shapes a mutation does not reach: deep nesting (block and flow), very long lines, huge sequences, a huge
    file, an empty file, a directory, binary garbage."""
    pipe = RawImagePipeline(False, device=-1)
    path = tmp_path / "p.yaml"
    docs = {
        "deep_block": "".join(" " * i + "k%d:\n" % i for i in range(20000)).encode(),
        "deep_flow": b"a: " + b"{a: " * 50000 + b"1" + b"}" * 50000,
        "deep_flow_open": b"a: " + b"{a: " * 200000,
        "long_line": b"debayer: {enabled: true, encoding: \"" + b"x" * (8 << 20) + b"\"}",
        "long_key": b"k" * (4 << 20) + b": 1",
        "huge_seq": b"camera_matrix:\n  data: [" + b"1.0, " * 2000000 + b"1.0]",
        "many_keys": b"".join(b"k%d: %d\n" % (i, i) for i in range(200000)),
        "unterminated_seq": b"camera_matrix:\n  data: [1, 2, 3\n" * 1000,
        "empty": b"",
        "only_comments": b"# nothing\n" * 1000,
        "nul_bytes": b"\x00" * 65536,
        "random": bytes(random.Random(7).randrange(256) for _ in range(65536)),
        "crlf": PARAMS.replace("\n", "\r\n").encode(),
        "bom": b"\xef\xbb\xbf" + PARAMS.encode(),
        "ints": b"flip:\n  enabled: true\n  angle: 1e300\nimage_width: -1e300\nimage_height: nan\n",
    }
    for name, data in docs.items():
        path.write_bytes(data)
        for loader in ("load_params", "load_camera_calibration", "load_color_calibration"):
            try:
                getattr(pipe, loader)(str(path))
            except REFUSED:
                pass
    for loader in ("load_params", "load_camera_calibration", "load_color_calibration", "load_ccc_model"):
        try:
            getattr(pipe, loader)(str(tmp_path))  # a directory
        except REFUSED:
            pass
    good = tmp_path / "good.yaml"
    good.write_text(PARAMS)
    pipe.load_params(str(good))


def test_ccc_model_loader_survives_mutations(rip_lib, tmp_path):
    """Binary model: int32 width, int32 height, width x height float32 filter, then bias (convolutional_color_constancy.cpp
    loadModel)."""
    rng = random.Random(11)
    n = 256
    nrng = np.random.default_rng(5)
    body = nrng.standard_normal((2, n, n)).astype(np.float32).tobytes()
    good = struct.pack("<ii", n, n) + body
    pipe = RawImagePipeline(False, device=-1)
    path = tmp_path / "model.bin"
    path.write_bytes(good)
    pipe.load_ccc_model(str(path))
    headers = [(0, 0), (-1, 256), (256, -1), (1 << 30, 1 << 30), (65536, 65536), (1, 1), (255, 257), (256, 255), (2147483647, 1), (-2147483648, -2147483648),
               (1024, 1024), (1025, 1024), (3, 5)]
    outcomes = {"ok": 0, "err": 0}
    for w, h in headers:
        for tail in (body, body[:1000], b"", body + body):
            path.write_bytes(struct.pack("<ii", w, h) + tail)
            try:
                pipe.load_ccc_model(str(path))
                outcomes["ok"] += 1
            except REFUSED:
                outcomes["err"] += 1
    for _ in range(60):
        data = bytearray(good)
        k = rng.randrange(4)
        if k == 0:
            del data[rng.randrange(len(data)):]
        elif k == 1:
            for _ in range(16):
                data[rng.randrange(len(data))] = rng.randrange(256)
        elif k == 2:
            data[8:8 + 4096] = struct.pack("<f", float("nan")) * 1024
        else:
            data[8:8 + 4096] = struct.pack("<f", float("inf")) * 1024
        path.write_bytes(bytes(data))
        try:
            pipe.load_ccc_model(str(path))
            outcomes["ok"] += 1
        except REFUSED:
            outcomes["err"] += 1
    assert outcomes["ok"] > 5 and outcomes["err"] > 20, outcomes
    path.write_bytes(good)
    pipe.load_ccc_model(str(path))


def test_setters_reject_or_accept_extreme_values(rip_lib):
    """Every numeric setter of the boundary with extreme arguments: an error or a stored value, no crash; then the table
    builders (gamma LUT, vignetting mask) run on whatever was accepted."""
    pipe = RawImagePipeline(False, device=-1)
    extremes = [0.0, -0.0, 1e-320, -1.0, 1e308, -1e308, float("inf"), float("-inf"), float("nan"), 1e-30, 255.0, 65536.0]
    for name in ("set_gamma_correction_k", "set_vignetting_correction_parameters", "set_white_balance_percentile", "set_white_balance_saturation_threshold",
                 "set_color_enhancer_hue_gain", "set_color_enhancer_saturation_gain", "set_color_enhancer_value_gain", "set_undistortion_balance",
                 "set_undistortion_fov_scale"):
        fn = getattr(pipe, name, None)
        if fn is None:
            continue
        import inspect
        nargs = len(inspect.signature(fn).parameters)
        for v in extremes:
            try:
                fn(*([v] * nargs))
            except REFUSED + (OverflowError,):
                pass
    for w, h in ((0, 0), (-1, 5), (1, 1), (7, 3), (65536, 1), (1, 65536)):
        for fn in (pipe.set_undistortion_image_size,) if hasattr(pipe, "set_undistortion_image_size") else ():
            try:
                fn(w, h)
            except REFUSED + (OverflowError,):
                pass
