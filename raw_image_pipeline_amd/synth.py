"""Seeded synthetic inputs shared by tests, bench.py and __graft_entry__.smoke().
numpy only; no reference data."""
import numpy as np

PATTERNS = {  # colour index (0=B,1=G,2=R) of the 2x2 cell, row-major
    "bayer_rggb8": ((2, 1), (1, 0)),
    "bayer_grbg8": ((1, 2), (0, 1)),
    "bayer_gbrg8": ((1, 0), (2, 1)),
    "bayer_bggr8": ((0, 1), (1, 2)),
}

# config/alphasense_color_calib_example.yaml:7 (values)
COLOR_MATRIX = [2.4276948, 0.21479778, -0.30818, 0.09277014, 1.1962607, -0.09772757, -0.24436986, -0.22239459, 2.099912]
# config/alphasense_calib_1.6mp_example.yaml:14 (values)
DIST_COEFFS = [-0.0480706813, 0.0129997684, -0.0112199955, 0.0026955514]


def mosaic(bgr, pattern):
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    cell = PATTERNS[pattern]
    for dy in range(2):
        for dx in range(2):
            out[dy::2, dx::2] = bgr[dy::2, dx::2, cell[dy][dx]]
    return out


def gen_scene_bgr(width, height, seed, tint=(0.70, 1.00, 0.55)):
    """RGB scene = low-frequency cosines + 8x6 colour patches + N(0,4) noise, times an illuminant
    tint (r,g,b); returned as uint8 BGR."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.empty((height, width, 3), np.float32)  # RGB
    for c in range(3):
        acc = np.full((height, width), 110.0, np.float32)
        for _ in range(3):
            fx, fy = rng.uniform(0.5, 3.0, 2)
            ph = rng.uniform(0, 2 * np.pi)
            acc += 30.0 * np.cos(2 * np.pi * (fx * xx / width + fy * yy / height) + ph).astype(np.float32)
        img[..., c] = acc
    patches = rng.uniform(20, 235, (6, 8, 3)).astype(np.float32)
    py = np.minimum((yy * 6 / height).astype(np.int32), 5)
    px = np.minimum((xx * 8 / width).astype(np.int32), 7)
    inner = (((yy * 6 / height) % 1.0) > 0.25) & (((xx * 8 / width) % 1.0) > 0.25)
    img = np.where(inner[..., None], patches[py, px], img)
    img += rng.normal(0, 4, img.shape).astype(np.float32)
    img *= np.asarray(tint, np.float32)
    rgb = np.clip(img, 0, 255).astype(np.uint8)
    return rgb[..., ::-1].copy()


def gen_frame(width, height, pattern="bayer_rggb8", seed=0, kind="scene", tint=(0.70, 1.00, 0.55)):
    """uint8 Bayer frame.  kind="scene": mosaicked gen_scene_bgr; kind="uniform": iid uniform bytes
    (worst-case gradients for the interpolating stages)."""
    if kind == "uniform":
        return np.random.default_rng(seed).integers(0, 256, (height, width), dtype=np.uint8)
    return mosaic(gen_scene_bgr(width, height, seed, tint), pattern)


def camera_model(width, height):
    """Equidistant camera scaled from the 1.6 MP Alphasense example to any W x H."""
    fx, cx = 0.48558 * width, 0.49431 * width
    fy, cy = 0.48510 * width, 0.48593 * height
    K = [fx, 0.0, cx, 0.0, fy, cy, 0.0, 0.0, 1.0]
    R = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
    P = [fx, 0.0, cx, 0.0, 0.0, fy, cy, 0.0, 0.0, 0.0, 1.0, 0.0]
    return dict(K=K, D=list(DIST_COEFFS), R=R, P=P, width=width, height=height)


def calibration_yaml(cam, model="equidistant"):
    """Camera calibration file text in the ROS camera_calibration_parsers layout the reference reads
    (undistortion.cpp:165-170)."""
    def seq(v):
        return "[" + ", ".join(repr(float(x)) for x in v) + "]"
    return ("image_width: %d\nimage_height: %d\ncamera_name: synthetic\n"
            "camera_matrix:\n  rows: 3\n  cols: 3\n  data: %s\n"
            "distortion_model: %s\n"
            "distortion_coefficients:\n  rows: 1\n  cols: 4\n  data: %s\n"
            "rectification_matrix:\n  rows: 3\n  cols: 3\n  data: %s\n"
            "projection_matrix:\n  rows: 3\n  cols: 4\n  data: %s\n"
            % (cam["width"], cam["height"], seq(cam["K"]), model, seq(cam["D"]), seq(cam["R"]), seq(cam["P"])))


def load_camera(pipe, cam, model="equidistant"):
    """Loads `cam` through the YAML loader (the only way the reference marks a calibration as
    available, undistortion.cpp:173-174)."""
    import os
    import tempfile
    fd, path = tempfile.mkstemp(suffix=".yaml", prefix="rip_calib_")
    try:
        with os.fdopen(fd, "w") as f:
            f.write(calibration_yaml(cam, model))
        pipe.load_camera_calibration(path)
    finally:
        os.unlink(path)


def ccc_model(sigma=3.0):
    """Synthetic CCC model in the default.bin layout: isotropic Gaussian filter centred at (0,0)
    with circular wrap, zero bias (float32 256x256 each)."""
    idx = np.arange(256)
    d = np.minimum(idx, 256 - idx).astype(np.float64)
    g = np.exp(-(d[:, None] ** 2 + d[None, :] ** 2) / (2 * sigma * sigma))
    return (g / g.sum()).astype(np.float32), np.zeros((256, 256), np.float32)


def configure_full_chain(pipe, width, height, wb_method="grey_world"):
    """BASELINE.json configs[1]: debayer + flip(180) + WB + colour calibration + gamma(custom,
    k=0.8) + vignetting(1.5, 1e-3, 1e-6) + undistortion (balance 0, fov_scale 1)."""
    cam = camera_model(width, height)
    pipe.set_debayer(True)
    pipe.set_flip(True)
    pipe.set_flip_angle(180)
    pipe.set_white_balance(True)
    pipe.set_white_balance_method(wb_method)
    pipe.set_white_balance_saturation_threshold(0.8, 0.2)
    pipe.set_white_balance_temporal_consistency(False)
    pipe.set_color_calibration(True)
    pipe.set_color_calibration_matrix(COLOR_MATRIX)
    pipe.set_color_calibration_bias([0.0, 0.0, 0.0])
    pipe.set_gamma_correction(True)
    pipe.set_gamma_correction_method("custom")
    pipe.set_gamma_correction_k(0.8)
    pipe.set_vignetting_correction(True)
    pipe.set_vignetting_correction_parameters(1.5, 1e-3, 1e-6)
    pipe.set_color_enhancer(False)
    pipe.set_undistortion(True)
    load_camera(pipe, cam)
    pipe.set_undistortion_balance(0.0)
    pipe.set_undistortion_fov_scale(1.0)
    return cam
