import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from raw_image_pipeline_amd import RawImagePipeline, synth
from helpers import cfg, configure
w, h = 640, 480
p = RawImagePipeline(False, "", "", "", device=0)
c = cfg(undistort=True, cam=synth.camera_model(w, h), balance=0.2, fov_scale=1.1)
configure(p, c)
yy, xx = np.mgrid[0:h, 0:w]
for name, img in (("x ramp", (xx // 4).astype(np.uint8)), ("const", np.full((h, w), 77, np.uint8))):
    p.set_tunable("remap_tiled", 1)
    a = p.process(img, "mono8").astype(int)
    p.set_tunable("remap_tiled", 0)
    b = p.process(img, "mono8").astype(int)
    print(os.environ.get("RIP_LIBRARY", "default")[-12:], name, "differing", int((a != b).sum()), "max", int(np.abs(a - b).max()))
    print("  row 100 tiled  ", a[100, 300:324].tolist())
    print("  row 100 generic", b[100, 300:324].tolist())
