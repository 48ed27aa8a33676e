import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from raw_image_pipeline_amd import synth
from raw_image_pipeline_amd.frontend import CameraStream
W,H=2448,2048
frame=synth.gen_frame(W,H,"bayer_rggb8",seed=1,kind="scene")
def t(f,n=10):
    f(); f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
base={"flip/enabled": True, "flip/angle": 180, "gamma_correction/enabled": True}
for extra in [{}, {"white_balance/enabled": True, "white_balance/method": "gray_world"}, {"vignetting_correction/enabled": True}]:
    cams=[CameraStream(dict(base, **extra), device=0) for _ in range(3)]
    print(extra, "one camera %.2f ms/frame"%t(lambda: cams[0].on_image(frame,"bayer_rggb8")),
          "| three alternating %.2f ms/frame"%(t(lambda: [c.on_image(frame,"bayer_rggb8") for c in cams])/3))
