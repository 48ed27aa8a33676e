import sys, cProfile, pstats
sys.path.insert(0,'/root/repo')
import numpy as np
from raw_image_pipeline_amd import synth
from raw_image_pipeline_amd.frontend import CameraStream
W,H=2448,2048
frame=synth.gen_frame(W,H,"bayer_rggb8",seed=1,kind="scene")
cam=CameraStream({"flip/enabled": True, "flip/angle": 180, "gamma_correction/enabled": True}, device=0)
for _ in range(3): cam.on_image(frame,"bayer_rggb8")
pr=cProfile.Profile(); pr.enable()
for _ in range(10): cam.on_image(frame,"bayer_rggb8")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
