import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from raw_image_pipeline_amd import RawImagePipeline, synth
W,H=2448,2048
p=RawImagePipeline(False,"","","",device=0)
synth.configure_full_chain(p,W,H)
frame=synth.gen_frame(W,H,"bayer_rggb8",seed=1,kind="scene")
def t(f,n=10):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
print("apply (taps default)   %.2f ms"%t(lambda: p.apply(frame.copy(),"bayer_rggb8")))
print("process                %.2f ms"%t(lambda: p.process(frame,"bayer_rggb8")))
print("get_dist_debayered     %.2f ms"%t(lambda: p.get_dist_debayered_image()))
print("get_dist_color         %.2f ms"%t(lambda: p.get_dist_color_image()))
print("get_processed          %.2f ms"%t(lambda: p.get_processed_image()))
print("frame.copy             %.2f ms"%t(lambda: frame.copy()))
big=np.empty((H,W,3),np.uint8)
print("np 15MB copy           %.2f ms"%t(lambda: big.copy()))
print("np 15MB empty+fill     %.2f ms"%t(lambda: np.empty((H,W,3),np.uint8).fill(1)))
import torch
s=torch.cuda.Stream()
p.set_stream(s)
print("apply on a torch stream %.2f ms"%t(lambda: p.apply(frame.copy(),"bayer_rggb8")))
print("process on a torch stream %.2f ms"%t(lambda: p.process(frame,"bayer_rggb8")))
p2=RawImagePipeline(False,"","","",device=0); synth.configure_full_chain(p2,W,H); p2.set_stream(torch.cuda.Stream())
def both():
    p.apply(frame.copy(),"bayer_rggb8"); p2.apply(frame.copy(),"bayer_rggb8")
print("two handles alternating %.2f ms per pair"%t(both))
from raw_image_pipeline_amd.frontend import CameraStream
cam=CameraStream({"flip/enabled": True, "flip/angle": 180, "gamma_correction/enabled": True}, device=0)
print("CameraStream.on_image  %.2f ms"%t(lambda: cam.on_image(frame,"bayer_rggb8")))
cam.pipe.set_stream(torch.cuda.Stream())
print("CameraStream.on_image on a torch stream %.2f ms"%t(lambda: cam.on_image(frame,"bayer_rggb8")))
