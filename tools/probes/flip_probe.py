import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from raw_image_pipeline_amd import RawImagePipeline, synth
W,H=2448,2048
base=synth.gen_frame(W,H,"bayer_rggb8",seed=1,kind="scene")
frames=torch.from_numpy(np.stack([np.roll(base,2*i,axis=0) for i in range(16)])).cuda()
for angle in (0,180,90,270):
    p=RawImagePipeline(False,"","","",device=0)
    synth.configure_full_chain(p,W,H)
    p.set_undistortion(False)
    p.set_flip(angle!=0); p.set_flip_angle(angle)
    p.set_stream(torch.cuda.current_stream())
    out=p.apply_device(frames,"bayer_rggb8")
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10): p.apply_device(frames,"bayer_rggb8",out=out)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/10/16
    print("flip %3d: %.1f us/frame  out %s"%(angle,dt*1e6,tuple(out.shape)))
