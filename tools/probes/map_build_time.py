import time, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from raw_image_pipeline_amd import RawImagePipeline, synth
for (w,h) in ((2448,2048),(3840,2160)):
    for mode in ("device","host"):
        if mode=="host": os.environ["RIP_MAPS_ON_HOST"]="1"
        else: os.environ.pop("RIP_MAPS_ON_HOST",None)
        p=RawImagePipeline(False,"","","",device=0)
        synth.load_camera(p, synth.camera_model(w,h))
        p.set_undistortion(True)
        ts=[]
        for rep in range(3):
            p.set_undistortion_balance(0.1*rep)   # dirties the maps
            t=time.perf_counter(); p.init_undistortion(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t)
        print(w,h,mode,"init_undistortion: %.1f ms (min of 3)"%(min(ts)*1e3))
