#!/bin/bash
# rip_submit / rip_collect against the ring depth, and the four-camera rig, under runtime settings / the in-flight limit
cat > /tmp/d4.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from raw_image_pipeline_amd import RawImagePipeline, synth
from raw_image_pipeline_amd.pipeline import host_alloc
W, H = 2448, 2048
frame = host_alloc((H, W)); frame[...] = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
for depth in (2, 3, 4, 6):
    p = RawImagePipeline(False, "", "", "", device=0)
    synth.configure_full_chain(p, W, H, "grey_world")
    p.set_ring_depth(depth)
    t = []
    for i in range(3 * depth):
        if len(t) == depth: p.collect(t.pop(0), copy=False)
        t.append(p.submit(frame, "bayer_rggb8"))
    t0 = time.perf_counter(); n = 300
    for i in range(n):
        if len(t) == depth: p.collect(t.pop(0), copy=False)
        t.append(p.submit(frame, "bayer_rggb8"))
    while t: p.collect(t.pop(0), copy=False)
    print("depth %d: %.3f ms/frame" % (depth, (time.perf_counter() - t0) / n * 1e3), flush=True)
PY
for e in "RIP_RING_INFLIGHT=0" "RIP_RING_INFLIGHT=3" "RIP_RING_INFLIGHT=2" "RIP_RING_INFLIGHT=4" "RIP_RING_INFLIGHT=0 AMD_DIRECT_DISPATCH=0"; do
  echo "=== $e"
  env $e python /tmp/d4.py 2>&1 | grep depth | tr '\n' ' '; echo
  env $e python tools/probes/rig_ring_probe.py 2>&1 | grep -A1 "RIG [234]"
done
