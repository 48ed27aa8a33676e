#!/usr/bin/env python3
"""Timeline of ONE resident frame per call on the reference's example configuration (1440 x 1080, ccc + undistortion):
which kernels a call launches, how long each runs and how long the device idles between them.
  single_frame_timeline.py run                  -- the workload (under rocprofv3 --kernel-trace)
  single_frame_timeline.py report <results.db>  -- median per-call timeline from the rocpd database"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CALLS = 200


def run():
    import numpy as np
    import torch
    from raw_image_pipeline_amd import RawImagePipeline, synth
    w, h = 1440, 1080
    p = RawImagePipeline(False, "", "", "", device=0)
    filt, bias = synth.ccc_model()
    p.set_ccc_model(filt, bias)
    p.set_white_balance(True)
    p.set_white_balance_method("ccc")
    p.set_gamma_correction(False)
    p.set_undistortion(True)
    synth.load_camera(p, synth.camera_model(w, h))
    p.set_undistortion_fov_scale(0.8)
    frame = torch.from_numpy(synth.gen_frame(w, h, "bayer_gbrg8", seed=3, kind="scene")[None]).cuda()
    p.set_stream(torch.cuda.current_stream())
    out = p.apply_device(frame, "bayer_gbrg8")
    for _ in range(CALLS + 20):
        p.apply_device(frame, "bayer_gbrg8", out=out)
    torch.cuda.synchronize()


def report(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # a call starts at the estimator's histogram kernel (or at the clear of its counters right before it)
    calls, cur_call = [], []
    for n, s, e in rows:
        if "ccc_hist" in n:
            carry = [cur_call.pop()] if cur_call and "fillBuffer" in cur_call[-1][0] else []
            if cur_call:
                calls.append(cur_call)
            cur_call = carry
        cur_call.append((n, s, e))
    calls.append(cur_call)
    calls = [c for c in calls if len(c) == max(set(len(x) for x in calls), key=[len(x) for x in calls].count)][20:20 + CALLS]
    k = len(calls[0])
    import statistics as st
    print("# %d calls of %d kernels; medians in microseconds" % (len(calls), k))
    total_run = total_gap = 0.0
    for i in range(k):
        dur = st.median((c[i][2] - c[i][1]) / 1e3 for c in calls)
        gap = st.median((c[i][1] - c[i - 1][2]) / 1e3 for c in calls) if i else 0.0
        total_run += dur
        total_gap += gap
        print("%-44s run %7.2f   idle before %6.2f" % (calls[0][i][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][-44:], dur, gap))
    span = st.median((c[-1][2] - c[0][1]) / 1e3 for c in calls)
    period = st.median((calls[j + 1][0][1] - calls[j][0][1]) / 1e3 for j in range(len(calls) - 1))
    print("first kernel start -> last kernel end %.2f (kernels %.2f + idle %.2f); call period %.2f" % (span, total_run, total_gap, period))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
