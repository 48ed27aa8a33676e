#!/usr/bin/env python3
"""Timing-only experiment switches of the ring remap (rip_remap.hip, -DRIP_EXPERIMENTS builds: Tunables::remap_exp), all in
ONE process on ONE handle and ONE output allocation -- the remap's duration has a level per allocation of the output
(EXPERIMENTS.md "the run-to-run spread"), so separate processes cannot resolve differences below 10 %.  The masks are
walked round-robin `--rounds` times; prints per-class kernel times of every trial and the per-mask medians.

usage: RIP_LIBRARY=raw_image_pipeline_amd/variants/exp.so remap_exp_probe.py --masks 0,1,2,4,6 [--tunable name=v1,v2]
       [--rounds 3] [--steps 6] [--workload config2]"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0")
    ap.add_argument("--tunable", default="", help="name=v1,v2,...: a second axis, walked inside every mask")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--size", default="2448x2048")
    args = ap.parse_args()
    import torch
    import bench
    from raw_image_pipeline_amd import RawImagePipeline

    width, height = (int(v) for v in args.size.split("x"))
    pipe = RawImagePipeline(False, "", "", "", device=0)
    pipe.set_stream(torch.cuda.current_stream())
    pattern, _ = bench.configure(pipe, args.workload, width, height)
    frames = torch.from_numpy(bench.make_frames(width, height, pattern, args.batch, 0)).cuda()
    orows, ocols, ocn, _ = pipe.query_output(height, width, 1, pattern)
    out = torch.empty((args.batch, orows, ocols, ocn), dtype=torch.uint8, device="cuda")
    masks = [int(m, 0) for m in args.masks.split(",")]
    tname, tvals = "", [None]
    if args.tunable:
        tname, _, v = args.tunable.partition("=")
        tvals = [int(x) for x in v.split(",")]
    results = {}
    for r in range(args.rounds):
        for m in masks:
            for tv in tvals:
                pipe.set_tunable("remap_exp", m)
                if tname:
                    pipe.set_tunable(tname, tv)
                for _ in range(2):
                    pipe.apply_device(frames, pattern, out=out)
                torch.cuda.synchronize()
                pipe.profile_begin(64 * args.steps + 8)
                for _ in range(args.steps):
                    pipe.apply_device(frames, pattern, out=out)
                torch.cuda.synchronize()
                prof = pipe.profile_end()
                ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}
                key = (m, tv)
                results.setdefault(key, []).append(ms)
                print("round %d mask %3d %s%s  %s" % (r, m, tname, "" if tv is None else "=%d" % tv, ms), flush=True)
    print("--- medians (ms per %d frames)" % args.batch)
    for (m, tv), lst in results.items():
        keys = sorted({k for d in lst for k in d})
        med = {k: round(statistics.median(d.get(k, 0.0) for d in lst), 4) for k in keys}
        print("mask %3d %s%s  %s" % (m, tname, "" if tv is None else "=%d" % tv, med), flush=True)


if __name__ == "__main__":
    main()
