#!/usr/bin/env python3
"""Timing-only experiment switches of the ring remap (rip_remap.hip, -DRIP_EXPERIMENTS builds: Tunables::remap_exp), all in
ONE process on ONE handle and ONE output allocation -- the remap's duration has a level per allocation of the output
(EXPERIMENTS.md "the run-to-run spread"), so separate processes cannot resolve differences below 10 %.  The masks are
walked round-robin `--rounds` times; prints per-class kernel times of every trial and the per-mask medians.

usage: RIP_LIBRARY=raw_image_pipeline_amd/variants/exp.so remap_exp_probe.py --masks 0,1,2,4,6 [--tunable name=v1,v2]
       [--rounds 3] [--steps 6] [--workload config2]
       remap_exp_probe.py --libs head=raw_image_pipeline_amd/variants/head.so,new= [--set new:remap_persistent=0] ...
--libs: several builds in the one process (name=path; an empty path is the tree's library), one handle each, the same input and
output tensors; --set lib:tunable=value pins a tunable on one of them (a library that does not know it is left alone)."""
import argparse
import os
import statistics
import time
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
    ap.add_argument("--libs", default="")
    ap.add_argument("--set", action="append", default=[])
    args = ap.parse_args()
    import torch
    import bench
    from raw_image_pipeline_amd import RawImagePipeline

    width, height = (int(v) for v in args.size.split("x"))
    from raw_image_pipeline_amd import pipeline as P
    libs = [("", os.environ.get("RIP_LIBRARY", ""))]
    if args.libs:
        libs = [tuple(x.split("=", 1)) for x in args.libs.split(",")]
    pipes = {}
    for name, path in libs:
        P._lib = None  # load_library caches one library per process; every handle keeps the one it was created with
        if path:
            os.environ["RIP_LIBRARY"] = os.path.abspath(path)
        else:
            os.environ.pop("RIP_LIBRARY", None)
        pp = RawImagePipeline(False, "", "", "", device=0)
        pp.set_stream(torch.cuda.current_stream())
        pattern, _ = bench.configure(pp, args.workload, width, height)
        pipes[name] = pp
    for spec in args.set:
        name, _, kv = spec.partition(":")
        k, _, v = kv.partition("=")
        try:
            pipes[name].set_tunable(k, int(v))
        except Exception as e:
            print("note: %s: %s" % (spec, e))
    pipe = pipes[libs[0][0]]
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
      for lname, pipe in pipes.items():
        for m in masks:
            for tv in tvals:
                try:
                    pipe.set_tunable("remap_exp", m)
                except Exception:
                    if m:
                        continue
                if tname:
                    try:
                        pipe.set_tunable(tname, tv)
                    except Exception:
                        pass
                for _ in range(2):
                    pipe.apply_device(frames, pattern, out=out)
                torch.cuda.synchronize()
                pipe.profile_begin(64 * args.steps * 40 + 8)
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    pipe.apply_device(frames, pattern, out=out)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / args.steps * 1e3
                prof = pipe.profile_end()
                ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}
                ms["wall"] = round(wall, 4)
                key = (lname, m, tv)
                results.setdefault(key, []).append(ms)
                print("round %d %s mask %3d %s%s  %s" % (r, lname, m, tname, "" if tv is None else "=%d" % tv, ms), flush=True)
    print("--- medians (ms per %d frames)" % args.batch)
    for (lname, m, tv), lst in results.items():
        keys = sorted({k for d in lst for k in d})
        med = {k: round(statistics.median(d.get(k, 0.0) for d in lst), 4) for k in keys}
        print("mask %3d %s %s%s  %s" % (m, lname, tname, "" if tv is None else "=%d" % tv, med), flush=True)


if __name__ == "__main__":
    main()
