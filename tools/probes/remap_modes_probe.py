#!/usr/bin/env python3
"""The remap of config 2 takes 1.82 ms per 256 frames in some processes and 1.99 ms in others, on one box and with one
binary (profiles/r05_bench_lines.jsonl: 1.82 1.82 1.89 1.99 1.99 1.99 2.00).  This probe asks what the mode follows:
several trials inside ONE process, each with a fresh handle (fresh device buffers of the library) and a fresh output
tensor, optionally shifted by a pad allocation; prints the per-class kernel times and where the buffers landed.
Run it several times (separate processes) for the other half of the answer.

Trial modes (--modes, one per trial, comma separated; <M> in MiB): n = nothing beside the output tensor; b<M> = a pad tensor
allocated BEFORE the output and kept alive; f<M> = the same pad, released to the driver again before the timed steps;
a<M> = pad allocated AFTER the output; g<M> = no pad, the output tensor M MiB larger than needed; o<M> = the output carved
M MiB (fractions allowed) into ONE buffer that lives for the whole process (same physical pages, only the offset moves).

usage: remap_modes_probe.py [--trials N] [--modes n,b1,f1,...] [--steps K] [--keep-handle]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=4)
    ap.add_argument("--modes", default="n")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--keep-handle", action="store_true", help="one handle for all trials: only the output tensor moves")
    args = ap.parse_args()
    import torch
    import bench
    from raw_image_pipeline_amd import RawImagePipeline

    width, height = 2448, 2048
    modes = args.modes.split(",")
    import time
    frames = None
    pipe = None
    arena = None
    for t in range(args.trials):
        mode = modes[t % len(modes)]
        kind, mib = mode[0], float(mode[1:] or 0)
        nbytes = int(mib * (1 << 20))
        pad = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if kind in "bf" and nbytes else None
        if pipe is None or not args.keep_handle:
            pipe = RawImagePipeline(False, "", "", "", device=0)
            pipe.set_stream(torch.cuda.current_stream())
            pattern, _ = bench.configure(pipe, "config2", width, height)
        if frames is None:
            frames = torch.from_numpy(bench.make_frames(width, height, pattern, args.batch, 0)).cuda()
        orows, ocols, ocn, _ = pipe.query_output(height, width, 1, pattern)
        need = args.batch * orows * ocols * ocn
        if kind == "o":
            if arena is None:
                arena = torch.empty(need + (80 << 20), dtype=torch.uint8, device="cuda")
            flat = arena
            out = flat[nbytes:nbytes + need].view(args.batch, orows, ocols, ocn)
        else:
            flat = torch.empty(need + (nbytes if kind == "g" else 0), dtype=torch.uint8, device="cuda")
            out = flat[:need].view(args.batch, orows, ocols, ocn)
        if kind == "a" and nbytes:
            pad = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        if kind == "f":
            del pad
            pad = None
            torch.cuda.empty_cache()
        for _ in range(2):
            pipe.apply_device(frames, pattern, out=out)
        torch.cuda.synchronize()
        pipe.profile_begin(64 * args.steps + 8)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.apply_device(frames, pattern, out=out)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        prof = pipe.profile_end()
        ms = {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]}
        print("trial %d mode %-5s frames %#x  out %#x  step %.4f ms  %s" % (t, mode, frames.data_ptr(), out.data_ptr(), wall, ms), flush=True)
        del out, flat, pad
        if not args.keep_handle:
            del pipe
            pipe = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
