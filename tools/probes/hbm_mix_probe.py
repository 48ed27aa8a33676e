#!/usr/bin/env python3
"""What does the box sustain for write-heavy streams?  The debayer-only chain writes three bytes for every byte it reads;
the streaming-copy rate (1 : 1) may not be its ceiling.  torch kernels only: fill (0 : 1), copy (1 : 1), a byte -> 3 byte
expand (1 : 3, the chain's mix), a 3 byte -> byte reduction (3 : 1), each on ~4 GB, best of 10."""
import time
import torch

def rate(fn, nbytes, reps=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return nbytes / best / 1e12

n = 1 << 30
a = torch.empty(n, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
print("fill   (0:1)  %.2f TB/s" % rate(lambda: a.fill_(7), 4 * n))
print("copy   (1:1)  %.2f TB/s" % rate(lambda: b.copy_(a), 8 * n))
print("read   (1:0)  %.2f TB/s" % rate(lambda: a.sum(), 4 * n))
src = torch.empty(n, dtype=torch.int32, device="cuda"); dst = torch.empty((n, 3), dtype=torch.int32, device="cuda")
print("expand (1:3)  %.2f TB/s" % rate(lambda: dst.copy_(src[:, None].expand(n, 3)), 16 * n))
out = torch.empty(n, dtype=torch.int32, device="cuda")
print("select (3:1 strided read of 1/3)  %.2f TB/s of useful bytes" % rate(lambda: out.copy_(dst[:, 0]), 8 * n))
