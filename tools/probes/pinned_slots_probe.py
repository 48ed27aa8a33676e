#!/usr/bin/env python3
"""Is the k-th pinned buffer of a process slower to DMA into?  Eight hipHostMalloc'ed 15 MB buffers and eight device
buffers; device -> host copy rate of every (host, device) pair on one stream, then with an upload running beside it."""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
N = 15 * 1024 * 1024
K = 8


def check(rc):
    assert rc == 0, rc


host, dev = [], []
for flags in (0,):
    for _ in range(K):
        p = ctypes.c_void_p()
        check(hip.hipHostMalloc(ctypes.byref(p), N, flags))
        ctypes.memset(p.value, 1, N)
        host.append(p.value)
        d = ctypes.c_void_p()
        check(hip.hipMalloc(ctypes.byref(d), N))
        dev.append(d.value)
s = ctypes.c_void_p()
check(hip.hipStreamCreateWithFlags(ctypes.byref(s), 1))
s2 = ctypes.c_void_p()
check(hip.hipStreamCreateWithFlags(ctypes.byref(s2), 1))


def rate(dst, src, kind, stream, reps=10):
    check(hip.hipMemcpyAsync(dst, src, N, kind, stream))
    check(hip.hipStreamSynchronize(stream))
    t0 = time.perf_counter()
    for _ in range(reps):
        check(hip.hipMemcpyAsync(dst, src, N, kind, stream))
    check(hip.hipStreamSynchronize(stream))
    return N * reps / (time.perf_counter() - t0) / 1e9


print("device -> host GB/s, row = host buffer k (allocation order), column = device buffer k")
for h in range(K):
    print("host %d  " % h + " ".join("%6.1f" % rate(host[h], dev[d], 2, s) for d in range(K)), flush=True)
print("host -> device GB/s by host buffer: " + " ".join("%6.1f" % rate(dev[0], host[h], 1, s) for h in range(K)))
