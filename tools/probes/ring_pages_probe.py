#!/usr/bin/env python3
"""Where do the pinned result buffers of the host-frame ring live?  Ring of 6 slots: for every slot the address of its
result view, the NUMA node of a few of its pages (move_pages in query mode), the smaps entry of its mapping, and the slot's
download time from RIP_DEBUG_RING."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RIP_DEBUG_RING"] = "1"
import numpy as np  # noqa: E402
from raw_image_pipeline_amd import RawImagePipeline, synth  # noqa: E402
from raw_image_pipeline_amd.pipeline import host_alloc  # noqa: E402

libc = ctypes.CDLL("libc.so.6", use_errno=True)
W, H = 2448, 2048
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frame = host_alloc((H, W))
frame[...] = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
p = RawImagePipeline(False, "", "", "", device=0)
synth.configure_full_chain(p, W, H, "grey_world")
p.set_ring_depth(DEPTH)
tickets, addr = [], {}
for i in range(6 * DEPTH):
    if len(tickets) == DEPTH:
        t = tickets.pop(0)
        v = p.collect(t, copy=False)
        addr[v.ctypes.data] = addr.get(v.ctypes.data, 0) + 1
    tickets.append(p.submit(frame, "bayer_rggb8"))
while tickets:
    v = p.collect(tickets.pop(0), copy=False)
    addr[v.ctypes.data] = addr.get(v.ctypes.data, 0) + 1


def nodes_of(a, n=8, stride=2 << 20):
    pages = (ctypes.c_void_p * n)(*[(a & ~4095) + k * stride for k in range(n)])
    status = (ctypes.c_int * n)()
    rc = libc.syscall(279, 0, n, pages, None, status, 0)  # move_pages, x86-64
    return list(status) if rc == 0 else "move_pages failed errno %d" % ctypes.get_errno()


smaps = open("/proc/self/smaps").read()
entries = re.split(r"\n(?=[0-9a-f]+-[0-9a-f]+ )", smaps)
print("numa nodes online:", open("/sys/devices/system/node/online").read().strip(), " THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
for a in sorted(addr):
    ent = ""
    for e in entries:
        m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ \S+\s*(.*)", e)
        if m and int(m.group(1), 16) <= a < int(m.group(2), 16):
            kv = dict(re.findall(r"(\w+):\s+(\d+) kB", e))
            ent = "map %s-%s %s %s size %s kB AnonHuge %s kB KernelPageSize %s" % (m.group(1), m.group(2), m.group(3), m.group(4), kv.get("Size"), kv.get("AnonHugePages"), kv.get("KernelPageSize"))
    print("result buffer at 0x%x (offset in 2 MB: %d KB): nodes %s; %s" % (a, (a % (2 << 20)) // 1024, nodes_of(a), ent))
