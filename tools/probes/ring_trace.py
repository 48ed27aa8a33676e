#!/usr/bin/env python3
"""ring_trace.py run <depth>  |  ring_trace.py report <db>: kernel + memory-copy timeline of rip_submit / rip_collect."""
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(depth):
    import numpy as np
    from raw_image_pipeline_amd import RawImagePipeline, synth
    from raw_image_pipeline_amd.pipeline import host_alloc
    W, H = 2448, 2048
    frame = host_alloc((H, W))
    frame[...] = synth.gen_frame(W, H, "bayer_rggb8", seed=1, kind="scene")
    p = RawImagePipeline(False, "", "", "", device=0)
    synth.configure_full_chain(p, W, H, "grey_world")
    p.set_ring_depth(depth)
    tickets = []
    for i in range(40):
        if len(tickets) == depth:
            p.collect(tickets.pop(0), copy=False)
        tickets.append(p.submit(frame, "bayer_rggb8"))
    while tickets:
        p.collect(tickets.pop(0), copy=False)


def report(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
    mc = [n for n in names if "memory_cop" in n.lower()]
    print("copy tables/views:", mc)
    view = "memory_copies" if "memory_copies" in names else mc[0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view).fetchall()]
    print("columns:", cols)
    rows = cur.execute("select * from %s order by start" % view).fetchall()
    ci = {c: i for i, c in enumerate(cols)}
    t0 = rows[0][ci["start"]]
    for r in rows[-40:]:
        dur = (r[ci["end"]] - r[ci["start"]]) / 1e3
        print("copy %-28s size %9s  start %10.1f us  dur %8.1f us" % (str(r[ci.get("name", 0)])[:28], r[ci["size"]] if "size" in ci else "?", (r[ci["start"]] - t0) / 1e3, dur))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2])
