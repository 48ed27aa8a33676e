#!/usr/bin/env python3
"""Instruction ledger of the fused chain's frame loop (one 8-pixel item of one frame per trip), priced with the measured
issue costs of tools/isa_cycles.py, for the kernel variants bench.py launches.  Writes profiles/chain_ledger.json (read by
bench.py for the `roofline.valu` record) and prints the per-opcode tables (committed as profiles/rNN_chain_isa_ledger.txt).

The frame loop is found in the hipcc -save-temps listing as the depth-2 loop of the kernel: every basic block LLVM annotates
with "in Loop: Header=<frame loop header>" plus the header itself.  That is a STATIC count: blocks behind wave-uniform
branches that the benchmark never takes (debayered tap, non-zero colour bias, abToXZ linear segment, image-edge fix-ups)
are inside the loop too; they are listed separately by the marker instructions they contain and subtracted for the
`executed` figure.  `--pmc-valu-per-wave N` (SQ_INSTS_VALU / SQ_WAVES of the same launch) cross-checks the executed count.

usage: chain_ledger.py [--clock GHz] [--pmc workload=valu_per_wave,iterations_per_wave ...] [--measured workload=valu_per_item ...]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_cycles as ic  # noqa: E402

# workload -> kernel variant substring (BITS, WB, threads): bench.py's configurations
VARIANTS = {"config2": "chain_fast_kernelILi7ELi1ELi512E", "chain": "chain_fast_kernelILi7ELi0ELi512E", "default_chain": "chain_fast_kernelILi3ELi1ELi256E",
            "config3": "chain_fast_kernelILi8ELi2ELi256E", "config5": "chain_fast_kernelILi0ELi0ELi256E"}
# blocks the benchmark never executes, recognised by an instruction only they contain
RARE_MARKERS = [("abToXZ linear segment", re.compile(r"0x4ded21")), ("colour bias != 0", None), ("debayered tap", None),
                ("colour enhancer with hue / value gains != 1", re.compile(r"rip_generic_hsv_gains"))]
LDS_CYCLES = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read_u8": 2, "ds_read_u16": 2, "ds_read2_b32": 4, "ds_read_b96": 8}


def compile_listing(tmp):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
             "-fno-slp-vectorize", "-D__HIP_PLATFORM_AMD__", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-x", "hip", "-c", os.path.join(ROOT, "raw_image_pipeline_amd", "csrc", "rip_chain.hip"),
             "-save-temps", "-o", "chain.o"]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(tmp, "rip_chain-hip-amdgcn-amd-amdhsa-gfx950.s")


def loop_blocks(lines):
    """Splits the kernel body into basic blocks and returns those of the deepest loop (the frame loop)."""
    blocks, cur = [], {"label": "entry", "comment": "", "ins": []}
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "comment": m.group(2) or "", "ins": []}
            continue
        m = re.match(r"^; %bb\.\d+:\s*(;.*)?$", l)
        if m:
            blocks.append(cur)
            cur = {"label": "", "comment": l, "ins": []}
            continue
        t = l.strip()
        if l.startswith("\t") and t and t[0] not in ".;":
            cur["ins"].append(t)
        elif t.startswith(";") and "in Loop" in t or "Loop Header" in t or "Parent Loop" in t:
            cur["comment"] += " " + t
    blocks.append(cur)
    depth = max([int(d) for b in blocks for d in re.findall(r"Depth=(\d+)", b["comment"])] or [0])
    if depth == 0:
        return blocks
    # round 5: the kernel holds two frame loops (fast_chunks<PLAIN = true / false>); the benchmark runs the first one in
    # code order (no debayered tap, zero colour bias)
    hdr = None
    for b in blocks:
        if re.search(r"Loop Header: Depth=%d" % depth, b["comment"]):
            hdr = b["label"].lstrip(".L")
            break
    return [b for b in blocks if ("Header=%s Depth=%d" % (hdr, depth)) in b["comment"] or
            (b["label"].lstrip(".L") == hdr and re.search(r"Loop Header: Depth=%d" % depth, b["comment"]))]


RARE_BLOCK = [re.compile(r"0x4ded21"), re.compile(r"^v_swap_b32")]


def is_rare(b):
    ins = b["ins"]
    if any(r.search(t) for t in ins for r in RARE_BLOCK):
        return True
    if sum(1 for t in ins if t.startswith("v_add_f32") and ic.SGPR_SRC.search(t.split(",", 1)[1])) >= 3:
        return True  # colour bias != 0
    if sum(1 for t in ins if t.startswith("v_perm_b32")) >= 6 and any("buffer_store_dwordx3" in t for t in ins):
        return True  # debayered tap
    if sum(1 for t in ins if re.match(r"v_lsh(l|r)rev_b32_e32 v\d+, 8, v\d+", t)) >= 6:
        return True  # image-border fix-ups of the demosaic
    if sum(1 for t in ins if t.startswith("v_cndmask_b32_e64")) >= 3 and len(ins) <= 12:
        return True  # first / last row fix-up
    if any(t.startswith("v_cmp_gt_i32") for t in ins) and any(t.startswith("s_and_saveexec") for t in ins) and len(ins) <= 4:
        return True  # per-lane tests of the abToXZ linear segment
    return False


def executed_path(blocks):
    """The blocks one trip of the frame loop runs in the benchmark: the walk from the loop header back to it that avoids the
    blocks recognised as never-taken (is_rare; the generic colour-enhancer gains between their marker instructions) and
    otherwise takes the LONGER side of every branch -- in the benchmark's configuration every optional block that is not one
    of the rare ones does run, and the structurizer's complementary `if (a) .. if (!a) ..` pairs would otherwise let a
    shortest walk skip both sides (round 5; replaces the neighbourhood heuristics of rounds 2-4).  Edges: fall-through unless
    the block ends in s_branch, plus every forward branch target inside the loop; the loop body is a DAG."""
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    n = len(blocks)
    succ = [[] for _ in range(n)]
    back = [False] * n
    for i, b in enumerate(blocks):
        fall = True
        for t in b["ins"]:
            m = re.match(r"s_(cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", t)
            if not m:
                continue
            j = index.get(m.group(2))
            if j == 0:
                back[i] = True
            elif j is not None and j > i:
                succ[i].append(j)
            if m.group(1) == "branch":
                fall = False
        if fall and i + 1 < n:
            succ[i].append(i + 1)
    rare = [is_rare(b) for b in blocks]
    inside = False
    for i, b in enumerate(blocks):  # the generic colour-enhancer gains: everything after the block that opens the bracket
        if inside:
            rare[i] = True
        if any("rip_generic_hsv_gains" in t for t in b["ins"]):
            inside, rare[i] = True, False
        if any("rip_generic_hsv_end" in t for t in b["ins"]):
            inside = False
    score = [None] * n
    nxt = [None] * n
    for i in range(n - 1, -1, -1):
        own = (-10 ** 6 if rare[i] else 0) + sum(1 for t in blocks[i]["ins"] if t.startswith("v_"))
        best, arg = (0, None) if back[i] else (None, None)
        for j in succ[i]:
            if score[j] is not None and (best is None or score[j] > best):
                best, arg = score[j], j
        if best is not None:
            score[i], nxt[i] = own + best, arg
    path, i = [], 0
    while i is not None:
        path.append(i)
        i = nxt[i]
    return [blocks[i] for i in path]


def price(blocks):
    cyc, cnt = collections.Counter(), collections.Counter()
    lds = 0
    for b in blocks:
        for t in b["ins"]:
            op = t.split()[0]
            c = ic.cost(t)
            key = op + (" [sgpr src]" if c == 4.3 and ic.FAST.match(op) else "")
            cnt[key] += 1
            cyc[key] += c
            if op.startswith("ds_"):
                lds += LDS_CYCLES.get(op, 2)
    return cyc, cnt, lds


def main():
    clock = 2.1
    pmc = {}
    measured = {}  # --measured workload=valu_per_item: the two-point hardware count (EXPERIMENTS.md, round 4)
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--clock":
            clock = float(args.pop(0))
        elif a == "--measured":
            while args and "=" in args[0]:
                k, v = args.pop(0).split("=")
                measured[k] = float(v)
        elif a == "--pmc":
            while args and "=" in args[0]:
                k, v = args.pop(0).split("=")
                pmc[k] = [float(x) for x in v.split(",")]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        listing = compile_listing(tmp)
        for wl, pat in VARIANTS.items():
            name, lines = ic.kernel_lines(listing, pat)
            blocks = loop_blocks(lines)
            taken = executed_path(blocks)
            taken_ids = set(id(b) for b in taken)
            rare = [b for b in blocks if id(b) not in taken_ids]
            cyc, cnt, lds = price(blocks)
            rcyc, rcnt, rlds = price(rare)
            valu = sum(v for k, v in cnt.items() if k.startswith("v_"))
            rvalu = sum(v for k, v in rcnt.items() if k.startswith("v_"))
            tot, rtot = sum(cyc.values()), sum(rcyc.values())
            rec = {"kernel": pat, "static_valu_instr": valu, "static_valu_cycles": round(tot, 1), "rare_valu_instr": rvalu,
                   "rare_valu_cycles": round(rtot, 1), "valu_cycles_per_item": round(tot - rtot, 1), "valu_instr_per_item": valu - rvalu,
                   "lds_instr_per_item": sum(v for k, v in cnt.items() if k.startswith("ds_")), "lds_cycles_per_item": lds - rlds,
                   "salu_instr_per_item": sum(v for k, v in cnt.items() if k.startswith("s_")), "pixels_per_item": 8, "clock_GHz": clock}
            if wl in pmc:
                per_wave, iters = pmc[wl]
                rec["pmc_valu_instr_per_item"] = round(per_wave / iters, 1)
            out[wl] = rec
            print("== %s: %s" % (wl, name[:100]))
            print("   frame loop, static: %d VALU instructions, %.0f issue cycles; never-taken blocks: %d / %.0f; executed estimate: %d / %.0f"
                  % (valu, tot, rvalu, rtot, valu - rvalu, tot - rtot))
            print("   per pixel-wave: %.1f VALU issue cycles, %.1f LDS cycles (conflict-free), %d LDS / %d SALU instructions per item"
                  % ((tot - rtot) / 8, (lds - rlds) / 8.0, rec["lds_instr_per_item"], rec["salu_instr_per_item"]))
            # per-opcode table of the blocks that are NOT recognised as never-taken (round 4: the table used to be the static
            # listing, which made the colour-bias adds, the 16 compares of the abToXZ linear segment and the other three
            # demosaic patterns look like executed work).  Still static inside those blocks: the four-way Bayer-pattern switch
            # contributes the merges of all patterns (about 60 instructions per item, 18 of them v_perm_b32) of which one
            # pattern's share runs -- the hardware count (`measured_valu_per_item`, two-point PMC measurement) is the
            # executed figure.
            rare_ids = set(id(b) for b in rare)
            ecyc, ecnt, _ = price([b for b in blocks if id(b) not in rare_ids])
            if wl in measured:
                rec["measured_valu_per_item"] = measured[wl]
                print("   executed per item, measured (SQ_INSTS_VALU at 16 and 8 frames per visit, setup separated): %.0f VALU instructions"
                      % measured[wl])
            print("   per opcode, never-taken blocks excluded:")
            for k, v in sorted(ecyc.items(), key=lambda kv: -kv[1])[:28]:
                if v:
                    print("      %-36s n=%-4d cycles=%.0f" % (k, ecnt[k], v))
    with open(os.path.join(ROOT, "profiles", "chain_ledger.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
