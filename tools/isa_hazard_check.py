#!/usr/bin/env python3
"""Static check of a hipcc -save-temps listing for a hazard the compiler cannot pad: on gfx90a and later the result of a
dot instruction (v_dot2* / v_dot4* / v_dot8*) needs THREE wait states before a different VALU instruction reads it (LLVM
GCNHazardRecognizer: DotWriteDifferentVALURead).  hipcc inserts s_nop for the instructions it emits, but it does not look
inside inline assembly, so an asm block whose first instructions read a fresh dot result executes on stale registers
(round 4: the one-channel remap gather).  This walks every kernel of the listing and reports each instruction INSIDE an
;;#ASMSTART ... ;;#ASMEND block that reads a VGPR written by a dot instruction fewer than three wait states earlier
(an s_nop N counts N + 1, every other instruction 1; a branch target resets nothing -- the check is per straight-line
listing order, which is conservative enough for the kernels here).

usage: isa_hazard_check.py file.hip [...]      (compiles each with the library's flags)   exit code 1 on a finding"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-fno-slp-vectorize", "-D__HIP_PLATFORM_AMD__"]
NEED = 3


def regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_listing(path):
    findings = []
    kernel, in_asm = "?", False
    fresh = {}  # vgpr -> wait states elapsed since a dot instruction wrote it
    for raw in open(path):
        line = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel, fresh = m.group(1), {}
            continue
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        operands = t[len(op):].split(";")[0]
        parts = [p.strip() for p in operands.split(",")]
        dst, srcs = (parts[0] if parts else ""), parts[1:]
        if in_asm and op.startswith("v_"):
            reads = [r for s in srcs for r in regs(s)]
            if op.endswith("c_i32_i16") or op.startswith(("v_fmac", "v_mac", "v_dot2c", "v_dot4c")):
                reads += regs(dst)  # accumulating forms read their destination
            for r in reads:
                if r in fresh and fresh[r] < NEED:
                    findings.append("%s: `%s` inside inline asm reads v%d %d wait state(s) after a dot instruction wrote it (needs %d)"
                                    % (kernel[:70], t, r, fresh[r], NEED))
        steps = 1
        mn = re.match(r"s_nop\s+(\d+)", t)
        if mn:
            steps = int(mn.group(1)) + 1
        for r in list(fresh):
            fresh[r] += steps
            if fresh[r] >= NEED:
                del fresh[r]
        if op.startswith("v_"):
            for r in regs(dst):
                fresh.pop(r, None)
            if op.startswith("v_dot"):
                for r in regs(dst):
                    fresh[r] = 0
    return findings


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        # every object the library links: the sources under the default flags, and the translation units build.py compiles a
        # second time under the contracted floating-point model (the *_fc1 kernels ship too)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raw_image_pipeline_amd"))
        import build as B
        jobs = []
        for src in sys.argv[1:]:
            src = os.path.abspath(src)
            jobs.append((src, [], ""))
            if os.path.basename(src) in B.FC1_SOURCES:
                jobs.append((src, ["-DRIP_FP_CONTRACT=1"], " (RIP_FP_CONTRACT=1)"))
        for n, (src, defs, tag) in enumerate(jobs):
            sub = os.path.join(tmp, str(n))
            os.makedirs(sub)
            extra = B.PER_SOURCE_FLAGS.get(os.path.basename(src), [])
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + defs + ["-x", "hip", "-c", src, "-save-temps", "-o", "x.o"], cwd=sub, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            lst = [f for f in os.listdir(sub) if f.endswith("gfx950.s") and os.path.splitext(os.path.basename(src))[0] in f]
            f = check_listing(os.path.join(sub, lst[0]))
            print("%s%s: %d finding(s)" % (os.path.basename(src), tag, len(f)))
            for line in f[:20]:
                print("   " + line)
            bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
