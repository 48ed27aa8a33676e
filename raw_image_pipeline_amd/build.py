"""Builds librip_hip.so (the C-ABI library, include/rip.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so it travels with the repo snapshot.
-ffp-contract=off is part of the numerical contract (OpenCV's separate float mul/add)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librip_hip.so")
SOURCES = ["rip_chain.hip", "rip_stats.hip", "rip_ccc.hip", "rip_remap.hip", "rip_maps.hip", "rip_fused.hip", "rip_probe.hip", "rip_host.cpp", "rip_api.cpp"]  # compiled in parallel
HEADERS = ["rip_kernels.hpp", "rip_device.hpp", "rip_chain_dev.hpp", "rip_remap_dev.hpp", "rip_tile.hpp", "rip_host.hpp", os.path.join("..", "..", "include", "rip.h")]
# per-source additions.  rip_chain.hip: LLVM's max-ILP machine scheduler -- the fused chain is bound by VALU issue and LDS at
# six waves per SIMD and gains 2.3 % from the extra instruction-level parallelism inside a wave (2.311 -> 2.257 ms per 256
# frames); the same strategy costs the memory-bound remap 3.5 % and the ccc kernels 8 %, so it is not a global flag.
PER_SOURCE_FLAGS = {"rip_chain.hip": ["-mllvm", "-amdgpu-sched-strategy=" + os.environ.get("RIP_CHAIN_SCHED", "max-ilp")] if os.environ.get("RIP_CHAIN_SCHED", "max-ilp") != "default" else []}  # RIP_CHAIN_SCHED: A/B builds of other strategies (tools/ab_chain.py)
# translation units compiled a second time under the contracted floating-point model (rip_device.hpp RIP_FP_CONTRACT; object
# <name>_fc1.o): the kernels rip_set_fp_contraction(1) selects.  -ffp-contract stays off: the fused forms are written out.
FC1_SOURCES = ["rip_chain.hip", "rip_fused.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return all(os.path.getmtime(d) <= t for d in deps)


# --asan: the HOST layer (rip_host.cpp, rip_api.cpp: YAML reader, loaders, table builders, frame ring, copy threads) under
# AddressSanitizer + UndefinedBehaviorSanitizer, compiled with g++ against GCC's runtimes; the device code is compiled by hipcc
# as always.  (Not clang's runtime: the ROCm build of compiler-rt intercepts hsa_amd_memory_pool_allocate for device-side ASan and
# aborts the first HIP allocation of a process on a GPU box -- "allocator is trying to allocate 0x400000 bytes" -- without
# xnack+ code objects.)  tools/README.md "Sanitizer build".
ASAN_OUT = os.path.join(HERE, "librip_hip_asan.so")
ASAN_HOST_CXX = os.environ.get("RIP_ASAN_CXX", "g++")
ASAN_HOST_FLAGS = ["-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__",
                   "-I/opt/rocm/include", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=undefined,float-cast-overflow",
                   "-fno-omit-frame-pointer", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]


def _gcc_lib(name):
    r = subprocess.run([ASAN_HOST_CXX, "-print-file-name=" + name], stdout=subprocess.PIPE, text=True)
    return os.path.realpath(r.stdout.strip())


def asan_runtime():
    """Path of GCC's shared ASan runtime (the LD_PRELOAD a python process needs to load librip_hip_asan.so)."""
    return _gcc_lib("libasan.so")


def build(force=False, verbose=False, out=None, extra_flags=None, tag="", asan=False):
    """out / extra_flags / tag: A/B variants for experiments (tools/ab_chain.py), e.g. extra_flags=["-DRIP_X=1"],
    out=".../variants/x.so"; the default build takes no extra flags beyond $RIP_EXTRA_FLAGS."""
    if asan:
        out, tag = out or ASAN_OUT, tag or "_asan"
        if not force and os.path.exists(out) and all(os.path.getmtime(os.path.join(CSRC, d)) <= os.path.getmtime(out) for d in SOURCES + HEADERS):
            return out
    if out is None and not force and up_to_date():
        return OUT
    out = out or OUT
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build" + tag)
    os.makedirs(bdir, exist_ok=True)
    for s, fc in [(s, 0) for s in SOURCES] + [(s, 1) for s in FC1_SOURCES]:
        obj = os.path.join(bdir, os.path.splitext(s)[0] + ("_fc1" if fc else "") + ".o")
        if asan and s.endswith(".cpp"):
            cmd = [ASAN_HOST_CXX] + ASAN_HOST_FLAGS + list(extra_flags or []) + ["-c", os.path.join(CSRC, s), "-o", obj]
        else:
            cmd = [hipcc()] + FLAGS + os.environ.get("RIP_EXTRA_FLAGS", "").split() + list(extra_flags or []) + PER_SOURCE_FLAGS.get(s, []) + (["-DRIP_FP_CONTRACT=1"] if fc else []) + ["-x", "hip", "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for s, pr in procs:
        log, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, log))
        if verbose and log.strip():
            print(log)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ([_gcc_lib("libasan.so"), _gcc_lib("libubsan.so"), "-lstdc++", "-lpthread"] if asan else [])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, asan="--asan" in sys.argv))
    if "--asan" in sys.argv:
        print("LD_PRELOAD=" + asan_runtime())
