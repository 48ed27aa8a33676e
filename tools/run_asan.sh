#!/bin/bash
# Runs a command (default: the CPU test suite) against the sanitizer builds of the host layer and of the oracle:
#   tools/run_asan.sh                      -> python -m pytest tests -q -m "not gpu" -x
#   tools/run_asan.sh python -m pytest tests/test_host_fuzz.py -q
# Builds librip_hip_asan.so / librip_oracle_asan.so (g++ / gcc, GCC's libasan + libubsan) first if they are missing (tools/README.md "Sanitizer build").
set -eu
cd "$(dirname "$0")/.."
python raw_image_pipeline_amd/build.py --asan > /dev/null
make -s -C oracle asan
rt=$(python -c "import sys; sys.path.insert(0, '.'); from raw_image_pipeline_amd import build as b; print(b.asan_runtime())")
export RIP_LIBRARY=$PWD/raw_image_pipeline_amd/librip_hip_asan.so RIP_ORACLE_ASAN=1
# detect_leaks=0: the interpreter itself leaks by design; protect_shadow_gap=0: the HIP runtime maps into ASan's shadow gap
export ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0:protect_shadow_gap=0:abort_on_error=1:halt_on_error=1}
export UBSAN_OPTIONS=${UBSAN_OPTIONS:-print_stacktrace=1:halt_on_error=1}
if [ $# -eq 0 ]; then set -- python -m pytest tests -q -m "not gpu" -x; fi
# libstdc++ right behind the runtime: python itself does not link it, and GCC's ASan resolves its __cxa_throw interceptor when it
# starts -- without this the first C++ exception of the process (ours: every refused call) dies in "CHECK failed: real___cxa_throw"
cxx=$(g++ -print-file-name=libstdc++.so.6)
# ASan's dlopen interceptor loses the caller's RUNPATH: torch finds its own libraries (libcaffe2_nvrtc.so at CUDA init) only with
# its lib directory on the search path
tl=$(python -c "import importlib.util, os; print(os.path.join(os.path.dirname(importlib.util.find_spec('torch').origin), 'lib'))" 2>/dev/null || true)
export LD_LIBRARY_PATH=${tl:+$tl:}${LD_LIBRARY_PATH:-}
LD_PRELOAD=$rt:$cxx${LD_PRELOAD:+:$LD_PRELOAD} exec "$@"
