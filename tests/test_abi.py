"""The C-ABI library loads, exports every symbol include/rip.h declares, and refuses to process
frames without a HIP device (no compute calls here: this file runs on the CPU-only box)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_whole_reference_surface():
    syms = declared_symbols()
    assert len(syms) >= 80
    # one entry point per reference setter / getter (raw_image_pipeline.hpp:59-137)
    for name in ["rip_apply", "rip_apply_device", "rip_load_params", "rip_load_camera_calibration", "rip_load_color_calibration",
                 "rip_init_undistortion", "rip_reset_white_balance_temporal_consistency", "rip_set_debayer_encoding",
                 "rip_set_white_balance_saturation_threshold", "rip_set_undistortion_projection_matrix",
                 "rip_get_rect_projection_matrix", "rip_is_color_enhancer_enabled", "rip_get_image"]:
        assert name in syms


def test_library_exports_every_declared_symbol(rip_lib):
    missing = [s for s in declared_symbols() if not hasattr(rip_lib, s)]
    assert not missing, "declared in include/rip.h but not exported: %s" % missing


def test_no_cpu_fallback_without_a_device(rip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raw_image_pipeline_amd import RawImagePipeline, RipError
    with pytest.raises(RipError, match="no CPU execution path"):
        RawImagePipeline(False, "", "", "", device=0)
    with pytest.raises(RipError, match="no CPU execution path"):
        RawImagePipeline(False)
    host = RawImagePipeline(False, "", "", "", device=-1)
    with pytest.raises(RipError, match="no CPU execution path"):
        host.process(np.zeros((8, 8), np.uint8), "bayer_rggb8")
    with pytest.raises((RipError, ValueError)):
        host.get_white_balance_info(1)


def test_version_and_error_strings(rip_lib):
    assert b"gfx950" in rip_lib.rip_version()
    h = C.c_void_p()
    st = rip_lib.rip_create(-1, 0, b"", b"", b"", None)
    assert st == 1 and b"null" in rip_lib.rip_last_error(None)
    assert rip_lib.rip_create(-1, 0, b"", b"", b"", C.byref(h)) == 0
    assert rip_lib.rip_set_color_calibration_matrix(h, (C.c_double * 3)(1, 2, 3), 3) == 1   # needs 9 values
    assert b"9 values" in rip_lib.rip_last_error(h)
    assert rip_lib.rip_load_ccc_model(h, b"/nonexistent/model.bin") == 3
    rip_lib.rip_destroy(h)


def test_copy_host_copies_every_size_and_survives_concurrent_callers(rip_lib):
    """rip_copy_host (the split memcpy behind rip_collect's / rip_get_image's deep copies): sizes around the 4 MB threshold
    and odd ones byte for byte, nothing written past the end, four threads copying at once (one at a time gets the pool,
    the others copy on their own thread)."""
    import ctypes as C
    import threading
    import numpy as np
    lib = rip_lib
    lib.rip_copy_host.restype = None
    lib.rip_copy_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(0)
    for n in (0, 1, 4095, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, 2448 * 2048 * 3, (16 << 20) + 3):
        src = rng.integers(0, 256, n, dtype=np.uint8)
        dst = np.zeros(n + 16, np.uint8)
        lib.rip_copy_host(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        assert np.array_equal(dst[:n], src) and not dst[n:].any(), n
    errors = []

    def worker(seed):
        r = np.random.default_rng(seed)
        for _ in range(8):
            n = int(r.integers(1 << 20, 12 << 20))
            src = r.integers(0, 256, n, dtype=np.uint8)
            dst = np.empty(n, np.uint8)
            lib.rip_copy_host(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), C.c_size_t(n))
            if not np.array_equal(dst, src):
                errors.append((seed, n))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_copy_host_in_a_forked_child_copies_on_its_own_thread(rip_lib):
    """The copy pool's worker threads do not exist in a forked child (and its mutexes may have been inherited locked): the
    child's rip_copy_host must fall back to a plain memcpy instead of waiting for workers."""
    import ctypes as C
    import os
    import numpy as np
    lib = rip_lib
    lib.rip_copy_host.restype = None
    lib.rip_copy_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    src = np.random.default_rng(1).integers(0, 256, 9 << 20, dtype=np.uint8)
    dst = np.empty_like(src)
    lib.rip_copy_host(dst.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), C.c_size_t(src.size))  # the pool exists now
    assert np.array_equal(dst, src)
    pid = os.fork()
    if pid == 0:
        out = np.empty_like(src)
        lib.rip_copy_host(out.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), C.c_size_t(src.size))
        os._exit(0 if np.array_equal(out, src) else 1)
    deadline = 30.0
    import time
    t0 = time.time()
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        if time.time() - t0 > deadline:
            os.kill(pid, 9)
            os.waitpid(pid, 0)
            raise AssertionError("the forked child hung in rip_copy_host")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
