"""Host-side mirror of the reference's Python binding (raw_image_pipeline_python/src/
raw_image_pipeline_python.cpp:14-74: class ``RawImagePipeline``, 2 constructors, snake_case methods)
implemented over the C-ABI of ``librip_hip.so`` (include/rip.h) with ctypes.

All per-frame work runs in the HIP kernels of csrc/rip_{chain,stats,ccc,remap}.hip; this module only marshals
numpy / torch buffers.  It never falls back to a CPU implementation: a missing library or a
missing GPU raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librip_hip.so")

RIP_OK, RIP_ERR_INVALID_ARGUMENT, RIP_ERR_ASSERT, RIP_ERR_IO, RIP_ERR_DEVICE, RIP_ERR_CAPACITY = range(6)
TAP_DEBAYERED, TAP_COLOR, TAP_PROCESSED = 1, 2, 4
IMAGE_DEBAYERED, IMAGE_COLOR, IMAGE_PROCESSED, IMAGE_RECT_MASK = 0, 1, 2, 3


class RipError(RuntimeError):
    """HIP/device failure (RIP_ERR_DEVICE, RIP_ERR_CAPACITY)."""


class RipAssertError(RipError):
    """Input the reference's OpenCV call would have asserted on (cv::Exception)."""


class RipIOError(RipError, IOError):
    """Malformed YAML / model file (YAML::Exception class of failures)."""


_lib = None


def load_library(path=None):
    """Loads librip_hip.so.  Raises if it has not been built (``python -m raw_image_pipeline_amd.build``
    or ``__graft_entry__.build()``): there is no other implementation to fall back to."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("RIP_LIBRARY") or LIB_PATH  # RIP_LIBRARY: A/B runs of two builds on one GPU box
    try:
        # torch bundles its own libamdhip64 (same soname as ROCm's): import it first so that this
        # library binds to the SAME HIP runtime -- device pointers and streams are shared with torch.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `python raw_image_pipeline_amd/build.py` "
                           "(hipcc, gfx950). The pipeline has no CPU fallback." % path)
    lib = C.CDLL(path)
    lib.rip_last_error.restype = C.c_char_p
    lib.rip_last_error.argtypes = [C.c_void_p]
    lib.rip_version.restype = C.c_char_p
    lib.rip_destroy.restype = None
    lib.rip_destroy.argtypes = [C.c_void_p]
    lib.rip_copy_host.restype = None
    lib.rip_copy_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    _lib = lib
    return lib


def _as_doubles(values, n=None):
    arr = (C.c_double * len(values))(*[float(v) for v in values])
    return arr, len(values)


def _wrap_pinned(raw, shape, dtype):
    """THE array over a ctypes byte buffer: built in one step, so that its ``base`` is the buffer object and not another
    ndarray.  numpy collapses the base chain of a view through every non-owning ndarray until it meets something that is not
    one: a slice / reshape / reversed view of this array therefore references THIS array -- the object the ``OutputPool``
    counts references of.  (Until round 5 this was ``np.frombuffer(raw).reshape(shape)``: views of the reshaped array
    referenced the hidden frombuffer array, the pool saw no reference and recycled page-locked memory a caller's slice still
    looked at -- ADVICE round 4.)"""
    return np.ndarray(tuple(shape), dtype=np.dtype(dtype), buffer=raw)


def host_alloc(shape, dtype=np.uint8):
    """A numpy array over page-locked host memory (rip_host_alloc): frames handed to ``submit`` from it are uploaded
    asynchronously without the staging copy a pageable frame gets -- and must stay untouched until their ``collect``.
    The memory is released with the array."""
    import weakref
    lib = load_library()
    lib.rip_host_alloc.restype = C.c_void_p
    lib.rip_host_alloc.argtypes = [C.c_size_t]
    lib.rip_host_free.restype = None
    lib.rip_host_free.argtypes = [C.c_void_p]
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib.rip_host_alloc(max(n, 1))
    if not ptr:
        raise MemoryError("rip_host_alloc(%d) failed" % n)
    raw = (C.c_uint8 * max(n, 1)).from_address(ptr)
    weakref.finalize(raw, lib.rip_host_free, C.c_void_p(ptr))  # the ctypes object is the base of every view below
    return _wrap_pinned(raw, shape, dtype)


class OutputPool:
    """Destination arrays for the deep copies a pipeline hands out (``RawImagePipeline.out_pool``).  A fresh 15 MB numpy
    array costs more than filling it -- the kernel zeroes every new page -- so the pool keeps the arrays it has handed out
    and hands one out again once NOBODY else references it any more (CPython reference count: only the pool's own list
    holds it; a view of an array keeps its base alive and therefore counts).  The caller still owns what it gets, for as long
    as it keeps it: an array in use is never recycled.  At most ``limit`` arrays per shape are remembered."""

    def __init__(self, limit=8, pinned=False):
        self.limit = int(limit)
        self.pinned = bool(pinned)  # page-locked arrays (host_alloc): ``submit(..., out=...)`` downloads straight into them
        self._arrays = {}
        self._refcount = getattr(sys, "getrefcount", None)

    def take(self, shape, dtype=np.uint8, or_none=False):
        """An array of this shape nobody else references.  A pinned pool whose ``limit`` arrays are all in use hands out an
        ordinary array (page-locked memory is only worth its allocation when it is recycled) -- or None with ``or_none``,
        for callers that need the page lock (``submit(..., out=)``)."""
        key = (tuple(shape), np.dtype(dtype).str)
        if key not in self._arrays and len(self._arrays) >= 4:
            # a camera whose image size keeps changing must not pile up page-locked memory: forget the arrays of the shapes
            # seen longest ago (whoever still holds one of them keeps it alive; it just is not recycled any more)
            for old in list(self._arrays)[:len(self._arrays) - 3]:
                del self._arrays[old]
        arrays = self._arrays.pop(key, [])
        self._arrays[key] = arrays  # most recently used last
        if self._refcount is not None:
            for i in range(len(arrays)):
                # references: the list, the call argument of getrefcount -- anything above is somebody else's
                if self._refcount(arrays[i]) == 2:
                    return arrays[i]
        if self.pinned and not (self._refcount is not None and len(arrays) < self.limit):
            return None if or_none else np.empty(shape, dtype)
        arr = host_alloc(shape, dtype) if self.pinned else np.empty(shape, dtype)
        if self._refcount is not None and len(arrays) < self.limit:
            arrays.append(arr)
        return arr


class RawImagePipeline:
    """Same surface as ``py_raw_image_pipeline.RawImagePipeline``.

    ``RawImagePipeline(use_gpu)`` mirrors the one-argument constructor (example params, example
    camera and colour calibration); ``RawImagePipeline(use_gpu, params_path, calibration_path,
    color_calibration_path)`` the four-argument one.  ``use_gpu`` is recorded only: every stage
    always runs on the MI355X and follows the reference's CPU/OpenCV arithmetic.
    """

    _ONE_ARG = object()

    def __init__(self, use_gpu=False, params_path=_ONE_ARG, calibration_path="", color_calibration_path="",
                 device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.last_encoding = None
        if params_path is RawImagePipeline._ONE_ARG:
            st = self._lib.rip_create_default(int(device), int(bool(use_gpu)), C.byref(self._h))
        else:
            st = self._lib.rip_create(int(device), int(bool(use_gpu)), (params_path or "").encode(),
                                      (calibration_path or "").encode(), (color_calibration_path or "").encode(),
                                      C.byref(self._h))
        if st != RIP_OK:
            msg = self._lib.rip_last_error(None).decode(errors="replace")
            self._h = C.c_void_p()
            self._raise(st, msg)
        self.device = int(device)
        self._torch_stream = None
        self.out_pool = None  # an OutputPool: collect() / the image getters then recycle the arrays of their deep copies
        self._given = {}       # ticket -> (out, tap_debayered, tap_color) arrays handed to submit()
        self._given_last = (None, None, None)  # ... of the frame collected last (what the image getters return)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.rip_destroy(h)
            self._h = C.c_void_p()

    close = __del__

    # ---- plumbing ------------------------------------------------------------------------------
    @staticmethod
    def _raise(st, msg):
        if st == RIP_ERR_INVALID_ARGUMENT:
            raise ValueError(msg)  # std::invalid_argument
        if st == RIP_ERR_ASSERT:
            raise RipAssertError(msg)
        if st == RIP_ERR_IO:
            raise RipIOError(msg)
        raise RipError(msg)

    def _check(self, st):
        if st != RIP_OK:
            self._raise(st, self._lib.rip_last_error(self._h).decode(errors="replace"))  # a message may quote bytes of a malformed file

    def _call(self, name, *args):
        self._check(getattr(self._lib, name)(self._h, *args))

    def set_stream(self, stream):
        """HIP stream handle (int / torch.cuda.Stream) the device work is enqueued on."""
        self._torch_stream = stream if hasattr(stream, "wait_stream") else None
        handle = getattr(stream, "cuda_stream", stream)
        self._call("rip_set_stream", C.c_void_p(int(handle) if handle else 0))

    # ---- frame API -------------------------------------------------------------------------------
    def query_output(self, rows, cols, channels, encoding):
        r, c, cn = C.c_int(), C.c_int(), C.c_int()
        enc = C.create_string_buffer(32)
        self._call("rip_query_output", int(rows), int(cols), int(channels), encoding.encode(), C.byref(r), C.byref(c),
                   C.byref(cn), enc)
        return r.value, c.value, cn.value, enc.value.decode()

    def query_taps(self, rows, cols, channels, encoding):
        """Geometry of the debayered / colour taps (post-flip, pre-undistortion): rows, cols, channels."""
        tr, tc, tcn = C.c_int(), C.c_int(), C.c_int()
        self._call("rip_query_taps", int(rows), int(cols), int(channels), encoding.encode(), C.byref(tr), C.byref(tc), C.byref(tcn))
        return tr.value, tc.value, tcn.value

    def process(self, image, encoding):
        """cv::Mat process(const cv::Mat&, std::string&): returns a new array; input untouched."""
        img = np.asarray(image)
        wide = img.dtype == np.uint16  # 16-bit Bayer extension (set_debayer_16bit): uint16 in, uint16 BGR out
        if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3):
            raise ValueError("image must be uint8 (or uint16 Bayer), HxW or HxWxC")
        if img.strides[-1] != img.itemsize or (img.ndim == 3 and img.strides[1] != img.shape[2] * img.itemsize):
            img = np.ascontiguousarray(img)
        rows, cols = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        orows, ocols, ocn, oenc = self.query_output(rows, cols, cn, encoding)
        if wide != oenc.endswith("16"):
            raise ValueError("dtype %s does not match encoding %s" % (img.dtype, encoding))
        oshape = (orows, ocols) if ocn == 1 else (orows, ocols, ocn)
        odtype = np.uint16 if wide else np.uint8
        out = self.out_pool.take(oshape, odtype) if self.out_pool is not None else np.empty(oshape, odtype)
        r, c, k = C.c_int(), C.c_int(), C.c_int()
        enc = C.create_string_buffer(32)
        self._given_last = (None, None, None)  # the image getters now belong to this frame
        self._call("rip_apply", img.ctypes.data_as(C.c_void_p), rows, cols, cn, C.c_size_t(img.strides[0]),
                   encoding.encode(), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes), C.byref(r), C.byref(c),
                   C.byref(k), enc)
        self.last_encoding = enc.value.decode()
        shape = (r.value, c.value) if k.value == 1 else (r.value, c.value, k.value)
        return out if out.shape == shape else out.reshape(shape)

    def apply(self, image, encoding):
        """bool apply(cv::Mat&, std::string&): returns the processed image; when it has the input's
        shape the input array is overwritten too (the reference re-seats the caller's Mat)."""
        out = self.process(image, encoding)
        if isinstance(image, np.ndarray) and image.shape == out.shape and image.flags.writeable:
            image[...] = out
        return out

    # ---- asynchronous host frames (rip_submit / rip_collect; no counterpart in the reference's binding) -------------
    def submit(self, image, encoding, out=None, tap_debayered=None, tap_color=None):
        """Enqueues upload + chain + download of one host frame and returns its ticket without waiting (rip_submit).  Up to
        ``set_ring_depth`` (default 3) frames may be in flight; frames are processed in submission order.

        ``out`` / ``tap_debayered`` / ``tap_color``: page-locked arrays (``host_alloc``, a pinned ``OutputPool``) of the
        result's / the taps' shape -- the downloads are written straight into them (rip_submit_to) and ``collect`` / the image
        getters return these very arrays: deep-copy semantics without a copy."""
        img = np.asarray(image)
        if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3):
            raise ValueError("image must be uint8 (or uint16 Bayer), HxW or HxWxC")
        if img.strides[-1] != img.itemsize or (img.ndim == 3 and img.strides[1] != img.shape[2] * img.itemsize):
            img = np.ascontiguousarray(img)
        rows, cols = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        t = C.c_uint64()
        given = (out, tap_debayered, tap_color)
        if all(g is None for g in given):
            self._call("rip_submit", img.ctypes.data_as(C.c_void_p), rows, cols, cn, C.c_size_t(img.strides[0]), encoding.encode(), C.byref(t))
            return t.value
        for g in given:
            if g is not None and not (isinstance(g, np.ndarray) and g.flags.c_contiguous and g.flags.writeable):
                raise ValueError("out / tap arrays must be writeable C-contiguous numpy arrays over page-locked memory")
        ptr = lambda g: g.ctypes.data_as(C.c_void_p) if g is not None else None
        taps = [g for g in (tap_debayered, tap_color) if g is not None]
        self._call("rip_submit_to", img.ctypes.data_as(C.c_void_p), rows, cols, cn, C.c_size_t(img.strides[0]), encoding.encode(),
                   ptr(out), C.c_size_t(out.nbytes if out is not None else 0), ptr(tap_debayered), ptr(tap_color),
                   C.c_size_t(min(g.nbytes for g in taps) if taps else 0), C.byref(t))
        self._given[t.value] = given
        return t.value

    def collect(self, ticket, copy=True):
        """Waits for the frame of ``ticket`` and returns its image (rip_collect).  ``copy=False`` returns a read-only view of
        the handle's pinned result buffer instead.  As rip.h says of rip_collect's view: it is valid until the NEXT ``collect``
        on this pipeline -- or until a ``submit`` finds every other slot in flight and takes this one (keep at most
        ``ring depth - 1`` frames in flight to hold on to a view while submitting); copy what must live longer."""
        r, c, k = C.c_int(), C.c_int(), C.c_int()
        enc = C.create_string_buffer(32)
        view = C.c_void_p()
        self._call("rip_collect", C.c_uint64(int(ticket)), None, C.c_size_t(0), C.byref(view), C.byref(r), C.byref(c), C.byref(k), enc)
        self.last_encoding = enc.value.decode()
        wide = self.last_encoding.endswith("16")
        n = r.value * c.value * k.value
        shape = (r.value, c.value) if k.value == 1 else (r.value, c.value, k.value)
        self._given_last = self._given.pop(int(ticket), (None, None, None))
        if self._given_last[0] is not None:  # the download went straight into the caller's array: it IS the deep copy
            g = self._given_last[0]
            return g if g.shape == tuple(shape) else g.reshape(shape)
        if copy:  # the library's copy (split over a few threads for whole frames), not numpy's
            dtype = np.uint16 if wide else np.uint8
            out = self.out_pool.take(shape, dtype) if self.out_pool is not None else np.empty(shape, dtype)
            self._lib.rip_copy_host(out.ctypes.data_as(C.c_void_p), view, C.c_size_t(out.nbytes))
            return out
        buf = (C.c_uint16 if wide else C.c_uint8) * n
        arr = np.frombuffer(buf.from_address(view.value), np.uint16 if wide else np.uint8).reshape(shape)
        arr.flags.writeable = False
        return arr

    def set_ring_depth(self, depth):
        self._call("rip_set_ring_depth", int(depth))

    def apply_device(self, frames, encoding, out=None, tap_debayered=None, tap_color=None):
        """Device-resident batch (rip_apply_device).  ``frames``: uint8 CUDA tensor [n, rows, cols]
        or [n, rows, cols, c] (torch) already in HBM; returns the output tensor [n, R, C(, 3)].
        Asynchronous on the handle's stream."""
        import torch
        if frames.dtype != torch.uint8 or not frames.is_cuda:
            raise ValueError("frames must be a uint8 CUDA tensor")
        if frames.dim() == 3:
            n, rows, cols = frames.shape
            cn = 1
        elif frames.dim() == 4:
            n, rows, cols, cn = frames.shape
        else:
            raise ValueError("frames must be [n, rows, cols] or [n, rows, cols, c]")
        if frames.stride(-1) != 1 or (frames.dim() == 4 and frames.stride(2) != cn):
            raise ValueError("pixels must be contiguous")
        in_step = frames.stride(1)
        in_frame = frames.stride(0) if n > 1 else in_step * rows
        orows, ocols, ocn, enc = self.query_output(rows, cols, cn, encoding)
        shape = (n, orows, ocols) if ocn == 1 else (n, orows, ocols, ocn)
        if out is None:
            out = torch.empty(shape, dtype=torch.uint8, device=frames.device)
        elif tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError("out must be a contiguous tensor of shape %s" % (shape,))
        if out.device != frames.device:
            raise ValueError("out must live on the device of frames")
        # taps hold the post-flip, pre-undistortion geometry (tightly packed): query it with undistortion off
        if tap_debayered is not None or tap_color is not None:
            tr, tc, tcn = C.c_int(), C.c_int(), C.c_int()
            self._call("rip_query_taps", int(rows), int(cols), int(cn), encoding.encode(), C.byref(tr), C.byref(tc), C.byref(tcn))
            tshape = (n, tr.value, tc.value) if tcn.value == 1 else (n, tr.value, tc.value, tcn.value)
            for t in (tap_debayered, tap_color):
                if t is not None and (t.dtype != torch.uint8 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != tshape
                                      or t.device != frames.device):
                    raise ValueError("tap tensors must be contiguous uint8 CUDA tensors of shape %s" % (tshape,))
        # The C call enqueues on the handle's stream; the tensors were produced / allocated on torch's current stream.
        # Order the two when they differ, and tell the caching allocator that the handle's stream uses the buffers.
        cur = torch.cuda.current_stream(frames.device)
        mine = self._torch_stream
        if mine is not None and mine.cuda_stream != cur.cuda_stream:
            mine.wait_stream(cur)
            for t in (frames, out, tap_debayered, tap_color):
                if t is not None:
                    t.record_stream(mine)
        with torch.cuda.device(frames.device):  # the C-ABI selects the handle's device; keep the caller's current device
            self._given_last = (None, None, None)
            self._call("rip_apply_device", C.c_void_p(frames.data_ptr()), C.c_size_t(in_step), C.c_size_t(in_frame), int(n),
                       int(rows), int(cols), int(cn), encoding.encode(), C.c_void_p(out.data_ptr()), C.c_size_t(0),
                       C.c_size_t(0), C.c_void_p(tap_debayered.data_ptr() if tap_debayered is not None else 0),
                       C.c_void_p(tap_color.data_ptr() if tap_color is not None else 0))
        self.last_encoding = enc
        return out

    def set_taps(self, mask):
        self._call("rip_set_taps", int(mask))

    def set_tap_download(self, mask):
        """Which of the kept taps (TAP_DEBAYERED | TAP_COLOR) ``submit`` downloads into pinned memory with the result
        (rip_set_tap_download; default none): what a front end that publishes them on every frame wants."""
        self._call("rip_set_tap_download", int(mask))

    def _get_image(self, which, copy=True):
        """``copy=False``: for a frame that came through ``collect`` a read-only view of the handle's pinned host memory
        (rip_get_image_view; same lifetime as ``collect(copy=False)``); frames of ``process`` only exist on the device and are
        copied as before."""
        r, c, k = C.c_int(), C.c_int(), C.c_int()
        given = {IMAGE_DEBAYERED: self._given_last[1], IMAGE_COLOR: self._given_last[2], IMAGE_PROCESSED: self._given_last[0]}.get(which)
        if given is not None:  # the tap of the collected frame was downloaded into the caller's own array (submit(..., tap_*=))
            return given
        if not copy:
            view = C.c_void_p()
            self._call("rip_get_image_view", which, C.byref(view), C.byref(r), C.byref(c), C.byref(k))
            if view.value:
                n = r.value * c.value * k.value
                arr = np.frombuffer((C.c_uint8 * n).from_address(view.value), np.uint8)
                arr = arr.reshape((r.value, c.value) if k.value == 1 else (r.value, c.value, k.value))
                arr.flags.writeable = False
                return arr
        self._call("rip_get_image", which, None, C.c_size_t(0), C.byref(r), C.byref(c), C.byref(k))
        if r.value == 0 or c.value == 0:
            return np.empty((0, 0), np.uint8)
        shape = (r.value, c.value) if k.value == 1 else (r.value, c.value, k.value)
        out = self.out_pool.take(shape) if self.out_pool is not None else np.empty(shape, np.uint8)
        self._call("rip_get_image", which, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(r), C.byref(c),
                   C.byref(k))
        return out

    def get_dist_debayered_image(self, copy=True):
        return self._get_image(IMAGE_DEBAYERED, copy)

    def get_dist_color_image(self, copy=True):
        return self._get_image(IMAGE_COLOR, copy)

    def get_rect_mask(self):
        return self._get_image(IMAGE_RECT_MASK)

    def get_processed_image(self, copy=True):
        return self._get_image(IMAGE_PROCESSED, copy)

    # ---- loaders ---------------------------------------------------------------------------------
    def load_params(self, path):
        self._call("rip_load_params", path.encode())

    def load_camera_calibration(self, path):
        self._call("rip_load_camera_calibration", path.encode())

    def load_color_calibration(self, path):
        self._call("rip_load_color_calibration", path.encode())

    def init_undistortion(self):
        self._call("rip_init_undistortion")

    def load_ccc_model(self, path):
        self._call("rip_load_ccc_model", path.encode())

    def set_ccc_model(self, filt, bias):
        filt = np.ascontiguousarray(filt, np.float32)
        bias = np.ascontiguousarray(bias, np.float32)
        if filt.shape != bias.shape or filt.ndim != 2:
            raise ValueError("filter and bias must be 2-D arrays of the same shape")
        self._call("rip_set_ccc_model", int(filt.shape[1]), int(filt.shape[0]), filt.ctypes.data_as(C.c_void_p),
                   bias.ctypes.data_as(C.c_void_p))

    def set_ccc_kalman_model(self, h, r):
        self._call("rip_set_ccc_kalman_model", C.c_double(h), C.c_double(r))

    # ---- other interfaces ------------------------------------------------------------------------
    def reset_white_balance_temporal_consistency(self):
        self._call("rip_reset_white_balance_temporal_consistency")

    def set_gpu(self, use_gpu):
        self._call("rip_set_gpu", int(bool(use_gpu)))

    def set_debug(self, debug):
        """While on, process()/apply() of an 8-bit frame also write the image after each of the eight modules to
        /tmp/0N_<module>.png (RIP_DEBUG_DIR replaces /tmp), min-max normalised like the reference's saveDebugImage
        (raw_image_pipeline.hpp:143-186)."""
        self._call("rip_set_debug", int(bool(debug)))

    def set_fp_contraction(self, mode):
        """Not in the reference: the floating-point contraction model of the float stages -- 0 (default) every product and
        sum rounded (OpenCV built for baseline x86-64), 1 the fused forms of an FMA-target build (aarch64 / Jetson).
        include/rip.h rip_set_fp_contraction."""
        self._call("rip_set_fp_contraction", int(mode))

    DEBUG_DUMP_NAMES = ("00_debayer", "01_flip", "02_white_balancing", "03_color_calibration", "04_gamma_correction",
                        "05_vignetting_correction", "06_color_enhancer", "07_undistortion")

    def debug_write_png(self, path, image, normalize=True):
        """Test hook: the PNG writer (and min-max normalisation) of the debug dumps on a host image (H x W or H x W x 3 BGR)."""
        a = np.ascontiguousarray(image, np.uint8)
        cn = 1 if a.ndim == 2 else a.shape[2]
        self._check(self._lib.rip_debug_write_png(self._h, str(path).encode(), a.ctypes.data_as(C.c_void_p), int(a.shape[0]),
                                                  int(a.shape[1]), int(cn), int(bool(normalize))))

    # ---- setters (names as in raw_image_pipeline_python.cpp:25-58) --------------------------------
    def set_debayer(self, enabled):
        self._call("rip_set_debayer", int(bool(enabled)))

    def set_debayer_16bit(self, enabled):
        """Extension: accept bayer_*16 frames (debayer + flip only, bgr16 out) instead of raising like the reference."""
        self._call("rip_set_debayer_16bit", int(bool(enabled)))

    def set_debayer_encoding(self, encoding):
        self._call("rip_set_debayer_encoding", encoding.encode())

    def set_flip(self, enabled):
        self._call("rip_set_flip", int(bool(enabled)))

    def set_flip_angle(self, angle):
        self._call("rip_set_flip_angle", int(angle))

    def set_white_balance(self, enabled):
        self._call("rip_set_white_balance", int(bool(enabled)))

    def set_white_balance_method(self, method):
        self._call("rip_set_white_balance_method", method.encode())

    def set_white_balance_percentile(self, percentile):
        self._call("rip_set_white_balance_percentile", C.c_double(percentile))

    def set_white_balance_saturation_threshold(self, bright_thr, dark_thr):
        self._call("rip_set_white_balance_saturation_threshold", C.c_double(bright_thr), C.c_double(dark_thr))

    def set_white_balance_temporal_consistency(self, enabled):
        self._call("rip_set_white_balance_temporal_consistency", int(bool(enabled)))

    def set_gamma_correction(self, enabled):
        self._call("rip_set_gamma_correction", int(bool(enabled)))

    def set_gamma_correction_method(self, method):
        self._call("rip_set_gamma_correction_method", method.encode())

    def set_gamma_correction_k(self, k):
        self._call("rip_set_gamma_correction_k", C.c_double(k))

    def set_vignetting_correction(self, enabled):
        self._call("rip_set_vignetting_correction", int(bool(enabled)))

    def set_vignetting_correction_parameters(self, scale, a2, a4):
        self._call("rip_set_vignetting_correction_parameters", C.c_double(scale), C.c_double(a2), C.c_double(a4))

    def set_color_enhancer(self, enabled):
        self._call("rip_set_color_enhancer", int(bool(enabled)))

    def set_color_enhancer_hue_gain(self, gain):
        self._call("rip_set_color_enhancer_hue_gain", C.c_double(gain))

    def set_color_enhancer_saturation_gain(self, gain):
        self._call("rip_set_color_enhancer_saturation_gain", C.c_double(gain))

    def set_color_enhancer_value_gain(self, gain):
        self._call("rip_set_color_enhancer_value_gain", C.c_double(gain))

    def set_color_calibration(self, enabled):
        self._call("rip_set_color_calibration", int(bool(enabled)))

    def set_color_calibration_matrix(self, matrix):
        arr, n = _as_doubles(list(np.asarray(matrix, dtype=np.float64).reshape(-1)))
        self._call("rip_set_color_calibration_matrix", arr, n)

    def set_color_calibration_bias(self, bias):
        arr, n = _as_doubles(list(np.asarray(bias, dtype=np.float64).reshape(-1)))
        self._call("rip_set_color_calibration_bias", arr, n)

    def set_undistortion(self, enabled):
        self._call("rip_set_undistortion", int(bool(enabled)))

    def set_undistortion_image_size(self, width, height):
        self._call("rip_set_undistortion_image_size", int(width), int(height))

    def set_undistortion_new_image_size(self, width, height):
        self._call("rip_set_undistortion_new_image_size", int(width), int(height))

    def set_undistortion_balance(self, balance):
        self._call("rip_set_undistortion_balance", C.c_double(balance))

    def set_undistortion_fov_scale(self, fov_scale):
        self._call("rip_set_undistortion_fov_scale", C.c_double(fov_scale))

    def _set_vec(self, name, values):
        arr, n = _as_doubles(list(np.asarray(values, dtype=np.float64).reshape(-1)))
        self._call(name, arr, n)

    def set_undistortion_camera_matrix(self, m):
        self._set_vec("rip_set_undistortion_camera_matrix", m)

    def set_undistortion_distortion_coeffs(self, d):
        self._set_vec("rip_set_undistortion_distortion_coefficients", d)

    def set_undistortion_distortion_model(self, model):
        self._call("rip_set_undistortion_distortion_model", model.encode())

    def set_undistortion_rectification_matrix(self, m):
        self._set_vec("rip_set_undistortion_rectification_matrix", m)

    def set_undistortion_projection_matrix(self, m):
        self._set_vec("rip_set_undistortion_projection_matrix", m)

    # ---- getters -----------------------------------------------------------------------------------
    def _flag(self, name):
        return bool(getattr(self._lib, name)(self._h))

    def is_debayer_enabled(self):
        return self._flag("rip_is_debayer_enabled")

    def is_flip_enabled(self):
        return self._flag("rip_is_flip_enabled")

    def is_white_balance_enabled(self):
        return self._flag("rip_is_white_balance_enabled")

    def is_color_calibration_enabled(self):
        return self._flag("rip_is_color_calibration_enabled")

    def is_gamma_correction_enabled(self):
        return self._flag("rip_is_gamma_correction_enabled")

    def is_vignetting_correction_enabled(self):
        return self._flag("rip_is_vignetting_correction_enabled")

    def is_color_enhancer_enabled(self):
        return self._flag("rip_is_color_enhancer_enabled")

    def is_undistortion_enabled(self):
        return self._flag("rip_is_undistortion_enabled")

    def get_dist_image_height(self):
        return int(self._lib.rip_get_dist_image_height(self._h))

    def get_dist_image_width(self):
        return int(self._lib.rip_get_dist_image_width(self._h))

    def get_rect_image_height(self):
        return int(self._lib.rip_get_rect_image_height(self._h))

    def get_rect_image_width(self):
        return int(self._lib.rip_get_rect_image_width(self._h))

    def _string(self, name):
        buf = C.create_string_buffer(64)
        self._call(name, buf, C.c_size_t(64))
        return buf.value.decode(errors="replace")

    def get_dist_distortion_model(self):
        return self._string("rip_get_dist_distortion_model")

    def get_rect_distortion_model(self):
        return self._string("rip_get_rect_distortion_model")

    def _matrix(self, name, shape):
        n = int(np.prod(shape))
        buf = (C.c_double * n)()
        self._call(name, buf)
        return np.array(buf, dtype=np.float64).reshape(shape)

    def get_color_calibration_matrix(self):
        return self._matrix("rip_get_color_calibration_matrix", (3, 3))

    def get_color_calibration_bias(self):
        return self._matrix("rip_get_color_calibration_bias", (4, 1))

    def get_dist_camera_matrix(self):
        return self._matrix("rip_get_dist_camera_matrix", (3, 3))

    def get_dist_distortion_coefficients(self):
        return self._matrix("rip_get_dist_distortion_coefficients", (1, 4))

    def get_dist_rectification_matrix(self):
        return self._matrix("rip_get_dist_rectification_matrix", (3, 3))

    def get_dist_projection_matrix(self):
        return self._matrix("rip_get_dist_projection_matrix", (3, 4))

    def get_rect_camera_matrix(self):
        return self._matrix("rip_get_rect_camera_matrix", (3, 3))

    def get_rect_distortion_coefficients(self):
        return self._matrix("rip_get_rect_distortion_coefficients", (1, 4))

    def get_rect_rectification_matrix(self):
        return self._matrix("rip_get_rect_rectification_matrix", (3, 3))

    def get_rect_projection_matrix(self):
        return self._matrix("rip_get_rect_projection_matrix", (3, 4))

    # ---- introspection ------------------------------------------------------------------------------
    def get_undistortion_maps(self):
        r, c = C.c_int(), C.c_int()
        self._call("rip_get_undistortion_maps", None, None, C.c_size_t(0), C.byref(r), C.byref(c))
        mx = np.empty((r.value, c.value), np.float32)
        my = np.empty((r.value, c.value), np.float32)
        self._call("rip_get_undistortion_maps", mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p),
                   C.c_size_t(mx.size), C.byref(r), C.byref(c))
        return mx, my

    def get_white_balance_info(self, n_frames=1):
        buf = np.empty((n_frames, 8), np.float32)
        self._call("rip_get_white_balance_info", buf.ctypes.data_as(C.c_void_p), int(n_frames))
        return buf

    def get_ccc_track(self, n_frames=1):
        """(n, 4) int32: raw argmax (x, y) and filtered (x, y) of the ccc estimator for the frames of the last batch."""
        buf = np.empty((n_frames, 4), np.int32)
        self._call("rip_get_ccc_track", buf.ctypes.data_as(C.c_void_p), int(n_frames))
        return buf

    def debug_plan_info(self, src_rows, src_cols):
        """Test hook (rip_debug_plan_info): dict describing the compiled remap plan of the current calibration."""
        info = (C.c_int * 9)()
        self._call("rip_debug_plan_info", int(src_rows), int(src_cols), info)
        keys = ("tiles_x", "tiles_y", "border_pixels", "max_lds_bytes", "max_rect_w", "max_rect_h", "on_device", "tile_w", "tile_h")
        return dict(zip(keys, [int(v) for v in info]))

    def set_tunable(self, name, value):
        """Launch tunable of this handle (rip_set_tunable: development / test hook; the library reads its environment
        overrides once, when the handle is created)."""
        self._call("rip_set_tunable", name.encode(), int(value))

    PROBE_KINDS = {"copy": 0, "read": 1, "fill": 2, "expand13": 3, "expand13_nt": 4, "copy12": 5, "expand13_wide": 6, "expand13_wide_nt": 7, "read_nt": 8, "expand13_coalesced": 9, "expand13_coalesced_nt": 10}

    def hbm_probe(self, kind, nbytes=1 << 30, reps=10):
        """GB/s (best of ``reps`` launches, bytes read + written) of one of the library's hand-written streaming kernels on
        this handle's device and stream (rip_debug_hbm_probe): the measured ceiling a kernel of that access shape can reach
        on this box."""
        out = C.c_double()
        self._call("rip_debug_hbm_probe", int(self.PROBE_KINDS.get(kind, kind)), C.c_size_t(int(nbytes)), int(reps), C.byref(out))
        return out.value

    KERNEL_CLASSES = ("stats", "ccc", "chain", "remap")

    def profile_begin(self, max_records):
        self._call("rip_profile_begin", int(max_records))

    def profile_end(self):
        """-> {class: (total_ms, launches)} from HIP events on the handle's stream."""
        ms = (C.c_double * 4)()
        cnt = (C.c_int * 4)()
        self._call("rip_profile_end", ms, cnt)
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNEL_CLASSES)}

    def debug_atan(self, values):
        """atan of the device map builder (double-double, rip_maps.hip) for an array of doubles."""
        a = np.ascontiguousarray(values, np.float64)
        out = np.empty_like(a)
        self._check(self._lib.rip_debug_atan(self._h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), int(a.size)))
        return out

    def get_vignetting_mask(self, rows, cols):
        """Host-built mask plane of precomputeVignettingMask (vignetting_correction.cpp:32-63)."""
        out = np.empty((int(rows), int(cols)), np.float32)
        self._check(self._lib.rip_get_vignetting_mask(self._h, int(rows), int(cols), out.ctypes.data_as(C.c_void_p),
                                                      C.c_size_t(out.size)))
        return out

    def get_table(self, which):
        buf = np.empty(4096, np.int32)
        n = self._lib.rip_get_table(self._h, int(which), buf.ctypes.data_as(C.c_void_p), 4096)
        if n < 0:
            raise ValueError("unknown table id")
        return buf[:n].copy()
