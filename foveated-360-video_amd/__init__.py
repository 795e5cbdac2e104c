"""foveated-360-video_amd -- MI355X-native frame-transform engine (host side, Python).

Thin ctypes binding over the C ABI of ``include/f360.h`` (``lib/libf360.so``, built
from ``csrc/`` with hipcc for gfx950).  The classes mirror the reference's C++
classes for this path -- same method names, argument order and units:

* ``SATEncoder``   -- /root/reference/src/sat_encoder.h:35-42
* ``SATDecoder``   -- /root/reference/src/sat_decoder.h:44-82
* ``ImageSampler`` -- /root/reference/src/image_sampler.h:53-101
* ``Projections``  -- /root/reference/src/projections.h:28-35
* ``Context``      -- replaces OpenCLManager, /root/reference/src/opencl_manager.h:8-22

Device buffers are plain integers (device addresses: ``DeviceBuffer.ptr`` or
``torch.Tensor.data_ptr()``), widths/heights are pixels, linesizes are bytes, the
gaze centre is two floats in [0, 1].

There is no CPU fallback: if ``libf360.so`` is missing, or no HIP device is
visible, the calls raise.  (The package is imported under its directory name via
``importlib.import_module("foveated-360-video_amd")`` or the ``f360_amd`` alias
module at the repository root, because the name contains hyphens.)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import (POINTER, byref, c_char_p, c_float, c_int, c_int16, c_size_t,
                    c_uint8, c_uint32, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
# F360_LIBRARY: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("F360_LIBRARY") or os.path.join(_HERE, "lib", "libf360.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

F360_OK = 0
F360_ERR_INVALID_ARG = -1
F360_ERR_NO_DEVICE = -2
F360_ERR_HIP = -3
F360_ERR_OOM = -4
F360_ERR_NOT_INITIALIZED = -5


class F360Error(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"f360 status {status}: {message}")
        self.status = status


def build_native(verbose: bool = False) -> str:
    """Compile csrc/ into lib/libf360.so (hipcc --offload-arch=gfx950)."""
    out = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise RuntimeError("building libf360.so failed")
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); every symbol include/f360.h declares
_SIGNATURES = {
    "f360_version": (c_int, []),
    "f360_last_error_string": (c_char_p, []),
    "f360_status_string": (c_char_p, [c_int]),
    "f360_device_count": (c_int, [POINTER(c_int)]),
    "f360_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "f360_ctx_create_on_stream": (c_int, [c_int, c_void_p, POINTER(c_void_p)]),
    "f360_ctx_destroy": (c_int, [c_void_p]),
    "f360_ctx_device": (c_int, [c_void_p, POINTER(c_int)]),
    "f360_ctx_stream": (c_int, [c_void_p, POINTER(c_void_p)]),
    "f360_sync": (c_int, [c_void_p]),
    "f360_malloc": (c_int, [c_void_p, c_size_t, POINTER(c_void_p)]),
    "f360_free": (c_int, [c_void_p, c_void_p]),
    "f360_memset": (c_int, [c_void_p, c_void_p, c_int, c_size_t]),
    "f360_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "f360_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "f360_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "f360_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    "f360_host_alloc_pinned": (c_int, [c_size_t, POINTER(c_void_p)]),
    "f360_host_free_pinned": (c_int, [c_void_p]),
    "f360_event_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "f360_event_destroy": (c_int, [c_void_p]),
    "f360_event_record": (c_int, [c_void_p, c_void_p]),
    "f360_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "f360_sat_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "f360_sat_encode_prepare": (c_int, [c_void_p, c_int, c_int]),
    "f360_sat_encode_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                      c_int, c_int]),
    "f360_sat_encode_batch_max": (c_int, []),
    "f360_sat_tables_alloc": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p)]),
    "f360_sat_tables_free": (c_int, [c_void_p, c_void_p]),
    "f360_sat_tables_report": (c_char_p, [c_void_p]),
    "f360_sat_encode_yuv420p_batch": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                              POINTER(c_void_p), POINTER(c_void_p), c_int, c_int,
                                              c_int, c_int, c_int]),
    "f360_yuv420p_to_rgb0": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_int]),
    "f360_rgb0_to_yuv420p": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p, c_int, c_int, c_int]),
    "f360_sat_encode_yuv420p": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_int, c_int, c_int, c_int]),
    "f360_satdec_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "f360_satdec_destroy": (c_int, [c_void_p]),
    "f360_satdec_initialize_grid": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "f360_satdec_export_grid": (c_int, [c_void_p, c_void_p]),
    "f360_satdec_sample_rect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_int, c_int, c_float, c_float]),
    "f360_satdec_sample_rect_batch": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_int,
                                              c_int, c_void_p, c_int, c_int, POINTER(c_float)]),
    "f360_satdec_sample_rect_frames": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_int,
                                               c_int, POINTER(c_void_p), c_int, c_int,
                                               POINTER(c_float)]),
    "f360_satdec_encode_sample_frames": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                                 POINTER(c_void_p), c_int, c_int, c_int, c_int,
                                                 c_int, c_int, c_int, POINTER(c_float)]),
    "f360_satdec_foveate_rect_frames": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p),
                                                c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                                POINTER(c_float)]),
    "f360_satdec_foveate_rect_frames_yuv420p": (c_int, [c_void_p, POINTER(c_void_p),
                                                        POINTER(c_void_p), POINTER(c_void_p),
                                                        POINTER(c_void_p), c_int, c_int, c_int,
                                                        c_int, c_int, c_int, c_int, c_int, c_int,
                                                        POINTER(c_float)]),
    "f360_satdec_encode_sample_frames_yuv420p": (c_int, [c_void_p, POINTER(c_void_p),
                                                         POINTER(c_void_p), POINTER(c_void_p),
                                                         POINTER(c_void_p), POINTER(c_void_p),
                                                         c_int, c_int, c_int, c_int, c_int, c_int,
                                                         c_int, c_int, c_int, POINTER(c_float)]),
    "f360_satdec_foveate_rect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                         c_int, c_int, c_float, c_float]),
    "f360_satdec_foveate_rect_yuv420p": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int,
                                                 c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                 c_int, c_int, c_int, c_float, c_float]),
    "f360_satdec_interpolate_rect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_void_p, c_int, c_int, c_int, c_float,
                                             c_float]),
    "f360_satdec_decode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int]),
    "f360_is_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "f360_is_destroy": (c_int, [c_void_p]),
    "f360_is_initialize_grid": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "f360_is_initialize_logpolar_grid": (c_int, [c_void_p, c_int, c_int, c_int, c_int]),
    "f360_is_export_grid": (c_int, [c_void_p, c_void_p]),
    "f360_is_export_logpolar_grid": (c_int, [c_void_p, c_void_p]),
    "f360_is_sample_rect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                    c_int, c_int, c_float, c_float]),
    "f360_is_sample_logpolar": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_int, c_int, c_int, c_float, c_float]),
    "f360_is_interpolate_logpolar": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_void_p, c_int, c_int, c_int, c_float,
                                             c_float]),
    "f360_is_logpolar_gaussian_blur": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_void_p]),
    "f360_gnomonic": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                              c_int, c_float, c_float]),
    "f360_ctx_set_option": (c_int, [c_void_p, c_char_p, c_int]),
    "f360_ctx_get_option": (c_int, [c_void_p, c_char_p, POINTER(c_int)]),
    "f360_kernel_count": (c_int, []),
    "f360_kernel_name": (c_char_p, [c_int]),
    "f360_ctx_profile_arm": (c_int, [c_void_p, c_int]),
    "f360_ctx_profile_read": (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int)]),
    "f360_ctx_profile_frames": (c_int, [c_void_p, c_int, POINTER(c_int)]),
    "f360_debug_walk_stats": (c_int, [c_void_p, c_void_p, c_int]),
    "f360_debug_walk_recoveries": (c_int, [c_void_p, POINTER(ctypes.c_uint)]),
    "f360_ctx_handoff_recoveries": (c_int, [c_void_p, POINTER(ctypes.c_uint)]),
    "f360_debug_gn_fast_sweep": (c_int, [c_void_p, c_int, ctypes.c_ulonglong, c_void_p, c_void_p]),
    "f360_debug_gnomonic_worklist": (c_int, [c_void_p, c_void_p]),
    "f360_ctx_profile_reset": (c_int, [c_void_p]),
    "f360_tables_satdec_grid_axis": (c_int, [c_void_p, c_int, c_int]),
    "f360_tables_is_grid_axis": (c_int, [c_void_p, c_int, c_int]),
    "f360_tables_logpolar_axes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "f360_expand_rect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                 c_int, c_float, c_float]),
    "f360_expand_logpolar": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                     c_int, c_int, c_float, c_float]),
    "f360_tables_interp_axis": (c_int, [c_void_p, c_int, c_int, c_int]),
    "f360_tables_yuv2rgb": (c_int, [c_void_p]),
}


def lib() -> ctypes.CDLL:
    """Load libf360.so (once).  Fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with __graft_entry__.build() or "
            f"`make -C {CSRC_DIR}`; this package has no CPU fallback")
    handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return handle


def _check(status: int) -> None:
    if status != F360_OK:
        msg = lib().f360_last_error_string()
        raise F360Error(status, msg.decode() if msg else "?")


def device_count() -> int:
    n = c_int(0)
    st = lib().f360_device_count(byref(n))
    return n.value if st == F360_OK else 0


def reduced_size(full: int) -> int:
    """16 * ceil(full / 1.8 / 16) -- /root/reference/src/run_satlogrectilinear.cc:368-369."""
    import math
    return 16 * math.ceil(full / 1.8 / 16)


class Context:
    """Replaces OpenCLManager: device + one in-order stream (opencl_manager.cc:7-67)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = c_void_p()
        if stream is None:
            _check(lib().f360_ctx_create(device, byref(self._h)))
        else:
            _check(lib().f360_ctx_create_on_stream(device, c_void_p(stream), byref(self._h)))
        self.device = device

    # OpenCLManager::InitializeContext() analogue for call-sequence parity
    def InitializeContext(self) -> int:
        return 0

    @property
    def handle(self) -> c_void_p:
        return self._h

    def finish(self) -> None:
        _check(lib().f360_sync(self._h))

    def set_option(self, key: str, value: int) -> None:
        _check(lib().f360_ctx_set_option(self._h, key.encode(), value))

    def get_option(self, key: str) -> int:
        v = c_int(0)
        _check(lib().f360_ctx_get_option(self._h, key.encode(), byref(v)))
        return v.value

    def yuv420p_to_rgb0(self, dst, dst_linesize: int, y, u, v, y_linesize: int,
                        u_linesize: int, v_linesize: int, width: int, height: int) -> None:
        """The sws_scale of VideoDecoder::GetFrame (video_decoder.cc:222-224) on the device."""
        _check(lib().f360_yuv420p_to_rgb0(self._h, _p(dst), dst_linesize, _p(y), _p(u), _p(v),
                                          y_linesize, u_linesize, v_linesize, width, height))

    def rgb0_to_yuv420p(self, y, u, v, y_linesize: int, u_linesize: int, v_linesize: int, src,
                        src_linesize: int, width: int, height: int) -> None:
        """The sws_scale of VideoEncoder::EncodeFrame (video_encoder.cc:380-395) on the device."""
        _check(lib().f360_rgb0_to_yuv420p(self._h, _p(y), _p(u), _p(v), y_linesize, u_linesize,
                                          v_linesize, _p(src), src_linesize, width, height))

    def expand_rect(self, dst, dst_w, dst_h, dst_linesize, src, src_w, src_h, src_linesize,
                    center_x, center_y) -> None:
        """ExpandSampledFrameRectCPU (sat_decoder.cc:555-616) on the device."""
        _check(lib().f360_expand_rect(self._h, _p(dst), dst_w, dst_h, dst_linesize, _p(src),
                                      src_w, src_h, src_linesize, center_x, center_y))

    def expand_logpolar(self, dst, dst_w, dst_h, dst_linesize, src, src_w, src_h, src_linesize,
                        center_x, center_y) -> None:
        """ExpandSampledFrameLogPolarCPU (image_sampler.cc:623-666) on the device."""
        _check(lib().f360_expand_logpolar(self._h, _p(dst), dst_w, dst_h, dst_linesize, _p(src),
                                          src_w, src_h, src_linesize, center_x, center_y))

    def profile_arm(self, calls: int) -> None:
        """Sample the next `calls` transform calls with HIP events around each kernel."""
        _check(lib().f360_ctx_profile_arm(self._h, calls))

    def profile_reset(self) -> None:
        _check(lib().f360_ctx_profile_reset(self._h))

    def profile_read(self) -> dict:
        """{kernel name: (total_ms, launches)} of the sampled calls since the last reset."""
        out = {}
        for k in range(lib().f360_kernel_count()):
            ms, n = ctypes.c_double(0), c_int(0)
            _check(lib().f360_ctx_profile_read(self._h, k, byref(ms), byref(n)))
            if n.value:
                out[lib().f360_kernel_name(k).decode()] = (ms.value, n.value)
        return out

    def profile_frames(self) -> dict:
        """{kernel name: frames} the sampled launches covered (a batched call's launch covers
        several frames)."""
        out = {}
        for k in range(lib().f360_kernel_count()):
            n = c_int(0)
            _check(lib().f360_ctx_profile_frames(self._h, k, byref(n)))
            if n.value:
                out[lib().f360_kernel_name(k).decode()] = n.value
        return out

    def debug_walk_stats(self, max_units: int = 4096):
        """(units, 4) uint64 array {start, end, slow waits, polls} of the last read-once encoder
        launch that ran with debug.ablate bit 8 (f360_debug_walk_stats)."""
        import numpy as np
        out = np.zeros((max_units, 8), dtype=np.uint64)
        n = lib().f360_debug_walk_stats(self._h, out.ctypes.data_as(c_void_p), max_units)
        if n < 0:
            _check(n)
        return out[:n]

    def handoff_recoveries(self) -> int:
        """f360_ctx_handoff_recoveries: strips of read-once launches that finished without their
        hand-off since the last call (the results are exact either way; time was lost)."""
        c = ctypes.c_uint(0)
        _check(lib().f360_ctx_handoff_recoveries(self._h, byref(c)))
        return c.value

    def debug_walk_recoveries(self) -> int:
        """Strips of read-once encoder launches that gave up waiting for their hand-off and
        finished alone (exactly) since the last call (f360_debug_walk_recoveries)."""
        c = ctypes.c_uint(0)
        _check(lib().f360_debug_walk_recoveries(self._h, byref(c)))
        return c.value

    def debug_gn_fast_sweep(self, kind: int, n: int = 1 << 30):
        """(largest |fast - double| seen, bound the guard assumes) of the gnomonic remap's float
        asin (kind 0: every float in [-1, 1]) or atan2 (kind 1: n argument pairs)."""
        worst, bound = ctypes.c_float(0), ctypes.c_float(0)
        _check(lib().f360_debug_gn_fast_sweep(self._h, kind, n, ctypes.byref(worst),
                                              ctypes.byref(bound)))
        return worst.value, bound.value

    def debug_gnomonic_worklist(self) -> int:
        """Pixels the last GnomonicProjection call resolved with the exact chain."""
        c = ctypes.c_uint(0)
        _check(lib().f360_debug_gnomonic_worklist(self._h, ctypes.byref(c)))
        return c.value

    def malloc(self, nbytes: int) -> "DeviceBuffer":
        return DeviceBuffer(self, nbytes)

    def upload(self, array) -> "DeviceBuffer":
        import numpy as np
        a = np.ascontiguousarray(array)
        buf = DeviceBuffer(self, a.nbytes)
        buf.copy_from_host(a)
        return buf

    def close(self) -> None:
        if self._h:
            lib().f360_ctx_destroy(self._h)
            self._h = c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DeviceBuffer:
    """cl::Buffer analogue (video_server.cc:224-232): device memory owned by the caller."""

    def __init__(self, ctx: Context, nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = c_void_p()
        _check(lib().f360_malloc(ctx.handle, self.nbytes, byref(p)))
        self.ptr = p.value

    def copy_from_host(self, array) -> None:  # cl::copy(queue, begin, end, buffer)
        import numpy as np
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        _check(lib().f360_memcpy_h2d(self.ctx.handle, c_void_p(self.ptr),
                                     a.ctypes.data_as(c_void_p), a.nbytes))

    def copy_to_host(self, dtype, shape):  # cl::copy(queue, buffer, begin, end)
        import numpy as np
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(lib().f360_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(c_void_p),
                                     c_void_p(self.ptr), out.nbytes))
        return out

    def fill(self, byte: int) -> None:
        _check(lib().f360_memset(self.ctx.handle, c_void_p(self.ptr), byte, self.nbytes))

    def free(self) -> None:
        if self.ptr:
            lib().f360_free(self.ctx.handle, c_void_p(self.ptr))
            self.ptr = 0

    def __int__(self) -> int:
        return self.ptr


class Event:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._h = c_void_p()
        _check(lib().f360_event_create(ctx.handle, byref(self._h)))

    def record(self) -> None:
        _check(lib().f360_event_record(self.ctx.handle, self._h))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = c_float(0)
        _check(lib().f360_event_elapsed_ms(self._h, stop._h, byref(ms)))
        return ms.value

    def destroy(self) -> None:
        if self._h:
            lib().f360_event_destroy(self._h)
            self._h = c_void_p()


def _p(x) -> c_void_p:
    return c_void_p(int(x))


class TablePool:
    """Tables allocated by f360_sat_tables_alloc: `.ptrs` (device pointers), `.report` (what was
    drawn and kept), `.free()`."""

    def __init__(self, ctx: "Context", width: int, height: int, count: int):
        self._ctx = ctx
        self._h = c_void_p()
        out = (c_void_p * count)()
        _check(lib().f360_sat_tables_alloc(ctx.handle, width, height, count, out, byref(self._h)))
        self.ptrs = [int(p) for p in out]
        self.report = lib().f360_sat_tables_report(self._h).decode()

    def read_table(self, k: int, shape):
        """Table k on the host, as uint32 of `shape` (tests)."""
        import numpy as np
        out = np.empty(shape, dtype=np.uint32)
        _check(lib().f360_memcpy_d2h(self._ctx.handle, out.ctypes.data_as(c_void_p),
                                     c_void_p(self.ptrs[k]), out.nbytes))
        return out

    def free(self) -> None:
        if self._h:
            _check(lib().f360_sat_tables_free(self._ctx.handle, self._h))
            self._h = c_void_p()


class SATEncoder:
    """sat_encoder.h:35-42.  ``SATEncoder()`` without a context is the CPU-only object of
    the reference (sat_encoder.cc:3): its GPU method reports and returns."""

    def __init__(self, cl_manager: Context | None = None):
        self.cl_manager = cl_manager

    def EncodeFrameGPU(self, cl_target_buffer, cl_source_buffer, source_width: int,
                       source_height: int, source_linesize: int) -> None:
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            "[SATEncoder::EncodeFrameGPU] Not initialized with a device context")
        _check(lib().f360_sat_encode(self.cl_manager.handle, _p(cl_target_buffer),
                                     _p(cl_source_buffer), source_width, source_height,
                                     source_linesize))

    def EncodeFramesGPU(self, cl_target_buffers, cl_source_buffers, source_width: int,
                        source_height: int, source_linesize: int) -> None:
        """EncodeFrameGPU for several frames of one geometry in shared launches
        (f360_sat_encode_batch): lists of device pointers, table k of source k.  Not in the
        reference."""
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            "[SATEncoder::EncodeFramesGPU] Not initialized with OpenCL")
        n = len(cl_target_buffers)
        if n != len(cl_source_buffers):
            raise ValueError("EncodeFramesGPU: as many tables as sources")
        sats = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        srcs = (c_void_p * n)(*[int(p) for p in cl_source_buffers])
        _check(lib().f360_sat_encode_batch(self.cl_manager.handle, n, sats, srcs, source_width,
                                           source_height, source_linesize))

    def AllocateTables(self, source_width: int, source_height: int, count: int) -> "TablePool":
        """`count` tables for EncodeFramesGPU calls of `count` frames, placed for the read-once
        encoder (f360_sat_tables_alloc: groups of tables are drawn, one launch is timed into each,
        the fastest are kept).  Not in the reference."""
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            "[SATEncoder::AllocateTables] Not initialized with OpenCL")
        return TablePool(self.cl_manager, source_width, source_height, count)

    def EncodeFramesYUV420PGPU(self, cl_target_buffers, planes, y_linesize: int, u_linesize: int,
                               v_linesize: int, source_width: int, source_height: int) -> None:
        """EncodeFrameYUV420PGPU for several frames in shared launches; `planes` is a list of
        (y, u, v) device pointers, one per table.  Not in the reference."""
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            "[SATEncoder::EncodeFramesYUV420PGPU] Not initialized with OpenCL")
        n = len(cl_target_buffers)
        if n != len(planes):
            raise ValueError("EncodeFramesYUV420PGPU: as many tables as plane triples")
        sats = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        ys = (c_void_p * n)(*[int(p[0]) for p in planes])
        us = (c_void_p * n)(*[int(p[1]) for p in planes])
        vs = (c_void_p * n)(*[int(p[2]) for p in planes])
        _check(lib().f360_sat_encode_yuv420p_batch(self.cl_manager.handle, n, sats, ys, us, vs,
                                                   y_linesize, u_linesize, v_linesize,
                                                   source_width, source_height))

    def EncodeFrameYUV420PGPU(self, cl_target_buffer, cl_y, cl_u, cl_v, y_linesize: int,
                              u_linesize: int, v_linesize: int, source_width: int,
                              source_height: int) -> None:
        """Table of the RGB0 frame sws_scale would make of the planes (video_decoder.cc:222),
        computed from the planes directly.  Not in the reference."""
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            "[SATEncoder::EncodeFrameYUV420PGPU] Not initialized with OpenCL")
        _check(lib().f360_sat_encode_yuv420p(self.cl_manager.handle, _p(cl_target_buffer),
                                             _p(cl_y), _p(cl_u), _p(cl_v), y_linesize,
                                             u_linesize, v_linesize, source_width,
                                             source_height))


class SATDecoder:
    """sat_decoder.h:44-82 (device methods)."""

    def __init__(self, cl_manager: Context | None = None):
        self.cl_manager = cl_manager
        self._h = c_void_p()
        if cl_manager is not None:
            _check(lib().f360_satdec_create(cl_manager.handle, byref(self._h)))

    def _need(self, what: str) -> None:
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            f"[SATDecoder::{what}] Not initialized with a device context")

    def InitializeGrid(self, target_width, target_height, source_width, source_height) -> None:
        self._need("InitializeGrid")
        _check(lib().f360_satdec_initialize_grid(self._h, target_width, target_height,
                                                 source_width, source_height))

    def export_grid(self, target_width, target_height):
        import numpy as np
        g = np.empty((target_height + 1, target_width + 1, 2), dtype=np.int16)
        _check(lib().f360_satdec_export_grid(self._h, g.ctypes.data_as(c_void_p)))
        return g

    def SampleFrameRectGPU(self, cl_target_buffer, target_width, target_height,
                           target_linesize, cl_source_buffer, codec_ctx, center_x,
                           center_y) -> None:
        """``codec_ctx`` is anything with ``.width``/``.height`` (or a (w, h) tuple): the
        reference reads only those two fields (sat_decoder.cc:328-329)."""
        self._need("SampleFrameRectGPU")
        w, h = (codec_ctx if isinstance(codec_ctx, tuple)
                else (codec_ctx.width, codec_ctx.height))
        _check(lib().f360_satdec_sample_rect(self._h, _p(cl_target_buffer), target_width,
                                             target_height, target_linesize,
                                             _p(cl_source_buffer), w, h, center_x, center_y))

    def SampleFrameRectGPUBatch(self, cl_target_buffers, target_width, target_height,
                                target_linesize, cl_source_buffer, codec_ctx, centers) -> None:
        """One launch for several gaze points against one table; `centers`
        is a list of (center_x, center_y), `cl_target_buffers` a list of device pointers."""
        self._need("SampleFrameRectGPUBatch")
        w, h = (codec_ctx if isinstance(codec_ctx, tuple)
                else (codec_ctx.width, codec_ctx.height))
        n = len(cl_target_buffers)
        ptrs = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_sample_rect_batch(self._h, ptrs, n, target_width,
                                                   target_height, target_linesize,
                                                   _p(cl_source_buffer), w, h, xy))

    def SampleFramesRectGPU(self, cl_target_buffers, target_width, target_height,
                            target_linesize, cl_source_buffers, codec_ctx, centers) -> None:
        """Frame k's table sampled at gaze k into target k, shared launches
        (f360_satdec_sample_rect_frames); lists of device pointers and of (cx, cy)."""
        self._need("SampleFramesRectGPU")
        w, h = (codec_ctx if isinstance(codec_ctx, tuple)
                else (codec_ctx.width, codec_ctx.height))
        n = len(cl_target_buffers)
        if n != len(cl_source_buffers) or n != len(centers):
            raise ValueError("SampleFramesRectGPU: as many targets as tables and gaze points")
        ptrs = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        sats = (c_void_p * n)(*[int(p) for p in cl_source_buffers])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_sample_rect_frames(self._h, ptrs, n, target_width,
                                                    target_height, target_linesize, sats, w, h,
                                                    xy))

    def EncodeSampleFramesGPU(self, cl_target_buffers, target_width, target_height,
                              target_linesize, cl_tables, cl_source_frames, source_width,
                              source_height, source_linesize, centers) -> None:
        """SATEncoder.EncodeFramesGPU + SampleFramesRectGPU for frames whose gaze is known before
        the encode (f360_satdec_encode_sample_frames; the reference's offline modes,
        run_satlogrectilinear.cc:932-938 -- its server reads the gaze after the encode and keeps
        the two calls): the same tables and reduced frames; the reduced pixels come out of the
        encoder's pass and no table is read back."""
        self._need("EncodeSampleFramesGPU")
        n = len(cl_target_buffers)
        if n != len(cl_tables) or n != len(cl_source_frames) or n != len(centers):
            raise ValueError("EncodeSampleFramesGPU: as many targets as tables, frames and gaze points")
        dsts = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        sats = (c_void_p * n)(*[int(p) for p in cl_tables])
        srcs = (c_void_p * n)(*[int(p) for p in cl_source_frames])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_encode_sample_frames(self._h, dsts, sats, srcs, n, target_width,
                                                      target_height, target_linesize,
                                                      source_width, source_height,
                                                      source_linesize, xy))

    def FoveateFramesRectGPU(self, cl_target_buffers, target_width, target_height,
                             target_linesize, cl_source_frames, source_width, source_height,
                             source_linesize, centers) -> None:
        """FoveateFrameRectGPU for a batch of frames (f360_satdec_foveate_rect_frames): the reduced
        frames of EncodeSampleFramesGPU without the tables."""
        self._need("FoveateFramesRectGPU")
        n = len(cl_target_buffers)
        if n != len(cl_source_frames) or n != len(centers):
            raise ValueError("FoveateFramesRectGPU: as many targets as frames and gaze points")
        dsts = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        srcs = (c_void_p * n)(*[int(p) for p in cl_source_frames])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_foveate_rect_frames(self._h, dsts, srcs, n, target_width,
                                                     target_height, target_linesize, source_width,
                                                     source_height, source_linesize, xy))

    def FoveateFramesRectYUV420PGPU(self, cl_target_buffers, target_width, target_height,
                                    target_linesize, planes, y_linesize, u_linesize, v_linesize,
                                    source_width, source_height, centers) -> None:
        """FoveateFramesRectGPU from planar YUV 4:2:0 frames (`planes`: list of (y, u, v))."""
        self._need("FoveateFramesRectYUV420PGPU")
        n = len(cl_target_buffers)
        if n != len(planes) or n != len(centers):
            raise ValueError("FoveateFramesRectYUV420PGPU: as many targets as frames and gaze points")
        dsts = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        ys = (c_void_p * n)(*[int(p[0]) for p in planes])
        us = (c_void_p * n)(*[int(p[1]) for p in planes])
        vs = (c_void_p * n)(*[int(p[2]) for p in planes])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_foveate_rect_frames_yuv420p(
            self._h, dsts, ys, us, vs, y_linesize, u_linesize, v_linesize, n, target_width,
            target_height, target_linesize, source_width, source_height, xy))

    def EncodeSampleFramesYUV420PGPU(self, cl_target_buffers, target_width, target_height,
                                     target_linesize, cl_tables, planes, y_linesize, u_linesize,
                                     v_linesize, source_width, source_height, centers) -> None:
        """EncodeSampleFramesGPU from planar YUV 4:2:0 frames: `planes` is a list of (y, u, v)
        device pointers (f360_satdec_encode_sample_frames_yuv420p)."""
        self._need("EncodeSampleFramesYUV420PGPU")
        n = len(cl_target_buffers)
        if n != len(cl_tables) or n != len(planes) or n != len(centers):
            raise ValueError("EncodeSampleFramesYUV420PGPU: as many targets as tables, frames and gaze points")
        dsts = (c_void_p * n)(*[int(p) for p in cl_target_buffers])
        sats = (c_void_p * n)(*[int(p) for p in cl_tables])
        ys = (c_void_p * n)(*[int(p[0]) for p in planes])
        us = (c_void_p * n)(*[int(p[1]) for p in planes])
        vs = (c_void_p * n)(*[int(p[2]) for p in planes])
        xy = (c_float * (2 * n))(*[float(v) for c in centers for v in c])
        _check(lib().f360_satdec_encode_sample_frames_yuv420p(
            self._h, dsts, sats, ys, us, vs, y_linesize, u_linesize, v_linesize, n, target_width,
            target_height, target_linesize, source_width, source_height, xy))

    def FoveateFrameRectGPU(self, cl_target_buffer, target_width, target_height,
                            target_linesize, cl_source_frame, source_width, source_height,
                            source_linesize, center_x, center_y) -> None:
        """EncodeFrameGPU + SampleFrameRectGPU in one pass for a gaze known up front
        (SURVEY.md 8f-1): same bytes, no table written or re-read."""
        self._need("FoveateFrameRectGPU")
        _check(lib().f360_satdec_foveate_rect(self._h, _p(cl_target_buffer), target_width,
                                              target_height, target_linesize,
                                              _p(cl_source_frame), source_width, source_height,
                                              source_linesize, center_x, center_y))

    def FoveateFrameRectYUV420PGPU(self, cl_target_buffer, target_width, target_height,
                                   target_linesize, cl_y, cl_u, cl_v, y_linesize, u_linesize,
                                   v_linesize, source_width, source_height, center_x,
                                   center_y) -> None:
        """FoveateFrameRectGPU from planar YUV 4:2:0.  Not in the reference."""
        _check(lib().f360_satdec_foveate_rect_yuv420p(
            self._h, _p(cl_target_buffer), target_width, target_height, target_linesize,
            _p(cl_y), _p(cl_u), _p(cl_v), y_linesize, u_linesize, v_linesize, source_width,
            source_height, center_x, center_y))

    def InterpolateFrameRectGPU(self, cl_target_buffer, target_width, target_height,
                                target_linesize, cl_source_buffer, source_width,
                                source_height, source_linesize, center_x, center_y) -> None:
        self._need("InterpolateFrameRectGPU")
        _check(lib().f360_satdec_interpolate_rect(self._h, _p(cl_target_buffer), target_width,
                                                  target_height, target_linesize,
                                                  _p(cl_source_buffer), source_width,
                                                  source_height, source_linesize, center_x,
                                                  center_y))

    def DecodeFrameGPU(self, cl_target_buffer, target_linesize, cl_source_buffer, width,
                       height) -> None:
        self._need("DecodeFrameGPU")
        _check(lib().f360_satdec_decode(self._h, _p(cl_target_buffer), target_linesize,
                                        _p(cl_source_buffer), width, height))

    def close(self) -> None:
        if self._h:
            lib().f360_satdec_destroy(self._h)
            self._h = c_void_p()


class ImageSampler:
    """image_sampler.h:53-101 (device methods; the image-pyramid pair has no kernel source
    in the reference and is out of scope)."""

    def __init__(self, cl_manager: Context | None = None):
        self.cl_manager = cl_manager
        self._h = c_void_p()
        if cl_manager is not None:
            _check(lib().f360_is_create(cl_manager.handle, byref(self._h)))

    def _need(self, what: str) -> None:
        if self.cl_manager is None:
            raise F360Error(F360_ERR_NOT_INITIALIZED,
                            f"[ImageSampler::{what}] Not initialized with a device context")

    def InitializeGrid(self, target_width, target_height, source_width, source_height) -> None:
        self._need("InitializeGrid")
        _check(lib().f360_is_initialize_grid(self._h, target_width, target_height,
                                             source_width, source_height))

    def InitializeLogpolarGrid(self, target_width, target_height, source_width,
                               source_height) -> None:
        self._need("InitializeLogpolarGrid")
        _check(lib().f360_is_initialize_logpolar_grid(self._h, target_width, target_height,
                                                      source_width, source_height))

    def export_grid(self, target_width, target_height):
        import numpy as np
        g = np.empty((target_height, target_width, 2), dtype=np.int16)
        _check(lib().f360_is_export_grid(self._h, g.ctypes.data_as(c_void_p)))
        return g

    def export_logpolar_grid(self, target_width, target_height):
        import numpy as np
        g = np.empty((target_height, target_width, 2), dtype=np.int16)
        _check(lib().f360_is_export_logpolar_grid(self._h, g.ctypes.data_as(c_void_p)))
        return g

    def SampleFrameRectGPU(self, cl_target_buffer, target_width, target_height,
                           target_linesize, cl_source_buffer, source_width, source_height,
                           source_linesize, center_x, center_y) -> None:
        self._need("SampleFrameRectGPU")
        _check(lib().f360_is_sample_rect(self._h, _p(cl_target_buffer), target_width,
                                         target_height, target_linesize, _p(cl_source_buffer),
                                         source_width, source_height, source_linesize,
                                         center_x, center_y))

    def SampleFrameLogPolarGPU(self, cl_target_buffer, target_width, target_height,
                               target_linesize, cl_source_buffer, source_width,
                               source_height, source_linesize, center_x, center_y) -> None:
        self._need("SampleFrameLogPolarGPU")
        _check(lib().f360_is_sample_logpolar(self._h, _p(cl_target_buffer), target_width,
                                             target_height, target_linesize,
                                             _p(cl_source_buffer), source_width, source_height,
                                             source_linesize, center_x, center_y))

    def InterpolateFrameLogPolarGPU(self, cl_target_buffer, target_width, target_height,
                                    target_linesize, cl_source_buffer, source_width,
                                    source_height, source_linesize, center_x,
                                    center_y) -> None:
        self._need("InterpolateFrameLogPolarGPU")
        _check(lib().f360_is_interpolate_logpolar(self._h, _p(cl_target_buffer), target_width,
                                                  target_height, target_linesize,
                                                  _p(cl_source_buffer), source_width,
                                                  source_height, source_linesize, center_x,
                                                  center_y))

    def ApplyLogPolarGaussianBlur(self, cl_target_buffer, target_width, target_height,
                                  target_linesize, cl_source_buffer) -> None:
        self._need("ApplyLogPolarGaussianBlur")
        _check(lib().f360_is_logpolar_gaussian_blur(self._h, _p(cl_target_buffer),
                                                    target_width, target_height,
                                                    target_linesize, _p(cl_source_buffer)))

    def close(self) -> None:
        if self._h:
            lib().f360_is_destroy(self._h)
            self._h = c_void_p()


class Projections:
    """projections.h:28-35.  Positional order follows projections.cc:51-55
    (target_width before target_height; the header swaps the two *names*)."""

    def __init__(self, cl_manager: Context):
        self.cl_manager = cl_manager

    def GnomonicProjection(self, cl_target_buffer, target_width, target_height,
                           target_linesize, cl_source_buffer, source_width, source_height,
                           source_linesize, center_x, center_y) -> None:
        _check(lib().f360_gnomonic(self.cl_manager.handle, _p(cl_target_buffer), target_width,
                                   target_height, target_linesize, _p(cl_source_buffer),
                                   source_width, source_height, source_linesize, center_x,
                                   center_y))


# ---- host-only table builders (no device needed; used by the CPU test tier) ----
def tables_satdec_grid_axis(n_out: int, n_src: int):
    import numpy as np
    g = np.empty(n_out + 1, dtype=np.int16)
    _check(lib().f360_tables_satdec_grid_axis(g.ctypes.data_as(c_void_p), n_out, n_src))
    return g


def tables_is_grid_axis(n_out: int, n_src: int):
    import numpy as np
    g = np.empty(n_out, dtype=np.int16)
    _check(lib().f360_tables_is_grid_axis(g.ctypes.data_as(c_void_p), n_out, n_src))
    return g


def tables_logpolar_axes(out_w: int, out_h: int):
    import numpy as np
    r = np.empty(out_w, dtype=np.float32)
    c = np.empty(out_h, dtype=np.float32)
    s = np.empty(out_h, dtype=np.float32)
    _check(lib().f360_tables_logpolar_axes(r.ctypes.data_as(c_void_p),
                                           c.ctypes.data_as(c_void_p),
                                           s.ctypes.data_as(c_void_p), out_w, out_h))
    return r, c, s


def tables_yuv2rgb() -> dict:
    """Constants of libswscale's yuv420p -> RGB converters (host side, no GPU)."""
    import numpy as np
    t = np.empty(16, dtype=np.int32)
    _check(lib().f360_tables_yuv2rgb(t.ctypes.data_as(c_void_p)))
    names = ("cy", "c0", "crv", "cbu", "cgu", "cgv", "r0", "gu0", "gv0", "b0", "yc", "vrc",
             "ubc", "vgc", "ugc", "yoff")
    return {n: int(v) for n, v in zip(names, t)}


def tables_interp_axis(value_range: int, n_full: int, n_reduced: int):
    import numpy as np
    t = np.empty((2 * value_range + 1, 4), dtype=np.int32)
    _check(lib().f360_tables_interp_axis(t.ctypes.data_as(c_void_p), value_range, n_full,
                                         n_reduced))
    return t
