"""ctypes binding of the CPU oracle (oracle/f360_oracle.c) for the tests, smoke() and
bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE: never imported by the product package."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libf360_oracle.so")

_lib = None


def build() -> str:
    out = subprocess.run(["make", "-C", ORACLE_DIR], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + out.stdout + out.stderr)
    return ORACLE_SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = ctypes.CDLL(ORACLE_SO)
        L.f360o_fnv1a64.restype = c_uint64
        L.f360o_fnv1a64.argtypes = [c_void_p, c_size_t]
        L.f360o_pipeline_encode_sample.restype = c_uint64
        L.f360o_pipeline_encode_sample.argtypes = [c_int, c_int, c_int, c_int, c_int, c_uint32,
                                                   POINTER(c_double)]
        L.f360o_pipeline_compute.restype = c_uint64
        L.f360o_pipeline_compute.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                             POINTER(c_double)]
        L.f360o_lcg_fill.argtypes = [c_void_p, c_size_t, c_uint32]
        L.f360o_sat_encode.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int]
        L.f360o_satdec_grid.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_satdec_grid_axes.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_satdec_sample_rect.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                               c_int, c_void_p, c_float, c_float]
        L.f360o_satdec_interpolate_rect.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int,
                                                    c_int, c_float, c_float]
        L.f360o_satdec_decode.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int]
        L.f360o_expand_rect.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                        c_int, c_float, c_float]
        L.f360o_expand_logpolar.argtypes = L.f360o_expand_rect.argtypes
        L.f360o_yuv_to_rgb_pixel.argtypes = [c_int, c_int, c_int, c_int, c_void_p]
        L.f360o_yuv420p_to_rgb0.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                            c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_rgb0_to_yuv420p.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                            c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_rgb2yuv_chroma_vfilter.argtypes = [c_void_p, c_void_p, c_int, c_int]
        L.f360o_rgb2yuv_coeffs.argtypes = [c_void_p]
        L.f360o_is_grid.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_is_sample_rect.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                           c_int, c_void_p, c_float, c_float]
        L.f360o_is_logpolar_grid.argtypes = [c_void_p, c_int, c_int, c_int, c_int]
        L.f360o_is_sample_logpolar.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                               c_int, c_int, c_void_p, c_float, c_float]
        L.f360o_is_interpolate_logpolar.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int,
                                                    c_int, c_float, c_float]
        L.f360o_is_logpolar_blur.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.f360o_gnomonic.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float,
                                     c_float]
        L.f360o_set_float_model.argtypes = [c_int]
        _lib = L
    return _lib


def _ptr(a: np.ndarray) -> c_void_p:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_void_p)


def set_float_model(model: int) -> None:
    lib().f360o_set_float_model(model)


def lcg_frame(width: int, height: int, seed: int, bpp: int = 4, linesize: int | None = None):
    """every byte in memory order from the LCG (4th byte included)."""
    linesize = linesize or width * bpp
    buf = np.empty(height * linesize, dtype=np.uint8)
    lib().f360o_lcg_fill(_ptr(buf), buf.size, seed)
    return buf.reshape(height, linesize)


def fnv1a64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().f360o_fnv1a64(_ptr(a), a.nbytes))


def sat_encode(frame: np.ndarray, width: int, height: int, linesize: int) -> np.ndarray:
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    sat = np.empty((height, width, 3), dtype=np.uint32)
    lib().f360o_sat_encode(_ptr(sat), _ptr(frame), width, height, linesize)
    return sat


def satdec_grid(out_w, out_h, src_w, src_h) -> np.ndarray:
    g = np.empty((out_h + 1, out_w + 1, 2), dtype=np.int16)
    lib().f360o_satdec_grid(_ptr(g), out_w, out_h, src_w, src_h)
    return g


def satdec_grid_axes(out_w, out_h, src_w, src_h):
    gx = np.empty(out_w + 1, dtype=np.int16)
    gy = np.empty(out_h + 1, dtype=np.int16)
    lib().f360o_satdec_grid_axes(_ptr(gx), _ptr(gy), out_w, out_h, src_w, src_h)
    return gx, gy


def satdec_sample_rect(dst: np.ndarray, out_w, out_h, out_linesize, sat, src_w, src_h, grid,
                       cx, cy) -> np.ndarray:
    """dst is modified in place (pre-filled by the caller) and returned."""
    sat = np.ascontiguousarray(sat, dtype=np.uint32)
    lib().f360o_satdec_sample_rect(_ptr(dst), out_w, out_h, out_linesize, _ptr(sat), src_w,
                                   src_h, _ptr(grid), cx, cy)
    return dst


def satdec_interpolate_rect(src: np.ndarray, out_w, out_h, src_w, src_h, cx, cy) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((out_h, out_w, 4), dtype=np.uint8)
    lib().f360o_satdec_interpolate_rect(_ptr(dst), out_w, out_h, _ptr(src), src_w, src_h, cx, cy)
    return dst


def satdec_decode(dst: np.ndarray, dst_linesize, sat, width, height) -> np.ndarray:
    sat = np.ascontiguousarray(sat, dtype=np.uint32)
    lib().f360o_satdec_decode(_ptr(dst), dst_linesize, _ptr(sat), width, height)
    return dst


def is_grid(out_w, out_h, src_w, src_h) -> np.ndarray:
    g = np.empty((out_h, out_w, 2), dtype=np.int16)
    lib().f360o_is_grid(_ptr(g), out_w, out_h, src_w, src_h)
    return g


def is_sample_rect(dst, out_w, out_h, out_linesize, src, src_w, src_h, src_linesize, grid, cx,
                   cy) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    lib().f360o_is_sample_rect(_ptr(dst), out_w, out_h, out_linesize, _ptr(src), src_w, src_h,
                               src_linesize, _ptr(grid), cx, cy)
    return dst


def is_logpolar_grid(out_w, out_h, src_w, src_h) -> np.ndarray:
    g = np.empty((out_h, out_w, 2), dtype=np.int16)
    lib().f360o_is_logpolar_grid(_ptr(g), out_w, out_h, src_w, src_h)
    return g


def is_sample_logpolar(dst, out_w, out_h, out_linesize, src, src_w, src_h, src_linesize, grid,
                       cx, cy) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    lib().f360o_is_sample_logpolar(_ptr(dst), out_w, out_h, out_linesize, _ptr(src), src_w,
                                   src_h, src_linesize, _ptr(grid), cx, cy)
    return dst


def is_interpolate_logpolar(src, out_w, out_h, src_w, src_h, cx, cy) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((out_h, out_w, 4), dtype=np.uint8)
    lib().f360o_is_interpolate_logpolar(_ptr(dst), out_w, out_h, _ptr(src), src_w, src_h, cx, cy)
    return dst


def is_logpolar_blur(src, w, h) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((h, w, 4), dtype=np.uint8)
    lib().f360o_is_logpolar_blur(_ptr(dst), w, h, _ptr(src))
    return dst


def gnomonic(src, dst_w, dst_h, src_w, src_h, cx, cy) -> np.ndarray:
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dst_h, dst_w, 4), dtype=np.uint8)
    lib().f360o_gnomonic(_ptr(dst), dst_w, dst_h, _ptr(src), src_w, src_h, cx, cy)
    return dst


def pipeline_encode_sample(frames, src_w, src_h, out_w, out_h, seed0):
    """(digest, seconds) of the CPU hot path -- bench.py's cpu_baseline ("port")."""
    sec = c_double(0.0)
    d = lib().f360o_pipeline_encode_sample(frames, src_w, src_h, out_w, out_h, seed0,
                                           ctypes.byref(sec))
    return int(d), sec.value


def rgb0_to_yuv420p(rgb0, width, height, model, pads=(0, 0, 0)):
    """rgb0: (height, linesize) uint8 rows of 4-byte pixels -> (y, u, v) planes (rows padded by
    `pads` bytes, filled with 0xEE so that untouched bytes show)."""
    rgb0 = np.ascontiguousarray(rgb0, dtype=np.uint8)
    y = np.full((height, width + pads[0]), 0xEE, dtype=np.uint8)
    u = np.full((height // 2, width // 2 + pads[1]), 0xEE, dtype=np.uint8)
    v = np.full((height // 2, width // 2 + pads[2]), 0xEE, dtype=np.uint8)
    lib().f360o_rgb0_to_yuv420p(_ptr(y), y.shape[1], _ptr(u), u.shape[1], _ptr(v), v.shape[1],
                                _ptr(rgb0), rgb0.shape[1], width, height, model)
    return y, u, v


def rgb2yuv_chroma_vfilter(height):
    """(filter[ch][size], pos[ch]) of libswscale's vertical chroma filter for RGB -> yuv420p."""
    ch = (height + 1) // 2
    f = np.zeros((ch, 8), dtype=np.int16)
    p = np.zeros(ch, dtype=np.int32)
    size = lib().f360o_rgb2yuv_chroma_vfilter(_ptr(f), _ptr(p), height, 8)
    assert size <= 8
    return f.reshape(-1)[:ch * size].reshape(ch, size).copy(), p


def rgb2yuv_coeffs():
    c = np.zeros(9, dtype=np.int32)
    lib().f360o_rgb2yuv_coeffs(_ptr(c))
    return [int(v) for v in c]


def pipeline_compute(frames, first_index, src_w, src_h, out_w, out_h):
    """(digest, seconds) of the CPU hot path over frames synthesised beforehand
    (frames: uint8 array [n, src_h, 4 * src_w]); the clock covers the compute only."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    sec = c_double(0.0)
    d = lib().f360o_pipeline_compute(_ptr(frames), frames.shape[0], first_index, src_w, src_h,
                                     out_w, out_h, ctypes.byref(sec))
    return int(d), sec.value


YUV_SWS_C, YUV_SWS_X86 = 0, 1


def yuv_to_rgb_pixel(model, Y, U, V):
    rgb = np.zeros(3, dtype=np.uint8)
    lib().f360o_yuv_to_rgb_pixel(model, Y, U, V, _ptr(rgb))
    return tuple(int(c) for c in rgb)


def yuv420p_to_rgb0(y, u, v, width, height, model, dst=None, dst_linesize=None) -> np.ndarray:
    """y: (height, y_linesize), u/v: (height/2, linesize) uint8 arrays."""
    y, u, v = (np.ascontiguousarray(p, dtype=np.uint8) for p in (y, u, v))
    if dst is None:
        dst_linesize = 4 * width
        dst = np.zeros((height, dst_linesize), dtype=np.uint8)
    lib().f360o_yuv420p_to_rgb0(_ptr(dst), dst_linesize, _ptr(y), y.shape[1], _ptr(u), u.shape[1],
                                _ptr(v), v.shape[1], width, height, model)
    return dst


def expand(kind, dst, dst_w, dst_h, dst_linesize, src, src_w, src_h, src_linesize, cx, cy):
    """kind: "rect" or "logpolar"; dst is modified in place (untouched pixels keep their value)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    fn = lib().f360o_expand_rect if kind == "rect" else lib().f360o_expand_logpolar
    fn(_ptr(dst), dst_w, dst_h, dst_linesize, _ptr(src), src_w, src_h, src_linesize, cx, cy)
    return dst
