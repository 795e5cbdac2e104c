"""GPU tier: csrc/cr_math.h on the device.  The fast asin / atan2 must return the CORRECTLY
ROUNDED float -- the oracle's definition of the OpenCL builtins: evaluate in double, round once --
for every argument they vouch for, and vouch for all but a few in 10,000; the gnomonic remap built
on them stays bit-exact against the oracle for every table layout, with and without them."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAZES = [(0.0, 0.0), (0.5, 0.5), (0.65, 0.75), (0.0, 1.0), (1.0, 1.0), (0.999, 0.5), (0.3, 0.2)]


def probe(f360, ctx, kind, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    n = a.size
    da = ctx.upload(a.view(np.uint8))
    db = ctx.upload(np.ascontiguousarray(b, dtype=np.float32).view(np.uint8)) if b is not None else None
    out, flag = ctx.malloc(4 * n), ctx.malloc(n)
    st = f360.lib().f360_debug_cr_math(ctx.handle, kind, ctypes.c_size_t(n), ctypes.c_void_p(da.ptr),
                                       ctypes.c_void_p(db.ptr if db else 0),
                                       ctypes.c_void_p(out.ptr), ctypes.c_void_p(flag.ptr))
    assert st == 0, f360.lib().f360_last_error_string()
    got = out.copy_to_host(np.float32, (n,))
    ok = flag.copy_to_host(np.uint8, (n,)).astype(bool)
    for x in (da, db, out, flag):
        if x is not None:
            x.free()
    return got, ok


def same_bits(x, y):
    return np.array_equal(x.view(np.uint32), y.view(np.uint32))


def test_fast_asin_is_correctly_rounded_where_it_vouches(f360, gpu_ctx):
    rng = np.random.default_rng(11)
    n = 1 << 24
    a = np.concatenate([
        rng.uniform(-1, 1, n // 2), 1 - np.exp2(-rng.uniform(0, 25, n // 8)),
        -1 + np.exp2(-rng.uniform(0, 25, n // 8)), rng.standard_normal(n // 8) * 1e-3,
        rng.standard_normal(n // 8) * np.exp2(rng.uniform(-120, -10, n // 8)),
        [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.nextafter(np.float32(1), np.float32(0)),
         np.float32(1e-45), np.float32(-1e-45), 1.0000001, -1.5, np.nan, np.inf, -np.inf]]).astype(np.float32)
    got, ok = probe(f360, gpu_ctx, 0, a)
    with np.errstate(invalid="ignore"):
        want = np.arcsin(a.astype(np.float64)).astype(np.float32)
        legal = np.abs(a) <= 1
    assert not ok[~legal].any()                      # |a| > 1, NaN, inf: never vouched for
    assert same_bits(got[ok], want[ok])              # sign of zero included
    assert ok[legal].mean() > 0.9998, ok[legal].mean()
    assert ok[-14:-6].all()                          # the exact special arguments are fast ones


def test_fast_atan2_is_correctly_rounded_where_it_vouches(f360, gpu_ctx):
    rng = np.random.default_rng(12)
    n = 1 << 24
    y = rng.standard_normal(n) * np.exp2(rng.integers(-30, 30, n))
    x = rng.standard_normal(n) * np.exp2(rng.integers(-30, 30, n))
    # near the octant boundaries and the axes
    m = n // 8
    ang = np.concatenate([rng.uniform(-np.pi, np.pi, m),
                          np.repeat(np.arange(-8, 9) * np.pi / 8, m // 17 + 1)[:m] + rng.standard_normal(m) * 1e-6])
    rad = np.exp2(rng.uniform(-20, 20, 2 * m))
    y = np.concatenate([y, rad * np.sin(ang), [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 1.0, 0.0, -0.0, np.inf, 1.0, np.nan]])
    x = np.concatenate([x, rad * np.cos(ang), [1.0, 1.0, -1.0, -1.0, 0.0, 0.0, -0.0, 0.0, -0.0, 1.0, np.inf, 1.0]])
    y, x = y.astype(np.float32), x.astype(np.float32)
    got, ok = probe(f360, gpu_ctx, 1, y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert same_bits(got[ok], want[ok])
    assert not ok[-6:].any()                         # x = -0 with y != 0, both zero, inf, NaN
    assert ok[-12:-6].all()
    assert ok[:-12].mean() > 0.9998, ok[:-12].mean()


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("table", [0, 1, 2])
def test_gnomonic_is_bit_exact_for_every_table_and_path(f360, gpu_ctx, oracle, fast, table):
    gpu_ctx.set_option("gnomonic.fast", fast)
    gpu_ctx.set_option("gnomonic.table", table)
    gpu_ctx.set_option("gnomonic.guard", 0)   # the exact chain on every pixel
    try:
        for (w, h, tw, th) in [(256, 128, 96, 64), (1920, 1080, 960, 540), (640, 320, 333, 117)]:
            frame = oracle.lcg_frame(w, h, 808).reshape(h, w, 4)
            proj = f360.Projections(gpu_ctx)
            src, dst = gpu_ctx.upload(frame), gpu_ctx.malloc(tw * th * 4)
            for (cx, cy) in GAZES:
                want = oracle.gnomonic(frame, tw, th, w, h, cx, cy)
                dst.fill(0x77)
                proj.GnomonicProjection(dst.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
                got = dst.copy_to_host(np.uint8, (th, tw, 4))
                assert int((got != want).any(axis=2).sum()) == 0, (w, h, tw, th, cx, cy)
            src.free()
            dst.free()
    finally:
        gpu_ctx.set_option("gnomonic.fast", 0)
        gpu_ctx.set_option("gnomonic.table", 1)
        gpu_ctx.set_option("gnomonic.guard", 1)
