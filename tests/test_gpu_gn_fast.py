"""The index-guarded gnomonic remap ("gnomonic.guard", csrc/gn_fast_math.h, projections.hip).

Two things make it exact, and both are tested on the device: the float asin / atan2 cores stay
inside the absolute error bounds the guard is built from (sweeps against double precision: every
float in [-1, 1]; 2^30 argument pairs), and with those bounds the accepted pixels carry the texel
of the exact chain -- the remap equals the oracle and the all-exact kernel for every gaze tried,
while only one or two pixels in a hundred take the exact chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fast_cores_stay_inside_half_their_bounds(f360, gpu_ctx):
    worst, bound = gpu_ctx.debug_gn_fast_sweep(0)
    assert 0 < worst <= 0.5 * bound, (worst, bound)
    worst, bound = gpu_ctx.debug_gn_fast_sweep(1, 1 << 30)
    assert 0 < worst <= 0.5 * bound, (worst, bound)


@pytest.mark.parametrize("w,h,tw,th", [(256, 128, 96, 64), (1920, 1080, 960, 540),
                                       (640, 320, 333, 117), (64, 32, 32, 32)])
def test_guarded_remap_matches_oracle(f360, gpu_ctx, oracle, w, h, tw, th):
    frame = oracle.lcg_frame(w, h, 811).reshape(h, w, 4)
    proj = f360.Projections(gpu_ctx)
    src, dst = gpu_ctx.upload(frame), gpu_ctx.malloc(tw * th * 4)
    rng = np.random.default_rng(w + tw)
    gazes = [(0.0, 0.0), (0.5, 0.5), (0.65, 0.75), (0.0, 1.0), (1.0, 1.0), (0.999, 0.5),
             (0.5, 0.0), (0.5, 1.0), (0.25, 0.5), (1.0, 0.5)]
    gazes += [(float(rng.uniform(0, 1)), float(rng.uniform(0, 1))) for _ in range(6)]
    gazes += [(-0.2, 1.3), (1.4, -0.3)]          # outside the frame: the launch is the exact one
    for (cx, cy) in gazes:
        want = oracle.gnomonic(frame, tw, th, w, h, cx, cy)
        dst.fill(0x77)
        proj.GnomonicProjection(dst.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
        got = dst.copy_to_host(np.uint8, (th, tw, 4))
        assert int((got != want).any(axis=2).sum()) == 0, (cx, cy)
    src.free()
    dst.free()


@pytest.mark.parametrize("w,h,tw,th", [(7680, 3840, 3840, 1920), (3840, 1920, 1920, 1080),
                                       (4096, 2048, 1001, 777)])
def test_guarded_remap_equals_exact_kernel_at_size(f360, gpu_ctx, w, h, tw, th):
    """Full-size sources (the guards scale with the source size): byte-identical to the kernel
    that runs the exact chain on every pixel, for a sweep of gazes including the poles and the
    seam; the worklist stays a small, non-empty fraction."""
    rng = np.random.default_rng(7)
    frame = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
    proj = f360.Projections(gpu_ctx)
    src = gpu_ctx.upload(frame)
    a, b = gpu_ctx.malloc(tw * th * 4), gpu_ctx.malloc(tw * th * 4)
    gazes = [(0.5, 0.5), (0.0, 0.0), (1.0, 1.0), (0.5, 0.0), (0.5, 1.0), (0.0, 0.5), (0.999, 0.5),
             (0.37, 0.61), (0.81, 0.13), (0.125, 0.875)]
    fractions = []
    try:
        gpu_ctx.set_option("debug.ablate", 512)   # count the pixels that take the exact chain
        for (cx, cy) in gazes:
            gpu_ctx.set_option("gnomonic.guard", 1)
            a.fill(0x11)
            proj.GnomonicProjection(a.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            fractions.append(gpu_ctx.debug_gnomonic_worklist() / (tw * th))
            gpu_ctx.set_option("gnomonic.guard", 0)
            b.fill(0x22)
            proj.GnomonicProjection(b.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            ga, gb = a.copy_to_host(np.uint8, (th, 4 * tw)), b.copy_to_host(np.uint8, (th, 4 * tw))
            assert np.array_equal(ga, gb), (cx, cy, int((ga != gb).sum()))
    finally:
        gpu_ctx.set_option("gnomonic.guard", 1)
        gpu_ctx.set_option("debug.ablate", 0)
    assert 0 < max(fractions) < 0.08, fractions
    for buf in (src, a, b):
        buf.free()


def test_guarded_remap_repeated_calls_and_geometry_changes(f360, gpu_ctx, oracle):
    """The worklist counters alternate between calls and the tables follow the target geometry."""
    w, h = 640, 320
    frame = oracle.lcg_frame(w, h, 5).reshape(h, w, 4)
    proj = f360.Projections(gpu_ctx)
    src = gpu_ctx.upload(frame)
    for (tw, th) in [(200, 100), (333, 117), (200, 100), (64, 64)]:
        dst = gpu_ctx.malloc(tw * th * 4)
        for k in range(5):
            cx, cy = 0.1 + 0.17 * k, 0.9 - 0.19 * k
            proj.GnomonicProjection(dst.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            got = dst.copy_to_host(np.uint8, (th, tw, 4))
            want = oracle.gnomonic(frame, tw, th, w, h, cx, cy)
            assert int((got != want).any(axis=2).sum()) == 0, (tw, th, k)
        dst.free()
    src.free()


def test_guarded_remap_along_the_seam(f360, gpu_ctx):
    """A view centred on a pole with the gaze a quarter turn from the seam: the centre row of the
    viewport (y == 0 for an even height) runs exactly along the source's left / right edge, where
    atan2 sits at +-pi/2 and su at the fmod wrap, beyond the clamp at 0.999.  (Found by
    scripts/gn_guard_soak.py: a clamped index must not be accepted within a guard of the wrap.)"""
    w, h = 7680, 3840
    rng = np.random.default_rng(3)
    src = gpu_ctx.upload(rng.integers(0, 256, (h, 4 * w), dtype=np.uint8))
    proj = f360.Projections(gpu_ctx)
    try:
        for (tw, th) in [(746, 526), (1024, 512), (333, 118)]:
            a, b = gpu_ctx.malloc(tw * th * 4), gpu_ctx.malloc(tw * th * 4)
            for cx in (0.75, 0.25, 0.0, 0.5, 1.0):
                for cy in (0.0, 1.0):
                    gpu_ctx.set_option("gnomonic.guard", 1)
                    proj.GnomonicProjection(a.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
                    gpu_ctx.set_option("gnomonic.guard", 0)
                    proj.GnomonicProjection(b.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
                    ga = a.copy_to_host(np.uint8, (th, 4 * tw))
                    gb = b.copy_to_host(np.uint8, (th, 4 * tw))
                    assert np.array_equal(ga, gb), (tw, th, cx, cy, int((ga != gb).sum()))
            a.free()
            b.free()
    finally:
        gpu_ctx.set_option("gnomonic.guard", 1)
    src.free()
