"""GPU tier: seeded random differential testing of the C-ABI path against the oracle -- random
frame sizes (down to 2x2), pixel formats, row paddings, reduced sizes, gaze points (inside and
outside [0,1]) and engine options.  Catches the edge cases fixed-size tests miss."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


SAMPLER_VARIANTS = (0, 1, 2)
DEFAULT_SAMPLER = 2
# F360_FUZZ_SCALE=10 runs ten times as many cases (a longer soak outside the regular tier)
SCALE = max(1, int(os.environ.get("F360_FUZZ_SCALE", "1")))


def reduced(n):
    return 16 * math.ceil(n / 1.8 / 16)


def random_geometry(rng):
    kind = rng.integers(0, 4)
    if kind == 0:   # tiny
        w, h = int(rng.integers(2, 24)), int(rng.integers(2, 24))
    elif kind == 1:  # thin / tall
        w, h = int(rng.integers(2, 40)), int(rng.integers(100, 700))
    elif kind == 2:  # wide
        w, h = int(rng.integers(300, 2100)), int(rng.integers(2, 60))
    else:
        w, h = int(rng.integers(40, 900)), int(rng.integers(40, 500))
    return w, h


def random_gaze(rng):
    if rng.random() < 0.25:
        return float(np.float32(rng.uniform(-0.6, 1.6))), float(np.float32(rng.uniform(-0.6, 1.6)))
    return float(np.float32(rng.uniform(0, 1))), float(np.float32(rng.uniform(0, 1)))


def test_random_sat_encode(f360, gpu_ctx, oracle):
    rng = np.random.default_rng(20240601)
    enc = f360.SATEncoder(gpu_ctx)
    old = {k: gpu_ctx.get_option(k) for k in ("sat.band_rows", "sat.sb_bands")}
    try:
        for case in range(60 * SCALE):
            w, h = random_geometry(rng)
            bpp = int(rng.choice([3, 4, 4, 4, 5]))
            pad = int(rng.choice([0, 0, 1, 4, 16])) if bpp * w > 8 else 0
            ls = w * bpp + pad
            if ls // w != bpp:   # keep the reference's bytes-per-pixel = linesize / width
                ls = w * bpp
            gpu_ctx.set_option("sat.band_rows", int(rng.choice([0, 8, 16, 32, 64])))
            gpu_ctx.set_option("sat.sb_bands", int(rng.choice([0, 1, 2, 3, 8])))
            frame = rng.integers(0, 256, (h, ls), dtype=np.uint8)
            if case % 7 == 0:
                frame[:] = 255
            src = gpu_ctx.upload(frame)
            sat = gpu_ctx.malloc(w * h * 12)
            sat.fill(0xEE)
            enc.EncodeFrameGPU(sat.ptr, src.ptr, w, h, ls)
            got = sat.copy_to_host(np.uint32, (h, w, 3))
            assert np.array_equal(got, oracle.sat_encode(frame, w, h, ls)), (case, w, h, bpp, ls)
            src.free()
            sat.free()
    finally:
        for k, v in old.items():
            gpu_ctx.set_option(k, v)


def test_random_sample_interpolate_fused(f360, gpu_ctx, oracle):
    rng = np.random.default_rng(77)
    for case in range(40 * SCALE):
        w, h = random_geometry(rng)
        w, h = max(w, 4), max(h, 4)
        if rng.random() < 0.5:
            rw, rh = reduced(w), reduced(h)
        else:
            rw, rh = int(rng.integers(2, 2 * w + 8)), int(rng.integers(2, 2 * h + 8))
        frame = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
        sat_h = oracle.sat_encode(frame, w, h, 4 * w)
        grid = oracle.satdec_grid(rw, rh, w, h)
        dec = f360.SATDecoder(gpu_ctx)
        dec.InitializeGrid(rw, rh, w, h)
        assert np.array_equal(dec.export_grid(rw, rh), grid), (case, w, h, rw, rh)
        sat, src = gpu_ctx.upload(sat_h), gpu_ctx.upload(frame)
        pad = int(rng.choice([0, 4, 12]))
        ls = 4 * rw + pad
        dst = gpu_ctx.malloc(rh * ls)
        for _ in range(3):
            cx, cy = random_gaze(rng)
            want = np.full((rh, ls), 0xA5, dtype=np.uint8)
            oracle.satdec_sample_rect(want, rw, rh, ls, sat_h, w, h, grid, cx, cy)
            for variant in SAMPLER_VARIANTS:
                gpu_ctx.set_option("sample.variant", variant)
                dst.fill(0xA5)
                dec.SampleFrameRectGPU(dst.ptr, rw, rh, ls, sat.ptr, (w, h), cx, cy)
                assert np.array_equal(dst.copy_to_host(np.uint8, (rh, ls)), want), \
                    (case, variant, w, h, rw, rh, cx, cy)
            gpu_ctx.set_option("sample.variant", DEFAULT_SAMPLER)
            dst.fill(0xA5)
            dec.FoveateFrameRectGPU(dst.ptr, rw, rh, ls, src.ptr, w, h, 4 * w, cx, cy)
            assert np.array_equal(dst.copy_to_host(np.uint8, (rh, ls)), want), (case, "fused", w, h, rw, rh)
            # un-warp a random reduced frame back to w x h
            red = rng.integers(0, 256, (rh, rw, 4), dtype=np.uint8)
            rsrc, full = gpu_ctx.upload(red), gpu_ctx.malloc(w * h * 4)
            dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, rsrc.ptr, rw, rh, 4 * rw, cx, cy)
            assert np.array_equal(full.copy_to_host(np.uint8, (h, w, 4)),
                                  oracle.satdec_interpolate_rect(red, w, h, rw, rh, cx, cy)), \
                (case, "interp", w, h, rw, rh, cx, cy)
            rsrc.free()
            full.free()
        for b in (sat, src, dst):
            b.free()
        dec.close()


def test_random_image_sampler_and_gnomonic(f360, gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    proj = f360.Projections(gpu_ctx)
    bad_lp = bad_gn = total_lp = total_gn = 0
    for case in range(25 * SCALE):
        w, h = random_geometry(rng)
        w, h = max(w, 4), max(h, 4)
        rw, rh = int(rng.integers(2, w + 40)), int(rng.integers(2, h + 40))
        bpp = int(rng.choice([3, 4]))
        frame = rng.integers(0, 256, (h, bpp * w), dtype=np.uint8)
        smp = f360.ImageSampler(gpu_ctx)
        smp.InitializeGrid(rw, rh, w, h)
        smp.InitializeLogpolarGrid(rw, rh, w, h)
        isg, lpg = oracle.is_grid(rw, rh, w, h), oracle.is_logpolar_grid(rw, rh, w, h)
        assert np.array_equal(smp.export_grid(rw, rh), isg)
        assert np.array_equal(smp.export_logpolar_grid(rw, rh), lpg)
        src, dst = gpu_ctx.upload(frame), gpu_ctx.malloc(rh * rw * 4)
        cx, cy = random_gaze(rng)
        for name, fn, ofn, g in (("rect", smp.SampleFrameRectGPU, oracle.is_sample_rect, isg),
                                 ("logpolar", smp.SampleFrameLogPolarGPU, oracle.is_sample_logpolar, lpg)):
            want = np.full((rh, 4 * rw), 0x3C, dtype=np.uint8)
            ofn(want, rw, rh, 4 * rw, frame, w, h, bpp * w, g, cx, cy)
            dst.fill(0x3C)
            fn(dst.ptr, rw, rh, 4 * rw, src.ptr, w, h, bpp * w, cx, cy)
            assert np.array_equal(dst.copy_to_host(np.uint8, (rh, 4 * rw)), want), (case, name, w, h, rw, rh)
        # float-transcendental kernels: count differing pixels (index flips), bar 1e-4
        red = rng.integers(0, 256, (rh, rw, 4), dtype=np.uint8)
        rsrc, full = gpu_ctx.upload(red), gpu_ctx.malloc(w * h * 4)
        smp.InterpolateFrameLogPolarGPU(full.ptr, w, h, 4 * w, rsrc.ptr, rw, rh, 4 * rw, cx, cy)
        got = full.copy_to_host(np.uint8, (h, w, 4))
        want = oracle.is_interpolate_logpolar(red, w, h, rw, rh, cx, cy)
        bad_lp += int((got != want).any(axis=2).sum())
        total_lp += w * h
        if bpp == 4:
            tw, th = int(rng.integers(1, 200)), int(rng.integers(1, 120))
            view = gpu_ctx.malloc(tw * th * 4)
            proj.GnomonicProjection(view.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            got = view.copy_to_host(np.uint8, (th, tw, 4))
            want = oracle.gnomonic(frame.reshape(h, w, 4), tw, th, w, h, cx, cy)
            bad_gn += int((got != want).any(axis=2).sum())
            total_gn += tw * th
            view.free()
        for b in (src, dst, rsrc, full):
            b.free()
        smp.close()
    assert bad_lp <= max(2, total_lp // 10000), (bad_lp, total_lp)
    assert bad_gn <= max(2, total_gn // 10000), (bad_gn, total_gn)


def test_random_planar_sources(f360, gpu_ctx, oracle):
    """Planar YUV 4:2:0: random sizes (width % 4, even height), plane paddings, both libswscale
    models and tilings -- converter, table and fused foveation against the oracle."""
    rng = np.random.default_rng(4242)
    enc = f360.SATEncoder(gpu_ctx)
    old = {k: gpu_ctx.get_option(k) for k in ("sat.band_rows", "sat.sb_bands", "yuv.model")}
    try:
        for case in range(30 * SCALE):
            w, h = random_geometry(rng)
            w, h = max(4, w // 4 * 4), max(2, h // 2 * 2)
            model = int(rng.integers(0, 2))
            gpu_ctx.set_option("yuv.model", model)
            gpu_ctx.set_option("sat.band_rows", int(rng.choice([0, 8, 16, 32, 64])))
            gpu_ctx.set_option("sat.sb_bands", int(rng.choice([-1, 0, 1, 2, 3])))
            ypad, cpad = int(rng.choice([0, 4, 8])), int(rng.choice([0, 2, 6]))
            y = rng.integers(0, 256, (h, w + ypad), dtype=np.uint8)
            u = rng.integers(0, 256, (h // 2, w // 2 + cpad), dtype=np.uint8)
            v = rng.integers(0, 256, (h // 2, w // 2 + cpad), dtype=np.uint8)
            if case % 5 == 0:
                y[:] = 255
            rgb0 = oracle.yuv420p_to_rgb0(y, u, v, w, h, model)
            want_sat = oracle.sat_encode(rgb0, w, h, 4 * w)
            dy, du, dv = gpu_ctx.upload(y), gpu_ctx.upload(u), gpu_ctx.upload(v)
            rgb = gpu_ctx.malloc(w * h * 4)
            sat = gpu_ctx.malloc(w * h * 12)
            sat.fill(0xEE)
            gpu_ctx.yuv420p_to_rgb0(rgb.ptr, 4 * w, dy.ptr, du.ptr, dv.ptr, y.shape[1], u.shape[1],
                                    v.shape[1], w, h)
            enc.EncodeFrameYUV420PGPU(sat.ptr, dy.ptr, du.ptr, dv.ptr, y.shape[1], u.shape[1],
                                      v.shape[1], w, h)
            assert np.array_equal(rgb.copy_to_host(np.uint8, (h, 4 * w)), rgb0), (case, w, h, model)
            assert np.array_equal(sat.copy_to_host(np.uint32, (h, w, 3)), want_sat), (case, w, h, model)
            if w >= 8 and h >= 8:
                rw, rh = reduced(w), reduced(h)
                dec = f360.SATDecoder(gpu_ctx)
                dec.InitializeGrid(rw, rh, w, h)
                grid = oracle.satdec_grid(rw, rh, w, h)
                cx, cy = random_gaze(rng)
                want = np.full((rh, 4 * rw), 0x5A, np.uint8)
                oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, cx, cy)
                red = gpu_ctx.malloc(rh * 4 * rw)
                red.fill(0x5A)
                dec.FoveateFrameRectYUV420PGPU(red.ptr, rw, rh, 4 * rw, dy.ptr, du.ptr, dv.ptr,
                                               y.shape[1], u.shape[1], v.shape[1], w, h, cx, cy)
                assert np.array_equal(red.copy_to_host(np.uint8, (rh, 4 * rw)), want), (case, w, h)
                red.free()
                dec.close()
            for b in (dy, du, dv, rgb, sat):
                b.free()
    finally:
        for k, v_ in old.items():
            gpu_ctx.set_option(k, v_)


def test_random_frames_in_shared_launches(f360, gpu_ctx, oracle):
    """EncodeFramesGPU + SampleFramesRectGPU on random geometries, batch sizes (1 .. 20: more
    than a launch holds), pixel formats and gaze points; every frame against the oracle."""
    rng = np.random.default_rng(4242)
    enc = f360.SATEncoder(gpu_ctx)
    for case in range(24 * SCALE):
        w, h = random_geometry(rng)
        w, h = max(w, 4), max(h, 4)
        if case % 3 == 0:
            w = (w + 3) // 4 * 4 + 4      # the vector / tile-streamer path
        n = int(rng.integers(1, 21))
        bpp = 4 if case % 4 else 3
        ls = w * bpp
        rw, rh = reduced(w), reduced(h)
        gpu_ctx.set_option("sample.variant", int(rng.choice([1, 2, 2])))
        # the read-once strip walker whenever the layout allows it (every other case), with
        # random depth and frames per launch; else the automatic choice (three kernels here)
        gpu_ctx.set_option("sat.walk", 1 if case % 2 else -1)
        gpu_ctx.set_option("sat.walk_frames", int(rng.choice([0, 1, 3, 7, 64])))
        frames = [rng.integers(0, 256, (h, ls), dtype=np.uint8) for _ in range(n)]
        gazes = [random_gaze(rng) for _ in range(n)]
        srcs = [gpu_ctx.upload(f) for f in frames]
        sats = [gpu_ctx.malloc(w * h * 12) for _ in range(n)]
        reds = [gpu_ctx.malloc(rw * rh * 4) for _ in range(n)]
        for r in reds:
            r.fill(0x5A)
        dec = f360.SATDecoder(gpu_ctx)
        dec.InitializeGrid(rw, rh, w, h)
        enc.EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, ls)
        dec.SampleFramesRectGPU([r.ptr for r in reds], rw, rh, 4 * rw, [s.ptr for s in sats],
                                (w, h), gazes)
        grid = oracle.satdec_grid(rw, rh, w, h)
        for k in range(n):
            sat_h = oracle.sat_encode(frames[k], w, h, ls)
            assert np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), sat_h), (case, k, w, h)
            want = np.full((rh, 4 * rw), 0x5A, dtype=np.uint8)
            oracle.satdec_sample_rect(want, rw, rh, 4 * rw, sat_h, w, h, grid, *gazes[k])
            assert np.array_equal(reds[k].copy_to_host(np.uint8, (rh, 4 * rw)), want), (case, k, w, h, gazes[k])
        for b in srcs + sats + reds:
            b.free()
        dec.close()
    gpu_ctx.set_option("sample.variant", DEFAULT_SAMPLER)
    gpu_ctx.set_option("sat.walk", -1)
    gpu_ctx.set_option("sat.walk_frames", 0)


def test_random_output_colour_step(f360, gpu_ctx, oracle):
    """f360_rgb0_to_yuv420p on random even sizes (8 rows up), row paddings of the source and of
    each plane, unaligned bases, both libswscale models; bytes outside the planes untouched."""
    rng = np.random.default_rng(99)
    for case in range(30 * SCALE):
        w = 2 * int(rng.integers(1, 700)) if case % 3 else 8 * int(rng.integers(1, 300))
        h = 2 * int(rng.integers(4, 150))
        model = int(rng.integers(0, 2))
        spad = int(rng.choice([0, 0, 4, 16, 20]))
        pads = tuple(int(rng.choice([0, 0, 1, 4, 8, 12])) for _ in range(3))
        off = int(rng.choice([0, 0, 4, 16]))
        src_h = rng.integers(0, 256, (h, 4 * w + spad), dtype=np.uint8)
        if case % 5 == 0:
            src_h[:] = rng.choice([0, 255])
        want = oracle.rgb0_to_yuv420p(src_h, w, h, model, pads=pads)
        gpu_ctx.set_option("yuv.model", model)
        raw = np.zeros(src_h.size + 64, dtype=np.uint8)
        raw[off:off + src_h.size] = src_h.reshape(-1)
        src = gpu_ctx.upload(raw)
        planes = [gpu_ctx.malloc(p.size) for p in want]
        for p in planes:
            p.fill(0xEE)
        gpu_ctx.rgb0_to_yuv420p(planes[0].ptr, planes[1].ptr, planes[2].ptr, want[0].shape[1],
                                want[1].shape[1], want[2].shape[1], src.ptr + off, src_h.shape[1],
                                w, h)
        for name, buf, ref in zip("yuv", planes, want):
            assert np.array_equal(buf.copy_to_host(np.uint8, ref.shape), ref), (case, name, w, h, model, spad, pads, off)
            buf.free()
        src.free()
    gpu_ctx.set_option("yuv.model", 1)
