"""Colour-space step in front of the path: the oracle's restatement of
libswscale's two yuv420p -> RGB converters, and the host constants of the HIP kernels.
CPU only; the GPU parity tests are in test_gpu_parity.py."""
import numpy as np
import pytest


def test_known_colours(oracle):
    """ITU-R 601 studio-range colour bars: both converters land within 3 of the ideal value
    (the C tables are 1.8 dark by construction, yuv2rgb.c:802,978)."""
    bars = {
        (16, 128, 128): (0, 0, 0),
        (235, 128, 128): (255, 255, 255),
        (81, 90, 240): (255, 0, 0),
        (145, 54, 34): (0, 255, 0),
        (41, 240, 110): (0, 0, 255),
        (210, 16, 146): (255, 255, 0),
    }
    for model in (oracle.YUV_SWS_C, oracle.YUV_SWS_X86):
        for yuv, want in bars.items():
            got = oracle.yuv_to_rgb_pixel(model, *yuv)
            assert max(abs(g - w) for g, w in zip(got, want)) <= 3, (model, yuv, got)


def test_grey_ramp_is_monotonic_and_neutral(oracle):
    for model in (oracle.YUV_SWS_C, oracle.YUV_SWS_X86):
        prev = -1
        for Y in range(256):
            r, g, b = oracle.yuv_to_rgb_pixel(model, Y, 128, 128)
            assert r == g == b
            assert r >= prev
            prev = r
        assert oracle.yuv_to_rgb_pixel(model, 0, 128, 128) == (0, 0, 0)
        assert oracle.yuv_to_rgb_pixel(model, 255, 128, 128) == (255, 255, 255)


def test_models_differ_by_rounding_only(oracle):
    """The two libswscale converters disagree by a few code values at most: whichever one the
    reference's FFmpeg build runs, the other is within this bound."""
    yy, uu, vv = np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 3), np.arange(0, 256, 3),
                             indexing="ij")
    n = yy.size
    # one chroma sample per pixel pair: both pixels of a pair carry the same luma
    y2 = np.zeros((2, 2 * n), np.uint8)
    y2[:, 0::2] = yy.ravel()
    y2[:, 1::2] = yy.ravel()
    u2 = uu.ravel()[None, :].astype(np.uint8)
    v2 = vv.ravel()[None, :].astype(np.uint8)
    a = oracle.yuv420p_to_rgb0(y2, u2, v2, 2 * n, 2, oracle.YUV_SWS_C).reshape(2, 2 * n, 4)
    b = oracle.yuv420p_to_rgb0(y2, u2, v2, 2 * n, 2, oracle.YUV_SWS_X86).reshape(2, 2 * n, 4)
    assert (a[..., 3] == 255).all() and (b[..., 3] == 255).all()
    d = np.abs(a[..., :3].astype(int) - b[..., :3].astype(int))
    assert d.max() <= 3, d.max()
    assert (a[0] == a[1]).all()  # both rows of a 2x2 block use the same chroma sample


def test_frame_layout(oracle):
    """2x2 chroma blocks, linesize padding untouched."""
    w, h = 6, 4
    rng = np.random.default_rng(5)
    y = rng.integers(0, 256, (h, 8), dtype=np.uint8)
    u = rng.integers(0, 256, (h // 2, 4), dtype=np.uint8)
    v = rng.integers(0, 256, (h // 2, 4), dtype=np.uint8)
    dst = np.full((h, 4 * w + 8), 0x5A, np.uint8)
    oracle.yuv420p_to_rgb0(y, u, v, w, h, oracle.YUV_SWS_X86, dst=dst, dst_linesize=4 * w + 8)
    assert (dst[:, 4 * w:] == 0x5A).all()
    for (py, px) in [(0, 0), (1, 1), (2, 5), (3, 4)]:
        want = oracle.yuv_to_rgb_pixel(oracle.YUV_SWS_X86, int(y[py, px]), int(u[py // 2, px // 2]),
                                       int(v[py // 2, px // 2]))
        assert tuple(dst[py, 4 * px:4 * px + 3]) == want
        assert dst[py, 4 * px + 3] == 255


def test_host_constants_reproduce_both_converters(f360, oracle):
    """The closed forms the HIP kernels evaluate (yuv_device.h), in numpy, against the oracle's
    literal table / 16-bit-lane restatement over a dense sample of the YUV cube."""
    k = f360.tables_yuv2rgb()
    Y, U, V = (a.astype(np.int64) for a in np.meshgrid(
        np.arange(256), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"))
    clip = lambda a: np.clip(a, 0, 255)
    # model 0
    base = k["c0"] + Y * k["cy"]
    r0 = clip((base + (k["r0"] + ((V * k["crv"]) >> 16)) * k["cy"]) >> 16)
    g0 = clip((base + (k["gu0"] + ((U * k["cgu"]) >> 16) + k["gv0"] +
                       ((V * k["cgv"]) >> 16)) * k["cy"]) >> 16)
    b0 = clip((base + (k["b0"] + ((U * k["cbu"]) >> 16)) * k["cy"]) >> 16)
    # model 1
    u, v = (U << 3) - 0x400, (V << 3) - 0x400
    yy = (((Y << 3) - k["yoff"]) * k["yc"]) >> 16
    r1 = clip(yy + ((v * k["vrc"]) >> 16))
    g1 = clip(yy + ((u * k["ugc"]) >> 16) + ((v * k["vgc"]) >> 16))
    b1 = clip(yy + ((u * k["ubc"]) >> 16))
    idx = np.random.default_rng(11).choice(Y.size, 4000, replace=False)
    for i in idx:
        yuv = (int(Y.flat[i]), int(U.flat[i]), int(V.flat[i]))
        assert oracle.yuv_to_rgb_pixel(oracle.YUV_SWS_C, *yuv) == \
            (int(r0.flat[i]), int(g0.flat[i]), int(b0.flat[i])), yuv
        assert oracle.yuv_to_rgb_pixel(oracle.YUV_SWS_X86, *yuv) == \
            (int(r1.flat[i]), int(g1.flat[i]), int(b1.flat[i])), yuv
    # 32-bit arithmetic on the device: every intermediate fits
    for a in (base, base + (k["r0"] + ((V * k["crv"]) >> 16)) * k["cy"], yy, u * k["ubc"]):
        assert np.abs(a).max() < 2 ** 31


# ------------------------------------------------------------------ output side: RGB0 -> yuv420p
def frame_of(rgb, w, h):
    fr = np.zeros((h, 4 * w), dtype=np.uint8)
    fr.reshape(h, w, 4)[:, :, :3] = rgb
    fr.reshape(h, w, 4)[:, :, 3] = 0x5A   # the pad byte must not matter
    return fr


def test_rgb2yuv_colour_bars_and_coefficients(oracle):
    """ITU-R 601 limited range: the values every video engineer knows, from both models."""
    assert oracle.rgb2yuv_coeffs() == [8414, 16519, 3208, -4865, -9528, 14392, 14392, -12061, -2332]
    bars = {(255, 255, 255): (235, 128, 128), (0, 0, 0): (16, 128, 128), (255, 0, 0): (81, 90, 240),
            (0, 255, 0): (145, 54, 34), (0, 0, 255): (41, 240, 110), (255, 255, 0): (210, 16, 146),
            (0, 255, 255): (170, 166, 16), (255, 0, 255): (106, 202, 222)}
    for model in (oracle.YUV_SWS_C, oracle.YUV_SWS_X86):
        for rgb, want in bars.items():
            y, u, v = oracle.rgb0_to_yuv420p(frame_of(rgb, 16, 8), 16, 8, model)
            assert (y == want[0]).all() and (u == want[1]).all() and (v == want[2]).all(), (model, rgb)


def test_rgb2yuv_vertical_chroma_filter_is_the_closed_form(oracle):
    """initFilter restated as it stands (bilinear, 2:1, chroma sited between the two rows) gives
    512 1536 1536 512 on rows 2c-1 .. 2c+2, folded at the frame borders -- the closed form the
    device kernel uses."""
    for h in list(range(8, 64, 2)) + [1080, 2144]:
        f, p = oracle.rgb2yuv_chroma_vfilter(h)
        ch = h // 2
        assert f.shape == (ch, 4)
        assert list(f[0]) == [2048, 1536, 512, 0] and p[0] == 0
        assert list(f[-1]) == [0, 512, 1536, 2048] and p[-1] == h - 4
        for c in range(1, ch - 1):
            assert list(f[c]) == [512, 1536, 1536, 512] and p[c] == 2 * c - 1, (h, c)


def test_rgb2yuv_luma_is_per_pixel_and_models_differ_by_at_most_one(oracle):
    rng = np.random.default_rng(5)
    w, h = 64, 24
    fr = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
    yc, uc, vc = oracle.rgb0_to_yuv420p(fr, w, h, oracle.YUV_SWS_C)
    yx, ux, vx = oracle.rgb0_to_yuv420p(fr, w, h, oracle.YUV_SWS_X86)
    px = fr.reshape(h, w, 4).astype(np.int64)
    y14 = (8414 * px[:, :, 0] + 16519 * px[:, :, 1] + 3208 * px[:, :, 2] + (32 << 14) + 256) >> 9
    assert np.array_equal(yc, ((2 * y14 + 64) >> 7).astype(np.uint8)) and np.array_equal(yc, yx)
    # chroma: the x86 scaler truncates each tap, never more than one code value below / above
    assert np.abs(uc.astype(int) - ux.astype(int)).max() <= 1
    assert np.abs(vc.astype(int) - vx.astype(int)).max() <= 1
    assert np.array_equal(uc[-1], ux[-1]) and np.array_equal(vc[-1], vx[-1])   # last row: C both
    # a vertically constant frame: chroma equals the one-row value whatever the taps
    fr2 = np.tile(fr[:1], (h, 1))
    _, u2, v2 = oracle.rgb0_to_yuv420p(fr2, w, h, oracle.YUV_SWS_C)
    assert (u2 == u2[0]).all() and (v2 == v2[0]).all()


def test_rgb2yuv_round_trip_through_the_input_side(oracle):
    """yuv420p -> RGB0 (input side) -> yuv420p (output side) returns to the same planes up to
    the rounding of two 8-bit conversions, on smooth content."""
    w, h = 64, 32
    yy, xx = np.mgrid[0:h, 0:w]
    y = (40 + xx * 2 + yy).astype(np.uint8)
    u = np.full((h // 2, w // 2), 118, dtype=np.uint8)
    v = np.full((h // 2, w // 2), 140, dtype=np.uint8)
    rgb = oracle.yuv420p_to_rgb0(y, u, v, w, h, oracle.YUV_SWS_C)
    y2, u2, v2 = oracle.rgb0_to_yuv420p(rgb, w, h, oracle.YUV_SWS_C)
    assert np.abs(y2.astype(int) - y.astype(int)).max() <= 2
    assert np.abs(u2.astype(int) - 118).max() <= 2 and np.abs(v2.astype(int) - 140).max() <= 2
