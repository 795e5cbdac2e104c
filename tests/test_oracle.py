"""CPU tier: the oracle against its golden vectors and against independent numpy restatements
of the integer parts; size-independent properties; float-model sensitivity of the tables."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def reduced(n):
    import math
    return 16 * math.ceil(n / 1.8 / 16)


def np_sat(frame, w, h, bpp):
    px = frame[:, : w * bpp].reshape(h, w, bpp)[:, :, :3].astype(np.uint64)
    return (px.cumsum(0).cumsum(1) & 0xFFFFFFFF).astype(np.uint32)


@pytest.fixture(scope="module")
def golden_small():
    return np.load(os.path.join(GOLD, "small.npz"))


@pytest.fixture(scope="module")
def golden_digests():
    with open(os.path.join(GOLD, "digests.json")) as f:
        return json.load(f)


def test_golden_small_reproduces(oracle, golden_small):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    now = mg.case_outputs(64, 32)
    assert set(now.keys()) == set(golden_small.files)
    for k in golden_small.files:
        assert np.array_equal(now[k], golden_small[k]), k


def test_golden_digests_256(oracle, golden_digests):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert mg.digests(mg.case_outputs(256, 128)) == golden_digests["cases"]["256x128"]


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 1920), (7680, 3840)])
def test_golden_digests_full_sizes(oracle, golden_digests, w, h):
    ent = golden_digests["cases"][f"{w}x{h}"]
    frame = oracle.lcg_frame(w, h, golden_digests["seed"])
    assert f"{oracle.fnv1a64(frame):016x}" == ent["frame"]
    sat = oracle.sat_encode(frame, w, h, 4 * w)
    assert f"{oracle.fnv1a64(sat):016x}" == ent["sat"]
    rw, rh = reduced(w), reduced(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    cx, cy = golden_digests["gazes"][2]
    red = np.full((rh, rw * 4), 0xA5, dtype=np.uint8)
    oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, cx, cy)
    assert f"{oracle.fnv1a64(red):016x}" == ent["sample_rect_2"]


@pytest.mark.parametrize("w,h,bpp,pad", [(64, 32, 4, 0), (37, 19, 4, 0), (40, 8, 3, 0),
                                         (130, 70, 4, 24), (1, 1, 4, 0), (5, 300, 3, 1)])
def test_sat_encode_equals_cumsum(oracle, w, h, bpp, pad):
    ls = w * bpp + pad
    # keep linesize / width == bpp as the reference computes it
    assert ls // w == bpp
    frame = oracle.lcg_frame(w, h, 7, bpp=bpp, linesize=ls)
    assert np.array_equal(oracle.sat_encode(frame, w, h, ls), np_sat(frame, w, h, bpp))


def test_sat_wraps_mod_2_32_at_8k(oracle):
    w, h = 7680, 3840
    white = np.full((h, 4 * w), 255, dtype=np.uint8)
    sat = oracle.sat_encode(white, w, h, 4 * w)
    assert int(sat[-1, -1, 0]) == (255 * w * h) % (1 << 32)
    assert 255 * w * h > (1 << 32)
    # box differences stay exact: every written reduced pixel is 255
    rw, rh = reduced(w), reduced(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    red = np.zeros((rh, rw * 4), dtype=np.uint8)
    oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, 0.5, 0.5)
    px = red.reshape(rh, rw, 4)
    # the reduced buffer spans +-W x +-H around the gaze, so rows whose box lies outside the
    # frame stay untouched (0); every written pixel must be exactly 255
    written = (px[:, :, :3] != 0).any(axis=2)
    assert (px[:, :, :3][written] == 255).all() and (px[:, :, 3] == 0).all()
    assert written[rh // 4: 3 * rh // 4].all() and 0.5 < written.mean() < 1.0


def test_decode_inverts_encode(oracle):
    w, h = 96, 40
    frame = oracle.lcg_frame(w, h, 99)
    sat = oracle.sat_encode(frame, w, h, 4 * w)
    dec = np.full((h, 4 * w), 0x11, dtype=np.uint8)
    oracle.satdec_decode(dec, 4 * w, sat, w, h)
    a, b = dec.reshape(h, w, 4), frame.reshape(h, w, 4)
    assert np.array_equal(a[:, :, :3], b[:, :, :3])
    assert (a[:, :, 3] == 0x11).all()  # pad byte untouched


def test_sampler_is_a_box_mean(oracle):
    """Independent numpy restatement of one sampled pixel from the raw image."""
    w, h = 256, 128
    rw, rh = reduced(w), reduced(h)
    frame = oracle.lcg_frame(w, h, 5)
    img = frame.reshape(h, w, 4)[:, :, :3].astype(np.int64)
    sat = oracle.sat_encode(frame, w, h, 4 * w)
    grid = oracle.satdec_grid(rw, rh, w, h)
    cx, cy = 0.5, 0.5
    red = np.zeros((rh, rw * 4), dtype=np.uint8)
    oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, cx, cy)
    red = red.reshape(rh, rw, 4)
    cxp, cyp = int(np.float32(cx) * np.float32(w)), int(np.float32(cy) * np.float32(h))
    rng = np.random.default_rng(0)
    for _ in range(200):
        i, j = int(rng.integers(1, rw - 1)), int(rng.integers(1, rh - 1))
        px, mx = cxp + grid[j + 1, i + 1, 0], cxp + grid[j + 1, i, 0]
        py, my = cyp + grid[j + 1, i + 1, 1], cyp + grid[j, i + 1, 1]
        if not (1 <= px < w and 0 <= mx < px and 1 <= py < h and 0 <= my < py):
            continue
        box = img[my + 1: py + 1, mx + 1: px + 1]
        want = box.sum(axis=(0, 1)) // (box.shape[0] * box.shape[1])
        assert np.array_equal(red[j, i, :3], want.astype(np.uint8)), (i, j)


def test_sampler_leaves_unwritten_bytes(oracle):
    w, h = 128, 64
    rw, rh = reduced(w), reduced(h)
    frame = oracle.lcg_frame(w, h, 3)
    sat = oracle.sat_encode(frame, w, h, 4 * w)
    grid = oracle.satdec_grid(rw, rh, w, h)
    red = np.full((rh, rw * 4 + 16), 0xA5, dtype=np.uint8)  # padded target rows
    oracle.satdec_sample_rect(red, rw, rh, rw * 4 + 16, sat, w, h, grid, 0.0, 1.0)
    assert (red[:, rw * 4:] == 0xA5).all()
    assert (red[:, 3: rw * 4: 4] == 0xA5).all()
    # gaze in a corner: some pixels fall outside the frame and stay untouched
    untouched = (red[:, : rw * 4].reshape(rh, rw, 4)[:, :, :3] == 0xA5).all(axis=2)
    assert untouched.any() and not untouched.all()


def test_interpolate_constant_image(oracle):
    w, h = 256, 128
    rw, rh = reduced(w), reduced(h)
    src = np.zeros((rh, rw, 4), dtype=np.uint8)
    src[:, :, 0], src[:, :, 1], src[:, :, 2], src[:, :, 3] = 10, 200, 77, 9
    for (cx, cy) in [(0.5, 0.5), (0.0, 0.0), (1.0, 1.0), (0.65, 0.75)]:
        out = oracle.satdec_interpolate_rect(src, w, h, rw, rh, cx, cy)
        assert (out[:, :, 0] == 10).all() and (out[:, :, 1] == 200).all()
        assert (out[:, :, 2] == 77).all() and (out[:, :, 3] == 0).all()


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 1920), (7680, 3840), (64, 32), (256, 128)])
def test_tables_insensitive_to_float_model(oracle, w, h):
    """The geometry tables of every benchmark config are identical whether OpenCL float
    builtins are modelled as correctly rounded or as glibc's float routines."""
    rw, rh = reduced(w), reduced(h)
    try:
        oracle.set_float_model(0)
        a = oracle.satdec_grid_axes(rw, rh, w, h)
        b = oracle.is_grid(rw, rh, w, h)
        oracle.set_float_model(1)
        a1 = oracle.satdec_grid_axes(rw, rh, w, h)
        b1 = oracle.is_grid(rw, rh, w, h)
    finally:
        oracle.set_float_model(0)
    assert np.array_equal(a[0], a1[0]) and np.array_equal(a[1], a1[1])
    assert np.array_equal(b, b1)


def test_grid_is_separable_and_monotone(oracle):
    w, h = 1920, 1080
    rw, rh = reduced(w), reduced(h)
    assert (rw, rh) == (1072, 608)  # parameters.h:8-9
    g = oracle.satdec_grid(rw, rh, w, h)
    assert (g[:, :, 0] == g[0:1, :, 0]).all() and (g[:, :, 1] == g[:, 0:1, 1]).all()
    assert (np.diff(g[0, :, 0].astype(int)) >= 1).all()
    assert (np.diff(g[:, 0, 1].astype(int)) >= 1).all()


def test_host_tables_match_oracle(oracle, f360):
    """Product host logic (csrc/host_tables.cpp) against the oracle, no GPU needed."""
    for (w, h) in [(64, 32), (256, 128), (1920, 1080), (3840, 1920), (7680, 3840)]:
        rw, rh = reduced(w), reduced(h)
        gx, gy = oracle.satdec_grid_axes(rw, rh, w, h)
        assert np.array_equal(f360.tables_satdec_grid_axis(rw, w), gx)
        assert np.array_equal(f360.tables_satdec_grid_axis(rh, h), gy)
        if w <= 1920:
            isg = oracle.is_grid(rw, rh, w, h)
            assert np.array_equal(f360.tables_is_grid_axis(rw, w), isg[0, :, 0])
            assert np.array_equal(f360.tables_is_grid_axis(rh, h), isg[:, 0, 1])
            r, c, s = f360.tables_logpolar_axes(rw, rh)
            lp = oracle.is_logpolar_grid(rw, rh, w, h)
            gxp = (r[None, :] * c[:, None]).astype(np.int32).astype(np.int16)
            gyp = (r[None, :] * s[:, None]).astype(np.int32).astype(np.int16)
            assert np.array_equal(gxp, lp[:, :, 0]) and np.array_equal(gyp, lp[:, :, 1])


def test_interp_axis_table_properties(f360):
    w, rw = 1920, 1072
    t = f360.tables_interp_axis(w, w, rw)
    d = np.arange(-w, w + 1)
    u, dcalc, dmin, du = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    assert (du == -np.sign(d)).all()
    assert (np.sign(u) == np.sign(d)).all()
    assert (np.abs(u) <= np.abs(d)).all()
    assert t[w].tolist() == [0, 0, 0, 0]
    # the forward map of the inverse never falls short of the pixel it came from
    inner = np.abs(d) <= w // 2
    assert (np.abs(dcalc[inner]) >= np.abs(d[inner])).all()
    assert (np.abs(dmin[inner]) <= np.abs(dcalc[inner])).all()


def test_expand_rect_inverts_the_sampler_lattice(oracle):
    """ExpandSampledFrameRectCPU scatters reduced pixel (i, j) to centre + f(u), f(v): the fovea
    (unit steps) is copied verbatim, other target pixels keep their value."""
    w, h = 256, 128
    rw, rh = 144, 80
    red = oracle.lcg_frame(rw, rh, 5)
    dst = np.full((h, 4 * w), 0x42, np.uint8)
    oracle.expand("rect", dst, w, h, 4 * w, red, rw, rh, 4 * rw, 0.5, 0.5)
    img, src = dst.reshape(h, w, 4), red.reshape(rh, rw, 4)
    touched = (img[:, :, :3] != 0x42).any(axis=2)
    assert touched.sum() <= rw * rh and touched.sum() > rw * rh // 4
    assert (img[:, :, 3] == 0x42).all()  # 3 bytes per pixel only
    for d in (-3, 0, 2):  # fovea: offset d from the centre maps to offset d
        assert np.array_equal(img[h // 2 + d, w // 2 + d, :3], src[rh // 2 + d, rw // 2 + d, :3])


def test_expand_logpolar_last_writer_wins(oracle):
    w, h = 128, 64
    rw, rh = 80, 48
    red = oracle.lcg_frame(rw, rh, 6)
    dst = np.zeros((h, 4 * w), np.uint8)
    oracle.expand("logpolar", dst, w, h, 4 * w, red, rw, rh, 4 * rw, 0.5, 0.5)
    img, src = dst.reshape(h, w, 4), red.reshape(rh, rw, 4)
    # column 0 has radius exp(0) = 1: at angle 0 it lands on (cx + 1, cy); the last row j whose
    # angle still truncates to that pixel owns it, and later columns with radius < 2 overwrite it
    assert img[h // 2, w // 2 + 1, :3].any()
    assert (img[:, :, 3] == 0).all()
