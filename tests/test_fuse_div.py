"""CPU tier: the float quotient of the one-pass encode + sample (sat_fuse_dev.h: walk_fuse_rows).
A reduced pixel is n / (w * h) with n <= 255 * w * h, truncated; for w * h <= 2048 the helper
wave computes it as trunc(fma(float(n), rcp(w) * rcp(h), 2^-12)).  The hardware reciprocal is good
to one ulp, so every combination of one-ulp errors is walked here, on the operands where a wrong
rounding would show: n = k * area (an exact quotient must not come out one lower) and
n = k * area - 1 (the largest fraction must not reach the next integer).  fma is emulated in
80-bit arithmetic (24 x 24-bit product plus 2^-12 is exact there), then rounded once."""
import numpy as np


def _ulp_step(x, steps):
    return (x.view(np.int32) + steps).view(np.float32)


def test_float_quotient_is_exact_for_boxes_up_to_2048():
    assert np.finfo(np.longdouble).nmant >= 63
    pairs = [(w, h) for w in range(1, 256) for h in range(1, 1024) if w * h <= 2048]
    w = np.array([p[0] for p in pairs], dtype=np.uint32)
    h = np.array([p[1] for p in pairs], dtype=np.uint32)
    area = (w * h).astype(np.uint64)
    k = np.arange(1, 256, dtype=np.uint64)
    worst = 0
    for dw in (-1, 0, 1):
        for dh in (-1, 0, 1):
            inv_w = _ulp_step(np.float32(1.0) / w.astype(np.float32), dw)
            inv_h = _ulp_step(np.float32(1.0) / h.astype(np.float32), dh)
            inv = (inv_w * inv_h).astype(np.float32)[:, None]
            for off in (0, -1):
                n = (area[:, None] * k[None, :]).astype(np.int64) + off
                want = n // area[:, None].astype(np.int64)
                nf = n.astype(np.float32)  # v_cvt_f32_u32: round to nearest even
                t = (nf.astype(np.longdouble) * inv.astype(np.longdouble) +
                     np.longdouble(2.0 ** -12)).astype(np.float32)
                got = np.floor(t).astype(np.int64)
                bad = int((got != want).sum())
                worst = max(worst, bad)
                assert bad == 0, (dw, dh, off, bad)
    assert worst == 0
