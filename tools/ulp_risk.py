#!/usr/bin/env python3
"""ulp_risk.py -- how much of the sampling tables hangs on the last bits of exp / pow / cos / sin.

The reference builds its grids inside OpenCL kernels (create_grid_kernel,
src/sat_decoder_sample_rect_kernel.cl:258-294, src/image_sampler_sample_rect_kernel.cl:73-80;
create_logpolar_grid_kernel, src/image_sampler_sample_logpolar_kernel.cl:5-39).  This repo -- the
product's host tables and the test oracle alike -- evaluates each float builtin as the correctly
rounded float of the double result.  OpenCL only bounds the builtins' error (exp <= 3 ulp,
pow <= 16 ulp, cos / sin <= 4 ulp in the full profile), so a given OpenCL device may return a
neighbouring float, and where the value then goes through a truncation the table entry can move
by one texel.

For every table entry of the benchmark geometries this script perturbs each builtin's result by
-K..+K ulp (all combinations) and counts the entries whose final integer can change.  That count
is the known parity risk of comparing against the restatement instead of the reference run on an
OpenCL device; DESIGN.md quotes the output of

    python tools/ulp_risk.py            # K = 4
"""
import argparse
import itertools
import json
import math

import numpy as np

f32 = np.float32


def reduced(n):
    return 16 * math.ceil(n / 1.8 / 16)


def nudge(x, k):
    """x (float32 array) moved by k ulp."""
    x = np.asarray(x, dtype=f32).copy()
    step = f32(np.inf) if k > 0 else f32(-np.inf)
    for _ in range(abs(k)):
        x = np.nextafter(x, step, dtype=f32)
    return x


def rect_axis(n_out, n_src, k_pow, k_exp, k_e1):
    """max(a, (int)(lambda * (exp(pow(2a/n, 4)) - 1))) for a = 0 .. n_out/2 + 1, float math
    (src/sat_decoder_sample_rect_kernel.cl:266-273), each builtin off by the given ulps."""
    a = np.arange(0, n_out // 2 + 2, dtype=np.int64)
    e1 = nudge(f32(np.exp(np.float64(f32(1.0)))), k_e1)
    lam = f32(n_src) / (e1 - f32(1.0))
    t = f32(2.0) * a.astype(f32) / f32(n_out)
    p = nudge(np.power(t.astype(np.float64), 4.0).astype(f32), k_pow)
    e = nudge(np.exp(p.astype(np.float64)).astype(f32), k_exp) - f32(1.0)
    v = (lam * e).astype(f32)
    return np.maximum(a, np.trunc(v).astype(np.int64))


def rect_axis_f64_margin(n_out, n_src):
    """The double flavour of the same expression (src/sat_decoder_interpolate_kernel.cl:56-65,
    compared with == there): distance of lambda * e from the next integer, in double ulps."""
    a = np.arange(0, n_out // 2 + 2, dtype=np.float64)
    lam = np.float64(f32(n_src) / (f32(np.exp(1.0)) - f32(1.0)))
    v = lam * (np.exp(np.power(2.0 * a / n_out, 4.0)) - 1.0)
    frac = np.minimum(v - np.floor(v), np.ceil(v) - v)
    ulps = frac / np.spacing(np.maximum(v, 1.0))
    live = v >= a  # where the max() takes this branch at all
    return float(ulps[live & (frac > 0)].min()) if np.any(live & (frac > 0)) else float("inf")


def sweep_rect(n_out, n_src, K):
    base = rect_axis(n_out, n_src, 0, 0, 0)
    moved = np.zeros(base.shape, dtype=bool)
    ks = range(-K, K + 1)
    for kp, ke, k1 in itertools.product(ks, ks, ks):
        moved |= rect_axis(n_out, n_src, kp, ke, k1) != base
    return int(moved.sum()), int(base.size)


def sweep_logpolar(out_w, out_h, K):
    """(int)(r * cos) / (int)(r * sin) with r = exp(10 * pow(i/n, 1)) (pow(x, 1) is exact in any
    conforming implementation, so only exp, cos and sin are perturbed)."""
    i = np.arange(out_w, dtype=np.int64)
    j = np.arange(out_h, dtype=np.int64)
    ang = ((j.astype(f32) / f32(out_h) * f32(2.0)).astype(np.float64) * 3.14159265359).astype(f32)
    x10 = f32(10.0) * (i.astype(f32) / f32(out_w))
    r0 = np.exp(x10.astype(np.float64)).astype(f32)
    c0 = np.cos(ang.astype(np.float64)).astype(f32)
    s0 = np.sin(ang.astype(np.float64)).astype(f32)

    def table(r, c, s):
        gx = np.trunc((r[None, :] * c[:, None]).astype(f32)).astype(np.int64)
        gy = np.trunc((r[None, :] * s[:, None]).astype(f32)).astype(np.int64)
        return gx, gy

    bx, by = table(r0, c0, s0)
    moved = np.zeros(bx.shape, dtype=bool)
    ks = range(-K, K + 1)
    rs = {k: nudge(r0, k) for k in ks}
    cs = {k: nudge(c0, k) for k in ks}
    ss = {k: nudge(s0, k) for k in ks}
    for kr, kt in itertools.product(ks, ks):
        gx, gy = table(rs[kr], cs[kt], ss[kt])
        moved |= (gx != bx) | (gy != by)
    # entries that land inside a frame of the sampled size at all (radius below its diagonal)
    return int(moved.sum()), int(moved.size)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ulps", type=int, default=4)
    args = ap.parse_args()
    K = args.ulps
    out = {"ulps": K, "rect_axes": [], "logpolar": []}
    for full in (1920, 1080, 3840, 7680):
        n_out = reduced(full)
        moved, total = sweep_rect(n_out, full, K)
        out["rect_axes"].append({"full": full, "reduced": n_out, "entries": total,
                                 "can_move": moved,
                                 "f64_min_margin_ulps": rect_axis_f64_margin(n_out, full)})
    for w, h in ((1920, 1080), (3840, 1920), (7680, 3840)):
        moved, total = sweep_logpolar(reduced(w), reduced(h), K)
        out["logpolar"].append({"full": [w, h], "reduced": [reduced(w), reduced(h)],
                                "entries": total, "can_move": moved})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
