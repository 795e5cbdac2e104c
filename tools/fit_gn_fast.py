#!/usr/bin/env python3
"""Coefficients of the FLOAT asin / atan cores behind the gnomonic remap's fast path
(csrc/gn_fast_math.h) and a host-side estimate of their absolute error.

    atan(t)  ~= t * Q(t^2),                         0 <= t <= 1        (t = min / max of |x|, |y|)
    asin(a)  ~= a + a * z * R(z), z = a^2,          |a| <= 0.5
    asin(a)  ~= pi/2 - 2 (s + s * z * R(z)),        z = (1 - |a|) / 2, s = sqrt(z), 0.5 < |a| <= 1

Q and R are Chebyshev interpolants computed in 80-bit arithmetic, rounded to float; the script then
evaluates them the way the kernel does -- float Horner, separate multiply and add -- on dense grids
against long double and prints the largest absolute error.  The device test
(tests/test_gpu_gn_fast.py) sweeps the device functions themselves (hardware rcp / sqrt); the
bound the kernel uses is a multiple of what both report.

    python tools/fit_gn_fast.py [--atan-terms 9] [--asin-terms 5] [--emit]
"""
import argparse

import numpy as np

ld = np.longdouble
f32 = np.float32


def cheb_fit(fun, zmax, terms):
    k = np.arange(terms, dtype=ld)
    nodes = np.cos(np.pi * (2 * k + 1) / (2 * terms)).astype(ld)
    z = (nodes + 1) * ld(zmax) / 2
    f = fun(z)
    A = np.concatenate([np.vander(z, terms, increasing=True).astype(ld), f[:, None]], axis=1)
    n = terms
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]
        A[i] = A[i] / A[i, i]
        for r in range(n):
            if r != i:
                A[r] = A[r] - A[r, i] * A[i]
    return A[:, n].astype(np.float64).astype(f32)


def atan_q(z):
    u = np.sqrt(z)
    return np.where(u > 0, np.arctan(u) / np.where(u > 0, u, 1), ld(1))


def asin_r(z):
    u = np.sqrt(z)
    safe = np.where(u > 0, u, 1)
    return np.where(u > 0, (np.arcsin(u) / safe - 1) / np.where(z > 0, z, 1), ld(1) / 6)


def horner32(z, c):
    q = np.full_like(z, c[-1], dtype=f32)
    for a in c[-2::-1]:
        q = (q * z).astype(f32)
        q = (q + a).astype(f32)
    return q


def atan_core32(t, c):
    t = t.astype(f32)
    z = (t * t).astype(f32)
    return (t * horner32(z, c)).astype(f32)


def asin32(a, c):
    a = a.astype(f32)
    s = np.abs(a)
    small = s <= f32(0.5)
    z = np.where(small, (a * a).astype(f32), ((f32(1) - s).astype(f32) * f32(0.5)).astype(f32)).astype(f32)
    r = horner32(z, c)
    base = np.where(small, a, np.sqrt(z).astype(f32)).astype(f32)
    p = (base + ((base * z).astype(f32) * r).astype(f32)).astype(f32)
    big = (f32(np.pi / 2) - (f32(2) * p).astype(f32)).astype(f32)
    return np.where(small, p, np.copysign(big, a)).astype(f32)


def estimate(atan_terms=9, asin_terms=5, n=4_000_001, seed=3):
    """(Q, R, max |atan core error|, max |asin error|) on grids of about n points each."""
    rng = np.random.default_rng(seed)
    q = cheb_fit(atan_q, 1.0002, atan_terms)
    t = np.concatenate([np.linspace(0, 1, n), rng.uniform(0, 1, n)]).astype(f32)
    e_at = float(np.max(np.abs(atan_core32(t, q).astype(ld) - np.arctan(t.astype(ld)))))
    r = cheb_fit(asin_r, 0.2502, asin_terms)
    a = np.concatenate([np.linspace(-1, 1, 2 * n), rng.uniform(-1, 1, n),
                        1 - np.logspace(-9, -1, 200_001), np.logspace(-9, -0.31, 200_001)]).astype(f32)
    e_as = float(np.max(np.abs(asin32(a, r).astype(ld) - np.arcsin(a.astype(ld)))))
    return q, r, e_at, e_as


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--atan-terms", type=int, default=9)
    ap.add_argument("--asin-terms", type=int, default=5)
    ap.add_argument("--emit", action="store_true")
    args = ap.parse_args()
    rng = np.random.default_rng(3)
    q = cheb_fit(atan_q, 1.0002, args.atan_terms)
    t = np.concatenate([np.linspace(0, 1, 4_000_001), rng.uniform(0, 1, 4_000_000)]).astype(f32)
    e_at = float(np.max(np.abs(atan_core32(t, q).astype(ld) - np.arctan(t.astype(ld)))))
    print(f"atan core, {args.atan_terms} terms: max |t Q(t^2) - atan t| on [0, 1] = {e_at:.3e}")
    r = cheb_fit(asin_r, 0.2502, args.asin_terms)
    a = np.concatenate([np.linspace(-1, 1, 8_000_001), rng.uniform(-1, 1, 4_000_000),
                        1 - np.logspace(-9, -1, 200_001), np.logspace(-9, -0.31, 200_001)]).astype(f32)
    e_as = float(np.max(np.abs(asin32(a, r).astype(ld) - np.arcsin(a.astype(ld)))))
    print(f"asin, {args.asin_terms} terms: max abs error on [-1, 1] = {e_as:.3e}")
    if args.emit:
        for name, c in (("kGnAtanQ", q), ("kGnAsinR", r)):
            print(f"constexpr float {name}[{len(c)}] = {{")
            for v in c:
                print(f"    {float(v).hex()}f,  // {float(v):+.9e}")
            print("};")


if __name__ == "__main__":
    main()
