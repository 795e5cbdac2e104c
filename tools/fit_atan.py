#!/usr/bin/env python3
"""Coefficients and error bound of the atan core behind the fast correctly-rounded float asin /
atan2 of the gnomonic remap (csrc/cr_math.h).

    atan(u) ~= u * Q(u^2),  |u| <= tan(pi/8),  Q of degree N-1 in z = u^2

Q is a Chebyshev interpolant of atan(sqrt(z))/sqrt(z) computed in 80-bit long double; the script
prints the coefficients as C hex-float literals, evaluates the polynomial the way the kernel does
(double Horner) on a dense grid plus random points against long double atan, and prints the
largest absolute error.  The kernel's rounding guard uses a bound several hundred times larger.

    python tools/fit_atan.py [--terms 13] [--emit]
"""
import argparse

import numpy as np

TAN_PI_8 = 0.41421356237309503  # slightly below sqrt(2) - 1 rounded up is fine: the kernel compares with this


def fit(terms):
    ld = np.longdouble
    zmax = ld(TAN_PI_8) ** 2 * ld(1.0002)  # a hair beyond the range the kernel uses
    # Chebyshev nodes on [0, zmax]
    k = np.arange(terms, dtype=ld)
    nodes = np.cos(np.pi * (2 * k + 1) / (2 * terms)).astype(ld)
    z = (nodes + 1) * zmax / 2
    u = np.sqrt(z)
    f = np.where(u > 0, np.arctan(u) / np.where(u > 0, u, 1), ld(1))
    # polynomial through the nodes, in the monomial basis of z (well conditioned enough at this
    # degree on a short interval; solved in long double)
    V = np.vander(z, terms, increasing=True).astype(ld)
    # Gaussian elimination in long double (numpy.linalg has no long double solver)
    A = np.concatenate([V, f[:, None]], axis=1)
    n = terms
    for i in range(n):
        p = i + int(np.argmax(np.abs(A[i:, i])))
        A[[i, p]] = A[[p, i]]
        A[i] = A[i] / A[i, i]
        for r in range(n):
            if r != i:
                A[r] = A[r] - A[r, i] * A[i]
    return A[:, n].astype(np.float64)


def poly_atan(u, c):
    """The kernel's evaluation order: Horner in z = u*u, then one multiply by u (all double)."""
    u = np.asarray(u, dtype=np.float64)
    z = u * u
    q = np.full_like(z, c[-1])
    for a in c[-2::-1]:
        q = q * z + a
    return u * q


def max_error(c, n=4_000_001, seed=1):
    u = np.linspace(0.0, TAN_PI_8, n)
    rng = np.random.default_rng(seed)
    u = np.concatenate([u, rng.uniform(0, TAN_PI_8, n)])
    want = np.arctan(u.astype(np.longdouble))
    return float(np.max(np.abs(poly_atan(u, c).astype(np.longdouble) - want)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--terms", type=int, default=13)
    ap.add_argument("--emit", action="store_true")
    args = ap.parse_args()
    c = fit(args.terms)
    err = max_error(c)
    print(f"terms {args.terms}: max |u*Q(u^2) - atan(u)| on [0, tan(pi/8)] = {err:.3e}")
    if args.emit:
        print("constexpr double kAtanQ[%d] = {" % len(c))
        for a in c:
            print(f"    {float(a).hex()},  // {a:+.17e}")
        print("};")


if __name__ == "__main__":
    main()
