"""CPU tier: the numerics behind csrc/cr_math.h (fast correctly rounded float asin / atan2).
The header's polynomial is the one tools/fit_atan.py derives; evaluated in the kernel's order it
stays within the error bound the rounding guard assumes -- with a factor of 50 to spare -- and a
numpy restatement of the whole routine (core, octant fix-ups, guard) returns the correctly rounded
float wherever its guard passes."""
import os
import re
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import fit_atan  # noqa: E402

HEADER = os.path.join(REPO, "foveated-360-video_amd", "csrc", "cr_math.h")
EPS = 1e-12


def header_coefficients():
    text = open(HEADER).read()
    body = text[text.index("cr_atan_poly(double u)"):text.index("return u * q;")]
    hexes = re.findall(r"(-?0x1\.[0-9a-f]+p[+-]?\d+)", body)
    return np.array([float.fromhex(h) for h in hexes][::-1])  # Horner order -> ascending


def test_header_holds_the_fitted_polynomial():
    c = header_coefficients()
    assert len(c) == 10
    assert np.array_equal(c, fit_atan.fit(10))
    assert float(re.search(r"kCrEps = ([0-9.e-]+);", open(HEADER).read()).group(1)) == EPS


def test_polynomial_error_is_far_inside_the_guard():
    c = header_coefficients()
    err = fit_atan.max_error(c, n=1_000_001)
    assert err < 1e-15, err
    # ... and relatively (results can be tiny: atan2(y, x) ~ y / x)
    u = np.concatenate([np.exp2(np.random.default_rng(3).uniform(-60, -1.27, 1_000_000)),
                        np.linspace(1e-6, fit_atan.TAN_PI_8, 1_000_001)])
    want = np.arctan(u.astype(np.longdouble))
    rel = np.abs(fit_atan.poly_atan(u, c).astype(np.longdouble) - want) / want
    assert float(rel.max()) < 2e-15 <= EPS / 500, float(rel.max())


def core_atan2(y, x, c):
    """cr_atan2_core in float64 numpy."""
    ax, ay = np.abs(x), np.abs(y)
    mn, mx = np.minimum(ax, ay), np.maximum(ax, ay)
    far = mn > fit_atan.TAN_PI_8 * mx
    num = np.where(far, mn - mx, mn)
    den = np.where(far, mn + mx, mx)
    r = fit_atan.poly_atan(num / den, c)
    r = np.where(far, r + 0.78539816339744831, r)
    r = np.where(ay > ax, 1.5707963267948966 - r, r)
    r = np.where(x < 0.0, 3.141592653589793 - r, r)
    return np.where(np.signbit(y), -r, r)


def guarded(r):
    f = r.astype(np.float32)
    e = EPS * np.abs(r)
    ok = ((r - e).astype(np.float32) == f) & ((r + e).astype(np.float32) == f)
    return f, ok


def test_restated_routines_round_correctly_where_the_guard_passes():
    c = header_coefficients()
    rng = np.random.default_rng(7)
    # atan2: magnitudes over 40 binades, all sign combinations
    n = 2_000_000
    y = (rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))).astype(np.float32)
    x = (rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))).astype(np.float32)
    r = core_atan2(y.astype(np.float64), x.astype(np.float64), c)
    exact = np.arctan2(y.astype(np.longdouble), x.astype(np.longdouble))
    assert float(np.max(np.abs(r.astype(np.longdouble) - exact) / np.abs(exact))) < 5e-15
    f, ok = guarded(r)
    assert ok.mean() > 0.9995
    assert np.array_equal(f[ok], exact.astype(np.float32)[ok])
    # asin: uniform, and crowded towards +-1 and 0
    a = np.concatenate([rng.uniform(-1, 1, n), 1 - np.exp2(-rng.uniform(0, 24, n // 4)),
                        -1 + np.exp2(-rng.uniform(0, 24, n // 4)),
                        rng.standard_normal(n // 4) * 1e-3]).astype(np.float32)
    a = a[np.abs(a) <= 1]
    d = a.astype(np.float64)
    r = core_atan2(d, np.sqrt((1.0 - d) * (1.0 + d)), c)
    a, d, r = a[a != 0], d[a != 0], r[a != 0]
    exact = np.arcsin(a.astype(np.longdouble))
    assert float(np.max(np.abs(r.astype(np.longdouble) - exact) / np.abs(exact))) < 5e-15
    f, ok = guarded(r)
    assert ok.mean() > 0.9995
    assert np.array_equal(f[ok], exact.astype(np.float32)[ok])
