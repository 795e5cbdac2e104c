"""CPU tier for the index-guarded gnomonic remap (csrc/gn_fast_math.h): the coefficients in the
header are the ones tools/fit_gn_fast.py derives, the polynomial cores evaluated the way the
kernel evaluates them (float Horner, here without the FMAs the device has) stay inside half the
bounds the guard is built from, and the guard arithmetic of projections.hip is what DESIGN.md
states.  The device functions themselves (hardware rcp / sqrt) are swept in test_gpu_gn_fast.py."""
import importlib.util
import os
import re

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "foveated-360-video_amd", "csrc", "gn_fast_math.h")


def _fit_tool():
    spec = importlib.util.spec_from_file_location("fit_gn_fast", os.path.join(REPO, "tools", "fit_gn_fast.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _header_hex_floats(func):
    text = open(HEADER).read()
    body = text[text.index(func):]
    body = body[:body.index("return")]
    vals = [float.fromhex(v.rstrip("f")) for v in re.findall(r"-?0x1\.[0-9a-f]+p[+-]?\d+f", body)]
    if "1.0f);" in body:  # the leading coefficient of Q is written as 1.0f
        vals.append(1.0)
    return vals


def test_header_coefficients_are_the_fitted_ones_and_errors_are_inside_the_bounds():
    tool = _fit_tool()
    q, r, e_at, e_as = tool.estimate(n=1_000_001)
    text = open(HEADER).read()
    e_asin = float(re.search(r"kGnEAsin = ([0-9.e-]+)f", text).group(1))
    e_atan2 = float(re.search(r"kGnEAtan2 = ([0-9.e-]+)f", text).group(1))
    # Horner order in the header: highest coefficient first
    hq = _header_hex_floats("gn_atan_q")
    hr = _header_hex_floats("gn_asin_r")
    assert np.allclose(hq, [float(v) for v in q[::-1]], rtol=0, atol=0), (hq, q)
    assert np.allclose(hr, [float(v) for v in r[::-1]], rtol=0, atol=0), (hr, r)
    # polynomial cores alone: well inside half the bounds (the rest of the budget is the
    # hardware reciprocal, the octant fix-ups and the pi constants, measured on the device)
    assert e_as < 0.5 * e_asin, (e_as, e_asin)
    assert e_at < 0.25 * e_atan2, (e_at, e_atan2)


def test_pi_constant_pairs():
    text = open(HEADER).read()
    pairs = dict(re.findall(r"(kPio2Hi|kPio2Lo|kPiHi|kPiLo) = (-?0x1\.[0-9a-f]+p[+-]?\d+)f", text))
    hi2, lo2 = float.fromhex(pairs["kPio2Hi"]), float.fromhex(pairs["kPio2Lo"])
    hi, lo = float.fromhex(pairs["kPiHi"]), float.fromhex(pairs["kPiLo"])
    assert hi2 == float(np.float32(np.pi / 2)) and hi == float(np.float32(np.pi))
    assert abs(hi2 + lo2 - np.pi / 2) < 1e-14 and abs(hi + lo - np.pi) < 2e-14
