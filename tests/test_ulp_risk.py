"""tools/ulp_risk.py counts table entries that hang on the last bits of the OpenCL builtins; its
unperturbed tables must be the oracle's own, or the count describes something else."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tools"), os.path.join(REPO, "tests")]

import oracle_binding as ob  # noqa: E402
import ulp_risk  # noqa: E402


def test_rect_axis_is_the_oracles_grid():
    ow, oh, sw, sh = 272, 160, 480, 270
    g = np.asarray(ob.is_grid(ow, oh, sw, sh)).reshape(oh, ow, 2).astype(np.int64)
    ax = ulp_risk.rect_axis(ow, sw, 0, 0, 0)
    ay = ulp_risk.rect_axis(oh, sh, 0, 0, 0)
    assert np.array_equal(np.abs(g[0, :, 0]), ax[np.abs(np.arange(ow) - ow // 2)])
    assert np.array_equal(np.abs(g[:, 0, 1]), ay[np.abs(np.arange(oh) - oh // 2)])


def test_a_perturbation_moves_few_entries():
    moved, total = ulp_risk.sweep_rect(272, 480, 1)
    assert 0 <= moved < total // 20


def test_logpolar_base_is_the_oracles_grid(monkeypatch):
    ow, oh = 144, 80
    g = np.asarray(ob.is_logpolar_grid(ow, oh, 256, 144)).reshape(oh, ow, 2).astype(np.int64)
    seen = {}
    real_trunc = np.trunc

    def spy(x):
        r = real_trunc(x)
        seen.setdefault("t", []).append(r)
        return r

    monkeypatch.setattr(ulp_risk.np, "trunc", spy)
    ulp_risk.sweep_logpolar(ow, oh, 0)
    gx, gy = seen["t"][0].astype(np.int64), seen["t"][1].astype(np.int64)
    # the table is int16 in the kernel; e^10 * cos stays inside it
    assert np.array_equal(gx, g[:, :, 0]) and np.array_equal(gy, g[:, :, 1])
