"""The PPM dump tools (counterpart of the reference's SaveFramePNG).  CPU tier:
the fixture dumper; GPU tier: the device dumper's reduced frame equals the oracle's bytes."""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:3] == b"P6\n"
    header, rest = data[3:].split(b"\n255\n", 1)
    w, h = (int(v) for v in header.split())
    assert len(rest) == 3 * w * h
    return np.frombuffer(rest, dtype=np.uint8).reshape(h, w, 3)


def test_golden_fixtures_dump_as_ppm(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "golden_to_ppm.py"),
                          os.path.join(REPO, "tests", "golden", "small.npz"), str(tmp_path)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    gold = np.load(os.path.join(REPO, "tests", "golden", "small.npz"))
    img = read_ppm(tmp_path / "sample_rect_1.ppm")
    want = gold["sample_rect_1"].reshape(32, 48, 4)[:, :, :3]
    assert np.array_equal(img, want)
    assert np.array_equal(read_ppm(tmp_path / "interp_rect_2.ppm"), gold["interp_rect_2"][:, :, :3])


@pytest.mark.gpu
def test_device_frames_dump_as_ppm(f360, gpu_ctx, oracle, tmp_path):
    subprocess.run(["make", "-C", os.path.join(REPO, "examples")], check=True, capture_output=True)
    w, h, cx, cy = 640, 320, 0.65, 0.75
    out = subprocess.run([os.path.join(REPO, "examples", "dump_frames_ppm"), str(tmp_path), str(w),
                          str(h), str(cx), str(cy)], capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    res = json.loads(out.stdout.strip().splitlines()[-1])
    rw, rh = 16 * math.ceil(w / 1.8 / 16), 16 * math.ceil(h / 1.8 / 16)
    assert res["ok"] and res["reduced"] == [rw, rh]
    src = read_ppm(tmp_path / "source.ppm")
    assert src.shape == (h, w, 3)
    frame = np.zeros((h, w, 4), dtype=np.uint8)
    frame[:, :, :3] = src
    sat = oracle.sat_encode(frame.reshape(h, 4 * w), w, h, 4 * w)
    red = np.zeros((rh, 4 * rw), dtype=np.uint8)
    oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, oracle.satdec_grid(rw, rh, w, h),
                              float(np.float32(cx)), float(np.float32(cy)))
    assert np.array_equal(read_ppm(tmp_path / "reduced.ppm"), red.reshape(rh, rw, 4)[:, :, :3])
    full = oracle.satdec_interpolate_rect(red.reshape(rh, rw, 4), w, h, rw, rh, float(np.float32(cx)),
                                          float(np.float32(cy)))
    assert np.array_equal(read_ppm(tmp_path / "unwarped.ppm"), full[:, :, :3])
