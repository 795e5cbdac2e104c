"""CPU tier: the gaze-trace reader/writer (include/f360/gaze_view_points.h) against the format the
reference parses (src/gaze_view_points.cc:5-33): `frame,<n>,forward,<x>,<y>,eye,<x>,<y>`, prediction =
previous sample, non-matching lines skipped."""
import math
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lissajous_trace(n):
    """cx = 0.5 + 0.45 sin(2 pi k / 97), cy = 0.5 + 0.35 sin(2 pi k / 61)."""
    return [(k, 0.5, 0.5, np.float32(0.5 + 0.45 * math.sin(2 * math.pi * k / 97)),
             np.float32(0.5 + 0.35 * math.sin(2 * math.pi * k / 61))) for k in range(n)]


def write_trace(path, samples, junk=True):
    with open(path, "w") as f:
        if junk:
            f.write("# synthetic Lissajous trace\n\n")
        for (k, vx, vy, gx, gy) in samples:
            f.write(f"frame,{k},forward,{vx:.9g},{vy:.9g},eye,{gx:.9g},{gy:.9g}\n")
        if junk:
            f.write("frame,x,forward,1,2\n")  # malformed: skipped


def test_gaze_trace_roundtrip(tmp_path):
    subprocess.run(["make", "-C", os.path.join(REPO, "examples"), "gaze_trace_tool"], check=True,
                   capture_output=True)
    tool = os.path.join(REPO, "examples", "gaze_trace_tool")
    samples = lissajous_trace(50) + [(50, -0.25, 1.5e0, 1e-3, 9.99e-1)]
    src, dst = tmp_path / "in.txt", tmp_path / "out.txt"
    write_trace(src, samples)
    out = subprocess.run([tool, str(src), str(dst)], capture_output=True, text=True, check=True)
    rows = [list(map(float, line.split())) for line in out.stdout.strip().splitlines()]
    assert len(rows) == len(samples)
    for n, (row, (k, vx, vy, gx, gy)) in enumerate(zip(rows, samples)):
        assert int(row[0]) == k
        # %.9g identifies a float exactly: compare as float32
        assert np.array_equal(np.float32(row[1:5]), np.float32([vx, vy, gx, gy]))
        prev = samples[n - 1] if n else samples[0]
        assert np.array_equal(np.float32(row[5:9]), np.float32(prev[1:5]))
    # the rewritten file parses to the same samples
    out2 = subprocess.run([tool, str(dst)], capture_output=True, text=True, check=True)
    assert out2.stdout == out.stdout
