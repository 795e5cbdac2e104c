#!/usr/bin/env python3
"""Debug aid: one frame through the band writer's one pass, the differing reduced pixels with
their boxes.  usage: python scripts/band_debug.py [w h cx cy]"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import f360_amd as f360
import oracle_binding as ob
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 512)
cx, cy = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.5, 0.5)
rw, rh = f360.reduced_size(w), f360.reduced_size(h)
frame = ob.lcg_frame(w, h, 300)
with f360.Context(0) as ctx:
    dec = f360.SATDecoder(ctx); dec.InitializeGrid(rw, rh, w, h)
    src, sat, red = ctx.upload(frame.reshape(-1)), ctx.malloc(w * h * 12), ctx.malloc(rw * rh * 4)
    red.fill(0xA5)
    dec.EncodeSampleFramesGPU([red.ptr], rw, rh, 4 * rw, [sat.ptr], [src.ptr], w, h, 4 * w, [(cx, cy)])
    got = red.copy_to_host(np.uint8, (rh, rw, 4))
    got_sat = sat.copy_to_host(np.uint32, (h, w, 3))
    dec.close()
want_sat = ob.sat_encode(frame, w, h, 4 * w)
print("table equal:", np.array_equal(got_sat, want_sat))
grid = ob.satdec_grid(rw, rh, w, h)
want = np.full((rh, 4 * rw), 0xA5, dtype=np.uint8)
ob.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, cx, cy)
want = want.reshape(rh, rw, 4)
gx, gy = ob.satdec_grid_axes(rw, rh, w, h)
def axis(c, dhi, dlo, size, wraps):
    hi, lo = c + int(dhi), c + int(dlo)
    if wraps:
        if hi >= size and lo >= size: hi -= size; lo -= size
        elif hi < 0 and lo < 0: hi += size; lo += size
    ok = (0 <= hi < size) or (0 <= lo < size)
    hi = min(max(hi, 1), size - 1); lo = min(max(lo, 0), hi - 1)
    return hi, lo, ok
cxp, cyp = int(np.float32(cx) * np.float32(w)), int(np.float32(cy) * np.float32(h))
bad = np.argwhere((got != want).any(axis=2))
print("differing pixels:", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:30])
for j, i in bad[:24]:
    print((int(j), int(i)), "want", want[j, i].tolist(), "got", got[j, i].tolist(),
          "box x", axis(cxp, gx[i + 1], gx[i], w, True), "box y", axis(cyp, gy[j + 1], gy[j], h, False))
