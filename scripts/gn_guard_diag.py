"""Which texel indices differ between the guarded and the all-exact gnomonic remap for one case
(index-coded source frames: the pixel value is its own x or y).
    python scripts/gn_guard_diag.py w h tw th cx cy"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
w, h, tw, th = (int(v) for v in sys.argv[1:5])
cx, cy = float(sys.argv[5]), float(sys.argv[6])
xs = np.broadcast_to(np.arange(w, dtype=np.uint32)[None, :], (h, w))
ys = np.broadcast_to(np.arange(h, dtype=np.uint32)[:, None], (h, w))
out = {}
with f360.Context(0) as ctx:
    proj = f360.Projections(ctx)
    dst = ctx.malloc(tw * th * 4)
    for name, arr in (("x", xs), ("y", ys)):
        src = ctx.upload(np.ascontiguousarray(arr).view(np.uint8).reshape(h, 4 * w))
        for guard in (1, 0):
            ctx.set_option("gnomonic.guard", guard)
            dst.fill(0xEE)
            proj.GnomonicProjection(dst.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            out[(name, guard)] = dst.copy_to_host(np.uint32, (th, tw)) & 0xFFFFFF
        src.free()
bad = (out[("x", 1)] != out[("x", 0)]) | (out[("y", 1)] != out[("y", 0)])
print("differing pixels", int(bad.sum()), "of", tw * th)
for (j, i) in np.argwhere(bad)[:40]:
    print(f"pixel ({i},{j}): guarded ({out[('x',1)][j,i]},{out[('y',1)][j,i]}) exact ({out[('x',0)][j,i]},{out[('y',0)][j,i]})")
