"""A few launches of the gnomonic remap at one viewport (for rocprofv3 runs).
    python scripts/gn_one.py [tw th [guard]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
tw, th = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 1920)
guard = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w, h = 7680, 3840
with f360.Context(0) as ctx:
    ctx.set_option("gnomonic.guard", guard)
    proj = f360.Projections(ctx)
    src = ctx.upload(np.random.default_rng(1).integers(0, 256, (h, 4 * w), dtype=np.uint8))
    view = ctx.malloc(tw * th * 4)
    for k in range(12):
        proj.GnomonicProjection(view.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, 0.3 + 0.02 * k, 0.45)
    ctx.finish()
