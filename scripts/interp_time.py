"""Un-warp timing at 8K for a few gaze points (events over 30 calls)."""
import sys, math
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import f360_amd as f360
w, h = 7680, 3840
rw, rh = 16 * math.ceil(w / 1.8 / 16), 16 * math.ceil(h / 1.8 / 16)
with f360.Context(0) as ctx:
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    dec = f360.SATDecoder(ctx)
    red = ctx.upload(np.random.default_rng(1).integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
    full = ctx.malloc(w * h * 4)
    e0, e1 = f360.Event(ctx), f360.Event(ctx)
    for k in range(3):
        dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4, 0.5)
    ctx.finish()
    e0.record()
    n = 30
    for k in range(n):
        dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4 + 0.01 * k, 0.5)
    e1.record()
    print("interpolate_rect 8K", sys.argv[1:], "%.1f us" % (1e3 * e0.elapsed_ms(e1) / n))
    dec.close()
