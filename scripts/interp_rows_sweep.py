import sys, os, math
sys.path[:0] = ['.', 'tests']
import numpy as np, f360_amd as f360
def reduced(n): return 16 * math.ceil(n / 1.8 / 16)
for (w, h) in [(7680, 3840), (3840, 1920), (1920, 1080)]:
    rw, rh = reduced(w), reduced(h)
    with f360.Context(0) as ctx:
        dec = f360.SATDecoder(ctx)
        red = ctx.upload(np.random.default_rng(0).integers(0, 256, (rh, rw * 4), dtype=np.uint8))
        full = ctx.malloc(w * h * 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        for rows in (0, 1, 2, 4, 8, 16):
            ctx.set_option("interp.rows", rows)
            dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4, 0.5); ctx.finish()
            e0.record()
            for k in range(20):
                dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4 + 0.01 * k, 0.5)
            e1.record()
            print(w, h, "rows", rows, round(1e3 * e0.elapsed_ms(e1) / 20, 2), "us")
        dec.close()
