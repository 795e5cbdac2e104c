"""Un-warp timing: staged vertical lerps (interp.staged) against per-pixel gathers, three sizes."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360


def reduced(n):
    return 16 * math.ceil(n / 1.8 / 16)


for (w, h) in [(7680, 3840), (3840, 1920), (1920, 1080)]:
    rw, rh = reduced(w), reduced(h)
    with f360.Context(0) as ctx:
        dec = f360.SATDecoder(ctx)
        red = ctx.upload(np.random.default_rng(1).integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
        full = ctx.malloc(w * h * 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        for staged in (0, 1, 0, 1):
            ctx.set_option("interp.staged", staged)
            for k in range(3):
                dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4, 0.5)
            ctx.finish()
            e0.record()
            n = 20
            for k in range(n):
                dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.3 + 0.02 * k, 0.45)
            e1.record()
            us = 1e3 * e0.elapsed_ms(e1) / n
            print(f"interpolate_rect {w}x{h} staged={staged}: {us:.1f} us  "
                  f"{(4 * rw * rh + 4 * w * h) / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
        dec.close()
