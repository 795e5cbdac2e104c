"""Point samplers of ImageSampler (log-rectilinear and log-polar forward warps) with and without
the XCD row bands ("is.xcd_bands"), by frame size.
    python scripts/is_time.py"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
for (w, h) in [(7680, 3840), (3840, 1920), (1920, 1080)]:
    rw, rh = 16 * math.ceil(w / 1.8 / 16), 16 * math.ceil(h / 1.8 / 16)
    with f360.Context(0) as ctx:
        smp = f360.ImageSampler(ctx)
        smp.InitializeGrid(rw, rh, w, h)
        smp.InitializeLogpolarGrid(rw, rh, w, h)
        frame = ctx.upload(np.random.default_rng(1).integers(0, 256, (h, 4 * w), dtype=np.uint8))
        red = ctx.malloc(rw * rh * 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        for name, fn in (("is_sample_rect", smp.SampleFrameRectGPU), ("is_sample_logpolar", smp.SampleFrameLogPolarGPU)):
            for bands in (0, 2, 1):
                ctx.set_option("is.xcd_bands", bands)
                for k in range(3):
                    fn(red.ptr, rw, rh, 4 * rw, frame.ptr, w, h, 4 * w, 0.5, 0.5)
                ctx.finish()
                e0.record()
                n = 30
                for k in range(n):
                    fn(red.ptr, rw, rh, 4 * rw, frame.ptr, w, h, 4 * w, 0.3 + 0.01 * k, 0.45)
                e1.record()
                print(f"{w}x{h} {name} is.xcd_bands={bands}: {1e3 * e0.elapsed_ms(e1) / n:.1f} us", flush=True)
        smp.close()
