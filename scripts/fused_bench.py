import sys, math
sys.path[:0] = ['.', 'tests']
import numpy as np, f360_amd as f360
def reduced(n): return 16 * math.ceil(n / 1.8 / 16)
for (w, h) in [(7680, 3840), (3840, 1920), (1920, 1080)]:
    rw, rh = reduced(w), reduced(h)
    with f360.Context(0) as ctx:
        enc, dec = f360.SATEncoder(ctx), f360.SATDecoder(ctx)
        dec.InitializeGrid(rw, rh, w, h)
        rng = np.random.default_rng(0)
        frames = [ctx.upload(rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)) for _ in range(6)]
        sat, red = ctx.malloc(w * h * 12), ctx.malloc(rw * rh * 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        def separate(k):
            enc.EncodeFrameGPU(sat.ptr, frames[k % 6].ptr, w, h, 4 * w)
            dec.SampleFrameRectGPU(red.ptr, rw, rh, 4 * rw, sat.ptr, (w, h), 0.4 + 0.01 * k, 0.5)
        def fused(k):
            dec.FoveateFrameRectGPU(red.ptr, rw, rh, 4 * rw, frames[k % 6].ptr, w, h, 4 * w, 0.4 + 0.01 * k, 0.5)
        for name, fn in (("encode+sample", separate), ("fused foveate", fused)):
            fn(0); ctx.finish()
            e0.record()
            for k in range(48): fn(k)
            e1.record()
            us_plain = 1e3 * e0.elapsed_ms(e1) / 48          # no per-kernel events
            ctx.profile_reset(); ctx.profile_arm(1000)
            e0.record()
            for k in range(24): fn(k)
            e1.record()
            us = 1e3 * e0.elapsed_ms(e1) / 24
            ctx.profile_arm(0)
            print(w, h, name, round(us_plain, 1), "us/frame", round(w * h / us_plain / 1e3, 1), "Gpix/s;",
                  "with per-kernel events", round(us, 1), "us:",
                  {k: round(1e3 * v[0] / v[1], 1) for k, v in ctx.profile_read().items()})
        dec.close()
