"""Log-polar un-warp timing: direct evaluation, with the inverse-map table, with the table and the
axis tables in LDS (is.lp_lds)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
for (w, h) in [(7680, 3840), (3840, 1920), (1920, 1080)]:
    rw, rh = 16 * math.ceil(w / 1.8 / 16), 16 * math.ceil(h / 1.8 / 16)
    with f360.Context(0) as ctx:
        smp = f360.ImageSampler(ctx)
        red = ctx.upload(np.random.default_rng(1).integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
        full = ctx.malloc(w * h * 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        for table, lds in ((0, 0), (1, 0), (1, 1), (1, 256), (1, 512), (1, 1024)):
            ctx.set_option("is.lp_table", table)
            ctx.set_option("is.lp_lds", lds)
            for k in range(2):
                smp.InterpolateFrameLogPolarGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4, 0.5)
            ctx.finish()
            e0.record()
            n = 20
            for k in range(n):
                smp.InterpolateFrameLogPolarGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.3 + 0.02 * k, 0.5)
            e1.record()
            print(f"{w}x{h} is.lp_table={table} is.lp_lds={lds}: {1e3 * e0.elapsed_ms(e1) / n:.1f} us", flush=True)
        smp.close()
