"""Gnomonic remap timing: per-geometry table layouts (none, five planes, two planes) x asin / atan2
through the library routines or through cr_math.h (gnomonic.fast)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
w, h = 7680, 3840
with f360.Context(0) as ctx:
    proj = f360.Projections(ctx)
    src = ctx.upload(np.random.default_rng(1).integers(0, 256, (h, 4 * w), dtype=np.uint8))
    e0, e1 = f360.Event(ctx), f360.Event(ctx)
    for (tw, th) in [(3840, 1920), (1920, 1080)]:
        view = ctx.malloc(tw * th * 4)
        for table, fast in ((0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)):
            ctx.set_option("gnomonic.table", table)
            ctx.set_option("gnomonic.fast", fast)
            for k in range(2):
                proj.GnomonicProjection(view.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, 0.5, 0.5)
            ctx.finish()
            e0.record()
            n = 20
            for k in range(n):
                proj.GnomonicProjection(view.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, 0.3 + 0.02 * k, 0.45)
            e1.record()
            print(f"gnomonic 8K -> {tw}x{th} gnomonic.table={table} gnomonic.fast={fast}: {1e3 * e0.elapsed_ms(e1) / n:.1f} us")
        view.free()
