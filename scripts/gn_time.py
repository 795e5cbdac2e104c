"""Gnomonic remap timing: per-geometry table layouts (none, five planes, two planes) x asin / atan2
through the library routines or through cr_math.h (gnomonic.fast), and the index-guarded remap
(gnomonic.guard: float evaluation + worklist; its exact resolve pass follows gnomonic.fast)."""
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
        for guard, table, fast in ((1, 1, 0), (1, 1, 1), (0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 0, 1), (0, 1, 1), (0, 2, 1)):
            ctx.set_option("gnomonic.guard", guard)
            ctx.set_option("debug.ablate", 0)
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
            ctx.finish()
            frac = 0.0
            if guard:   # one more, counted launch (the count is a debug atomic per workgroup)
                ctx.set_option("debug.ablate", 512)
                proj.GnomonicProjection(view.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, 0.5, 0.45)
                frac = ctx.debug_gnomonic_worklist() / (tw * th)
            print(f"gnomonic 8K -> {tw}x{th} gnomonic.guard={guard} gnomonic.table={table} gnomonic.fast={fast}: "
                  f"{1e3 * e0.elapsed_ms(e1) / n:.1f} us" + (f" (exact chain for {100 * frac:.2f} % of the pixels)" if guard else ""), flush=True)
        view.free()
