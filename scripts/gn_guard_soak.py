"""Soak of the index-guarded gnomonic remap: random source / viewport geometries and gazes, the
guarded launch against the kernel that runs the exact chain on every pixel, byte for byte.
    python scripts/gn_guard_soak.py [seconds] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
cases = pixels = bad = 0
worst = None
with f360.Context(0) as ctx:
    proj = f360.Projections(ctx)
    while time.time() - t0 < budget:
        kind = rng.integers(0, 4)
        if kind == 0:      # small and odd
            w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
            tw, th = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        elif kind == 1:    # video sizes
            w, h = [(1920, 1080), (3840, 1920), (4096, 2048), (7680, 3840), (8192, 4096)][rng.integers(0, 5)]
            tw, th = int(rng.integers(64, 2200)), int(rng.integers(64, 1300))
        elif kind == 2:    # extreme aspect
            w, h = int(rng.integers(2000, 16000)), int(rng.integers(2, 64))
            tw, th = int(rng.integers(1, 3000)), int(rng.integers(1, 40))
        else:              # tall
            w, h = int(rng.integers(2, 64)), int(rng.integers(2000, 12000))
            tw, th = int(rng.integers(1, 64)), int(rng.integers(1, 3000))
        frame = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
        src = ctx.upload(frame)
        a, b = ctx.malloc(tw * th * 4), ctx.malloc(tw * th * 4)
        for g in range(4):
            cx, cy = float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
            if g == 3:
                cx, cy = float(rng.choice([0.0, 0.25, 0.5, 0.75, 1.0])), float(rng.choice([0.0, 0.5, 1.0]))
            ctx.set_option("gnomonic.guard", 1)
            a.fill(0x11)
            proj.GnomonicProjection(a.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            ctx.set_option("gnomonic.guard", 0)
            b.fill(0x22)
            proj.GnomonicProjection(b.ptr, tw, th, 4 * tw, src.ptr, w, h, 4 * w, cx, cy)
            ga, gb = a.copy_to_host(np.uint8, (th, tw, 4)), b.copy_to_host(np.uint8, (th, tw, 4))
            n = int((ga != gb).any(axis=2).sum())
            cases += 1
            pixels += tw * th
            if n:
                bad += n
                worst = worst or []
                if len(worst) < 12:
                    worst.append((w, h, tw, th, cx, cy, n))
        for buf in (src, a, b):
            buf.free()
print({"cases": cases, "pixels": pixels, "differing_pixels": bad, "first_failures": worst,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
