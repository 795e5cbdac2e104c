"""Soak of the staged un-warp ("interp.staged") against the per-pixel gathers: random frame /
reduced geometries and gazes, byte for byte.
    python scripts/interp_staged_soak.py [seconds] [seed]"""
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
    while time.time() - t0 < budget:
        kind = rng.integers(0, 3)
        if kind == 0:
            w, h = int(rng.integers(2, 600)), int(rng.integers(2, 400))
            rw, rh = int(rng.integers(2, 700)), int(rng.integers(2, 500))
        elif kind == 1:
            w, h = [(1920, 1080), (3840, 1920), (2560, 1440), (7680, 3840)][rng.integers(0, 4)]
            rw, rh = 16 * -(-w // 29), 16 * -(-h // 29)
        else:
            w, h = int(rng.integers(256, 9000)), int(rng.integers(2, 48))
            rw, rh = int(rng.integers(2, 6000)), int(rng.integers(2, 64))
        dec = f360.SATDecoder(ctx)
        red = ctx.upload(rng.integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
        a, b = ctx.malloc(w * h * 4), ctx.malloc(w * h * 4)
        for g in range(4):
            cx, cy = float(rng.uniform(-0.1, 1.1)), float(rng.uniform(-0.1, 1.1))
            if g == 3:
                cx, cy = float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([0.0, 0.5, 1.0]))
            try:
                ctx.set_option("interp.staged", 1)
                a.fill(0x11)
                dec.InterpolateFrameRectGPU(a.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, cx, cy)
                ctx.set_option("interp.staged", 0)
                b.fill(0x11)
                dec.InterpolateFrameRectGPU(b.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, cx, cy)
            except f360.F360Error:
                continue   # (a geometry the entry point refuses: both paths refuse it alike)
            ga, gb = a.copy_to_host(np.uint8, (h, w, 4)), b.copy_to_host(np.uint8, (h, w, 4))
            n = int((ga != gb).any(axis=2).sum())
            cases += 1
            pixels += w * h
            if n:
                bad += n
                worst = worst or []
                if len(worst) < 12:
                    worst.append((w, h, rw, rh, cx, cy, n))
        for buf in (red, a, b):
            buf.free()
        dec.close()
print({"cases": cases, "pixels": pixels, "differing_pixels": bad, "first_failures": worst,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
