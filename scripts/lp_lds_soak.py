"""Soak of the LDS-table log-polar un-warp ("is.lp_lds") against the plain table kernel: random
frame / reduced-buffer geometries and gazes, byte for byte.
    python scripts/lp_lds_soak.py [seconds] [seed]"""
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
        kind = rng.integers(0, 4)
        if kind == 0:      # small and odd
            w, h = int(rng.integers(2, 400)), int(rng.integers(2, 300))
            rw, rh = int(rng.integers(2, 500)), int(rng.integers(1, 400))
        elif kind == 1:    # video sizes with the 1.8 rule
            w, h = [(1920, 1080), (3840, 1920), (2560, 1440), (4096, 2048)][rng.integers(0, 4)]
            rw, rh = 16 * -(-w // 29), 16 * -(-h // 29)
        elif kind == 2:    # reduced buffer larger than the frame, extreme aspect
            w, h = int(rng.integers(64, 2000)), int(rng.integers(2, 64))
            rw, rh = int(rng.integers(2, 4000)), int(rng.integers(1, 3000))
        else:
            w, h = int(rng.integers(2, 64)), int(rng.integers(64, 2000))
            rw, rh = int(rng.integers(2, 6000)), int(rng.integers(1, 100))
        smp = f360.ImageSampler(ctx)
        red = ctx.upload(rng.integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
        a, b = ctx.malloc(w * h * 4), ctx.malloc(w * h * 4)
        for g in range(4):
            cx, cy = float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
            if g == 3:
                cx, cy = float(rng.choice([0.0, 0.5, 1.0])), float(rng.choice([0.0, 0.5, 1.0]))
            ctx.set_option("is.lp_lds", int(rng.choice([1, 256, 512, 1024])))
            a.fill(0x11)
            smp.InterpolateFrameLogPolarGPU(a.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, cx, cy)
            ctx.set_option("is.lp_lds", 0)
            b.fill(0x22)
            smp.InterpolateFrameLogPolarGPU(b.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, cx, cy)
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
        smp.close()
print({"cases": cases, "pixels": pixels, "differing_pixels": bad, "first_failures": worst,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
