"""Soak of the two streaming siblings rebuilt in round 4, byte for byte, on random geometries:
  * rgb0_to_yuv420p: the row walker (runs of 1 .. 64 chroma rows, both libswscale models, padded
    source rows and planes) against the kernel with one chroma row per thread ("yuv.r2y_rows" -1);
  * decode: the LDS-direct row streamer against a host cumsum-inverse of the table (the frame the
    table was encoded from must come back in bytes 0..2 of every pixel, byte 3 untouched).
    python scripts/siblings_soak.py [seconds] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
cases = {"r2y": 0, "decode": 0}
pixels = bad = 0
worst = []
with f360.Context(0) as ctx:
    enc, dec = f360.SATEncoder(ctx), f360.SATDecoder(ctx)
    while time.time() - t0 < budget:
        # ---- rgb0 -> yuv420p
        kind = rng.integers(0, 3)
        if kind == 0:
            w, h = 8 * int(rng.integers(1, 160)), 2 * int(rng.integers(4, 200))
        elif kind == 1:
            w, h = 8 * int(rng.integers(100, 1100)), 2 * int(rng.integers(4, 40))
        else:
            w, h = 8 * int(rng.integers(1, 40)), 2 * int(rng.integers(200, 1500))
        spad = 16 * int(rng.integers(0, 3))
        ypad, cpad = 8 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
        src_h = rng.integers(0, 256, (h, 4 * w + spad), dtype=np.uint8)
        if rng.integers(0, 6) == 0:
            src_h[:] = rng.choice([0, 255])
        src = ctx.upload(src_h)
        yl, cl = w + ypad, w // 2 + cpad
        outs = []
        model = int(rng.integers(0, 2))
        ctx.set_option("yuv.model", model)
        for rows in (-1, int(rng.choice([1, 2, 3, 4, 5, 8, 16, 31, 64]))):
            ctx.set_option("yuv.r2y_rows", rows)
            planes = [ctx.malloc(h * yl), ctx.malloc(h // 2 * cl), ctx.malloc(h // 2 * cl)]
            for p in planes:
                p.fill(0xEE)
            ctx.rgb0_to_yuv420p(planes[0].ptr, planes[1].ptr, planes[2].ptr, yl, cl, cl, src.ptr,
                                4 * w + spad, w, h)
            outs.append([planes[0].copy_to_host(np.uint8, (h, yl)), planes[1].copy_to_host(np.uint8, (h // 2, cl)),
                         planes[2].copy_to_host(np.uint8, (h // 2, cl))])
            for p in planes:
                p.free()
        src.free()
        cases["r2y"] += 1
        pixels += w * h
        n = sum(int((a != b).sum()) for a, b in zip(*outs))
        if n:
            bad += n
            if len(worst) < 12:
                worst.append(("r2y", w, h, spad, ypad, cpad, model, rows, n))
        # ---- decode
        kind = rng.integers(0, 3)
        if kind == 0:
            w, h = 4 * int(rng.integers(1, 300)), int(rng.integers(1, 300))
        elif kind == 1:
            w, h = 4 * int(rng.integers(200, 2100)), int(rng.integers(1, 80))
        else:
            w, h = 4 * int(rng.integers(1, 90)), int(rng.integers(200, 2500))
        frame = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
        if rng.integers(0, 6) == 0:
            frame[:] = 255
        fsrc, sat, out = ctx.upload(frame), ctx.malloc(w * h * 12), ctx.malloc(w * h * 4)
        enc.EncodeFrameGPU(sat.ptr, fsrc.ptr, w, h, 4 * w)
        fill = int(rng.integers(0, 256))
        out.fill(fill)
        dec.DecodeFrameGPU(out.ptr, 4 * w, sat.ptr, w, h)
        got = out.copy_to_host(np.uint8, (h, w, 4))
        want = frame.reshape(h, w, 4).copy()
        want[:, :, 3] = fill
        cases["decode"] += 1
        pixels += w * h
        n = int((got != want).any(axis=2).sum())
        if n:
            bad += n
            if len(worst) < 12:
                worst.append(("decode", w, h, n))
        for b in (fsrc, sat, out):
            b.free()
    ctx.set_option("yuv.r2y_rows", 0)
    ctx.set_option("yuv.model", 1)
    dec.close()
print({"cases": cases, "pixels": pixels, "differing": bad, "first_failures": worst or None,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
