"""Soak of the one-pass encode + sample (f360_satdec_encode_sample_frames with the read-once
encoder forced on) against the two calls it replaces, byte for byte, on random geometries, frame
counts, gaze points (inside, on and beyond every edge), padded targets; a third of the calls from
planar YUV 4:2:0 frames (both libswscale models); every call also through the no-table form
(FoveateFramesRect[YUV420P]GPU):
    python scripts/fuse_soak.py [seconds] [seed] [big | band | bandbig]
("big": frames of 2560x1280 to 7680x3840, up to 5 per call -- dozens of strips, thousands of rows;
"band" / "bandbig": the one-pass form of the three-kernel encoder's table writer instead
(csrc/sat_band_fuse.hip: read-once encoder off, "fuse.band" 2 so that single frames take it too),
band height and the side stream drawn per call)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
big = len(sys.argv) > 3 and sys.argv[3] in ("big", "bandbig")
band = len(sys.argv) > 3 and sys.argv[3] in ("band", "bandbig")
rng = np.random.default_rng(seed)
t0 = last_note = time.time()
calls = frames_done = bad = planar_cases = 0
worst = []
with f360.Context(0) as ctx:
    ctx.set_option("sat.walk", 0 if band else 1)
    if band:
        ctx.set_option("fuse.band", 2)
    enc = f360.SATEncoder(ctx)
    while time.time() - t0 < budget:
        kind = rng.integers(0, 4) if not big else 9
        if big:
            w = 256 * int(rng.integers(10, 31)) - 4 * int(rng.integers(0, 3))
            h = 8 * int(rng.integers(100, 481)) - int(rng.integers(0, 8))
        elif kind == 0:
            w, h = 4 * int(rng.integers(1, 300)), int(rng.integers(2, 200))
        elif kind == 1:
            w, h = 4 * int(rng.integers(200, 1100)), int(rng.integers(2, 120))
        elif kind == 2:
            w, h = 4 * int(rng.integers(16, 200)), int(rng.integers(100, 1200))
        else:
            w, h = 256 * int(rng.integers(1, 12)), 8 * int(rng.integers(1, 60))
        n = int(rng.integers(1, 14)) if not big else int(rng.integers(1, 6))
        if band:
            ctx.set_option("sat.band_rows", int(rng.choice([0, 0, 16, 32, 64])))
            ctx.set_option("sat.pipeline", int(rng.integers(0, 2)))
            ctx.set_option("debug.fuse_force", int(rng.choice([0, 0, 0, 1, 2, 3])))
        rw, rh = f360.reduced_size(w), f360.reduced_size(h)
        tpad = 4 * int(rng.integers(0, 5))
        tl = 4 * rw + tpad
        gazes = []
        for _ in range(n):
            m = rng.integers(0, 4)
            if rng.integers(0, 40) == 0:  # far outside: nothing, or only the wrapped columns, is processed
                g = (float(rng.uniform(-16, 16)), float(rng.uniform(-16, 16)))
            elif m == 0:
                g = (float(rng.uniform(0, 1)), float(rng.uniform(0, 1)))
            elif m == 1:
                g = (float(rng.uniform(-1.5, 2.5)), float(rng.uniform(-1.5, 2.5)))
            elif m == 2:
                g = (float(rng.choice([0.0, 1.0, 0.5, 1 / w, 1 - 1 / w])), float(rng.choice([0.0, 1.0, 0.5, 1 / h, 1 - 1 / h])))
            else:  # the fovea on a strip boundary
                g = (float(np.clip((256 * int(rng.integers(0, max(1, w // 256) + 1)) + int(rng.integers(-2, 3))) / w, -15, 15)),
                     float(rng.uniform(0, 1)))
            gazes.append(g)
        dec = f360.SATDecoder(ctx)
        dec.InitializeGrid(rw, rh, w, h)
        fr = rng.integers(0, 256, (n, h, 4 * w), dtype=np.uint8)
        if rng.integers(0, 8) == 0:
            fr[0] = 255
        srcs = [ctx.upload(fr[k].reshape(-1)) for k in range(n)]
        sats_a = [ctx.malloc(w * h * 12) for _ in range(n)]
        sats_b = [ctx.malloc(w * h * 12) for _ in range(n)]
        reds_a = [ctx.malloc(rh * tl) for _ in range(n)]
        reds_b = [ctx.malloc(rh * tl) for _ in range(n)]
        fill = int(rng.integers(0, 256))
        for b in reds_a + reds_b:
            b.fill(fill)
        for b in sats_b:
            b.fill(0xEE)
        planar = h % 2 == 0 and rng.integers(0, 3) == 0  # a third of the cases from planes
        if planar:
            planar_cases += 1
            model = int(rng.integers(0, 2))
            ctx.set_option("yuv.model", model)
            cw = w // 2
            pl = [tuple(ctx.upload(rng.integers(0, 256, shp, dtype=np.uint8))
                        for shp in ((h, w), (h // 2, cw), (h // 2, cw))) for _ in range(n)]
            ptrs = [(a.ptr, b.ptr, c.ptr) for (a, b, c) in pl]
            enc.EncodeFramesYUV420PGPU([b.ptr for b in sats_a], ptrs, w, cw, cw, w, h)
            dec.SampleFramesRectGPU([b.ptr for b in reds_a], rw, rh, tl, [b.ptr for b in sats_a], (w, h), gazes)
            dec.EncodeSampleFramesYUV420PGPU([b.ptr for b in reds_b], rw, rh, tl, [b.ptr for b in sats_b],
                                             ptrs, w, cw, cw, w, h, gazes)
            reds_c = [ctx.malloc(rh * tl) for _ in range(n)]
            for b in reds_c:
                b.fill(fill)
            dec.FoveateFramesRectYUV420PGPU([b.ptr for b in reds_c], rw, rh, tl, ptrs, w, cw, cw, w, h, gazes)
            for k in range(n):
                if not np.array_equal(reds_c[k].copy_to_host(np.uint8, (rh, tl)),
                                      reds_a[k].copy_to_host(np.uint8, (rh, tl))):
                    bad += 1
                    if len(worst) < 10:
                        worst.append(("no-table planar", w, h, n, k, gazes[k], tpad, model))
            srcs = srcs + [p for t in pl for p in t] + reds_c
        else:
            enc.EncodeFramesGPU([b.ptr for b in sats_a], [b.ptr for b in srcs], w, h, 4 * w)
            dec.SampleFramesRectGPU([b.ptr for b in reds_a], rw, rh, tl, [b.ptr for b in sats_a], (w, h), gazes)
            dec.EncodeSampleFramesGPU([b.ptr for b in reds_b], rw, rh, tl, [b.ptr for b in sats_b],
                                      [b.ptr for b in srcs], w, h, 4 * w, gazes)
            # ... and the reduced frames alone (no tables): the same bytes once more
            reds_c = [ctx.malloc(rh * tl) for _ in range(n)]
            for b in reds_c:
                b.fill(fill)
            dec.FoveateFramesRectGPU([b.ptr for b in reds_c], rw, rh, tl, [b.ptr for b in srcs], w, h,
                                     4 * w, gazes)
            for k in range(n):
                if not np.array_equal(reds_c[k].copy_to_host(np.uint8, (rh, tl)),
                                      reds_a[k].copy_to_host(np.uint8, (rh, tl))):
                    bad += 1
                    if len(worst) < 10:
                        worst.append(("no-table", w, h, n, k, gazes[k], tpad))
            srcs = srcs + reds_c
        for k in range(n):
            ra, rb = reds_a[k].copy_to_host(np.uint8, (rh, tl)), reds_b[k].copy_to_host(np.uint8, (rh, tl))
            ta, tb = sats_a[k].copy_to_host(np.uint32, (h, w, 3)), sats_b[k].copy_to_host(np.uint32, (h, w, 3))
            d = int((ra != rb).sum()) + int((ta != tb).sum())
            if d:
                bad += 1
                if len(worst) < 10:
                    worst.append((w, h, n, k, gazes[k], tpad, int((ra != rb).sum()), int((ta != tb).sum())))
        calls += 1
        frames_done += n
        if time.time() - last_note > 60:  # (a silent GPU job looks hung to the runner)
            last_note = time.time()
            print(f"... {calls} calls, {frames_done} frames, {bad} differing", file=sys.stderr, flush=True)
        for b in srcs + sats_a + sats_b + reds_a + reds_b:
            b.free()
        dec.close()
    rec = ctx.debug_walk_recoveries()
print({"calls": calls, "planar_calls": planar_cases, "frames": frames_done, "differing_frames": bad, "first_failures": worst or None,
       "handoff_recoveries": rec, "seconds": round(time.time() - t0, 1), "seed": seed})
sys.exit(1 if bad else 0)
