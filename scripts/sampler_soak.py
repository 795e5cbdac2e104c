"""Soak of the SAT sampler's batched tile streamer (SampleFramesRectGPU, the default options)
against the per-pixel kernel on single frames (sample.variant 0): random frame / reduced
geometries, batch sizes, target paddings and gazes (inside, on and outside the frame), byte for
byte including the untouched fourth byte and the skipped pixels.
    python scripts/sampler_soak.py [seconds] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import f360_amd as f360

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
enc = f360.SATEncoder(ctx)
t0 = time.time()
cases = frames_done = bad = 0
worst = None
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    if kind == 0:
        w, h = 4 * int(rng.integers(2, 200)), int(rng.integers(4, 400))
        rw, rh = int(rng.integers(2, w + 30)), int(rng.integers(2, h + 30))
    elif kind == 1:
        w, h = [(1920, 1080), (2560, 1440), (3840, 1920), (1280, 720)][rng.integers(0, 4)]
        rw, rh = 16 * -(-w // 29), 16 * -(-h // 29)
    else:
        w, h = 4 * int(rng.integers(300, 2200)), int(rng.integers(8, 120))
        rw, rh = int(rng.integers(16, 5000)), int(rng.integers(2, 100))
    count = int(rng.integers(1, 41))
    while count * w * h * 12 > 300e6:
        count = max(1, count // 2)
    src = torch.empty((count, h, 4 * w), dtype=torch.uint8, device=dev)
    src.random_(0, 256)
    sat = torch.empty((count, h, w, 3), dtype=torch.int32, device=dev)
    enc.EncodeFramesGPU([sat[k].data_ptr() for k in range(count)], [src[k].data_ptr() for k in range(count)], w, h, 4 * w)
    pad = 4 * int(rng.integers(0, 5))
    tls = 4 * rw + pad
    dec = f360.SATDecoder(ctx)
    try:
        dec.InitializeGrid(rw, rh, w, h)
    except f360.F360Error:
        dec.close()
        continue
    centers = [(float(rng.uniform(-0.2, 1.2)), float(rng.uniform(-0.2, 1.2))) if rng.integers(0, 4) == 0
               else (float(rng.uniform(0, 1)), float(rng.uniform(0, 1))) for _ in range(count)]
    if count > 2:
        centers[0], centers[1] = (0.0, 0.0), (1.0, 1.0)
    a = torch.full((count, rh, tls), 0xA5, dtype=torch.uint8, device=dev)
    b = torch.full((count, rh, tls), 0xA5, dtype=torch.uint8, device=dev)
    ctx.set_option("sample.variant", 2)
    dec.SampleFramesRectGPU([a[k].data_ptr() for k in range(count)], rw, rh, tls,
                            [sat[k].data_ptr() for k in range(count)], (w, h), centers)
    ctx.set_option("sample.variant", 0)
    for k in range(count):
        dec.SampleFrameRectGPU(b[k].data_ptr(), rw, rh, tls, sat[k].data_ptr(), (w, h), centers[k][0], centers[k][1])
    ctx.set_option("sample.variant", 2)
    ctx.finish()
    cases += 1
    frames_done += count
    if not torch.equal(a, b):
        bad += 1
        worst = worst or []
        if len(worst) < 12:
            worst.append((w, h, rw, rh, count, tls, int((a != b).sum())))
    dec.close()
    del src, sat, a, b
print({"cases": cases, "frames": frames_done, "bad_cases": bad, "first_failures": worst,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
