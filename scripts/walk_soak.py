"""Soak of the read-once batched encoder against the three kernels: random frame geometries, batch
sizes, row paddings and encoder options (kernel variant, batches in flight, frames per launch);
every table compared in full on the host against numpy's own summed-area table of the frame
(mod 2^32), so both encoders are checked, not only against each other.
    python scripts/walk_soak.py [seconds] [seed]"""
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
t0 = time.time()
cases = frames_done = bad = planar_cases = 0
worst = None
ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
enc = f360.SATEncoder(ctx)
while time.time() - t0 < budget:
    kind = rng.integers(0, 3)
    if kind == 0:
        w, h = 4 * int(rng.integers(1, 300)), int(rng.integers(1, 300))
    elif kind == 1:
        w, h = 4 * int(rng.integers(200, 2100)), int(rng.integers(1, 70))
    else:
        w, h = 4 * int(rng.integers(1, 80)), int(rng.integers(300, 3000))
    count = int(rng.integers(1, 41))
    while count * w * h * 12 > 400e6:
        count = max(1, count // 2)
    pad = 16 * int(rng.integers(0, 3))
    if pad >= w:   # (bytes per pixel = linesize / width, as in the reference: keep it at 4)
        pad = 0
    linesize = 4 * w + pad
    src = torch.empty((count, h, linesize), dtype=torch.uint8, device=dev)
    src.random_(0, 256)
    if rng.integers(0, 8) == 0:
        src[int(rng.integers(0, count))].fill_(255)
    tabs = [torch.zeros((count, h, w, 3), dtype=torch.int32, device=dev) for _ in range(2)]
    # (now and then with hand-off waits one to a few polls long: strips drop off the chain and
    # finish alone -- the recovery path must give the same tables)
    opts = {"sat.walk_frames": int(rng.choice([0, 0, 1, 3, 8, 64])),
            "debug.walk_spin": int(rng.choice([0, 0, 0, 1, 7, 40]))}
    for k, v in opts.items():
        ctx.set_option(k, v)
    if rng.integers(0, 4) == 0 and w % 8 == 0 and h % 2 == 0:
        # planar YUV 4:2:0 source, either libswscale model: the read-once encoder against the
        # three kernels (the conversion itself has its own parity tests against the oracle)
        ctx.set_option("yuv.model", int(rng.integers(0, 2)))
        yl, cl = w + 16 * int(rng.integers(0, 2)), w // 2 + 16 * int(rng.integers(0, 2))
        ys = torch.empty((count, h, yl), dtype=torch.uint8, device=dev).random_(0, 256)
        us = torch.empty((count, h // 2, cl), dtype=torch.uint8, device=dev).random_(0, 256)
        vs = torch.empty((count, h // 2, cl), dtype=torch.uint8, device=dev).random_(0, 256)
        for walk in (1, 0):
            ctx.set_option("sat.walk", walk)
            enc.EncodeFramesYUV420PGPU([tabs[1 - walk][k].data_ptr() for k in range(count)],
                                       [(ys[k].data_ptr(), us[k].data_ptr(), vs[k].data_ptr()) for k in range(count)],
                                       yl, cl, cl, w, h)
        ctx.finish()
        cases += 1
        frames_done += count
        planar_cases += 1
        if not torch.equal(tabs[0], tabs[1]):
            bad += 1
            worst = worst or []
            if len(worst) < 12:
                worst.append(("yuv420p", w, h, count, yl, cl, opts))
        del src, tabs, ys, us, vs
        continue
    for walk in (1, 0):
        ctx.set_option("sat.walk", walk)
        enc.EncodeFramesGPU([tabs[1 - walk][k].data_ptr() for k in range(count)],
                            [src[k].data_ptr() for k in range(count)], w, h, linesize)
    ctx.finish()
    px = src[:, :, :4 * w].reshape(count, h, w, 4)[..., :3].to(torch.int64)
    want = px.cumsum(dim=2).cumsum(dim=1).to(torch.int32)  # wraps like uint32
    ok_walk = bool(torch.equal(tabs[0], want))
    ok_three = bool(torch.equal(tabs[1], want))
    cases += 1
    frames_done += count
    if not (ok_walk and ok_three):
        bad += 1
        worst = worst or []
        if len(worst) < 12:
            worst.append((w, h, count, linesize, opts, ok_walk, ok_three))
    del src, tabs, px, want
for k, v in (("sat.walk", -1), ("sat.walk_frames", 0), ("debug.walk_spin", 0)):
    ctx.set_option(k, v)
print({"cases": cases, "planar_cases": planar_cases, "frames": frames_done, "bad_cases": bad, "first_failures": worst,
       "seconds": round(time.time() - t0, 1)})
sys.exit(1 if bad else 0)
