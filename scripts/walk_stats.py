#!/usr/bin/env python3
"""Per-strip timing of one read-once encoder launch (debug.ablate bit 8): when each strip owner
started and ended, how many hand-off waits took the slow path.
    python scripts/walk_stats.py [--frames 64] [--opt key=value ...]"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=3840)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import numpy as np
    import torch
    import f360_amd as f360
    dev = torch.device("cuda", 0)
    w, h, n = args.width, args.height, args.frames
    frames = torch.empty((n, h, 4 * w), dtype=torch.uint8, device=dev)
    for k in range(n):
        frames[k].random_(0, 256)
    sats = [torch.empty((h, w, 3), dtype=torch.int32, device=dev) for _ in range(n)]
    ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_option("sat.walk", 1)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.set_option("debug.ablate", ctx.get_option("debug.ablate") | 256)
    enc = f360.SATEncoder(ctx)
    for _ in range(2):
        enc.EncodeFramesGPU([s.data_ptr() for s in sats], [frames[k].data_ptr() for k in range(n)], w, h, 4 * w)
    torch.cuda.synchronize()
    raw = ctx.debug_walk_stats(n * 64)
    cycles = (raw[:, 2] >> np.uint64(16)).astype(np.float64)
    raw[:, 2] &= np.uint64(0xffff)
    st = raw.astype(np.float64)
    ticks = st[:, 1] - st[:, 0]
    ok = (ticks > 0) & (cycles > 0)
    mhz = float(np.median(cycles[ok] / ticks[ok] * 100.0)) if ok.any() else None
    nstrips = (w + 255) // 256
    st = st[: n * nstrips].reshape(n, nstrips, 4)
    t0 = st[:, :, 0].min()
    start = (st[:, :, 0] - t0) / 100.0   # us
    end = (st[:, :, 1] - t0) / 100.0
    out = {"frames": n, "strips": nstrips, "launch_us": round(float(end.max()), 1),
           "start_us_by_strip_mean": [round(float(x), 1) for x in start.mean(axis=0)],
           "end_us_by_strip_mean": [round(float(x), 1) for x in end.mean(axis=0)],
           "end_us_min_max": [round(float(end.min()), 1), round(float(end.max()), 1)],
           "busy_us_by_strip_mean": [round(float(x), 1) for x in (end - start).mean(axis=0)],
           "slow_waits_by_strip_mean": [round(float(x), 1) for x in st[:, :, 2].mean(axis=0)],
           "polls_by_strip_mean": [round(float(x), 1) for x in st[:, :, 3].mean(axis=0)],
           "batches_per_strip": (h + 7) // 8, "shader_mhz_median": mhz}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
