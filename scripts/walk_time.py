#!/usr/bin/env python3
"""A/B of the batched encoders inside one process: the three kernels (sat.walk=0) against the
read-once strip walker (sat.walk=1) for several frame counts and walker depths.
    python scripts/walk_time.py [--width 7680 --height 3840] [--counts 16,32,64] [--reps 5]"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=3840)
    ap.add_argument("--counts", default="16,32,64")
    ap.add_argument("--depths", default="2,3")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--opt", action="append", default=[])
    args = ap.parse_args()
    import torch
    import f360_amd as f360
    dev = torch.device("cuda", 0)
    w, h = args.width, args.height
    counts = [int(c) for c in args.counts.split(",")]
    nmax = max(counts)
    frames = torch.empty((nmax, h, 4 * w), dtype=torch.uint8, device=dev)
    for k in range(nmax):
        frames[k].random_(0, 256)
    sats = [torch.empty((h, w, 3), dtype=torch.int32, device=dev) for _ in range(nmax)]
    ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    enc = f360.SATEncoder(ctx)
    fp = [frames[k].data_ptr() for k in range(nmax)]
    sp = [s.data_ptr() for s in sats]

    def run(n):
        enc.EncodeFramesGPU(sp[:n], fp[:n], w, h, 4 * w)

    def timed(n):
        run(n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            run(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps / n * 1e6

    ctx.set_option("sat.walk", 0)
    run(min(counts))
    torch.cuda.synchronize()
    ref = [s.clone() for s in sats[:2]]
    for n in counts:
        ctx.set_option("sat.walk", 0)
        print(json.dumps({"encoder": "three kernels", "frames": n, "us_per_frame": round(timed(n), 2)}), flush=True)
        for d in (2,):  # (batches a strip owner rotates through: one value since round 4)
            ctx.set_option("sat.walk", 1)
            us = timed(n)
            same = all(torch.equal(ref[k], sats[k]) for k in range(2))
            enc_bytes = 16 * w * h
            print(json.dumps({"encoder": "walk", "depth": d, "frames": n, "us_per_frame": round(us, 2),
                              "frac_of_8TBs": round(enc_bytes / (us * 1e-6) / 8e12, 4),
                              "equal_to_three_kernels": same}), flush=True)
    ctx.finish()


if __name__ == "__main__":
    main()
