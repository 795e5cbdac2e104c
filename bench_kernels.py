#!/usr/bin/env python3
"""bench_kernels.py -- every kernel of the engine on its own (HIP events on the launch stream),
with the algorithmic bytes DESIGN.md section 4 assigns to it and the fraction of 8 TB/s reached.

    python bench_kernels.py [--width 7680 --height 3840] [--reps 20]
"""
import argparse
import json
import math
import os
import sys

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import numpy as np  # noqa: E402


def reduced(n):
    return 16 * math.ceil(n / 1.8 / 16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=3840)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--opt", action="append", default=[], help="key=value engine option")
    args = ap.parse_args()
    import f360_amd as f360
    w, h = args.width, args.height
    rw, rh = reduced(w), reduced(h)
    rng = np.random.default_rng(1)
    with f360.Context(0) as ctx:
        for kv in args.opt:
            key, val = kv.split("=")
            ctx.set_option(key, int(val))
        enc, dec = f360.SATEncoder(ctx), f360.SATDecoder(ctx)
        smp, proj = f360.ImageSampler(ctx), f360.Projections(ctx)
        dec.InitializeGrid(rw, rh, w, h)
        smp.InitializeGrid(rw, rh, w, h)
        smp.InitializeLogpolarGrid(rw, rh, w, h)
        nbuf = 4  # rotate inputs so nothing stays cache-resident between repetitions
        frames = [ctx.upload(rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)) for _ in range(nbuf)]
        sat, full = ctx.malloc(w * h * 12), ctx.malloc(w * h * 4)
        red, red2 = ctx.malloc(rw * rh * 4), ctx.malloc(rw * rh * 4)
        view = ctx.malloc((w // 2) * (h // 2) * 4)
        enc.EncodeFrameGPU(sat.ptr, frames[0].ptr, w, h, 4 * w)
        dec.SampleFrameRectGPU(red.ptr, rw, rh, 4 * rw, sat.ptr, (w, h), 0.5, 0.5)
        ys = [ctx.upload(rng.integers(0, 256, (h, w), dtype=np.uint8)) for _ in range(nbuf)]
        us_ = [ctx.upload(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)) for _ in range(nbuf)]
        vs = [ctx.upload(rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)) for _ in range(nbuf)]
        yuv_args = lambda k: (ys[k % nbuf].ptr, us_[k % nbuf].ptr, vs[k % nbuf].ptr, w, w // 2, w // 2)
        cases = [
            ("sat_encode (3 kernels)", 16 * w * h,
             lambda k: enc.EncodeFrameGPU(sat.ptr, frames[k % nbuf].ptr, w, h, 4 * w)),
            ("yuv420p_to_rgb0", w * h * 3 // 2 + 4 * w * h,
             lambda k: ctx.yuv420p_to_rgb0(full.ptr, 4 * w, *yuv_args(k), w, h)),
            ("sat_encode from yuv420p (3 kernels)", w * h * 3 // 2 + 12 * w * h,
             lambda k: enc.EncodeFrameYUV420PGPU(sat.ptr, *yuv_args(k), w, h)),
            ("foveate_rect (fused encode+sample, 5 kernels)",
             4 * w * h + 4 * rw * rh,
             lambda k: dec.FoveateFrameRectGPU(red.ptr, rw, rh, 4 * rw, frames[k % nbuf].ptr, w, h, 4 * w,
                                               0.4 + 0.01 * k, 0.5)),
            ("foveate_rect from yuv420p", w * h * 3 // 2 + 4 * rw * rh,
             lambda k: dec.FoveateFrameRectYUV420PGPU(red.ptr, rw, rh, 4 * rw, *yuv_args(k), w, h,
                                                      0.4 + 0.01 * k, 0.5)),
            ("rgb0_to_yuv420p (reduced frame)", 4 * rw * rh + rw * rh * 3 // 2,
             lambda k: ctx.rgb0_to_yuv420p(oy.ptr, ou.ptr, ov.ptr, rw, rw // 2, rw // 2, red.ptr,
                                           4 * rw, rw, rh)),
            ("rgb0_to_yuv420p (full frame)", 4 * w * h + w * h * 3 // 2,
             lambda k: ctx.rgb0_to_yuv420p(fy.ptr, fu.ptr, fv.ptr, w, w // 2, w // 2,
                                           frames[k % nbuf].ptr, 4 * w, w, h)),
            ("sample_rect (SAT)", 12 * (rw + 1) * (rh + 1) + 4 * rw * rh,
             lambda k: dec.SampleFrameRectGPU(red.ptr, rw, rh, 4 * rw, sat.ptr, (w, h), 0.4 + 0.01 * k, 0.5)),
            ("sample_rect, 8 gaze points against one table (per gaze)", 12 * (rw + 1) * (rh + 1) + 4 * rw * rh,
             lambda k: dec.SampleFrameRectGPUBatch([r.ptr for r in reds8], rw, rh, 4 * rw, sat.ptr, (w, h),
                                                   [(0.3 + 0.05 * c + 0.01 * k, 0.5) for c in range(8)])),
            ("interpolate_rect", 4 * rw * rh + 4 * w * h,
             lambda k: dec.InterpolateFrameRectGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.4 + 0.01 * k, 0.5)),
            ("decode (SAT -> RGB0)", 12 * w * h + 4 * w * h,
             lambda k: dec.DecodeFrameGPU(full.ptr, 4 * w, sat.ptr, w, h)),
            ("is_sample_rect (point)", 8 * rw * rh,
             lambda k: smp.SampleFrameRectGPU(red2.ptr, rw, rh, 4 * rw, frames[k % nbuf].ptr, w, h, 4 * w, 0.5, 0.5)),
            ("is_sample_logpolar", 8 * rw * rh,
             lambda k: smp.SampleFrameLogPolarGPU(red2.ptr, rw, rh, 4 * rw, frames[k % nbuf].ptr, w, h, 4 * w, 0.5, 0.5)),
            ("is_interpolate_logpolar", 4 * rw * rh + 4 * w * h,
             lambda k: smp.InterpolateFrameLogPolarGPU(full.ptr, w, h, 4 * w, red2.ptr, rw, rh, 4 * rw, 0.5, 0.5)),
            ("is_blur", 8 * rw * rh,
             lambda k: smp.ApplyLogPolarGaussianBlur(red.ptr, rw, rh, 4 * rw, red2.ptr)),
            ("gnomonic (to w/2 x h/2)", 8 * (w // 2) * (h // 2),
             lambda k: proj.GnomonicProjection(view.ptr, w // 2, h // 2, 2 * w, frames[k % nbuf].ptr, w, h, 4 * w, 0.5, 0.5)),
        ]
        # the read-once batched encoder: enough frames to fill the device (32 at 8K), per frame
        nwalk = max(32, -(-960 // ((w + 255) // 256))) if w * h >= 3840 * 1920 else 0
        if nwalk:
            wsats = [ctx.malloc(12 * w * h) for _ in range(nwalk)]
            wsrc = [frames[k % nbuf].ptr for k in range(nwalk)]
            cases.insert(1, (f"sat_encode, {nwalk} frames per call (read-once sat_walk_kernel; per frame)",
                             16 * w * h,
                             lambda k: enc.EncodeFramesGPU([s.ptr for s in wsats], wsrc, w, h, 4 * w)))
        reds8 = [ctx.malloc(4 * rw * rh) for _ in range(8)]
        oy, ou, ov = ctx.malloc(rw * rh), ctx.malloc(rw * rh // 4), ctx.malloc(rw * rh // 4)
        fy, fu, fv = ctx.malloc(w * h), ctx.malloc(w * h // 4), ctx.malloc(w * h // 4)
        e0, e1 = f360.Event(ctx), f360.Event(ctx)
        out = []
        for name, nbytes, fn in cases:
            fn(0)
            ctx.finish()
            e0.record()
            for k in range(args.reps):
                fn(k)
            e1.record()
            us = 1e3 * e0.elapsed_ms(e1) / args.reps
            if "(per gaze)" in name:
                us /= 8
            if "per frame)" in name:
                us /= nwalk
            out.append({"kernel": name, "us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 1),
                        "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4)})
        print(json.dumps({"frame": [w, h], "reduced": [rw, rh], "reps": args.reps, "kernels": out}))
        dec.close()
        smp.close()


if __name__ == "__main__":
    main()
