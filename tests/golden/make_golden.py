#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle.

PARITY UNPINNED: the reference ships no tests or fixtures for this path and cannot be built or
run here (see oracle/f360_oracle.h), so these vectors were produced by the oracle itself.  They
freeze the oracle's behaviour (any later change to oracle/ or to the host tables shows up as a
golden mismatch) and they are what the GPU parity tests are compared with on the GPU box, where
/root/reference does not exist.

Inputs follow SURVEY.md 8(c)/(d): LCG frames (seed 12345), the gaze constants of the
reference's own call sites (run_satlogrectilinear.cc:88-89,179-180,264-265) plus edge gazes.

    python tests/golden/make_golden.py          # rewrites small.npz and digests.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

GAZES = [(0.0, 0.0), (0.5, 0.5), (0.65, 0.75), (0.0, 1.0), (1.0, 1.0), (0.999, 0.5)]
SEED = 12345


def reduced(n):
    import math
    return 16 * math.ceil(n / 1.8 / 16)


def case_outputs(w, h, seed=SEED, gazes=GAZES, with_nonseparable=True):
    """All oracle outputs for one frame size, as a dict of arrays."""
    rw, rh = reduced(w), reduced(h)
    frame = ob.lcg_frame(w, h, seed)
    out = {"frame_digest": np.uint64(ob.fnv1a64(frame))}
    sat = ob.sat_encode(frame, w, h, 4 * w)
    out["sat"] = sat
    gx, gy = ob.satdec_grid_axes(rw, rh, w, h)
    out["satdec_gx"], out["satdec_gy"] = gx, gy
    grid = ob.satdec_grid(rw, rh, w, h)
    isg = ob.is_grid(rw, rh, w, h)
    out["is_gx"], out["is_gy"] = isg[0, :, 0].copy(), isg[:, 0, 1].copy()
    lpg = ob.is_logpolar_grid(rw, rh, w, h)
    out["logpolar_grid"] = lpg
    for k, (cx, cy) in enumerate(gazes):
        red = np.full((rh, rw * 4), 0xA5, dtype=np.uint8)
        ob.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, cx, cy)
        out[f"sample_rect_{k}"] = red
        out[f"interp_rect_{k}"] = ob.satdec_interpolate_rect(red, w, h, rw, rh, cx, cy)
        pt = np.full((rh, rw * 4), 0x5A, dtype=np.uint8)
        ob.is_sample_rect(pt, rw, rh, 4 * rw, frame, w, h, 4 * w, isg, cx, cy)
        out[f"is_sample_rect_{k}"] = pt
        lp = np.full((rh, rw * 4), 0x3C, dtype=np.uint8)
        ob.is_sample_logpolar(lp, rw, rh, 4 * rw, frame, w, h, 4 * w, lpg, cx, cy)
        out[f"sample_logpolar_{k}"] = lp
        if with_nonseparable:
            out[f"interp_logpolar_{k}"] = ob.is_interpolate_logpolar(lp, w, h, rw, rh, cx, cy)
            out[f"gnomonic_{k}"] = ob.gnomonic(frame, w // 2, h, w, h, cx, cy)
    dec = np.full((h, w * 4), 0x11, dtype=np.uint8)
    out["decode"] = ob.satdec_decode(dec, 4 * w, sat, w, h)
    out["blur"] = ob.is_logpolar_blur(out["sample_logpolar_1"], rw, rh)
    return out


def digests(d):
    return {k: f"{ob.fnv1a64(np.ascontiguousarray(v)):016x}" for k, v in d.items()
            if isinstance(v, np.ndarray) and v.ndim > 0}


def main():
    ob.set_float_model(0)
    small = case_outputs(64, 32)
    np.savez_compressed(os.path.join(HERE, "small.npz"), **small)
    table = {"64x32": digests(small), "256x128": digests(case_outputs(256, 128))}
    # full benchmark sizes: digests only (SAT + sampler + separable tables), 3 gazes
    for (w, h) in [(1920, 1080), (3840, 1920), (7680, 3840)]:
        rw, rh = reduced(w), reduced(h)
        frame = ob.lcg_frame(w, h, SEED)
        sat = ob.sat_encode(frame, w, h, 4 * w)
        gx, gy = ob.satdec_grid_axes(rw, rh, w, h)
        grid = ob.satdec_grid(rw, rh, w, h)
        ent = {"frame": f"{ob.fnv1a64(frame):016x}", "sat": f"{ob.fnv1a64(sat):016x}",
               "satdec_gx": f"{ob.fnv1a64(gx):016x}", "satdec_gy": f"{ob.fnv1a64(gy):016x}"}
        for k, (cx, cy) in enumerate(GAZES[:3]):
            red = np.full((rh, rw * 4), 0xA5, dtype=np.uint8)
            ob.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, cx, cy)
            ent[f"sample_rect_{k}"] = f"{ob.fnv1a64(red):016x}"
            ent[f"interp_rect_{k}"] = f"{ob.fnv1a64(ob.satdec_interpolate_rect(red, w, h, rw, rh, cx, cy)):016x}"
        if w == 7680:
            # planar source (the first 1.5*w*h bytes of the LCG stream as Y, U, V planes with
            # tight rows) through the x86 libswscale model, then encode + sample
            buf = frame.reshape(-1)
            y = buf[:w * h].reshape(h, w)
            u = buf[w * h:w * h + w * h // 4].reshape(h // 2, w // 2)
            v = buf[w * h + w * h // 4:w * h + w * h // 2].reshape(h // 2, w // 2)
            ysat = ob.sat_encode(ob.yuv420p_to_rgb0(y, u, v, w, h, ob.YUV_SWS_X86), w, h, 4 * w)
            ent["yuv_x86_sat"] = f"{ob.fnv1a64(ysat):016x}"
            for k, (cx, cy) in enumerate(GAZES[:3]):
                red = np.full((rh, rw * 4), 0xA5, dtype=np.uint8)
                ob.satdec_sample_rect(red, rw, rh, 4 * rw, ysat, w, h, grid, cx, cy)
                ent[f"yuv_x86_sample_rect_{k}"] = f"{ob.fnv1a64(red):016x}"
        # all-255 frame: the 8K table wraps mod 2^32 (255*7680*3840 > 2^32)
        white = np.full((h, 4 * w), 255, dtype=np.uint8)
        ent["sat_white"] = f"{ob.fnv1a64(ob.sat_encode(white, w, h, 4 * w)):016x}"
        table[f"{w}x{h}"] = ent
        print(f"{w}x{h} done", flush=True)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump({"seed": SEED, "gazes": GAZES, "digest": "fnv1a64", "cases": table}, f,
                  indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "small.npz"), os.path.join(HERE, "digests.json"))


if __name__ == "__main__":
    main()
