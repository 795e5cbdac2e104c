"""Where the one-pass encode + sample spends its time, per strip position: one launch of 23 8K
frames with debug.ablate bit 8 (per-unit clocks and wait counts of the strip owners).
    python scripts/fuse_stats.py [frames]
(the helper's timed sections cost a few hundred cycles per row themselves: compare strips, do not
read the cycle counts as absolute)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import f360_amd as f360

n = int(sys.argv[1]) if len(sys.argv) > 1 else 23
w, h = 7680, 3840
rw, rh = f360.reduced_size(w), f360.reduced_size(h)
dev = torch.device("cuda", 0)
frames = torch.empty((n, h, 4 * w), dtype=torch.uint8, device=dev)
for k in range(n):
    frames[k].random_(0, 256)
sats = [torch.empty((h, w, 3), dtype=torch.int32, device=dev) for _ in range(n)]
reds = torch.zeros((n, rh, 4 * rw), dtype=torch.uint8, device=dev)
ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
dec = f360.SATDecoder(ctx)
dec.InitializeGrid(rw, rh, w, h)
# one gaze for every frame: strip positions mean the same thing in all of them
gazes = [(0.5, 0.5)] * n
args = ([reds[k].data_ptr() for k in range(n)], rw, rh, 4 * rw, [s.data_ptr() for s in sats],
        [frames[k].data_ptr() for k in range(n)], w, h, 4 * w, gazes)
for mode in (1, 0):
    ctx.set_option("fuse.walk", mode)
    ctx.set_option("debug.ablate", 0)
    dec.EncodeSampleFramesGPU(*args)
    ctx.set_option("debug.ablate", 256)
    if mode == 0:
        f360.SATEncoder(ctx).EncodeFramesGPU(args[4], args[5], w, h, 4 * w)
    else:
        dec.EncodeSampleFramesGPU(*args)
    ctx.finish()
    st = ctx.debug_walk_stats(4096)
    ns = 30
    u = len(st) // ns * ns
    st = st[:u].reshape(-1, ns, 8)
    dur = (st[:, :, 1] - st[:, :, 0]).astype(np.float64) / 100.0   # us
    slow = (st[:, :, 2] & 0xffff).astype(np.float64)
    hand = (st[:, :, 3] & 0xffffffff).astype(np.float64)
    box = (st[:, :, 3] >> 32).astype(np.float64)
    print("one pass" if mode else "encode only", "launch us", round(float((st[:, :, 1].max() - st[:, :, 0].min()) / 100.0), 1))
    hw, hk, hr, hn = (st[:, :, k].astype(np.float64) for k in (4, 5, 6, 7))
    print(" strip: walk us | slow hand-off waits | hand-off polls | slot polls | helper: boxes, rows, wait cycles per row, work cycles per row (means over frames)")
    for s_ in range(ns):
        rows = max(hr[:, s_].mean(), 1.0)
        print(f"  {s_:2d}: {dur[:, s_].mean():8.1f} {slow[:, s_].mean():7.1f} {hand[:, s_].mean():9.1f} {box[:, s_].mean():9.1f}"
              f" | {hn[:, s_].mean():6.0f} {hr[:, s_].mean():6.0f} {hw[:, s_].mean() / rows:8.0f} {hk[:, s_].mean() / rows:8.0f}")
dec.close()
