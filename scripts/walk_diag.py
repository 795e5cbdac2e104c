"""One batched-encode case in detail: both encoders against numpy, first mismatches.
    python scripts/walk_diag.py w h count linesize [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import f360_amd as f360
w, h, count, linesize = (int(v) for v in sys.argv[1:5])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda", 0)
ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
enc = f360.SATEncoder(ctx)
g = torch.Generator(device=dev)
g.manual_seed(seed)
src = torch.randint(0, 256, (count, h, linesize), dtype=torch.uint8, device=dev, generator=g)
host = src.cpu().numpy()
want = host[:, :, :4 * w].reshape(count, h, w, 4)[..., :3].astype(np.uint64).cumsum(axis=2).cumsum(axis=1).astype(np.uint32)
for walk in (1, 0):
    ctx.set_option("sat.walk", walk)
    for rep in range(3):
        tab = torch.zeros((count, h, w, 3), dtype=torch.int32, device=dev)
        enc.EncodeFramesGPU([tab[k].data_ptr() for k in range(count)], [src[k].data_ptr() for k in range(count)], w, h, linesize)
        ctx.finish()
        got = tab.cpu().numpy().view(np.uint32)
        bad = np.argwhere(got != want)
        print(f"sat.walk={walk} rep {rep}: {len(bad)} differing entries of {got.size}", "first:", bad[:3].tolist() if len(bad) else "",
              [(int(got[tuple(b)]), int(want[tuple(b)])) for b in bad[:3]])
        if len(bad):
            fr = np.unique(bad[:, 0]); rows = np.unique(bad[:, 1])
            print("   frames", fr[:10].tolist(), "rows", rows[:5].tolist(), "...", rows[-3:].tolist(), "n rows", len(rows))
