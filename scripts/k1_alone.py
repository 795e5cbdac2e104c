"""Times sat_reduce_kernel by itself (debug.ablate bit 8: the encoder stops after K1)."""
import sys, time
sys.path.insert(0, ".")
import torch
import f360_amd as f360

dev = torch.device("cuda", 0)
w, h = 7680, 3840
ctx = f360.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
enc = f360.SATEncoder(ctx)
frames = torch.randint(0, 256, (8, h, w * 4), dtype=torch.uint8, device=dev)
ys = torch.randint(0, 256, (8, h, w), dtype=torch.uint8, device=dev)
us = torch.randint(0, 256, (8, h // 2, w // 2), dtype=torch.uint8, device=dev)
vs = torch.randint(0, 256, (8, h // 2, w // 2), dtype=torch.uint8, device=dev)
sat = torch.empty((h, w, 3), dtype=torch.int32, device=dev)
for abl, sb in ((8, 2), (8, 1), (8 | 1, 2), (8 | 2, 2), (8 | 3, 2)):
    ctx.set_option("debug.ablate", abl)
    ctx.set_option("sat.sb_bands", sb)
    for src in ("rgb0", "yuv420p"):
        def call(k):
            if src == "rgb0":
                enc.EncodeFrameGPU(sat.data_ptr(), frames[k].data_ptr(), w, h, 4 * w)
            else:
                enc.EncodeFrameYUV420PGPU(sat.data_ptr(), ys[k].data_ptr(), us[k].data_ptr(),
                                          vs[k].data_ptr(), w, w // 2, w // 2, w, h)
        for k in range(8):
            call(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for i in range(n):
            call(i % 8)
        torch.cuda.synchronize()
        print(f"ablate {abl:2d} sb_bands {sb} {src:8s} K1 alone: {(time.perf_counter() - t0) / n * 1e6:7.1f} us")
