"""One geometry of the log-polar un-warp, a few launches (for rocprofv3 --pmc runs).
    python scripts/lp_one.py [width height [is.lp_lds]]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import f360_amd as f360
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 3840)
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rw, rh = 16 * math.ceil(w / 1.8 / 16), 16 * math.ceil(h / 1.8 / 16)
with f360.Context(0) as ctx:
    ctx.set_option("is.lp_lds", lds)
    smp = f360.ImageSampler(ctx)
    red = ctx.upload(np.random.default_rng(1).integers(0, 256, (rh, 4 * rw), dtype=np.uint8))
    full = ctx.malloc(w * h * 4)
    for k in range(6):
        smp.InterpolateFrameLogPolarGPU(full.ptr, w, h, 4 * w, red.ptr, rw, rh, 4 * rw, 0.3 + 0.05 * k, 0.5)
    ctx.finish()
    smp.close()
