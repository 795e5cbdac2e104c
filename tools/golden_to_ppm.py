#!/usr/bin/env python3
"""Writes the image-shaped entries of tests/golden/small.npz (or any .npz of uint8 frames) as
binary PPM files for a visual check -- the fixture-side counterpart of the reference's PNG dumps
(src/save_frame.h:15).  RGB0 entries (rows of 4 * width bytes, or [h, w, 4]) lose their pad byte.

    python tools/golden_to_ppm.py [tests/golden/small.npz] [outdir]
"""
import os
import sys

import numpy as np


def write_ppm(path, rgb):
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb[:, :, :3], dtype=np.uint8).tobytes())


def as_image(a):
    """[h, w, 4] / [h, w, 3] as is; [h, 4 * w] RGB0 rows reshaped; anything else: None."""
    if a.dtype != np.uint8:
        return None
    if a.ndim == 3 and a.shape[2] in (3, 4):
        return a
    if a.ndim == 2 and a.shape[1] % 4 == 0 and a.shape[1] >= 8:
        return a.reshape(a.shape[0], a.shape[1] // 4, 4)
    return None


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "tests", "golden", "small.npz")
    out = sys.argv[2] if len(sys.argv) > 2 else "golden_ppm"
    os.makedirs(out, exist_ok=True)
    n = 0
    with np.load(src) as z:
        for key in z.files:
            img = as_image(z[key])
            if img is None:
                continue
            write_ppm(os.path.join(out, key + ".ppm"), img)
            n += 1
    print(f"{n} images written to {out}/")
    return 0


if __name__ == "__main__":
    sys.exit(main())
