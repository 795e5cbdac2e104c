#!/usr/bin/env python3
"""Derives profiles/pmc_traffic.json from rocprofv3 PMC passes (scripts/pmc_s4.sh / prof.sh
output directories) and stamps it with the hash of the kernel sources it was measured on:
bench.py reports `roofline.traffic` only while that hash matches the tree it runs from.

    python scripts/pmc_traffic.py <out.json> <pmc dir(s) with FETCH_SIZE> <pmc dir(s) with WRITE_SIZE> [yuv fetch, yuv write [band fetch, band write]]
    (a "dir" may be several directories joined with ':' -- the default command's passes and the
    one-frame-per-call passes; the default command's passes with and without --one-pass)
    python scripts/pmc_traffic.py --hash          # the hash of the current csrc/
"""
import csv
import glob
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    """sha256 over the HIP / C++ sources of the engine (file names and contents)."""
    root = os.path.join(REPO, "foveated-360-video_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(root, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def means(paths, counter):
    acc = {}
    files = []
    for path in paths.split(":"):
        files += glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            # keep the template arguments: the benchmark's after-run variants launch other
            # instantiations of the same kernels (emit mode, planar sources)
            name = (row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("f360::sat::", "")
                    .split("(")[0].replace("void ", ""))
            acc.setdefault(name.strip(), []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    if sys.argv[1:] == ["--hash"]:
        print(csrc_hash())
        return
    out, fetch_dir, write_dir = sys.argv[1:4]
    size = "7680x3840"
    fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    doc = {"_note": ("HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                     "(separate passes, KB units); FETCH_SIZE doubled for the kernels that read with "
                     "wide coalesced 16-byte loads, as MI355X_MICROARCH.md prescribes for gfx950 "
                     "(confirmed on the reducer: 57.7 MB reported for a 118 MB frame); WRITE_SIZE as "
                     "is.  bench.py uses it only while csrc_sha equals the hash of its own csrc/."),
           "csrc_sha": csrc_hash()}
    # (instantiation measured, key in the file, FETCH_SIZE factor, frames one launch covered):
    # the default command's kernels first -- the read-once encoder (a step's 64 frames in two
    # launches of 32), the batch tile streamer with 16 --, then the single-frame kernels (--frames-per-call 1 passes);
    # an entry is per FRAME (per-launch counters divided by the frames of the launch), bench.py
    # multiplies by the frames its own launches cover.  RGB0 source = 1, LDS-staged stores = 1,
    # ring of 3 slots, byte stores.
    for inst, kernel, dbl, fpl in (("sat_walk_kernel<1, 2, false>", "sat_walk_kernel", 2, 32),
                                   ("sample_rect_stream_batch_kernel<3>", "sample_rect_kernel", 2, 16),
                                   ("sat_write_kernel<1, 1>", "sat_write_kernel", 2, 1),
                                   ("sat_reduce_kernel<1>", "sat_reduce_kernel", 2, 1),
                                   ("sat_carry_kernel", "sat_carry_kernel", 2, 1),
                                   ("sample_rect_stream_kernel<3, false>", "sample_rect_kernel", 2, 1)):
        if inst in fetch and inst in write and kernel not in doc:
            doc[kernel] = {size: int(1024 * (dbl * fetch[inst] + write[inst]) / fpl),
                           "_fetch_kb": round(fetch[inst] / fpl, 1),
                           "_write_kb": round(write[inst] / fpl, 1),
                           "_instance": inst, "_frames_per_launch": fpl}
    if len(sys.argv) >= 6:
        yf, yw = means(sys.argv[4], "FETCH_SIZE"), means(sys.argv[5], "WRITE_SIZE")
        # the planar default command is one pass as well (its plan and fix-up kernels are the
        # RGB0 command's); x86 rounding model = 3
        if "sat_walk_kernel<3, 2, true>" in yf and "sat_walk_kernel<3, 2, true>" in yw:
            doc.setdefault("sat_walk_kernel", {})[size + ":yuv420p:one_pass"] = int(
                1024 * (2 * yf["sat_walk_kernel<3, 2, true>"] + yw["sat_walk_kernel<3, 2, true>"]) / 32)
            for kernel in ("walk_fuse_plan_kernel", "walk_fuse_fix_kernel"):
                if kernel + "<0>" in yf and kernel + "<0>" in yw:
                    doc.setdefault(kernel, {})[size + ":yuv420p:one_pass"] = int(
                        1024 * (yf[kernel + "<0>"] + yw[kernel + "<0>"]) / 32)
        # planar source, x86 rounding model = 3
        for inst, kernel, fpl in (("sat_walk_kernel<3, 2, false>", "sat_walk_kernel", 32),
                                  ("sat_write_kernel<3, 1>", "sat_write_kernel", 1),
                                  ("sat_reduce_kernel<3>", "sat_reduce_kernel", 1)):
            if inst in yf and inst in yw:
                doc.setdefault(kernel, {})[size + ":yuv420p"] = int(1024 * (2 * yf[inst] + yw[inst]) / fpl)
    # the default command's one-pass kernels (EncodeSampleFramesGPU): the strip walker with helper
    # waves, its row-plan kernel and the fix-up of the boxes that straddle two strips
    for inst, kernel, dbl in (("sat_walk_kernel<1, 2, true>", "sat_walk_kernel", 2),
                              ("walk_fuse_plan_kernel<0>", "walk_fuse_plan_kernel", 1),
                              ("walk_fuse_fix_kernel<0>", "walk_fuse_fix_kernel", 1)):
        if inst in fetch and inst in write:
            e = doc.setdefault(kernel, {})
            e[size + ":one_pass"] = int(1024 * (dbl * fetch[inst] + write[inst]) / 32)
            e["_one_pass_fetch_kb"] = round(fetch[inst] / 32, 1)
            e["_one_pass_write_kb"] = round(write[inst] / 32, 1)
    # the band writer's one pass (8 frames per call, one frame per launch of the three kernels; the
    # plan kernel covers the call's 8 frames; the fix-up kernel is shared with the strip walker's
    # one pass and is told apart by the frames of its launch -- see `means`)
    if len(sys.argv) >= 8:
        bf, bw = means(sys.argv[6], "FETCH_SIZE"), means(sys.argv[7], "WRITE_SIZE")
        for inst, kernel, dbl, fpl in (("sat_write_fuse_kernel<1>", "sat_write_fuse_kernel", 2, 1),
                                       ("band_fuse_plan_kernel", "walk_fuse_plan_kernel", 1, 8),
                                       ("walk_fuse_fix_kernel<0>", "walk_fuse_fix_kernel", 1, 8)):
            if inst in bf and inst in bw:
                e = doc.setdefault(kernel, {})
                e[size + ":band_one_pass"] = int(1024 * (dbl * bf[inst] + bw[inst]) / fpl)
                e["_band_fetch_kb"] = round(bf[inst] / fpl, 1)
                e["_band_write_kb"] = round(bw[inst] / fpl, 1)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
