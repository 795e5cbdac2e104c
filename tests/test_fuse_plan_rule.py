"""CPU tier: the row-plan RULE of the one-pass encode + sample (sat_fuse.hip:
walk_fuse_plan_kernel), restated and checked by simulation.

A strip owner keeps ONE snapshot of its table row.  The plan kernel marks reduced row j as emitted
by the walk -- EMIT at table row hi(j), SNAP at lo(j) -- iff, among the rows j+1..j+3, no processed
row has its lo strictly inside (lo(j), hi(j)), and among j-1..j-3 no processed row has the same hi.
The claim behind the window of three: the grid offsets increase strictly, so lo and hi are
non-decreasing in j and only the clamps at the frame's top and bottom edge can break
lo(j+1) = hi(j).  This test does not trust the claim: for many geometries and a dense sweep of
gaze rows it builds the plan with the rule, then WALKS the table rows the way an owner does
(emit first, then snapshot) and requires that every emitted row finds exactly the snapshot of its own
lo, that no table row is asked to emit two reduced rows, and that what is left over is a handful.
The boxes come from the oracle's grid through a restatement of sample_axis (csrc/fov_maps.h)."""
import numpy as np
import pytest


def _axis(c, d_hi, d_lo, size):
    hi, lo = c + d_hi, c + d_lo
    ok = (0 <= hi < size) or (0 <= lo < size)
    hi = min(max(hi, 1), size - 1)
    lo = min(max(lo, 0), hi - 1)
    return hi, lo, ok


def _plan(gy, cyp, h, band_rows=0):
    rh = len(gy) - 1
    box = [_axis(cyp, int(gy[j + 1]), int(gy[j]), h) for j in range(rh)]
    emit, snap, left = {}, set(), []
    for j, (hi, lo, ok) in enumerate(box):
        if not ok:
            continue
        fused = True
        for d in (1, 2, 3):
            if j + d < rh:
                nh, nl, nok = box[j + d]
                if nok and lo < nl < hi:
                    fused = False
            if j - d >= 0:
                ph, pl, pok = box[j - d]
                if pok and ph == hi:
                    fused = False
        # the band writer's one pass (walk_fuse_plan_kernel<true>): an owner holds one band and
        # starts with the row above it as its snapshot, so all of a box's rows lie in one band
        if band_rows and (lo + 1) // band_rows != hi // band_rows:
            fused = False
        if fused:
            assert hi not in emit, ("two reduced rows emit at one table row", j, emit[hi], hi)
            emit[hi] = j
            snap.add(lo)
        else:
            left.append(j)
    return box, emit, snap, left


@pytest.mark.parametrize("w,h", [(7680, 3840), (3840, 1920), (1920, 1080), (1336, 203), (520, 66),
                                 (256, 128), (64, 9), (4096, 17)])
def test_every_emitted_row_finds_the_snapshot_of_its_own_lo(oracle, w, h):
    import f360_amd as f360
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = np.asarray(oracle.satdec_grid(rw, rh, w, h)).reshape(rh + 1, rw + 1, 2)
    gy = grid[:, 1, 1].astype(np.int64)
    assert (np.diff(gy) > 0).all() and np.diff(gy).max() <= 1023   # what the host checks
    centres = sorted(set(list(range(-h - 3, 2 * h + 4, max(1, h // 97))) +
                         [-h, -h + 1, -1, 0, 1, 2, h // 2, h - 2, h - 1, h, h + 1, 2 * h - 1, 2 * h]))
    most_left = 0
    for cyp in centres:
        box, emit, snap, left = _plan(gy, cyp, h)
        processed = [j for j, b in enumerate(box) if b[2]]
        assert sorted(list(emit.values()) + left) == processed
        most_left = max(most_left, len(left))
        snapshot_row = None      # an owner starts with zeros: "the row above the frame"
        for y in range(h):
            if y in emit:
                hi, lo, _ = box[emit[y]]
                assert hi == y and snapshot_row == lo, (w, h, cyp, emit[y], hi, lo, snapshot_row)
            if y in snap:
                snapshot_row = y
    assert most_left <= 4, most_left


@pytest.mark.parametrize("w,h,band_rows", [(7680, 3840, 64), (3840, 1920, 16), (1920, 1080, 16),
                                           (1336, 203, 16), (520, 66, 32), (256, 128, 64),
                                           (4096, 17, 16), (7680, 3840, 32)])
def test_band_owners_find_their_snapshots(oracle, w, h, band_rows):
    """The band form of the rule (sat_band_fuse.hip): every tile walks ITS band's rows with the
    row above the band as its first snapshot; every emitted row must find the snapshot of its own
    lo, and what is left over is at most one row per band boundary plus the clamped edge rows."""
    import f360_amd as f360
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = np.asarray(oracle.satdec_grid(rw, rh, w, h)).reshape(rh + 1, rw + 1, 2)
    gy = grid[:, 1, 1].astype(np.int64)
    nbands = (h + band_rows - 1) // band_rows
    centres = sorted(set(list(range(-h - 3, 2 * h + 4, max(1, h // 61))) +
                         [-1, 0, 1, h // 2, h - 1, h, band_rows - 1, band_rows, band_rows + 1]))
    most_left = 0
    for cyp in centres:
        box, emit, snap, left = _plan(gy, cyp, h, band_rows)
        processed = [j for j, b in enumerate(box) if b[2]]
        assert sorted(list(emit.values()) + left) == processed
        most_left = max(most_left, len(left))
        for band in range(nbands):
            y0, y1 = band * band_rows, min((band + 1) * band_rows, h)
            snapshot_row = y0 - 1   # the writer's prologue: the table row above the band
            for y in range(y0, y1):
                if y in emit:
                    hi, lo, _ = box[emit[y]]
                    assert hi == y and snapshot_row == lo, (w, h, cyp, band, emit[y], hi, lo, snapshot_row)
                if y in snap:
                    snapshot_row = y
    assert most_left <= nbands + 3, (most_left, nbands)
