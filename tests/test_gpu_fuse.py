"""GPU tier: encode + sample in one pass (f360_satdec_encode_sample_frames, sat_walk_kernel<.., true>)
against the CPU oracle.  The call must leave, byte for byte, what SATEncoder::EncodeFrameGPU
followed by SATDecoder::SampleFrameRectGPU leave: the whole table of every frame, and in every
reduced frame the three colour bytes of the processed pixels and nothing else (targets are
pre-filled).  A strip owner's helper wave emits the pixels whose box lies inside the strip from
differences of table rows the owner still holds; boxes that straddle two strips are put together
by a second kernel from the two strips' halves, and the reduced rows at the frame's clamped top
and bottom edge are sampled from the finished table -- so the gaze points below put the fovea on
strip boundaries, on the seam, on and beyond every edge."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAZES = [(0.5, 0.5), (0.0, 0.0), (1.0, 1.0), (0.65, 0.75), (0.999, 0.001), (0.25, 0.5),
         (0.5, 0.0), (0.5, 1.0), (-0.3, 0.4), (1.7, 0.5), (0.5, -0.8), (0.5, 2.2), (0.126, 0.874),
         (0.3333, 0.6667), (0.0039, 0.9961), (0.75, 0.25)]


@pytest.fixture
def walk_ctx(gpu_ctx):
    gpu_ctx.set_option("sat.walk", 1)
    yield gpu_ctx
    gpu_ctx.set_option("sat.walk", -1)
    gpu_ctx.set_option("fuse.walk", 1)


def _run(f360, ctx, oracle, w, h, gazes, seed=300, fill=0xA5, frames=None, tpad=0):
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    n = len(gazes)
    if frames is None:
        frames = [oracle.lcg_frame(w, h, seed + k) for k in range(n)]
    dec = f360.SATDecoder(ctx)
    dec.InitializeGrid(rw, rh, w, h)
    srcs = [ctx.upload(np.ascontiguousarray(f).reshape(-1)) for f in frames]
    sats = [ctx.malloc(w * h * 12) for _ in range(n)]
    tl = 4 * rw + tpad
    reds = [ctx.malloc(rh * tl) for _ in range(n)]
    for b in sats:
        b.fill(0xEE)
    for b in reds:
        b.fill(fill)
    dec.EncodeSampleFramesGPU([b.ptr for b in reds], rw, rh, tl, [b.ptr for b in sats],
                              [b.ptr for b in srcs], w, h, 4 * w, gazes)
    grid = oracle.satdec_grid(rw, rh, w, h)
    bad = []
    for k in range(n):
        want_sat = oracle.sat_encode(frames[k], w, h, 4 * w)
        if not np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want_sat):
            bad.append(("table", k))
        want = np.full((rh, tl), fill, dtype=np.uint8)
        oracle.satdec_sample_rect(want, rw, rh, tl, want_sat, w, h, grid, *gazes[k])
        got = reds[k].copy_to_host(np.uint8, (rh, tl))
        if not np.array_equal(got, want):
            rows = np.nonzero((got != want).any(axis=1))[0]
            cols = np.nonzero((got != want).any(axis=0))[0] // 4
            bad.append(("reduced", k, gazes[k], int((got != want).sum()), rows[:6].tolist(),
                        sorted(set(cols.tolist()))[:6]))
    for b in srcs + sats + reds:
        b.free()
    dec.close()
    return bad


@pytest.mark.parametrize("w,h", [(1024, 512), (1920, 1080), (1336, 203), (256, 128), (520, 66),
                                 (2048, 96)])
def test_encode_sample_matches_oracle(f360, walk_ctx, oracle, w, h):
    assert _run(f360, walk_ctx, oracle, w, h, GAZES) == []


def test_encode_sample_special_frames(f360, walk_ctx, oracle):
    """An all-255 frame (largest sums, boxes of 255 exactly), an all-zero one, padded targets."""
    w, h = 1536, 320
    frames = [np.full((h, 4 * w), 255, dtype=np.uint8), np.zeros((h, 4 * w), dtype=np.uint8),
              oracle.lcg_frame(w, h, 9)]
    assert _run(f360, walk_ctx, oracle, w, h, [(0.5, 0.5), (0.1, 0.9), (0.8, 0.3)], frames=frames,
                tpad=32, fill=0x3C) == []


def test_encode_sample_gaze_sweep(f360, walk_ctx, oracle):
    """A fine sweep of the gaze across two strip boundaries and both vertical edges."""
    w, h = 768, 160
    gazes = [(x / 96.0, y) for x in range(20, 76) for y in (0.02, 0.5, 0.97)]
    for k in range(0, len(gazes), 24):
        assert _run(f360, walk_ctx, oracle, w, h, gazes[k:k + 24], seed=k) == []


def test_encode_sample_falls_back_to_the_two_calls(f360, gpu_ctx, oracle):
    """Below the read-once encoder's frame count, and with the switch off, the call is the two
    calls -- same bytes."""
    assert _run(f360, gpu_ctx, oracle, 640, 320, GAZES[:3]) == []
    gpu_ctx.set_option("sat.walk", 1)
    gpu_ctx.set_option("fuse.walk", 0)
    try:
        assert _run(f360, gpu_ctx, oracle, 640, 320, GAZES[:3]) == []
    finally:
        gpu_ctx.set_option("sat.walk", -1)
        gpu_ctx.set_option("fuse.walk", 1)


def test_encode_sample_8k_against_the_two_calls(f360, gpu_ctx):
    """BASELINE's size: 23 frames of 7680x3840 (the smallest batch the read-once encoder takes by
    itself), Lissajous gaze; tables and reduced frames equal to what the two calls write."""
    w, h, n = 7680, 3840, 23
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    rng = np.random.default_rng(8)
    base = rng.integers(0, 256, (h, 4 * w), dtype=np.uint8)
    dec = f360.SATDecoder(gpu_ctx)
    dec.InitializeGrid(rw, rh, w, h)
    enc = f360.SATEncoder(gpu_ctx)
    srcs = []
    for k in range(n):
        f = np.roll(base, 977 * k, axis=1)
        if k == 5:
            f = np.full_like(base, 255)
        srcs.append(gpu_ctx.upload(f.reshape(-1)))
    gazes = [(0.5 + 0.45 * np.sin(2 * np.pi * k / 97), 0.5 + 0.35 * np.sin(2 * np.pi * k / 61))
             for k in range(n)]
    gazes[3], gazes[4] = (0.0, 0.0), (1.0, 1.0)
    sats = [gpu_ctx.malloc(w * h * 12) for _ in range(n)]
    reds_a = [gpu_ctx.malloc(rw * rh * 4) for _ in range(n)]
    reds_b = [gpu_ctx.malloc(rw * rh * 4) for _ in range(n)]
    for b in reds_a + reds_b:
        b.fill(0x5A)
    enc.EncodeFramesGPU([b.ptr for b in sats], [b.ptr for b in srcs], w, h, 4 * w)
    dec.SampleFramesRectGPU([b.ptr for b in reds_a], rw, rh, 4 * rw, [b.ptr for b in sats], (w, h),
                            gazes)
    want_sat_sums = []
    for k in (0, 5, 22):
        t = sats[k].copy_to_host(np.uint32, (h, w, 3))
        want_sat_sums.append((int(t.sum(dtype=np.uint64)), t[-1, -1].tolist(), t[1234, 4321].tolist()))
    for b in sats:
        b.fill(0)
    dec.EncodeSampleFramesGPU([b.ptr for b in reds_b], rw, rh, 4 * rw, [b.ptr for b in sats],
                              [b.ptr for b in srcs], w, h, 4 * w, gazes)
    assert gpu_ctx.debug_walk_recoveries() == 0
    for k in range(n):
        a = reds_a[k].copy_to_host(np.uint8, (rh, 4 * rw))
        b = reds_b[k].copy_to_host(np.uint8, (rh, 4 * rw))
        assert np.array_equal(a, b), f"reduced frame {k} (gaze {gazes[k]}) differs"
    for idx, k in enumerate((0, 5, 22)):
        t = sats[k].copy_to_host(np.uint32, (h, w, 3))
        assert (int(t.sum(dtype=np.uint64)), t[-1, -1].tolist(), t[1234, 4321].tolist()) == \
            want_sat_sums[idx], f"table {k} differs"
    for b in srcs + sats + reds_a + reds_b:
        b.free()
    dec.close()


def test_encode_sample_many_frames_three_launches(f360, walk_ctx, oracle):
    """140 frames of two strips: three launches of the one-pass kernel, each with its own row
    plans and side rows in the shared scratch."""
    rng = np.random.default_rng(3)
    gazes = [(float(rng.uniform(-0.2, 1.2)), float(rng.uniform(-0.2, 1.2))) for _ in range(140)]
    assert _run(f360, walk_ctx, oracle, 512, 64, gazes) == []


def test_encode_sample_recarved_and_mixed_with_plain_encodes(f360, walk_ctx, oracle):
    """The one-pass scratch (plans, side rows) grows with the geometry and is shared with the plain
    read-once encoder's hand-off buffers: small, large, small again, a plain batched encode in
    between."""
    from test_gpu_walk import _encode_batch_and_check
    assert _run(f360, walk_ctx, oracle, 520, 66, GAZES[:5]) == []
    assert _run(f360, walk_ctx, oracle, 2048, 96, GAZES[:9]) == []
    assert _encode_batch_and_check(f360, walk_ctx, oracle, 1024, 64, 6) == []
    assert _run(f360, walk_ctx, oracle, 520, 66, GAZES[5:12]) == []
    assert _run(f360, walk_ctx, oracle, 1024, 512, GAZES[:4], seed=77) == []


def test_encode_sample_survives_a_missing_hand_off(f360, walk_ctx, oracle):
    """A strip owner whose hand-off never comes finishes alone (test_gpu_walk.py); its helper wave
    must neither notice nor wait: same tables, same reduced frames, recoveries counted."""
    walk_ctx.set_option("debug.walk_mute", 2)
    walk_ctx.set_option("debug.walk_spin", 64)
    try:
        walk_ctx.debug_walk_recoveries()
        assert _run(f360, walk_ctx, oracle, 1024, 96, GAZES[:3]) == []
        assert walk_ctx.debug_walk_recoveries() >= 1
    finally:
        walk_ctx.set_option("debug.walk_mute", 0)
        walk_ctx.set_option("debug.walk_spin", 0)


def test_cpp_encode_sample_frames_example(f360, gpu_ctx, oracle):
    """SATDecoder::EncodeSampleFramesGPU through include/f360/sat_decoder.h: the example's digests
    of the tables / reduced frames equal the oracle's for the per-frame call sequence."""
    import json
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "examples", "run_satlogrectilinear_synth")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(repo, "examples")], check=True)
    w, h, n = 768, 192, 6
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    out = subprocess.run([exe, "encode_sample_frames", str(w), str(h), str(n)], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout.strip().splitlines()[-1])
    grid = oracle.satdec_grid(rw, rh, w, h)
    digest = 0
    for k in range(n):
        sat = oracle.sat_encode(oracle.lcg_frame(w, h, 12345 + k), w, h, 4 * w)
        if k == 0:
            assert got["sat"] == f"{oracle.fnv1a64(sat):016x}"
        red = np.full((rh, 4 * rw), 0xA5, dtype=np.uint8)
        cx = float(np.float32(0.25) + np.float32(0.5) * np.float32(k) / np.float32(n))
        oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, cx, 0.5)
        digest ^= (oracle.fnv1a64(red) + k) & 0xFFFFFFFFFFFFFFFF
    assert got["rect"] == f"{digest:016x}"


def test_one_pass_refuses_to_allocate_under_stream_capture(f360, gpu_ctx):
    """Like the plain read-once encoder: the first call of a geometry allocates (hand-off
    granules, row plans, side rows) and must say so instead of doing it inside a capture."""
    import torch
    dev = torch.device("cuda", 0)
    w, h, n = 1024, 128, 4
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        frames = torch.randint(0, 256, (n, h, 4 * w), dtype=torch.uint8, device=dev)
        sats = torch.zeros((n, h, w, 3), dtype=torch.int32, device=dev)
        reds = torch.zeros((n, rh, 4 * rw), dtype=torch.uint8, device=dev)
        want = torch.zeros_like(reds)
    stream.synchronize()
    ctx = f360.Context(0, stream=stream.cuda_stream)
    ctx.set_option("sat.walk", 1)
    dec = f360.SATDecoder(ctx)
    dec.InitializeGrid(rw, rh, w, h)
    gazes = [(0.3, 0.6)] * n
    args = (rw, rh, 4 * rw, [sats[k].data_ptr() for k in range(n)],
            [frames[k].data_ptr() for k in range(n)], w, h, 4 * w, gazes)
    g = torch.cuda.CUDAGraph()
    refused = False
    with torch.cuda.graph(g, stream=stream):
        try:
            dec.EncodeSampleFramesGPU([reds[k].data_ptr() for k in range(n)], *args)
        except f360.F360Error as e:
            refused = "captured" in str(e)
    assert refused
    stream.synchronize()
    # warmed up eagerly, the same call is capturable and replays to the same bytes
    dec.EncodeSampleFramesGPU([want[k].data_ptr() for k in range(n)], *args)
    ctx.finish()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=stream):
        dec.EncodeSampleFramesGPU([reds[k].data_ptr() for k in range(n)], *args)
    for _ in range(3):
        reds.zero_()
        g2.replay()
        stream.synchronize()
        assert torch.equal(reds, want)
    dec.close()
    ctx.close()


def test_config4_batch_through_the_one_pass_call(f360, oracle):
    """BASELINE config 4 the way bench.py runs it by default: 32 of its frames (31 LCG frames + the
    all-255 frame whose sums wrap mod 2^32) resident, ONE EncodeSampleFramesGPU call; every table
    and every reduced frame equals the ORACLE's by digest."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_configs
    res = bench_configs.config4_batched(f360, oracle, quick=True, one_pass=True)
    assert "EncodeSampleFramesGPU" in res["workload"] and "sat_walk_kernel" in res["workload"], res
    assert res["bad_frames"] == [], res


def test_two_one_pass_launches_share_the_device(f360, oracle):
    """Two contexts (two streams, two threads) run one-pass launches at the same time.  A workgroup
    of the one-pass kernel takes a CU's LDS for itself, so the two launches compete for CUs; a strip
    owner only ever waits for a unit whose workgroup already runs and a helper only for its own
    owner, so both launches drain and both are right -- tables and reduced frames."""
    import threading
    w, h, n = 2304, 200, 40   # 9 strips x 40 frames = 360 units = 90 workgroups per launch
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    results = {}

    def worker(tag, seed):
        rng = np.random.default_rng(seed)
        with f360.Context(0) as ctx:
            ctx.set_option("sat.walk", 1)
            dec = f360.SATDecoder(ctx)
            dec.InitializeGrid(rw, rh, w, h)
            frames = [oracle.lcg_frame(w, h, seed + k) for k in range(n)]
            gazes = [(float(rng.uniform(-0.1, 1.1)), float(rng.uniform(-0.1, 1.1))) for _ in range(n)]
            srcs = [ctx.upload(f) for f in frames]
            sats = [ctx.malloc(w * h * 12) for _ in range(n)]
            reds = [ctx.malloc(rw * rh * 4) for _ in range(n)]
            bad = []
            for rep in range(4):
                for b in sats:
                    b.fill(rep + 1)
                for b in reds:
                    b.fill(0x40 + rep)
                dec.EncodeSampleFramesGPU([b.ptr for b in reds], rw, rh, 4 * rw, [b.ptr for b in sats],
                                          [b.ptr for b in srcs], w, h, 4 * w, gazes)
                ctx.finish()
                for k in (0, n // 2, n - 1):
                    want_sat = oracle.sat_encode(frames[k], w, h, 4 * w)
                    want = np.full((rh, 4 * rw), 0x40 + rep, dtype=np.uint8)
                    oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, *gazes[k])
                    if not (np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want_sat) and
                            np.array_equal(reds[k].copy_to_host(np.uint8, (rh, 4 * rw)), want)):
                        bad.append((rep, k))
            for b in srcs + sats + reds:
                b.free()
            dec.close()
            results[tag] = (bad, ctx.debug_walk_recoveries())

    threads = [threading.Thread(target=worker, args=(t, 700 + 100 * t)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == {0: ([], 0), 1: ([], 0)}, results


@pytest.mark.parametrize("model,w,h,pad", [(1, 1024, 512, (0, 0, 0)), (0, 1024, 512, (0, 0, 0)),
                                           (1, 1336, 202, (8, 4, 2)), (1, 1920, 1080, (0, 0, 0)),
                                           (0, 520, 66, (4, 2, 6))])
def test_encode_sample_from_planes_matches_oracle(f360, walk_ctx, oracle, model, w, h, pad):
    """EncodeSampleFramesYUV420PGPU: the planar strip owners (conversion in registers, snapshot in
    LDS, no source-pixel rows) against the oracle: the table of the oracle-converted frame and its
    reduced frame at the gaze; both libswscale models, padded planes, ragged geometries."""
    rng = np.random.default_rng(2000 + w + h + model)
    cw = (w + 1) // 2
    gazes = GAZES[:10]
    n = len(gazes)
    planes = [(rng.integers(0, 256, (h, w + pad[0]), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw + pad[1]), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw + pad[2]), dtype=np.uint8)) for _ in range(n)]
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    walk_ctx.set_option("yuv.model", model)
    dec = f360.SATDecoder(walk_ctx)
    dec.InitializeGrid(rw, rh, w, h)
    dev = [tuple(walk_ctx.upload(p) for p in pl) for pl in planes]
    sats = [walk_ctx.malloc(w * h * 12) for _ in range(n)]
    reds = [walk_ctx.malloc(rw * rh * 4) for _ in range(n)]
    for b in sats:
        b.fill(0xEE)
    for b in reds:
        b.fill(0x77)
    y0, u0, v0 = planes[0]
    dec.EncodeSampleFramesYUV420PGPU([b.ptr for b in reds], rw, rh, 4 * rw, [b.ptr for b in sats],
                                     [(a.ptr, b.ptr, c.ptr) for (a, b, c) in dev], y0.shape[1],
                                     u0.shape[1], v0.shape[1], w, h, gazes)
    grid = oracle.satdec_grid(rw, rh, w, h)
    bad = []
    for k in range(n):
        y, u, v = planes[k]
        want_sat = oracle.sat_encode(oracle.yuv420p_to_rgb0(y, u, v, w, h, model), w, h, 4 * w)
        want = np.full((rh, 4 * rw), 0x77, dtype=np.uint8)
        oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, *gazes[k])
        if not np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want_sat):
            bad.append(("table", k))
        if not np.array_equal(reds[k].copy_to_host(np.uint8, (rh, 4 * rw)), want):
            bad.append(("reduced", k, gazes[k]))
    walk_ctx.set_option("yuv.model", 1)
    for b in sats + reds + [p for t in dev for p in t]:
        b.free()
    dec.close()
    assert bad == []


def _foveate_frames(f360, ctx, oracle, w, h, gazes, seed=400, fill=0x6B):
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    n = len(gazes)
    frames = [oracle.lcg_frame(w, h, seed + k) for k in range(n)]
    dec = f360.SATDecoder(ctx)
    dec.InitializeGrid(rw, rh, w, h)
    srcs = [ctx.upload(f) for f in frames]
    reds = [ctx.malloc(rw * rh * 4) for _ in range(n)]
    for b in reds:
        b.fill(fill)
    dec.FoveateFramesRectGPU([b.ptr for b in reds], rw, rh, 4 * rw, [b.ptr for b in srcs], w, h,
                             4 * w, gazes)
    grid = oracle.satdec_grid(rw, rh, w, h)
    bad = []
    for k in range(n):
        want = np.full((rh, 4 * rw), fill, dtype=np.uint8)
        oracle.satdec_sample_rect(want, rw, rh, 4 * rw, oracle.sat_encode(frames[k], w, h, 4 * w),
                                  w, h, grid, *gazes[k])
        got = reds[k].copy_to_host(np.uint8, (rh, 4 * rw))
        if not np.array_equal(got, want):
            rows = np.nonzero((got != want).any(axis=1))[0]
            bad.append((k, gazes[k], int((got != want).sum()), rows[:6].tolist()))
    for b in srcs + reds:
        b.free()
    dec.close()
    return bad


@pytest.mark.parametrize("w,h", [(1024, 512), (1336, 203), (520, 66), (2048, 96)])
def test_foveate_frames_without_tables_matches_oracle(f360, walk_ctx, oracle, w, h):
    """FoveateFramesRectGPU: the one-pass strip walker with its table stores off.  The reduced rows
    the walk cannot emit (boxes clamped at the frame's top and bottom edge) have no table to be
    sampled from and are summed from the source pixels instead -- the gaze points put them there."""
    assert _foveate_frames(f360, walk_ctx, oracle, w, h, GAZES) == []


def test_foveate_frames_few_frames_take_the_single_frame_call(f360, gpu_ctx, oracle):
    assert _foveate_frames(f360, gpu_ctx, oracle, 640, 320, GAZES[:4]) == []


@pytest.mark.parametrize("model,w,h", [(1, 1024, 512), (0, 1336, 202), (1, 520, 66)])
def test_foveate_frames_from_planes_without_tables(f360, walk_ctx, oracle, model, w, h):
    """FoveateFramesRectYUV420PGPU: planar strip owners, no tables; the leftover rows at the clamped
    edges are summed from pixels converted in the fix-up kernel with the selected libswscale model."""
    rng = np.random.default_rng(3000 + w + model)
    cw = (w + 1) // 2
    gazes = GAZES[:12]
    n = len(gazes)
    planes = [(rng.integers(0, 256, (h, w), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw), dtype=np.uint8)) for _ in range(n)]
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    walk_ctx.set_option("yuv.model", model)
    dec = f360.SATDecoder(walk_ctx)
    dec.InitializeGrid(rw, rh, w, h)
    dev = [tuple(walk_ctx.upload(p) for p in pl) for pl in planes]
    reds = [walk_ctx.malloc(rw * rh * 4) for _ in range(n)]
    for b in reds:
        b.fill(0x19)
    dec.FoveateFramesRectYUV420PGPU([b.ptr for b in reds], rw, rh, 4 * rw,
                                    [(a.ptr, b.ptr, c.ptr) for (a, b, c) in dev], w, cw, cw, w, h,
                                    gazes)
    grid = oracle.satdec_grid(rw, rh, w, h)
    bad = []
    for k in range(n):
        y, u, v = planes[k]
        sat = oracle.sat_encode(oracle.yuv420p_to_rgb0(y, u, v, w, h, model), w, h, 4 * w)
        want = np.full((rh, 4 * rw), 0x19, dtype=np.uint8)
        oracle.satdec_sample_rect(want, rw, rh, 4 * rw, sat, w, h, grid, *gazes[k])
        if not np.array_equal(reds[k].copy_to_host(np.uint8, (rh, 4 * rw)), want):
            bad.append((k, gazes[k]))
    walk_ctx.set_option("yuv.model", 1)
    for b in reds + [p for t in dev for p in t]:
        b.free()
    dec.close()
    assert bad == []
