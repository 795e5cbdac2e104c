"""GPU tier: encode + sample in one pass for calls too small for the strip walker
(f360_satdec_encode_sample_frames -> sat_write_fuse_kernel, csrc/sat_band_fuse.hip) against the CPU
oracle: from ONE frame per call (the per-frame loop of the reference's offline tool,
run_satlogrectilinear.cc:926-938) to the 22 8K frames below the read-once encoder's threshold
(8 per rank when BASELINE config 4's 64 frames are sharded over 8 GPUs).

The call must leave, byte for byte, what SATEncoder::EncodeFrameGPU followed by
SATDecoder::SampleFrameRectGPU leave: the whole table of every frame, and in every reduced frame the
three colour bytes of the processed pixels and nothing else (targets are pre-filled).  A tile of the
table writer emits the pixels whose box lies inside its band and its strip; boxes that straddle two
strips come from the side rows, boxes that cross a band boundary from the finished table -- so the
gaze points below put the fovea on strip boundaries, on band boundaries, on the seam, on and beyond
every edge, and the options sweep the band height."""
import numpy as np
import pytest

from test_gpu_fuse import GAZES, _run

pytestmark = pytest.mark.gpu


def _ran_band_writer(ctx, call, calls=1):
    """Runs `call` as a sampled call and returns the kernels it launched."""
    ctx.profile_reset()
    ctx.profile_arm(calls)
    out = call()
    ctx.finish()
    return out, set(ctx.profile_read().keys())


@pytest.mark.parametrize("w,h", [(1024, 512), (1920, 1080), (1336, 203), (256, 128), (520, 66),
                                 (2048, 96), (3840, 1920)])
def test_band_one_pass_matches_oracle(f360, gpu_ctx, oracle, w, h):
    n = 16 if w * h < 4_000_000 else 6
    bad, kernels = _ran_band_writer(gpu_ctx, lambda: _run(f360, gpu_ctx, oracle, w, h, GAZES[:n]))
    assert bad == []
    assert "sat_write_fuse_kernel" in kernels and "sample_rect_kernel" not in kernels, kernels


def test_band_one_pass_one_frame_per_call(f360, gpu_ctx, oracle):
    """The offline tool's shape: one frame, one gaze, one call -- for every gaze of the list.  By
    default such a call is the two calls (faster for a single frame); "fuse.band" 2 sends it
    through the band writer, and both must leave the oracle's bytes."""
    bad, kernels = _ran_band_writer(gpu_ctx, lambda: _run(f360, gpu_ctx, oracle, 1920, 1080, [GAZES[3]]),
                                    calls=2)
    assert bad == [] and "sat_write_fuse_kernel" not in kernels and "sample_rect_kernel" in kernels
    gpu_ctx.set_option("fuse.band", 2)
    try:
        for k, g in enumerate(GAZES):
            bad, kernels = _ran_band_writer(gpu_ctx, lambda: _run(f360, gpu_ctx, oracle, 1920, 1080, [g],
                                                                  seed=40 + k))
            assert bad == [], g
            assert "sat_write_fuse_kernel" in kernels
    finally:
        gpu_ctx.set_option("fuse.band", 1)


def test_band_one_pass_special_frames(f360, gpu_ctx, oracle):
    """An all-255 frame (largest sums, boxes of 255 exactly), an all-zero one, padded targets."""
    w, h = 1536, 320
    frames = [np.full((h, 4 * w), 255, dtype=np.uint8), np.zeros((h, 4 * w), dtype=np.uint8),
              oracle.lcg_frame(w, h, 9), oracle.lcg_frame(w, h, 10)]
    bad, kernels = _ran_band_writer(gpu_ctx, lambda: _run(
        f360, gpu_ctx, oracle, w, h, [(0.5, 0.5), (0.1, 0.9), (0.8, 0.3), (0.3, 0.2)], frames=frames,
        tpad=32, fill=0x3C))
    assert bad == [] and "sat_write_fuse_kernel" in kernels


def test_band_one_pass_gaze_sweep(f360, gpu_ctx, oracle):
    """A fine sweep of the gaze across two strip boundaries, and of its row across band
    boundaries (16-row bands at this size) and both vertical edges."""
    w, h = 768, 160
    gazes = [(x / 96.0, y) for x in range(20, 76) for y in (0.02, 0.5, 0.97)]
    gazes += [(0.37, y / 160.0) for y in range(-3, 164)]
    for k in range(0, len(gazes), 16):
        assert _run(f360, gpu_ctx, oracle, w, h, gazes[k:k + 16], seed=k) == []


@pytest.mark.parametrize("band_rows", [16, 32, 64])
def test_band_one_pass_band_heights(f360, gpu_ctx, oracle, band_rows):
    gpu_ctx.set_option("sat.band_rows", band_rows)
    try:
        assert _run(f360, gpu_ctx, oracle, 1920, 1080, GAZES[:8]) == []
        assert _run(f360, gpu_ctx, oracle, 1336, 203, GAZES[8:]) == []
    finally:
        gpu_ctx.set_option("sat.band_rows", 0)


@pytest.mark.parametrize("force", [1, 2, 3])
def test_band_one_pass_rare_branches(f360, gpu_ctx, oracle, force):
    """debug.fuse_force: side rows of one pixel (the fix-up then takes every row whole) and no
    listed leftover rows (the fix-up finds the rows the plan did not mark by itself)."""
    gpu_ctx.set_option("debug.fuse_force", force)
    try:
        assert _run(f360, gpu_ctx, oracle, 1024, 512, GAZES[:8]) == []
        assert _run(f360, gpu_ctx, oracle, 520, 66, GAZES[8:]) == []
    finally:
        gpu_ctx.set_option("debug.fuse_force", 0)


@pytest.mark.parametrize("force", [1, 2, 3, 4, 7])
def test_walker_one_pass_rare_branches(f360, gpu_ctx, oracle, force):
    """The same forced branches through the strip walker's one pass (the whole-row
    fix-up, the unlisted-rows path and -- bit 2 -- the helpers' tail loop for strips that own more
    pixels than their registers hold were never exercised)."""
    gpu_ctx.set_option("debug.fuse_force", force)
    gpu_ctx.set_option("sat.walk", 1)
    try:
        assert _run(f360, gpu_ctx, oracle, 1024, 512, GAZES[:8]) == []
        assert _run(f360, gpu_ctx, oracle, 1336, 203, GAZES[8:]) == []
    finally:
        gpu_ctx.set_option("debug.fuse_force", 0)
        gpu_ctx.set_option("sat.walk", -1)


def test_band_switch_off_is_the_two_calls(f360, gpu_ctx, oracle):
    gpu_ctx.set_option("fuse.band", 0)
    try:
        bad, kernels = _ran_band_writer(gpu_ctx, lambda: _run(f360, gpu_ctx, oracle, 640, 320, GAZES[:5]),
                                        calls=2)  # (the encode call and the sample call)
        assert bad == []
        assert "sat_write_fuse_kernel" not in kernels and "sample_rect_kernel" in kernels
    finally:
        gpu_ctx.set_option("fuse.band", 1)


def test_band_one_pass_unaligned_sources_take_the_two_calls(f360, gpu_ctx, oracle):
    """RGB24 frames are not the band writer's: the call is the two calls, same bytes."""
    w, h = 640, 320
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    frame = oracle.lcg_frame(w, h, 5, bpp=3)
    dec = f360.SATDecoder(gpu_ctx)
    dec.InitializeGrid(rw, rh, w, h)
    src, sat, red = gpu_ctx.upload(frame), gpu_ctx.malloc(w * h * 12), gpu_ctx.malloc(rw * rh * 4)
    red.fill(0x11)
    dec.EncodeSampleFramesGPU([red.ptr], rw, rh, 4 * rw, [sat.ptr], [src.ptr], w, h, 3 * w,
                              [(0.4, 0.6)])
    want_sat = oracle.sat_encode(frame, w, h, 3 * w)
    want = np.full((rh, 4 * rw), 0x11, dtype=np.uint8)
    oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, oracle.satdec_grid(rw, rh, w, h),
                              0.4, 0.6)
    assert np.array_equal(sat.copy_to_host(np.uint32, (h, w, 3)), want_sat)
    assert np.array_equal(red.copy_to_host(np.uint8, (rh, 4 * rw)), want)
    for b in (src, sat, red):
        b.free()
    dec.close()


@pytest.mark.parametrize("n", [1, 8, 16, 22])
def test_band_one_pass_8k_against_the_oracle(f360, gpu_ctx, oracle, n):
    """BASELINE's size, the frame counts below the read-once encoder's threshold (23): every table
    and every reduced frame against the ORACLE, incl. an all-255 frame (sums wrap mod 2^32), gazes
    with the fovea's edge on strip boundaries (multiples of 256 columns) and on band boundaries
    (multiples of 64 rows), on the seam and in the corners."""
    w, h = 7680, 3840
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    special = [(2560 / w, 1920 / h), (0.0, 0.0), (1.0, 1.0), ((2560 + 1600) / w, (1920 + 800) / h),
               (1 / w, 64 / h), (0.999, 0.5), (256 * 11 / w, 64 * 17 / h), (0.5, 63 / h)]
    gazes = [special[k] if k < len(special) else
             (0.5 + 0.45 * np.sin(2 * np.pi * k / 97), 0.5 + 0.35 * np.sin(2 * np.pi * k / 61))
             for k in range(n)]
    dec = f360.SATDecoder(gpu_ctx)
    dec.InitializeGrid(rw, rh, w, h)
    frames, srcs = [], []
    for k in range(n):
        f = (np.full((h, 4 * w), 255, dtype=np.uint8) if k == min(2, n - 1) and n > 1
             else oracle.lcg_frame(w, h, 500 + k).reshape(h, 4 * w))
        frames.append(f)
        srcs.append(gpu_ctx.upload(f.reshape(-1)))
    sats = [gpu_ctx.malloc(w * h * 12) for _ in range(n)]
    reds = [gpu_ctx.malloc(rw * rh * 4) for _ in range(n)]
    for b in sats:
        b.fill(0xEE)
    for b in reds:
        b.fill(0x5A)
    gpu_ctx.profile_reset()
    gpu_ctx.profile_arm(1)
    gpu_ctx.set_option("fuse.band", 2)  # (n == 1 too)
    try:
        dec.EncodeSampleFramesGPU([b.ptr for b in reds], rw, rh, 4 * rw, [b.ptr for b in sats],
                                  [b.ptr for b in srcs], w, h, 4 * w, gazes)
        gpu_ctx.finish()
    finally:
        gpu_ctx.set_option("fuse.band", 1)
    assert "sat_write_fuse_kernel" in gpu_ctx.profile_read()
    for k in range(n):
        want_sat = oracle.sat_encode(frames[k].reshape(-1), w, h, 4 * w)
        got_sat = sats[k].copy_to_host(np.uint32, (h, w, 3))
        assert np.array_equal(got_sat, want_sat), f"table {k}"
        want = np.full((rh, 4 * rw), 0x5A, dtype=np.uint8)
        oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, *gazes[k])
        got = reds[k].copy_to_host(np.uint8, (rh, 4 * rw))
        if not np.array_equal(got, want):
            rows = np.nonzero((got != want).any(axis=1))[0]
            cols = np.nonzero((got != want).any(axis=0))[0] // 4
            raise AssertionError(f"reduced frame {k} (gaze {gazes[k]}): {int((got != want).sum())} bytes "
                                 f"differ, rows {rows[:8].tolist()}, columns {sorted(set(cols.tolist()))[:8]}")
        del want_sat, got_sat
    for b in srcs + sats + reds:
        b.free()
    dec.close()


def test_band_one_pass_replays_from_a_hip_graph(f360, oracle):
    """A pipelined band one-pass call uses the context's side stream between its first and its
    last kernel (fork after the plan kernel, join before the fix-up): captured into a graph it
    must refuse to allocate under the capture, and -- warmed up eagerly -- replay to the oracle's
    bytes, tables and reduced frames."""
    import torch
    dev = torch.device("cuda", 0)
    w, h, n = 1536, 256, 6
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    host = [oracle.lcg_frame(w, h, 70 + k) for k in range(n)]
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        frames = torch.from_numpy(np.stack(host)).to(dev)
        sats = torch.zeros((n, h, w, 3), dtype=torch.int32, device=dev)
        reds = torch.zeros((n, rh, 4 * rw), dtype=torch.uint8, device=dev)
    stream.synchronize()
    ctx = f360.Context(0, stream=stream.cuda_stream)
    dec = f360.SATDecoder(ctx)
    dec.InitializeGrid(rw, rh, w, h)
    gazes = [GAZES[k] for k in range(n)]
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
    ctx.profile_reset()
    ctx.profile_arm(1)
    dec.EncodeSampleFramesGPU([reds[k].data_ptr() for k in range(n)], *args)   # eager warm-up
    ctx.finish()
    assert "sat_write_fuse_kernel" in ctx.profile_read()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=stream):
        dec.EncodeSampleFramesGPU([reds[k].data_ptr() for k in range(n)], *args)
    grid = oracle.satdec_grid(rw, rh, w, h)
    for rep in range(3):
        reds.fill_(0x5A)
        sats.fill_(-1)
        stream.synchronize()
        g2.replay()
        stream.synchronize()
        for k in range(n):
            want_sat = oracle.sat_encode(host[k], w, h, 4 * w)
            assert np.array_equal(sats[k].cpu().numpy().view(np.uint32), want_sat), (rep, k)
            want = np.full((rh, 4 * rw), 0x5A, dtype=np.uint8)
            oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, *gazes[k])
            assert np.array_equal(reds[k].cpu().numpy(), want), (rep, k)
    dec.close()
    ctx.close()


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("w,h,n", [(1024, 512, 6), (1336, 202, 5), (3840, 1920, 4)])
def test_band_one_pass_from_planes(f360, gpu_ctx, oracle, w, h, n, model):
    """The decoder's planar YUV 4:2:0 frames through the band writer's one pass
    (EncodeSampleFramesYUV420PGPU with 4 .. 22 frames): tables and reduced frames equal the oracle's
    on the RGB0 frame its libswscale restatement makes of the same planes, both rounding models."""
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    gpu_ctx.set_option("yuv.model", model)
    try:
        dec = f360.SATDecoder(gpu_ctx)
        dec.InitializeGrid(rw, rh, w, h)
        planes = [(oracle.lcg_frame(w, h, 600 + k, bpp=1), oracle.lcg_frame(w // 2, h // 2, 700 + k, bpp=1),
                   oracle.lcg_frame(w // 2, h // 2, 800 + k, bpp=1)) for k in range(n)]
        planes[0] = tuple(np.full_like(p, 255) for p in planes[0])
        bufs = [tuple(gpu_ctx.upload(p.reshape(-1)) for p in planes[k]) for k in range(n)]
        sats = [gpu_ctx.malloc(w * h * 12) for _ in range(n)]
        reds = [gpu_ctx.malloc(rw * rh * 4) for _ in range(n)]
        for b in sats:
            b.fill(0xEE)
        for b in reds:
            b.fill(0x5A)
        gazes = GAZES[:n]
        gpu_ctx.profile_reset()
        gpu_ctx.profile_arm(1)
        dec.EncodeSampleFramesYUV420PGPU([b.ptr for b in reds], rw, rh, 4 * rw, [b.ptr for b in sats],
                                         [tuple(b.ptr for b in bufs[k]) for k in range(n)], w, w // 2, w // 2,
                                         w, h, gazes)
        gpu_ctx.finish()
        assert "sat_write_fuse_kernel" in gpu_ctx.profile_read()
        for k in range(n):
            frame = oracle.yuv420p_to_rgb0(*planes[k], w, h, model)
            want_sat = oracle.sat_encode(frame.reshape(-1), w, h, 4 * w)
            assert np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want_sat), (k, "table")
            want = np.full((rh, 4 * rw), 0x5A, dtype=np.uint8)
            oracle.satdec_sample_rect(want, rw, rh, 4 * rw, want_sat, w, h, grid, *gazes[k])
            assert np.array_equal(reds[k].copy_to_host(np.uint8, (rh, 4 * rw)), want), (k, gazes[k])
        for group in bufs:
            for b in group:
                b.free()
        for b in sats + reds:
            b.free()
        dec.close()
    finally:
        gpu_ctx.set_option("yuv.model", 1)
