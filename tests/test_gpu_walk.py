"""GPU tier: the read-once batched SAT encoder (sat_walk_kernel, option "sat.walk") against the
CPU oracle.  One wave walks a 256-pixel strip of one frame from top to bottom and takes the row
prefixes of the strips to its left through tagged 8-byte hand-off granules; the table must be
the oracle's bit for bit -- any stale, torn or early-read granule shows up as a wrong sum in
every row below it.  The cases force the kernel on geometries the automatic choice would give to
the three-kernel encoder (few strips, one strip, ragged edges, heights that are not a multiple
of the 8-row batch) and run it under uneven load (many frames of different cost in flight,
a second stream hammering memory, re-used and re-carved hand-off buffers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode_batch_and_check(f360, ctx, oracle, w, h, n, pad=0, seed=900, frames=None):
    ls = 4 * w + pad
    if frames is None:
        frames = [oracle.lcg_frame(w, h, seed + k, bpp=4, linesize=ls) for k in range(n)]
    srcs = [ctx.upload(np.ascontiguousarray(f).reshape(-1)) for f in frames]
    sats = [ctx.malloc(w * h * 12) for _ in range(n)]
    for s in sats:
        s.fill(0xEE)
    f360.SATEncoder(ctx).EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, ls)
    bad = []
    for k in range(n):
        got = sats[k].copy_to_host(np.uint32, (h, w, 3))
        if not np.array_equal(got, oracle.sat_encode(frames[k], w, h, ls)):
            bad.append(k)
    for b in srcs + sats:
        b.free()
    return bad


@pytest.fixture
def walk_ctx(gpu_ctx):
    """A context that takes the read-once kernel for every batched encode, whatever its size."""
    gpu_ctx.set_option("sat.walk", 1)
    yield gpu_ctx
    gpu_ctx.set_option("sat.walk", -1)


@pytest.mark.parametrize("w,h,n,pad", [
    (1336, 203, 5, 0),      # ragged last strip, height not a multiple of 8
    (1920, 1080, 3, 0),     # the reference's operating point
    (256, 128, 19, 16),     # one strip per frame: no hand-off at all; padded rows
    (260, 9, 2, 0),         # two strips, the second 4 pixels wide; two batches, the second 1 row
    (4, 1, 1, 0), (8, 7, 3, 0), (512, 8, 4, 0), (516, 17, 70, 0),   # 70 frames: two launches
    (3840, 64, 2, 0),       # 15 strips, 8 batches
    (1024, 512, 40, 0),     # 160 units in flight
])
def test_walk_encode_matches_oracle(f360, walk_ctx, oracle, w, h, n, pad):
    assert _encode_batch_and_check(f360, walk_ctx, oracle, w, h, n, pad) == []


def test_walk_repeated_and_recarved(f360, walk_ctx, oracle):
    """The hand-off granules are never cleared between launches: the launch serial in their tag
    tells this launch's from older ones.  Same geometry many times, then another geometry whose
    strips land on the old granules, then the first again; a single-frame encode (the
    three-kernel path, which shares nothing with it) in between."""
    for rep in range(6):
        assert _encode_batch_and_check(f360, walk_ctx, oracle, 1024, 64, 6, seed=10 * rep) == []
    assert _encode_batch_and_check(f360, walk_ctx, oracle, 520, 130, 3) == []
    assert _encode_batch_and_check(f360, walk_ctx, oracle, 2048, 40, 9) == []   # larger: re-carve
    frame = oracle.lcg_frame(640, 48, 5)
    src, sat = walk_ctx.upload(frame), walk_ctx.malloc(640 * 48 * 12)
    f360.SATEncoder(walk_ctx).EncodeFrameGPU(sat.ptr, src.ptr, 640, 48, 4 * 640)
    assert np.array_equal(sat.copy_to_host(np.uint32, (48, 640, 3)),
                          oracle.sat_encode(frame, 640, 48, 4 * 640))
    src.free()
    sat.free()
    assert _encode_batch_and_check(f360, walk_ctx, oracle, 1024, 64, 6, seed=77) == []


def test_walk_special_frames(f360, walk_ctx, oracle):
    """All-255 (largest row prefixes: the 24-bit payload of a granule), all-zero and a frame
    whose 4th byte is garbage."""
    w, h = 4096, 96
    frames = [np.full((h, 4 * w), 255, dtype=np.uint8), np.zeros((h, 4 * w), dtype=np.uint8),
              oracle.lcg_frame(w, h, 3)]
    assert _encode_batch_and_check(f360, walk_ctx, oracle, w, h, 3, frames=frames) == []


def test_walk_under_uneven_load(f360, oracle):
    """A second context on its own stream streams through memory while the strip owners hand
    their prefixes along: hand-offs must not depend on timing or placement."""
    with f360.Context(0) as a, f360.Context(0) as b:
        a.set_option("sat.walk", 1)
        w, h, n = 2304, 256, 24
        frames = [oracle.lcg_frame(w, h, 300 + k) for k in range(n)]
        srcs = [a.upload(f) for f in frames]
        sats = [a.malloc(w * h * 12) for _ in range(n)]
        big_src = b.upload(oracle.lcg_frame(3840, 1920, 1))
        big_sat = b.malloc(3840 * 1920 * 12)
        enc_a, enc_b = f360.SATEncoder(a), f360.SATEncoder(b)
        for rep in range(4):
            for s in sats:
                s.fill(rep)
            for _ in range(6):
                enc_b.EncodeFrameGPU(big_sat.ptr, big_src.ptr, 3840, 1920, 4 * 3840)
            enc_a.EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, 4 * w)
            for _ in range(6):
                enc_b.EncodeFrameGPU(big_sat.ptr, big_src.ptr, 3840, 1920, 4 * 3840)
            for k in range(n):
                assert np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)),
                                      oracle.sat_encode(frames[k], w, h, 4 * w)), (rep, k)
            b.finish()
        for x in srcs + sats:
            x.free()
        big_src.free()
        big_sat.free()


def test_walk_equals_three_kernel_encoder_at_8k(f360, gpu_ctx, oracle, golden_digests=None):
    """Full size: three 8K frames (a seeded one, the golden one, the all-255 frame whose sums
    wrap mod 2^32) through both batched encoders; digests equal, and the golden frame's is the
    committed one."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digests.json")) as f:
        dig = json.load(f)
    w, h = 7680, 3840
    ent = dig["cases"][f"{w}x{h}"]
    frames = [oracle.lcg_frame(w, h, 7), oracle.lcg_frame(w, h, dig["seed"]),
              np.full((h, 4 * w), 255, dtype=np.uint8)]
    srcs = [gpu_ctx.upload(f) for f in frames]
    sats = [gpu_ctx.malloc(w * h * 12) for _ in frames]
    enc = f360.SATEncoder(gpu_ctx)
    out = {}
    for walk in (0, 1):   # three kernels, sat_walk_kernel
        gpu_ctx.set_option("sat.walk", walk)
        for s in sats:
            s.fill(0x5A)
        enc.EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, 4 * w)
        out[walk] = [f"{oracle.fnv1a64(s.copy_to_host(np.uint32, (h, w, 3))):016x}" for s in sats]
    gpu_ctx.set_option("sat.walk", -1)
    for b in srcs + sats:
        b.free()
    assert out[0] == out[1]
    assert out[1][1] == ent["sat"] and out[1][2] == ent["sat_white"]


@pytest.mark.parametrize("model", [0, 1])
@pytest.mark.parametrize("w,h,pad,n", [(256, 64, (0, 0, 0), 3), (260, 38, (4, 2, 6), 5),
                                       (1920, 1080, (0, 0, 0), 2), (1028, 22, (0, 0, 0), 70)])
def test_walk_encode_from_planes(f360, walk_ctx, oracle, model, w, h, pad, n):
    """Planar YUV 4:2:0 frames through the read-once encoder (conversion in registers, both
    libswscale models): every table is the oracle table of the oracle-converted frame.  Heights
    that are not a multiple of the 8-row batch, padded planes, 70 frames (two launches)."""
    rng = np.random.default_rng(1000 + w + h)
    cw = (w + 1) // 2

    def planes_of():
        return (rng.integers(0, 256, (h, w + pad[0]), dtype=np.uint8),
                rng.integers(0, 256, (h // 2, cw + pad[1]), dtype=np.uint8),
                rng.integers(0, 256, (h // 2, cw + pad[2]), dtype=np.uint8))
    walk_ctx.set_option("yuv.model", model)
    planes = [planes_of() for _ in range(n)]
    dev = [tuple(walk_ctx.upload(p) for p in pl) for pl in planes]
    sats = [walk_ctx.malloc(w * h * 12) for _ in range(n)]
    for s in sats:
        s.fill(0xEE)
    y0, u0, v0 = planes[0]
    f360.SATEncoder(walk_ctx).EncodeFramesYUV420PGPU(
        [s.ptr for s in sats], [(a.ptr, b.ptr, c.ptr) for (a, b, c) in dev], y0.shape[1],
        u0.shape[1], v0.shape[1], w, h)
    bad = []
    for k in range(n):
        y, u, v = planes[k]
        want = oracle.sat_encode(oracle.yuv420p_to_rgb0(y, u, v, w, h, model), w, h, 4 * w)
        if not np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want):
            bad.append(k)
    walk_ctx.set_option("yuv.model", 1)
    for b in sats + [p for t in dev for p in t]:
        b.free()
    assert bad == []


def _planes_batch_and_check(f360, ctx, oracle, w, h, n, model=1, seed=77):
    rng = np.random.default_rng(seed)
    cw = w // 2
    planes = [(rng.integers(0, 256, (h, w), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw), dtype=np.uint8),
               rng.integers(0, 256, (h // 2, cw), dtype=np.uint8)) for _ in range(n)]
    dev = [tuple(ctx.upload(p) for p in pl) for pl in planes]
    sats = [ctx.malloc(w * h * 12) for _ in range(n)]
    for s in sats:
        s.fill(0xEE)
    ctx.set_option("yuv.model", model)
    f360.SATEncoder(ctx).EncodeFramesYUV420PGPU(
        [s.ptr for s in sats], [(a.ptr, b.ptr, c.ptr) for (a, b, c) in dev], w, cw, cw, w, h)
    bad = []
    for k in range(n):
        y, u, v = planes[k]
        want = oracle.sat_encode(oracle.yuv420p_to_rgb0(y, u, v, w, h, model), w, h, 4 * w)
        if not np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)), want):
            bad.append(k)
    for b in sats + [p for t in dev for p in t]:
        b.free()
    return bad


def test_a_missing_hand_off_costs_time_not_a_result(f360, oracle):
    """The failure path of the hand-off chain, driven for real.  "debug.walk_mute" makes one unit
    publish nothing and "debug.walk_spin" shortens the bound of a wait from ~0.1 s to a few
    polls: the right neighbour's wait times out, the strip leaves the chain, recomputes the row
    sums to its left from the source for the rest of its rows and publishes correct prefixes
    itself.  The grid drains, every table is the oracle's bit for bit, the recovery is counted,
    and the next call on the same context (state words re-armed by the launch itself) is clean."""
    with f360.Context(0) as ctx:
        ctx.set_option("sat.walk", 1)
        assert ctx.debug_walk_recoveries() == 0
        for mute_unit, (w, h, n) in [(1, (1336, 203, 3)),      # strip 1 of frame 0 of 6 strips
                                     (9, (1024, 64, 6)),       # strip 1 of frame 2 of 4 strips
                                     (4, (2048, 50, 70))]:     # two launches: once in each
            ctx.set_option("debug.walk_spin", 48)
            ctx.set_option("debug.walk_mute", mute_unit + 1)
            assert _encode_batch_and_check(f360, ctx, oracle, w, h, n, seed=31 * mute_unit) == []
            ctx.finish()   # reports nothing: the tables are right
            assert ctx.debug_walk_recoveries() >= 1, (mute_unit, w, h, n)
            # planar sources recover through the same path (conversion in registers)
            ctx.set_option("debug.walk_mute", 2)   # strip 1 of frame 0 of 5 strips
            for model in (0, 1):
                assert _planes_batch_and_check(f360, ctx, oracle, 1028, 22, 5, model) == []
                assert ctx.debug_walk_recoveries() >= 1
            # without the fault and with the shipped bound: exact, and nothing to recover
            ctx.set_option("debug.walk_mute", 0)
            ctx.set_option("debug.walk_spin", 0)
            assert _encode_batch_and_check(f360, ctx, oracle, w, h, n, seed=5) == []
            assert ctx.debug_walk_recoveries() == 0
        # every wait one poll long: strips drop off the chain all over the frame, at any batch
        ctx.set_option("debug.walk_spin", 1)
        assert _encode_batch_and_check(f360, ctx, oracle, 2304, 256, 24, seed=8) == []
        assert _planes_batch_and_check(f360, ctx, oracle, 1920, 1080, 2) == []
        many = ctx.debug_walk_recoveries()
        ctx.set_option("debug.walk_spin", 0)
        assert _encode_batch_and_check(f360, ctx, oracle, 2304, 256, 24, seed=9) == []
        assert ctx.debug_walk_recoveries() == 0
        print(f"recoveries with one-poll waits: {many}")


def test_table_pool_allocates_usable_tables(f360, oracle):
    """f360_sat_tables_alloc: tables for batched calls, drawn in groups and kept by measured write
    rate.  Whatever it keeps must be `count` distinct, 16-byte aligned, non-overlapping tables
    that the encoders fill correctly; a call below the read-once threshold gets plain
    allocations; freeing gives everything back."""
    with f360.Context(0) as ctx:
        enc = f360.SATEncoder(ctx)
        for (w, h, n, walks) in [(2304, 256, 80, True), (7680, 64, 33, True), (640, 48, 3, False)]:
            pool = enc.AllocateTables(w, h, n)
            assert len(pool.ptrs) == n and len(set(pool.ptrs)) == n and pool.report
            assert ("kept" in pool.report) == walks, pool.report
            spans = sorted(pool.ptrs)
            assert all(p % 16 == 0 for p in spans)
            assert all(b - a >= w * h * 12 for a, b in zip(spans, spans[1:]))
            frames = [oracle.lcg_frame(w, h, 40 + k) for k in range(n)]
            srcs = [ctx.upload(f) for f in frames]
            enc.EncodeFramesGPU(pool.ptrs, [s.ptr for s in srcs], w, h, 4 * w)
            for k in (0, n // 2, n - 1):
                assert np.array_equal(pool.read_table(k, (h, w, 3)),
                                      oracle.sat_encode(frames[k], w, h, 4 * w)), (w, h, n, k)
            for s in srcs:
                s.free()
            pool.free()
            pool.free()   # idempotent


def test_table_pool_respects_its_memory_bound(f360, oracle):
    """"sat.pool_mb": the pool never holds more than the bound while it draws (the report says how
    much it held at most) -- here room for the tables asked for plus ONE group, so every further
    draw first gives the slowest group back -- and what it keeps is still right."""
    import re
    w, h, n = 2304, 256, 128          # 9 strips, at most 64 frames per launch: two groups of 64
    with f360.Context(0) as ctx:
        enc = f360.SATEncoder(ctx)
        group_mb = 64 * w * h * 12 / 2 ** 20
        ctx.set_option("sat.pool_mb", int(3 * group_mb + 64 * w * h * 4 / 2 ** 20 + 2))
        pool = enc.AllocateTables(w, h, n)
        m = re.search(r"at most (\d+) held at once: ([0-9.]+) of ([0-9.]+) GB allowed, (\d+) given back", pool.report)
        assert m, pool.report
        assert int(m.group(1)) == 3 and float(m.group(2)) <= float(m.group(3)) + 1e-3, pool.report
        assert int(m.group(4)) >= 1, pool.report      # (at least five draws are looked at)
        assert len(set(pool.ptrs)) == n
        frames = [oracle.lcg_frame(w, h, 900 + k) for k in (0, 1)]
        srcs = [ctx.upload(f) for f in frames]
        enc.EncodeFramesGPU(pool.ptrs, [srcs[k % 2].ptr for k in range(n)], w, h, 4 * w)
        for k in (0, 63, 64, n - 1):
            assert np.array_equal(pool.read_table(k, (h, w, 3)), oracle.sat_encode(frames[k % 2], w, h, 4 * w))
        for b in srcs:
            b.free()
        pool.free()
        ctx.set_option("sat.pool_mb", 0)


def test_walker_refuses_to_allocate_under_stream_capture(f360, oracle):
    """The first read-once call on a context allocates its hand-off buffers (and a larger call
    re-allocates them): illegal while the stream is being captured, so it is refused with a
    message instead of breaking the capture; after an eager warm-up the capture works
    (test_walker_launch_replays_from_a_hip_graph)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test without a GPU")
    dev = torch.device("cuda", 0)
    w, h, n = 1280, 72, 4
    frames = torch.zeros((n, h, 4 * w), dtype=torch.uint8, device=dev)
    sats = [torch.zeros((h, w, 3), dtype=torch.int32, device=dev) for _ in range(n)]
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        ctx = f360.Context(0, stream=side.cuda_stream)
        ctx.set_option("sat.walk", 1)
        enc = f360.SATEncoder(ctx)
        g = torch.cuda.CUDAGraph()
        refused = False
        with torch.cuda.graph(g, stream=side):
            try:
                enc.EncodeFramesGPU([s.data_ptr() for s in sats],
                                    [frames[k].data_ptr() for k in range(n)], w, h, 4 * w)
            except f360.F360Error as e:
                refused = "captured" in str(e)
        assert refused
    ctx.close()


def test_config4_batch_through_the_automatic_choice(f360, oracle):
    """BASELINE config 4 the way bench.py runs it: 32 of its frames (31 LCG frames + the all-255
    frame whose sums wrap mod 2^32) resident, one EncodeFramesGPU call -- 960 strips: the
    automatic choice takes the read-once encoder -- and one SampleFramesRectGPU call; every
    table and every reduced frame equals the oracle's by digest."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_configs
    res = bench_configs.config4_batched(f360, oracle, quick=True)
    assert "sat_walk_kernel" in res["workload"], res
    assert res["bad_frames"] == [], res


def test_two_walker_launches_share_the_device(f360, oracle):
    """Two contexts (two streams) run read-once launches at the same time: workgroups of one
    launch occupy SIMDs the other's later tickets are waiting for.  A strip only ever waits for a
    unit whose workgroup is already running, so both drain and both are right."""
    import threading
    w, h, n = 2304, 200, 40   # 9 strips x 40 frames = 360 units per launch
    results = {}

    def worker(tag, seed):
        with f360.Context(0) as ctx:
            ctx.set_option("sat.walk", 1)
            frames = [oracle.lcg_frame(w, h, seed + k) for k in range(n)]
            srcs = [ctx.upload(f) for f in frames]
            sats = [ctx.malloc(w * h * 12) for _ in range(n)]
            enc = f360.SATEncoder(ctx)
            bad = []
            for rep in range(5):
                for s in sats:
                    s.fill(rep + 1)
                enc.EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, 4 * w)
                enc.EncodeFramesGPU([s.ptr for s in sats], [s.ptr for s in srcs], w, h, 4 * w)
                ctx.finish()
                for k in (0, n // 2, n - 1):
                    if not np.array_equal(sats[k].copy_to_host(np.uint32, (h, w, 3)),
                                          oracle.sat_encode(frames[k], w, h, 4 * w)):
                        bad.append((rep, k))
            for b in srcs + sats:
                b.free()
            results[tag] = bad

    threads = [threading.Thread(target=worker, args=(t, 500 + 100 * t)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == {0: [], 1: []}, results


def test_walker_launch_replays_from_a_hip_graph(f360, oracle):
    """Nothing a read-once launch needs comes from the host per launch -- ticket, retirement count
    and the serial that tags its hand-off granules live in device memory and are advanced by the
    launches themselves -- so a captured launch (kernel arguments frozen) replays correctly, any
    number of times, on new inputs."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test without a GPU")
    w, h, n = 1280, 72, 6
    dev = torch.device("cuda", 0)
    frames = torch.zeros((n, h, 4 * w), dtype=torch.uint8, device=dev)
    sats = [torch.zeros((h, w, 3), dtype=torch.int32, device=dev) for _ in range(n)]
    side = torch.cuda.Stream(dev)

    def put(seed):
        host = [oracle.lcg_frame(w, h, seed + k) for k in range(n)]
        for k in range(n):
            frames[k].copy_(torch.from_numpy(host[k]).to(dev).reshape(h, 4 * w))
        return host

    with torch.cuda.stream(side):
        ctx = f360.Context(0, stream=side.cuda_stream)
        ctx.set_option("sat.walk", 1)
        enc = f360.SATEncoder(ctx)
        fp = [frames[k].data_ptr() for k in range(n)]
        sp = [s.data_ptr() for s in sats]
        put(10)
        enc.EncodeFramesGPU(sp, fp, w, h, 4 * w)   # eager warm-up: allocates the hand-off buffers
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            enc.EncodeFramesGPU(sp, fp, w, h, 4 * w)
        for rep in range(4):
            host = put(100 * (rep + 1))
            for s in sats:
                s.fill_(-1)
            torch.cuda.synchronize(dev)
            g.replay()
            torch.cuda.synchronize(dev)
            for k in range(n):
                assert np.array_equal(sats[k].cpu().numpy().view(np.uint32),
                                      oracle.sat_encode(host[k], w, h, 4 * w)), (rep, k)
        # and an eager launch after the replays still finds its state in order
        host = put(999)
        enc.EncodeFramesGPU(sp, fp, w, h, 4 * w)
        side.synchronize()
        assert np.array_equal(sats[n - 1].cpu().numpy().view(np.uint32),
                              oracle.sat_encode(host[n - 1], w, h, 4 * w))
    ctx.close()
