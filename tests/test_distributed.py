"""CPU tier: the N>1 path (frame sharding + the final counter reduction) with world_size 2 over
gloo.  Each rank runs the ORACLE on its shard here (no GPU in this tier); what is under test is
the sharding arithmetic and the reduction the benchmark uses, not the kernels."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total_frames, out_dir):
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "tests")]
    import torch.distributed as dist
    import oracle_binding as ob
    import f360_amd
    from importlib import import_module
    sharding = import_module("foveated-360-video_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = 128, 64
    rw, rh = f360_amd.reduced_size(w), f360_amd.reduced_size(h)
    grid = ob.satdec_grid(rw, rh, w, h)
    digests = {}
    for g in sharding.shard_range(total_frames, world, rank):
        frame = ob.lcg_frame(w, h, sharding.frame_seed(g))
        sat = ob.sat_encode(frame, w, h, 4 * w)
        red = np.zeros((rh, 4 * rw), dtype=np.uint8)
        ob.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, 0.5, 0.5)
        digests[g] = ob.fnv1a64(red)
    elapsed, pixels = sharding.reduce_run(1.0 + rank, float(len(digests) * w * h))
    # what bench.py prints as per_rank (rank 1 pretends its encoder fell back to three kernels)
    per_rank = sharding.gather_run(1.0 + rank, len(digests), encoder=1 - rank, recoveries=rank)
    assert per_rank == [(r, len(sharding.shard_range(total_frames, world, r)), 1.0 + r, 1 - r, r)
                        for r in range(world)], per_rank
    np.save(os.path.join(out_dir, f"rank{rank}.npy"),
            np.array([[g, d & 0xFFFFFFFF, d >> 32] for g, d in digests.items()] +
                     [[-1, int(elapsed * 1000), int(pixels)]], dtype=np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from importlib import import_module
    sharding = import_module("foveated-360-video_amd.sharding")
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                rg = sharding.shard_range(total, world, r)
                seen.extend(rg)
                assert len(rg) in (total // world, total // world + 1)
            assert seen == list(range(total))
    assert list(sharding.shard_range(64, 8, 3)) == list(range(24, 32))  # batch 64 -> 8 per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_two_rank_gloo_run(tmp_path, oracle, f360):
    import torch.multiprocessing as mp
    total, world = 7, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    rows = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    frames = {int(g): (int(lo) | (int(hi) << 32)) for g, lo, hi in rows if g >= 0}
    assert sorted(frames) == list(range(total))  # every frame exactly once
    # same digests as a single process computes
    w, h = 128, 64
    rw, rh = f360.reduced_size(w), f360.reduced_size(h)
    grid = oracle.satdec_grid(rw, rh, w, h)
    for g in range(total):
        sat = oracle.sat_encode(oracle.lcg_frame(w, h, 1 + g), w, h, 4 * w)
        red = np.zeros((rh, 4 * rw), dtype=np.uint8)
        oracle.satdec_sample_rect(red, rw, rh, 4 * rw, sat, w, h, grid, 0.5, 0.5)
        assert oracle.fnv1a64(red) == frames[g], g
    for g, a, b in rows:
        if g < 0:  # every rank sees max(elapsed) = 2.0 s and the total pixel count
            assert (int(a), int(b)) == (2000, total * w * h)
