"""Frame-batch sharding across the GPUs of one node.

Frames (and gaze points) are independent, so the hot path shards with no data-path collective:
rank r of N owns a contiguous block of the frame batch.  The only collective is the reduction
of the run's counters -- MAX of the elapsed time, SUM of the pixels -- over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The message is a
few doubles, so it is latency-bound and the xGMI link bandwidth is irrelevant.
"""
from __future__ import annotations


def shard_range(total: int, world: int, rank: int) -> range:
    """Contiguous block of `total` frames owned by `rank` (blocks differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def frame_seed(global_frame: int) -> int:
    """LCG seed of a frame of the synthetic batch (seeds 1..N)."""
    return 1 + global_frame


def reduce_run(elapsed_s: float, pixels: float, device=None):
    """(max elapsed, total pixels) over all ranks; identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return elapsed_s, pixels
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    p = torch.tensor([pixels], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(t.item()), float(p.item())


ENCODERS = ("three kernels (reduce, carry, write)", "read-once (sat_walk_kernel)",
            "fused (emit mode)", "one pass (sat_walk_kernel with helper waves: tables + reduced frames)",
            "one pass (three kernels, the table writer sat_write_fuse_kernel emits the reduced frame)")


def gather_run(elapsed_s: float, frames: int, device=None, encoder: int = 0, recoveries: int = 0):
    """[(rank, frames, elapsed, encoder, recoveries)] of every rank, on every rank (one
    all_gather of four doubles): the benchmark prints it as `per_rank`, so a straggler -- and a
    rank whose encode calls fell back to the three-kernel encoder (`encoder`: index into
    ENCODERS) or whose strip hand-offs timed out -- is visible.  Identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [(0, int(frames), float(elapsed_s), int(encoder), int(recoveries))]
    mine = torch.tensor([float(frames), elapsed_s, float(encoder), float(recoveries)],
                        dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [(r, int(t[0].item()), float(t[1].item()), int(t[2].item()), int(t[3].item()))
            for r, t in enumerate(out)]
