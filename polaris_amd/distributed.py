"""One-process-per-GPU frame assembly used by bench.py: the row-block partition of
tracer/scheduler.go (equal speeds) and the gather-to-primary exchange of renderer/default.go:191
expressed with torch.distributed (backend nccl = RCCL on the GPU box, gloo in the CPU tests)."""
from __future__ import annotations


def naive_rows(n_tracers: int, frame_h: int, speeds=None) -> list[int]:
    """tracer/scheduler.go:83-106 assignBlocksBasedOnSpeed (all speeds equal unless given)."""
    speeds = [1] * n_tracers if speeds is None else list(speeds)
    scaler = float(frame_h) / float(sum(speeds))
    rows = [int(max(1.0, float(s) * scaler)) for s in speeds]
    if sum(rows) < frame_h:
        rows[0] += frame_h - sum(rows)
    return rows


def block_of(rank: int, rows: list[int]) -> tuple[int, int]:
    """(block_y, block_h) of tracer `rank`: running sum, renderer/default.go:127-136."""
    return sum(rows[:rank]), rows[rank]


def gather_strips(strip, rows, frame_w, dist, rank, dst=0):
    """Gather every rank's accumulator strip (rows[r]*frame_w, 4) on `dst`.  Returns the list of
    strips on dst (index = rank), None elsewhere."""
    import torch

    world = len(rows)
    if world == 1:
        return [strip]
    bufs = [torch.empty((rows[i] * frame_w, 4), dtype=strip.dtype, device=strip.device) for i in range(world)] if rank == dst else None
    dist.gather(strip, bufs, dst=dst)
    return bufs
