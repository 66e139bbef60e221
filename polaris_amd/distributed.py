"""One-process-per-GPU frame assembly (used by bench.py and by tests/test_distributed_cpu.py).

The reference renders a frame on N devices inside one process: `Schedule` -> one worker per tracer
`Trace`s its row block -> `primary.MergeOutput(tracer)` (renderer/default.go:106-196,
tracer/opencl/resources.go:108-124).  Under the driver's launch contract every GPU has its own
process, so the single exchange step of the path -- the gather of the blocks' accumulator strips to
the primary -- is a `torch.distributed` gather (backend nccl = RCCL on the GPU box, gloo in the CPU
tests); there is no other data-path collective.

`StripExchange` keeps that gather OFF the critical path: `post()` starts the gather of frame i
asynchronously and returns at once, so every rank goes straight on to trace frame i + 1; `wait()`
(called one frame later) hands rank 0 the gathered strips of frame i.  Two sets of buffers alternate.
"""
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


class StripExchange:
    """Gather-to-primary of the per-rank accumulator strips, one frame of look-ahead.

    rows[r] = block height of rank r; a strip is (rows[r] * frame_w, 4) float32.  Collectives want
    equally sized pieces, so strips travel padded to the tallest block (the naive scheduler gives the
    remainder rows to tracer 0, tracer/scheduler.go:101-103).

    device: torch device of the strip buffers ("cpu" in the gloo tests).  `via_host=True` stages
    device strips through host memory (gloo cannot move device tensors): test mode for two ranks
    sharing one GPU.
    """

    def __init__(self, dist, rank: int, rows: list[int], frame_w: int, device, dst: int = 0, via_host: bool = False, depth: int = 2):
        import torch

        self.dist, self.rank, self.rows, self.W, self.dst = dist, rank, list(rows), frame_w, dst
        self.world = len(rows)
        self.via_host = via_host
        self.max_rows = max(rows)
        n = self.max_rows * frame_w
        self.strips = [torch.zeros((n, 4), dtype=torch.float32, device=device) for _ in range(depth)]
        self.gathered = None
        if rank == dst:
            self.gathered = [torch.empty((self.world, n, 4), dtype=torch.float32, device=device) for _ in range(depth)]
        self._host = None
        if via_host:
            self._host = [torch.empty((n, 4), dtype=torch.float32) for _ in range(depth)]
            self._host_g = [[torch.empty((n, 4), dtype=torch.float32) for _ in range(self.world)] for _ in range(depth)] if rank == dst else None
        self._turn = 0
        self.uniform = len(set(rows)) == 1

    def post(self, fill):
        """fill(strip_tensor) must leave this rank's block rows in strip_tensor[: rows[rank] * W] before it
        returns (bench.py: polaris_hip_export_block, which synchronises its own stream).  Starts the gather
        and returns a ticket for wait()."""
        i = self._turn % len(self.strips)
        self._turn += 1
        strip = self.strips[i]
        fill(strip)
        if self.world == 1:
            return (i, None)
        if self.via_host:
            self._host[i].copy_(strip)  # synchronous D2H
            work = self.dist.gather(self._host[i], self._host_g[i] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        else:
            glist = [self.gathered[i][r] for r in range(self.world)] if self.rank == self.dst else None
            work = self.dist.gather(strip, glist, dst=self.dst, async_op=True)
        return (i, work)

    def wait(self, ticket):
        """Completes the gather of `ticket`.  On dst returns [(block_y, block_h, tensor)] -- ONE entry
        covering the whole frame when all blocks are equally tall (the gathered buffer IS the frame then),
        else one per rank; elsewhere returns None."""
        import torch

        i, work = ticket
        if self.world == 1:
            return [(0, self.rows[0], self.strips[i])]
        if work is not None:
            work.wait()
        # On the nccl backend Work.wait() only makes torch's CURRENT STREAM wait for the collective; the host returns at once.
        # The strip buffer is refilled two posts later from the tracer's own (non-blocking) HIP stream, which has no ordering
        # against RCCL's stream -- so EVERY rank, not only dst, blocks here until its part of the gather has really finished.
        # That also bounds the skew between ranks to the depth of the exchange.
        if self.strips[i].is_cuda and not self.via_host:
            torch.cuda.current_stream(self.strips[i].device).synchronize()
        if self.rank != self.dst:
            return None
        g = self.gathered[i]
        if self.via_host:
            for r in range(self.world):
                g[r].copy_(self._host_g[i][r])
        if g.is_cuda:
            torch.cuda.current_stream(g.device).synchronize()  # the tracer merges on its own HIP stream: data must have landed
        if self.uniform:
            return [(0, self.rows[0] * self.world, g)]
        out, y = [], 0
        for r in range(self.world):
            out.append((y, self.rows[r], g[r]))
            y += self.rows[r]
        return out
