"""One-process-per-GPU frame assembly (used by bench.py and by tests/test_distributed_cpu.py).

The reference renders a frame on N devices inside one process: `Schedule` -> one worker per tracer
`Trace`s its row block -> `primary.MergeOutput(tracer)` (renderer/default.go:106-196,
tracer/opencl/resources.go:108-124).  Under the driver's launch contract every GPU has its own
process, so the single exchange step of the path -- the blocks' accumulator strips travelling to
the primary -- is a set of `torch.distributed` point-to-point transfers (backend nccl = RCCL on the GPU
box, gloo in the CPU tests): every rank sends exactly its rows, the primary receives them at their place
in the frame.  There is no other data-path transfer; the scheduler feedback below is 16 bytes per rank.

`StripExchange` keeps the transfers OFF the critical path: `post()` starts those of frame i
asynchronously and returns at once, so every rank goes straight on to trace frame i + 1; `wait()`
(called one frame later) hands rank 0 the assembled frame i.  Two sets of buffers alternate.
`SchedulerFeedback` runs the reference's block scheduler identically on every rank from all-gathered
(rows, time) pairs, also one frame behind.
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

    A frame's blocks are row ranges of the frame, so rank dst keeps whole-FRAME buffers and every strip lands at its own
    rows: rank r sends exactly rows[r] * frame_w float4 (point-to-point: nothing is padded, the blocks may be as uneven as the
    scheduler likes and may change from frame to frame), dst receives each at row offset sum(rows[:r]), and its own block is
    exported straight into the buffer.  The assembled frame is merged with ONE polaris_hip_merge_device.  Several sets of
    buffers alternate (`depth`); a set is reused only after its exchange has completed ON THE DEVICE on every rank.

    device: torch device of the buffers ("cpu" in the gloo tests).  `via_host=True` stages device strips through host memory
    (gloo cannot move device tensors): test mode for ranks sharing one GPU.
    """

    def __init__(self, dist, rank: int, world: int, frame_w: int, frame_h: int, device, dst: int = 0, via_host: bool = False, depth: int = 2):
        import torch

        self.dist, self.rank, self.world, self.W, self.H, self.dst = dist, rank, world, frame_w, frame_h, dst
        self.via_host = via_host
        n = frame_h * frame_w
        # dst: whole frames.  Elsewhere: one strip -- sized for the whole frame too, because the scheduler, not this class,
        # decides how many rows a rank gets (16 bytes per pixel: 4 MB at 512 x 512)
        self.bufs = [torch.zeros((n, 4), dtype=torch.float32, device=device) for _ in range(depth)]
        self._host = [torch.zeros((n, 4), dtype=torch.float32) for _ in range(depth)] if via_host else None
        self._turn = 0

    def post(self, fill, rows):
        """rows[r] = block height of rank r in THIS frame (the same list on every rank).  fill(tensor) must leave this
        rank's block rows in tensor[: rows[rank] * W] before it returns (bench.py: polaris_hip_export_block, which
        synchronises its own stream).  Starts the exchange and returns a ticket for wait()."""
        import torch

        rows = list(rows)
        assert len(rows) == self.world and sum(rows) == self.H and min(rows) >= 1, rows
        i = self._turn % len(self.bufs)
        self._turn += 1
        buf = self.bufs[i]
        W = self.W
        y = sum(rows[: self.rank])
        mine = buf[y * W:(y + rows[self.rank]) * W] if self.rank == self.dst else buf[: rows[self.rank] * W]
        fill(mine)
        if self.world == 1:
            return (i, [], rows)
        wire = buf
        if self.via_host:
            wire = self._host[i]
            if self.rank == self.dst:
                wire[y * W:(y + rows[self.rank]) * W].copy_(mine)  # synchronous D2H
            else:
                wire[: rows[self.rank] * W].copy_(mine)
        ops = []
        if self.rank == self.dst:
            yy = 0
            for r in range(self.world):
                if r != self.dst:
                    ops.append(self.dist.P2POp(self.dist.irecv, wire[yy * W:(yy + rows[r]) * W], r))
                yy += rows[r]
        else:
            ops.append(self.dist.P2POp(self.dist.isend, wire[: rows[self.rank] * W], self.dst))
        works = self.dist.batch_isend_irecv(ops)  # (nccl: one grouped launch; gloo: the individual operations)
        return (i, works, rows)

    def wait(self, ticket):
        """Completes the exchange of `ticket`.  On dst returns [(0, frame_h, frame tensor)] -- the assembled frame; elsewhere
        None."""
        import torch

        i, works, rows = ticket
        buf = self.bufs[i]
        for w in works:
            w.wait()
        # On the nccl backend Work.wait() only makes torch's CURRENT STREAM wait for the transfer; the host returns at once.
        # The buffers are refilled `depth` posts later from the tracer's own (non-blocking) HIP stream, which has no ordering
        # against RCCL's stream -- so EVERY rank, not only dst, blocks here until its part of the exchange has really finished.
        # That also bounds the skew between ranks to the depth of the exchange.
        if buf.is_cuda and not self.via_host:
            torch.cuda.current_stream(buf.device).synchronize()
        if self.rank != self.dst:
            return None
        if self.via_host and self.world > 1:
            buf.copy_(self._host[i])
            if buf.is_cuda:
                torch.cuda.current_stream(buf.device).synchronize()  # the tracer merges on its own HIP stream: data must have landed
        return [(0, self.H, buf)]


class SchedulerFeedback:
    """The reference's perfect scheduler (tracer/scheduler.go:50-80, restated in polaris_amd/host/scheduler.cpp) across
    processes: every rank publishes (block height, trace time) of a frame with an asynchronous all_gather -- 16 bytes per rank,
    completed one frame later like the strips -- and every rank feeds the same numbers to the same scheduler, so all arrive at
    the same rows without a synchronous step.  Frame f + 2 is scheduled from frame f's times; the first two frames use the
    naive split (the scheduler's own first frame, scheduler.go:52-56)."""

    def __init__(self, dist, rank: int, world: int, frame_h: int, device, kind: str = "perfect"):
        import torch

        from . import host_api

        self.dist, self.rank, self.world, self.H, self.kind = dist, rank, world, frame_h, kind
        self.rows = naive_rows(world, frame_h)
        self._sched = None
        if kind == "perfect" and world > 1:
            self._sched = host_api.Scheduler(host_api.PERFECT, [1] * world)
            self._sched.schedule(frame_h)  # its first frame: the naive split
        self._dev = device
        self._torch = torch
        self._pending = []

    def publish(self, rows, trace_ms: float):
        """After a frame's Trace: this rank's block height and time go out (asynchronously)."""
        if self._sched is None:
            return
        t = self._torch
        mine = t.tensor([int(rows[self.rank]), int(trace_ms * 1e6)], dtype=t.int64, device=self._dev)
        if mine.is_cuda:  # nccl: one output tensor, so that reading the result back is ONE device-to-host copy
            out = t.zeros((self.world, 2), dtype=t.int64, device=self._dev)
            work = self.dist.all_gather_into_tensor(out, mine, async_op=True)
        else:             # gloo (tests)
            out = [t.zeros(2, dtype=t.int64) for _ in range(self.world)]
            work = self.dist.all_gather(out, mine, async_op=True)
        self._pending.append((work, out, mine))

    def next_rows(self):
        """The rows of the NEXT frame: from the newest feedback that has been published at least one frame ago."""
        if self._sched is None:
            return self.rows
        while len(self._pending) > 1:  # everything but the frame just published has had a whole frame's time to complete
            work, out, _ = self._pending.pop(0)
            work.wait()
            got = out.cpu().tolist() if not isinstance(out, list) else [o.tolist() for o in out]  # (.cpu() orders itself behind the collective on the current stream)
            self.rows = self._sched.schedule(self.H, block_h=[g[0] for g in got], render_ns=[max(1, g[1]) for g in got])
        return self.rows

    def drain(self):
        for work, out, _ in self._pending:
            work.wait()
        self._pending = []
