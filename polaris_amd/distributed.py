"""One-process-per-GPU frame assembly (used by bench.py and by tests/test_distributed_cpu.py).

The reference renders a frame on N devices inside one process: `Schedule` -> one worker per tracer
`Trace`s its row block -> `primary.MergeOutput(tracer)` (renderer/default.go:106-196,
tracer/opencl/resources.go:108-124); all devices share ONE OpenCL context, so the primary's
aggregateAccumulator kernel reads a secondary's traceAccumulator directly (renderer/default.go:225-229,
tracer/opencl/tracer.go:279-286).  Under the driver's launch contract every GPU has its own process.

`PeerExchange` (the default) is that same read across processes: every rank publishes HIP-IPC handles of
its trace accumulator ring ONCE (polaris_hip_ipc_export), the primary maps them (polaris_hip_ipc_open), and
per frame its merge stream runs k_aggregate straight over the peer-mapped rows (polaris_hip_merge_ipc: an
xGMI peer read; no copy, no staging strip, no RCCL).  torch.distributed carries CONTROL only: the
handles at set-up over gloo; the 32 bytes per rank and frame (frame number, ring slot, rows, time) go through a shared-memory
mailbox (`ShmMailbox`: ranks of one host) or, where that cannot be mapped, a gloo all_gather.

`StripExchange` + `SchedulerFeedback` are the fallback when an IPC mapping cannot be opened: the strips
travel as `torch.distributed` point-to-point transfers (backend nccl = RCCL, gloo in the CPU tests): every
rank sends exactly its rows, the primary receives them at their place in the frame.

`StripExchange` keeps the transfers OFF the critical path: `post()` starts those of frame i
asynchronously and returns at once, so every rank goes straight on to trace frame i + 1; `wait()`
(called one frame later) hands rank 0 the assembled frame i.  Two sets of buffers alternate.
`SchedulerFeedback` runs the reference's block scheduler identically on every rank from all-gathered
(rows, time) pairs, also one frame behind.
"""
from __future__ import annotations


def device_for_rank(local_rank: int, visible: int) -> tuple[int, str]:
    """Which HIP device index a rank of a one-process-per-GPU launch uses, and why.  The reference has ONE process that sees every
    device and gives each a tracer (renderer/default.go:204-256); a launcher hands every rank either all GPUs -- then LOCAL_RANK
    names the rank's own -- or, masked per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES), exactly one, which is index 0 in
    EVERY rank.  Anything else (fewer visible devices than ranks, more than one) is oversubscribed: the ranks wrap around, and
    bench.py's `config.distinct_gpus` shows it."""
    if visible < 1:
        raise RuntimeError("no HIP device is visible to this rank")
    if local_rank < visible:
        return local_rank, "LOCAL_RANK"
    if visible == 1:
        return 0, "the one visible device (per-rank visibility mask, or ranks sharing a GPU)"
    return local_rank % visible, f"LOCAL_RANK modulo the {visible} visible devices (more ranks than devices)"


def gather_setup_errors(dist, rank: int, world: int, err: str, group=None) -> list[str]:
    """Every rank's set-up verdict ("" = fine) reaches every rank BEFORE the first collective that would otherwise hang on the
    rank that failed: one all_gather_object over the (bounded-timeout) control group.  Returns the non-empty messages, prefixed
    with their rank, in rank order -- the same list on every rank."""
    got = [None] * world
    dist.all_gather_object(got, err or "", group=group)
    return [f"rank {r}: {e}" for r, e in enumerate(got) if e]


def naive_rows(n_tracers: int, frame_h: int, speeds=None) -> list[int]:
    """tracer/scheduler.go:83-106 assignBlocksBasedOnSpeed (all speeds equal unless given)."""
    speeds = [1] * n_tracers if speeds is None else list(speeds)
    scaler = float(frame_h) / float(sum(speeds))
    rows = [int(max(1.0, float(s) * scaler)) for s in speeds]
    if sum(rows) < frame_h:
        rows[0] += frame_h - sum(rows)
    return rows


def fit_rows(rows: list[int], frame_h: int) -> list[int]:
    """The reference's perfect scheduler gives every tracer at least one row AFTER it has floored the shares
    (scheduler.go:70-76), so with very unequal speeds the blocks can add up to MORE than the frame (it only tops up when they
    add up to less); the reference's last block then runs off the frame.  Every rank applies the same fix to the same numbers:
    rows are taken back from the tallest blocks, one at a time.  A no-op for every assignment that fits."""
    rows = [int(v) for v in rows]
    while sum(rows) > frame_h and max(rows) > 1:
        rows[max(range(len(rows)), key=lambda i: (rows[i], -i))] -= 1
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

    def __init__(self, dist, rank: int, world: int, frame_w: int, frame_h: int, device, dst: int = 0, via_host: bool = False, depth: int = 2, group=None):
        import torch

        self.dist, self.rank, self.world, self.W, self.H, self.dst = dist, rank, world, frame_w, frame_h, dst
        self.via_host, self.group = via_host, group  # group: the process group the strips travel on (None = the default one)
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
                    ops.append(self.dist.P2POp(self.dist.irecv, wire[yy * W:(yy + rows[r]) * W], r, group=self.group))
                yy += rows[r]
        else:
            ops.append(self.dist.P2POp(self.dist.isend, wire[: rows[self.rank] * W], self.dst, group=self.group))
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

    def __init__(self, dist, rank: int, world: int, frame_h: int, device, kind: str = "perfect", group=None):
        import torch

        from . import host_api

        self.dist, self.rank, self.world, self.H, self.kind, self.group = dist, rank, world, frame_h, kind, group
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
            work = self.dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True)
        else:             # gloo (tests)
            out = [t.zeros(2, dtype=t.int64) for _ in range(self.world)]
            work = self.dist.all_gather(out, mine, group=self.group, async_op=True)
        self._pending.append((work, out, mine))

    def next_rows(self):
        """The rows of the NEXT frame: from the newest feedback that has been published at least one frame ago."""
        if self._sched is None:
            return self.rows
        while len(self._pending) > 1:  # everything but the frame just published has had a whole frame's time to complete
            work, out, _ = self._pending.pop(0)
            work.wait()
            got = out.cpu().tolist() if not isinstance(out, list) else [o.tolist() for o in out]  # (.cpu() orders itself behind the collective on the current stream)
            self.rows = fit_rows(self._sched.schedule(self.H, block_h=[g[0] for g in got], render_ns=[max(1, g[1]) for g in got]), self.H)
        return self.rows

    def drain(self):
        for work, out, _ in self._pending:
            work.wait()
        self._pending = []


class ShmMailbox:
    """The per-frame control message of `PeerExchange` -- four integers per rank -- through SHARED MEMORY instead of a gloo all_gather, for
    ranks that live on one host (the launch contract: N GPUs of ONE node).  What an asynchronous torch.distributed collective costs the
    calling thread depends on the host: ~18 us to start and ~10 us to complete on the GPU box's EPYC, ~0.2 ms + ~0.05 ms in an 8-CPU
    container -- next to an 8-GPU headline frame of ~1.8 ms per rank, in sequence with the Trace on every rank
    (profiles/r06_exchange_overhead.txt).  Here a post is three numpy stores (1.4 us) and a completed wait three loads (8 us), with no
    worker thread of a communication library in between.

    One file in /dev/shm, created by the primary and mapped by every rank: SLOTS x world records of six int64 --
    [v0, v1, v2, v3, check, sequence + 1].  post(seq, values) writes the payload and the check word, THEN the sequence word (x86 keeps
    stores in order; the check word = xor of the payload and the sequence catches a torn read on anything weaker: such a record just
    reads as "not there yet"); wait(seq) polls until every rank's record of that sequence number is in place.  A record is reused
    SLOTS posts later; by the exchange's own ordering (finish(f - 1) before post(f)) every rank has read sequence f before any rank
    posts f + 2, so four slots leave a margin.  A wait that does not complete within `timeout_s` raises instead of hanging (a rank
    that died), like the bounded gloo group it replaces.  Set-up is collective over `dist` and all-or-nothing: if any rank cannot
    map the file (different hosts or containers, no /dev/shm) every rank keeps the gloo message."""

    SLOTS = 4

    def __init__(self, rank: int, world: int, path: str, create: bool, timeout_s: float = 120.0):
        import numpy as np

        self.rank, self.world, self.path, self.timeout_s, self.owner = rank, world, path, timeout_s, create
        n = self.SLOTS * world * 6
        if create:
            import atexit

            with open(path, "wb") as f:
                f.write(b"\0" * (n * 8))
            atexit.register(self.close)      # (a run that ends on an exception must not leave the file behind)
        import mmap

        with open(path, "r+b") as f:
            self._mm = mmap.mmap(f.fileno(), n * 8)
        self.a = np.frombuffer(self._mm, dtype=np.int64).reshape(self.SLOTS, world, 6)   # (a plain ndarray: np.memmap pays a Python call per slice)

    @staticmethod
    def setup(dist, rank: int, world: int, group=None, primary: int = 0, timeout_s: float = 120.0):
        """Collective.  Returns a mailbox on every rank, or None on every rank."""
        import os
        import socket
        import uuid

        def host_id():
            try:
                with open("/proc/sys/kernel/random/boot_id") as f:
                    return socket.gethostname() + ":" + f.read().strip()
            except OSError:
                return socket.gethostname()

        name = [None]
        box = None
        if rank == primary:
            path = f"/dev/shm/polaris_ctl_{os.getpid()}_{uuid.uuid4().hex[:12]}"
            try:
                box = ShmMailbox(rank, world, path, True, timeout_s)
                name = [(path, host_id())]
            except OSError:
                name = [None]
        dist.broadcast_object_list(name, src=primary, group=group)
        ok = box is not None
        if rank != primary and name[0] is not None and name[0][1] == host_id():
            try:
                box = ShmMailbox(rank, world, name[0][0], False, timeout_s)
                ok = True
            except (OSError, ValueError):
                ok = False
        oks = [None] * world
        dist.all_gather_object(oks, bool(ok), group=group)
        if all(oks):
            return box
        if box is not None:
            box.close()
        return None

    def post(self, seq: int, values) -> None:
        s, r = seq % self.SLOTS, self.rank
        v = [int(x) for x in values]
        rec = self.a[s, r]
        rec[0:4] = v
        rec[4] = v[0] ^ v[1] ^ v[2] ^ v[3] ^ (seq + 1)
        rec[5] = seq + 1            # last: the record is there

    def wait(self, seq: int):
        import time

        import numpy as np

        s = seq % self.SLOTS
        spins, deadline = 0, None
        while True:
            if bool((self.a[s, :, 5] == seq + 1).all()):
                got = np.array(self.a[s, :, :5])
                if bool(((got[:, 0] ^ got[:, 1] ^ got[:, 2] ^ got[:, 3] ^ (seq + 1)) == got[:, 4]).all()):
                    return got[:, :4].tolist()
            spins += 1
            if spins < 200:
                time.sleep(0)
                continue
            if deadline is None:
                deadline = time.monotonic() + self.timeout_s
            elif time.monotonic() > deadline:
                rec = np.array(self.a[s])
                missing = [r for r in range(self.world) if int(rec[r, 5]) != seq + 1 or int(rec[r, 0] ^ rec[r, 1] ^ rec[r, 2] ^ rec[r, 3] ^ (seq + 1)) != int(rec[r, 4])]
                raise RuntimeError(f"control message {seq}: nothing from rank(s) {missing} within {self.timeout_s:.0f} s")
            time.sleep(50e-6)

    def close(self):
        import os

        path, self.a = self.path, None
        self._mm = None             # (unmapped when the last view goes)
        if self.owner and path:
            try:
                os.unlink(path)
            except OSError:
                pass
        self.path = ""


class PeerExchange:
    """Row blocks merged into the primary by PEER READS of every rank's trace accumulator, one frame behind the tracing.

    `port` is the tracer side (bench.py: HipPort over the C ABI; the CPU test: a shared-memory stand-in):
        export(depth) -> bytes         this rank's ring, as a blob another process can open     polaris_hip_ipc_export
        open(blob) -> peer             (primary) map a peer's ring                               polaris_hip_ipc_open
        close(peer)                                                                              polaris_hip_ipc_close
        slot() -> int                  ring slot the last Trace wrote                            polaris_hip_trace_slot
        begin_frame()                  (primary) the Reset stage of the frame being assembled    polaris_hip_reset_frame
        merge_peer(peer, slot, y, h)   (primary) frame rows [y, y+h) += peer ring[slot] rows     polaris_hip_merge_ipc
        merge_self(slot, y, h)         (primary) its own block, from its own ring               polaris_hip_merge_slot
        end_frame()                    (primary) wait for the merges + tone-map                  polaris_hip_sync_framebuffer

    Per frame f every rank runs   rows = next_rows();  Trace(block of rows);  finish(ticket f-1);  ticket f = post(rows, ms).
    `post` starts an asynchronous all_gather of (f, slot, rows[rank], ns) -- the ONLY per-frame message, 32 bytes per rank
    over gloo; `finish` completes it and, on the primary, merges frame f-1 from the slots the ranks named.  The same numbers
    feed the reference's block scheduler (tracer/scheduler.go) identically on every rank: frame f+1 is scheduled from frame
    f-1's times, exactly the lag of SchedulerFeedback.

    Why a ring of depth 3 is enough (and 2 is not).  A rank's Trace f writes slot f % depth.  The primary reads the slots of
    frame g inside finish(g), and posts g+1 only AFTER that (finish before post) -- so "all_gather(g+1) has completed" tells
    every rank that the primary is done with frame g's slots (end_frame is synchronous: the merge kernels have run).  A
    rank completes all_gather(g+1) in ITS finish(g+1), during step g+2, i.e. before its Trace g+3 -- the first Trace that
    comes round to slot g % 3 again.  With depth 2, Trace g+2 would reuse the slot one step before that is known.
    """

    MIN_DEPTH = 3

    def __init__(self, dist, rank: int, world: int, frame_w: int, frame_h: int, port, scheduler: str = "naive", depth: int = 3, group=None, primary: int = 0,
                 control: str = "shm", control_timeout_s: float = 120.0):
        import torch

        from . import host_api

        assert depth >= self.MIN_DEPTH, "the ring must hold the frame being traced, the frame being merged and one in between"
        self.dist, self.group, self.rank, self.world, self.W, self.H = dist, group, rank, world, frame_w, frame_h
        self.port, self.depth, self.primary, self.kind = port, depth, primary, scheduler
        self._torch = torch
        self.rows = naive_rows(world, frame_h)
        self._sched = None
        if scheduler == "perfect" and world > 1:
            self._sched = host_api.Scheduler(host_api.PERFECT, [1] * world)
            self._sched.schedule(frame_h)  # its first frame: the naive split (scheduler.go:52-56)
        self._frame = 0
        self._finished = -1   # newest frame whose all_gather this rank has completed
        self._peers = {}
        self.opened = False
        # the per-frame message: "shm" = a shared-memory mailbox where every rank can map it (ranks of one host), else the gloo all_gather
        self._want_control, self._control_timeout, self._mail = control, control_timeout_s, None
        self.control = "gloo"

    def setup(self) -> bool:
        """Exchange the rings' IPC blobs (once); the primary maps every peer.  Returns True on every rank iff every ring was
        exported AND every mapping opened -- otherwise nothing stays open and the caller falls back to StripExchange (in this
        process, no re-launch).  A rank whose export fails (hipIpcGetMemHandle refused, no memory for the ring's extra slots)
        still takes part in the collective -- it gathers None plus the error text -- so nobody is left waiting in it."""
        try:
            mine = (self.port.export(self.depth), "")
        except Exception as e:
            mine = (None, f"rank {self.rank}: export failed: {e}")
        got = [None] * self.world
        self.dist.all_gather_object(got, mine, group=self.group)
        blobs = [g[0] for g in got]
        failed = [g[1] for g in got if g[0] is None]
        ok, why = not failed, "; ".join(failed)
        if ok and self.rank == self.primary:
            try:
                for r in range(self.world):
                    if r != self.primary:
                        self._peers[r] = self.port.open(blobs[r])
            except Exception as e:  # hipIpcOpenMemHandle refused (no peer access, containers without a shared /dev/kfd view ...)
                ok, why = False, str(e)
                for p in self._peers.values():
                    self.port.close(p)
                self._peers = {}
        verdict = [ok, why]
        self.dist.broadcast_object_list(verdict, src=self.primary, group=self.group)
        self.opened, self.why_not = bool(verdict[0]), verdict[1]
        if self.opened and self._want_control == "shm" and self.world > 1:
            self._mail = ShmMailbox.setup(self.dist, self.rank, self.world, group=self.group, primary=self.primary, timeout_s=self._control_timeout)
            self.control = "shm" if self._mail is not None else "gloo"
        return self.opened

    def peers(self) -> dict:
        """(primary) rank -> the peer handle port.open() returned for that rank's ring."""
        return dict(self._peers)

    def set_scheduler(self, kind: str):
        """Start over with another block scheduler (between two timed regions; nothing may be pending).  The mappings stay."""
        from . import host_api

        self.kind = kind
        self.rows = naive_rows(self.world, self.H)
        self._sched = None
        if kind == "perfect" and self.world > 1:
            self._sched = host_api.Scheduler(host_api.PERFECT, [1] * self.world)
            self._sched.schedule(self.H)

    def next_rows(self):
        return self.rows

    def post(self, rows, own_ms: float):
        """After Trace f and finish(f-1): announce (f, slot, rows[rank], time).  Returns the ticket for finish()."""
        t = self._torch
        f = self._frame
        assert self._finished >= f - 1, "finish(f-1) comes before post(f): the primary's post tells the ranks their slots of f-1 are free"
        self._frame += 1
        if self._mail is not None:
            self._mail.post(f, (f, int(self.port.slot()), int(rows[self.rank]), max(1, int(own_ms * 1e6))))
            return (f, None, None, list(rows))
        mine = t.tensor([f, int(self.port.slot()), int(rows[self.rank]), max(1, int(own_ms * 1e6))], dtype=t.int64)
        out = [t.zeros(4, dtype=t.int64) for _ in range(self.world)]
        work = self.dist.all_gather(out, mine, group=self.group, async_op=True) if self.world > 1 else None
        if work is None:
            out[0].copy_(mine)
        return (f, work, out, list(rows))

    def wait(self, ticket):
        """Complete frame f's message (blocks until the SLOWEST rank has posted frame f: not this rank's work) and feed the
        block scheduler.  Returns what merge() needs."""
        f, work, out, rows = ticket
        if out is None:
            got = self._mail.wait(f)
        else:
            if work is not None:
                work.wait()
            got = [o.tolist() for o in out]
        assert all(g[0] == f for g in got), f"ranks disagree about the frame number: {got}"
        assert [g[2] for g in got] == rows, f"ranks disagree about the rows of frame {f}: {got} vs {rows}"
        self._finished = f
        if self._sched is not None:
            self.rows = fit_rows(self._sched.schedule(self.H, block_h=[g[2] for g in got], render_ns=[g[3] for g in got]), self.H)
        return got, rows

    def merge(self, got, rows) -> float:
        """The primary merges the frame: every block read where it lies (Reset stage, one merge per block, tone-map).  Returns
        the seconds this took -- the primary's own work on top of its Trace, the part of finish() a caller may bill to the
        rank's render time (the wait for the other ranks' messages is not); 0.0 on the other ranks."""
        if self.rank != self.primary:
            return 0.0
        import time

        t = time.perf_counter()
        self.port.begin_frame()
        y = 0
        for r in range(self.world):
            if r == self.primary:
                self.port.merge_self(got[r][1], y, rows[r])
            else:
                self.port.merge_peer(self._peers[r], got[r][1], y, rows[r])
            y += rows[r]
        self.port.end_frame()
        return time.perf_counter() - t

    def finish(self, ticket) -> float:
        """wait() + merge(); returns merge()'s seconds."""
        return self.merge(*self.wait(ticket))

    def close(self):
        for p in self._peers.values():
            self.port.close(p)
        self._peers = {}
        if self._mail is not None:
            self._mail.close()
            self._mail = None


class HipPort:
    """PeerExchange's tracer side over the C ABI (polaris_amd.tracer.HipTracer)."""

    def __init__(self, tracer, make_req):
        self.tr, self.make_req = tracer, make_req

    def export(self, depth): return self.tr.ipc_export(depth)
    def open(self, blob): return self.tr.ipc_open(blob)
    def close(self, peer): self.tr.ipc_close(peer)
    def slot(self): return self.tr.trace_slot()
    def begin_frame(self): self.tr.reset_frame()
    def merge_peer(self, peer, slot, y, h): self.tr.merge_ipc(peer, slot, self.make_req(y, h))
    def merge_self(self, slot, y, h): self.tr.merge_slot(self.tr, slot, self.make_req(y, h))
    def end_frame(self): self.tr.SyncFramebuffer(self.make_req(0, self.tr._H))
