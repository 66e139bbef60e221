"""(Second half of the file: PeerExchange, the default exchange -- peer reads of every rank's ring -- with shared-memory
segments standing in for the HIP IPC mappings.)

The N>1 path of bench.py on CPU: 2 ranks over gloo partition the frame by row block, each renders
its block (the CPU oracle stands in for the GPU tracer -- it is the checker, used here as test
infrastructure only), and the strips travel to rank 0 through polaris_amd.distributed.StripExchange
-- the SAME classes, used the same way (next_rows -> trace -> publish -> post -> wait one frame later),
that bench.py runs over RCCL.  Several frames with different seeds are in flight one behind the other;
every assembled frame must equal the per-block oracle results bit for bit (uneven blocks: 31 rows ->
[16, 15]; with the perfect scheduler the rows change from frame to frame with the ranks' made-up times)."""
import os
import socket
import sys

import numpy as np

from conftest import ROOT

W, H, SPP, B, FRAMES = 40, 31, 2, 4, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path, scheduler):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import SchedulerFeedback, StripExchange, block_of

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    ex = StripExchange(dist, rank, world, W, H, "cpu")
    fb = SchedulerFeedback(dist, rank, world, H, "cpu", kind=scheduler)
    orc = ob.Oracle("oracle")
    frames, pending, all_rows = [], [], []

    def finish(ticket):
        parts = ex.wait(ticket)
        if rank == 0:
            frame = np.zeros((H, W, 4), np.float32)
            for y, h, t in parts:  # what bench.py hands to polaris_hip_merge_device
                frame[y:y + h] += t.numpy().reshape(-1, 4)[: h * W].reshape(h, W, 4)
            frames.append(frame)

    for f in range(FRAMES):
        rows = fb.next_rows()
        all_rows.append(list(rows))
        by, bh = block_of(rank, rows)
        seeds = scenes.make_seeds(SPP, B, base=100 + f)
        acc, _, _ = orc.trace(sc, ob.make_request(W, H, spp=SPP, bounces=B, block_y=by, block_h=bh), seeds)
        fb.publish(rows, (3.0 if rank == 1 else 1.0) * bh)  # made-up times: rank 1 takes three times as long per row

        def fill(strip, acc=acc, by=by, bh=bh):
            strip[: bh * W].copy_(torch.from_numpy(np.ascontiguousarray(acc[by:by + bh].reshape(-1, 4))))

        ticket = ex.post(fill, rows)
        while pending:
            finish(pending.pop(0))
        pending.append(ticket)
    while pending:
        finish(pending.pop(0))
    fb.drain()
    if rank == 0:
        np.save(out_path, np.stack(frames))
        np.save(out_path + ".rows.npy", np.array(all_rows))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("scheduler", ["naive", "perfect"])
def test_two_rank_row_block_gather(built, tmp_path, scheduler):
    import torch.multiprocessing as mp

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import block_of, naive_rows

    out = str(tmp_path / "frames.npy")
    mp.spawn(_worker, args=(2, _free_port(), out, scheduler), nprocs=2, join=True)
    frames = np.load(out)
    all_rows = np.load(out + ".rows.npy").tolist()
    assert frames.shape == (FRAMES, H, W, 4)
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    assert all_rows[0] == naive_rows(2, H) == [16, 15] and all_rows[1] == [16, 15]   # feedback arrives two frames later
    if scheduler == "naive":
        assert all(r == [16, 15] for r in all_rows)
    else:  # rank 1 reported three times the time per row: from frame 2 on it gets about a quarter of the frame (scheduler.go:50-80)
        assert all(sum(r) == H and min(r) >= 1 for r in all_rows)
        assert all_rows[2][1] < 12 and all_rows[-1][1] <= 9, all_rows
    orc = ob.Oracle("oracle")
    for f in range(FRAMES):
        seeds = scenes.make_seeds(SPP, B, base=100 + f)
        expect = np.zeros((H, W, 3), np.float32)
        for r in range(2):
            by, bh = block_of(r, all_rows[f])
            a, _, _ = orc.trace(sc, ob.make_request(W, H, spp=SPP, bounces=B, block_y=by, block_h=bh), seeds)
            expect[by:by + bh] = a[by:by + bh, :, :3]
        assert np.array_equal(frames[f][..., :3].view(np.uint32), expect.view(np.uint32)), f


def test_naive_rows_is_the_reference_schedule():
    """tracer/scheduler_test.go:16-20 known answers (speeds -> rows at H=10) + the equal-speed split bench.py uses."""
    from polaris_amd.distributed import naive_rows

    assert naive_rows(2, 10, [1, 2]) == [4, 6]
    assert naive_rows(2, 10, [2, 1]) == [7, 3]
    assert naive_rows(2, 10, [1, 1000]) == [1, 9]
    assert naive_rows(8, 512) == [64] * 8 and naive_rows(8, 1080) == [135] * 8 and naive_rows(2, 97) == [49, 48]


def test_fit_rows_repairs_an_over_assigned_frame(built):
    """scheduler.go:70-76 clamps every share to one row AFTER flooring and only tops up a frame that is short: very unequal
    speeds over-assign it.  fit_rows takes rows back from the tallest blocks and leaves every assignment that fits alone."""
    from polaris_amd import host_api
    from polaris_amd.distributed import fit_rows

    s = host_api.Scheduler(host_api.PERFECT, [1] * 4)
    s.schedule(8)
    over = s.schedule(8, block_h=[2, 2, 2, 2], render_ns=[1, 1000, 1000, 1000])
    assert over == [7, 1, 1, 1]                      # the reference's arithmetic: 10 rows for an 8-row frame
    assert fit_rows(over, 8) == [5, 1, 1, 1]
    assert fit_rows([16, 15], 31) == [16, 15] and fit_rows([3, 3, 3], 8) == [2, 3, 3] and fit_rows([1, 1, 1], 2) == [1, 1, 1]


# ---- PeerExchange: the default exchange of bench.py (peer reads through IPC mappings) ---------------------------------------
class ShmPort:
    """PeerExchange's tracer side without a GPU: the ring is `depth` shared-memory segments (the role of the IPC-mapped trace
    accumulators), a "Trace" writes the block's rows into the next slot after POISONING the whole slot (what the clear at the
    start of polaris_hip_trace does to a slot the primary might still be reading), the primary reads peers' rows through its
    attachments.  A slot read too early or too late shows as NaNs or as another frame's rows."""

    def __init__(self, rank, W, H):
        self.rank, self.W, self.H = rank, W, H
        self.segs, self.views, self.pos, self.depth = [], [], 0, 1
        self.frames, self.frame = [], None
        self.attached = []

    def export(self, depth):
        from multiprocessing import shared_memory

        self.depth = depth
        for i in range(depth):
            seg = shared_memory.SharedMemory(create=True, size=self.H * self.W * 16)
            self.segs.append(seg)
            self.views.append(np.ndarray((self.H, self.W, 4), np.float32, buffer=seg.buf))
            self.views[-1][:] = 0
        return ",".join(seg.name for seg in self.segs).encode()

    def open(self, blob):
        from multiprocessing import shared_memory

        segs = [shared_memory.SharedMemory(name=n) for n in blob.decode().split(",")]
        self.attached.append(segs)
        return [np.ndarray((self.H, self.W, 4), np.float32, buffer=seg.buf) for seg in segs]

    def close(self, peer):
        pass

    def trace(self, acc, by, bh):
        self.pos = (self.pos + 1) % self.depth
        self.views[self.pos][:] = np.nan
        self.views[self.pos][by:by + bh] = acc[by:by + bh]

    def slot(self):
        return self.pos

    def begin_frame(self):
        self.frame = np.zeros((self.H, self.W, 4), np.float32)

    def merge_peer(self, peer, slot, y, h):
        self.frame[y:y + h] += peer[slot][y:y + h]

    def merge_self(self, slot, y, h):
        self.frame[y:y + h] += self.views[slot][y:y + h]

    def end_frame(self):
        self.frames.append(self.frame.copy())

    def release(self, unlink):
        self.views = []
        for segs in self.attached:
            for seg in segs:
                seg.close()
        for seg in self.segs:
            seg.close()
            if unlink:
                seg.unlink()


def _synthetic_frame(f, Hh):
    """What frame f must look like once assembled (fuzz mode): exact in float32, different in every pixel, channel and frame."""
    y, x, c = np.meshgrid(np.arange(Hh, dtype=np.float32), np.arange(W, dtype=np.float32), np.arange(4, dtype=np.float32), indexing="ij")
    return (np.float32(f) * 4096.0 + y * 32.0 + x * 0.25 + c * 0.0625).astype(np.float32)


def _fuzz_worker(rank, world, port, out_path, scheduler, Hh, frames, control="shm"):
    """PeerExchange under a RANDOMISED schedule: every rank sleeps a seeded random time before and after
    every "Trace" of every frame (0 - 20 ms, most of them short, so that any rank may be the one running ahead or lagging in any
    frame), the ring is poisoned at the start of every Trace, and with the perfect scheduler the made-up times are random too, so the
    rows change every frame.  The traced block is a synthetic pattern: the protocol is what is under test, not the tracer."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import random
    import time

    import torch.distributed as dist

    from polaris_amd.distributed import PeerExchange, block_of

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    shm = ShmPort(rank, W, Hh)
    px = PeerExchange(dist, rank, world, W, Hh, shm, scheduler=scheduler, control=control)
    assert px.setup()
    assert px.control == control, (px.control, control)     # (the ranks of this test share a host: the mailbox must have been mapped)
    rng = random.Random(20261004 + 7919 * rank)
    delays = (0.0, 0.0, 0.0, 0.0005, 0.001, 0.002, 0.004, 0.008, 0.02)
    pending, all_rows = [], []
    for f in range(frames):
        rows = px.next_rows()
        all_rows.append(list(rows))
        by, bh = block_of(rank, rows)
        time.sleep(rng.choice(delays))
        shm.trace(_synthetic_frame(f, Hh), by, bh)
        time.sleep(rng.choice(delays))
        while pending:
            px.finish(pending.pop(0))
        pending.append(px.post(rows, rng.uniform(0.5, 4.0) * bh))   # made-up times: the perfect scheduler moves rows every frame
    while pending:
        px.finish(pending.pop(0))
    if rank == 0:
        np.save(out_path, np.stack(shm.frames))
        np.save(out_path + ".rows.npy", np.array(all_rows))
    dist.barrier()
    px.close()
    shm.release(unlink=False)
    dist.barrier()
    shm.release(unlink=True)
    dist.destroy_process_group()


def _peer_worker(rank, world, port, out_path, scheduler, slow_rank, H=H, fail_export_on=-1):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import time

    import torch.distributed as dist

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import PeerExchange, block_of

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    shm = ShmPort(rank, W, H)
    if rank == fail_export_on:
        def refuse(depth):
            raise RuntimeError("hipIpcGetMemHandle: refused (test)")
        shm.export = refuse
    px = PeerExchange(dist, rank, world, W, H, shm, scheduler=scheduler)
    opened = px.setup()
    if fail_export_on >= 0:  # every rank must learn of the failure (nobody hangs in the collective) and nothing stays open
        assert not opened and f"rank {fail_export_on}: export failed" in px.why_not and not px._peers, (opened, px.why_not)
        if rank == 0:
            np.save(out_path, np.zeros(1))
        dist.barrier()
        shm.release(unlink=True)
        dist.destroy_process_group()
        return
    assert opened
    orc = ob.Oracle("oracle")
    pending, all_rows, merge_s = [], [], []
    for f in range(FRAMES + 2):
        rows = px.next_rows()
        all_rows.append(list(rows))
        by, bh = block_of(rank, rows)
        acc, _, _ = orc.trace(sc, ob.make_request(W, H, spp=SPP, bounces=B, block_y=by, block_h=bh), scenes.make_seeds(SPP, B, base=100 + f))
        shm.trace(acc, by, bh)
        if rank == slow_rank:
            time.sleep(0.15)   # the other ranks run ahead as far as the protocol lets them
        while pending:
            merge_s.append(px.finish(pending.pop(0)))
        # made-up times: rank 1 takes three times as long per row.  (bench.py bills a rank its Trace + the seconds finish()
        # RETURNS -- the primary's merges -- never the wait for the slowest rank inside finish().)
        pending.append(px.post(rows, (3.0 if rank == 1 else 1.0) * bh))
    while pending:
        merge_s.append(px.finish(pending.pop(0)))
    # the held-back rank costs everybody 0.15 s per frame in wait(); what finish() returns -- numpy adds of a 40-pixel-wide
    # frame on the primary, nothing elsewhere -- must not contain it (summed over the frames, so that one scheduling hiccup
    # of an oversubscribed test machine does not matter: with the wait inside, the sum would be >= 0.15 s x frames)
    assert all(t == 0.0 for t in merge_s) if rank != 0 else sum(merge_s) < 0.5 * 0.15 * len(merge_s), merge_s
    if rank == 0:
        np.save(out_path, np.stack(shm.frames))
        np.save(out_path + ".rows.npy", np.array(all_rows))
    dist.barrier()
    px.close()
    shm.release(unlink=False)
    dist.barrier()
    shm.release(unlink=True)
    dist.destroy_process_group()


def _check_peer_frames(out, world, H, scheduler):
    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import block_of, naive_rows

    frames = np.load(out)
    all_rows = np.load(out + ".rows.npy").tolist()
    n = FRAMES + 2
    assert frames.shape == (n, H, W, 4) and np.isfinite(frames).all()
    first = naive_rows(world, H)
    assert all_rows[0] == first and all_rows[1] == first   # frame f + 1 is scheduled from frame f - 1
    if scheduler == "naive":
        assert all(r == first for r in all_rows)
    else:  # rank 1 reported three times the time per row: it ends up with about a third of an equal share (scheduler.go:50-80)
        assert all(len(r) == world and sum(r) == H and min(r) >= 1 for r in all_rows), all_rows
        assert all_rows[2][1] < first[1] and all_rows[-1][1] <= max(1, (first[1] * 3) // 5), all_rows
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    orc = ob.Oracle("oracle")
    for f in range(n):
        seeds = scenes.make_seeds(SPP, B, base=100 + f)
        expect = np.zeros((H, W, 3), np.float32)
        for r in range(world):
            by, bh = block_of(r, all_rows[f])
            a, _, _ = orc.trace(sc, ob.make_request(W, H, spp=SPP, bounces=B, block_y=by, block_h=bh), seeds)
            expect[by:by + bh] = a[by:by + bh, :, :3]
        assert np.array_equal(frames[f][..., :3].view(np.uint32), expect.view(np.uint32)), f
    return all_rows


@pytest.mark.parametrize("scheduler,slow_rank", [("naive", 0), ("naive", 1), ("perfect", 0)])
def test_two_rank_peer_read_exchange(built, tmp_path, scheduler, slow_rank):
    """Two processes, the primary reads the other's rows out of a shared ring one frame behind the tracing -- with one rank held
    back so that the other runs as far ahead as the protocol allows (a ring slot reused before the primary has read it would
    put NaNs or another frame's rows into the assembled frame).  Every frame == the per-block oracle, bit for bit."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "frames.npy")
    mp.spawn(_peer_worker, args=(2, _free_port(), out, scheduler, slow_rank), nprocs=2, join=True)
    all_rows = _check_peer_frames(out, 2, H, scheduler)
    assert all_rows[0] == [16, 15]


@pytest.mark.parametrize("world,height,scheduler,slow_rank", [(8, 61, "naive", 3), (8, 64, "perfect", 0), (4, 31, "perfect", 3)])
def test_four_and_eight_rank_peer_read_exchange(built, tmp_path, world, height, scheduler, slow_rank):
    """What the driver's SCALE run launches at N = 4 and N = 8, without a GPU: eight (four) processes, seven (three) peer rings
    mapped by the primary, eight merges per frame, the all_gather_object of eight blobs, the block scheduler over eight
    (rows, ns) pairs (renderer/default.go:127-136,188-191; tracer/scheduler.go:50-106) -- a frame height the ranks do not
    divide (61 rows -> [12, 7, 7, 7, 7, 7, 7, 7]: the remainder goes to tracer 0, scheduler.go:98-104) and one they do, one
    rank in the MIDDLE held back so that seven peers run ahead of it as far as the ring's depth-3 argument allows.  Every
    assembled frame == the per-block oracle bit for bit."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "frames.npy")
    mp.spawn(_peer_worker, args=(world, _free_port(), out, scheduler, slow_rank, height), nprocs=world, join=True)
    all_rows = _check_peer_frames(out, world, height, scheduler)
    if (world, height) == (8, 61):
        assert all_rows[0] == [12, 7, 7, 7, 7, 7, 7, 7]


def test_a_failed_export_on_one_rank_reaches_every_rank(built, tmp_path):
    """PeerExchange.setup() with the LAST rank's export raising (hipIpcGetMemHandle refused, no memory for the extra ring slots):
    that rank still takes part in the collective, every rank gets verdict False with the reason, nothing stays mapped -- the
    caller (bench.py) then falls back to the strip transfers on every rank together."""
    import torch.multiprocessing as mp

    out = str(tmp_path / "none.npy")
    mp.spawn(_peer_worker, args=(3, _free_port(), out, "naive", -1, H, 2), nprocs=3, join=True)
    assert os.path.exists(out)


def test_peer_exchange_refuses_a_ring_that_is_too_short():
    from polaris_amd.distributed import PeerExchange

    with pytest.raises(AssertionError, match="ring"):
        PeerExchange(None, 0, 2, 8, 8, None, depth=2)


@pytest.mark.parametrize("scheduler,control", [("naive", "shm"), ("perfect", "shm"), ("perfect", "gloo")])
def test_eight_rank_exchange_under_random_per_frame_delays(built, tmp_path, scheduler, control):
    """100 frames x 8 ranks, a seeded random delay per rank PER FRAME on both sides of every Trace (not one rank held back by a
    constant): whoever runs ahead or lags changes from frame to frame, so a ring slot reused one frame too early, a merge from the
    wrong slot, or ranks disagreeing about the rows would show as NaNs (the poisoned slot) or as another frame's pattern.  Every
    assembled frame must be exactly the frame's pattern (renderer/default.go:127-136,188-191: a worker per tracer, merges as they
    finish).  The per-frame message travels through the shared-memory mailbox (ranks of one host: the default) or the gloo all_gather."""
    import glob

    import torch.multiprocessing as mp

    world, Hh, frames = 8, 61, 100
    out = str(tmp_path / "fuzz.npy")
    before = set(glob.glob("/dev/shm/polaris_ctl_*"))
    mp.spawn(_fuzz_worker, args=(world, _free_port(), out, scheduler, Hh, frames, control), nprocs=world, join=True)
    assert set(glob.glob("/dev/shm/polaris_ctl_*")) <= before          # the primary removed its mailbox
    got = np.load(out)
    all_rows = np.load(out + ".rows.npy").tolist()
    assert got.shape == (frames, Hh, W, 4) and np.isfinite(got).all()
    assert all(len(r) == world and sum(r) == Hh and min(r) >= 1 for r in all_rows)
    if scheduler == "perfect":
        assert len({tuple(r) for r in all_rows}) > 10          # the rows really moved
    for f in range(frames):
        assert np.array_equal(got[f], _synthetic_frame(f, Hh)), f


def test_mailbox_records_timeouts_and_torn_reads(tmp_path):
    """ShmMailbox on its own (two views of one file in this process): a record is there once its sequence word is, a record whose check
    word does not fit (a torn read on a weakly ordered machine) counts as not there, slots are reused four posts later, and a wait for a
    rank that never posts raises after the timeout instead of hanging."""
    from polaris_amd.distributed import ShmMailbox

    path = str(tmp_path / "box")
    a = ShmMailbox(0, 2, path, True, timeout_s=0.3)
    b = ShmMailbox(1, 2, path, False, timeout_s=0.3)
    for seq in range(11):                                   # wraps the four slots twice
        a.post(seq, (seq, 1, 30 + seq, 10 ** 12 + seq))
        b.post(seq, (seq, 2, 31 - seq, 7))
        assert a.wait(seq) == b.wait(seq) == [[seq, 1, 30 + seq, 10 ** 12 + seq], [seq, 2, 31 - seq, 7]]
    a.post(11, (11, 0, 1, 2))
    with pytest.raises(RuntimeError, match=r"nothing from rank\(s\) \[1\]"):
        a.wait(11)
    b.post(11, (11, 0, 3, 4))
    b.a[11 % 4, 1, 2] = 99                                  # payload and check word no longer agree
    with pytest.raises(RuntimeError, match="nothing from rank"):
        a.wait(11)
    b.post(11, (11, 0, 3, 4))
    assert a.wait(11) == [[11, 0, 1, 2], [11, 0, 3, 4]]
    b.close()
    a.close()
    assert not os.path.exists(path)


def test_device_for_rank_covers_the_launch_shapes():
    """Every rank sees all GPUs (LOCAL_RANK picks), every rank masked to one device (index 0 everywhere), more ranks than
    devices (wrap around -- bench.py's distinct_gpus then says so), no device at all."""
    from polaris_amd.distributed import device_for_rank

    assert [device_for_rank(r, 8)[0] for r in range(8)] == list(range(8)) and device_for_rank(5, 8)[1] == "LOCAL_RANK"
    assert [device_for_rank(r, 1)[0] for r in range(8)] == [0] * 8
    assert device_for_rank(0, 1)[1] == "LOCAL_RANK" and "one visible device" in device_for_rank(3, 1)[1]
    assert [device_for_rank(r, 2)[0] for r in range(4)] == [0, 1, 0, 1] and "more ranks than devices" in device_for_rank(3, 2)[1]
    with pytest.raises(RuntimeError, match="no HIP device"):
        device_for_rank(0, 0)


def _setup_error_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import json

    import torch.distributed as dist

    from polaris_amd.distributed import gather_setup_errors

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    a = gather_setup_errors(dist, rank, world, "")                                                # everybody fine
    b = gather_setup_errors(dist, rank, world, "hipErrorNoDevice: boom" if rank == 1 else "")       # one rank failed
    json.dump([a, b], open(os.path.join(out_dir, f"r{rank}.json"), "w"))
    dist.destroy_process_group()


def test_setup_errors_reach_every_rank(tmp_path):
    import json

    import torch.multiprocessing as mp

    mp.spawn(_setup_error_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        a, b = json.load(open(tmp_path / f"r{r}.json"))
        assert a == [] and b == ["rank 1: hipErrorNoDevice: boom"]
