"""The N>1 path of bench.py on CPU: 2 ranks over gloo partition the frame by row block, each
renders its block (the CPU oracle stands in for the GPU tracer -- it is the checker, used here as
test infrastructure only), rank 0 gathers the strips and assembles the frame accumulator.  The
assembled frame must equal the per-block oracle results bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import block_of, gather_strips, naive_rows

    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, H, spp, B = 40, 30, 2, 4
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    seeds = scenes.make_seeds(spp, B)
    rows = naive_rows(world, H)
    by, bh = block_of(rank, rows)
    acc, _, _ = ob.Oracle("oracle").trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
    strip = torch.from_numpy(np.ascontiguousarray(acc[by:by + bh].reshape(-1, 4)))
    strips = gather_strips(strip, rows, W, dist, rank)
    if rank == 0:
        frame = np.zeros((H, W, 4), np.float32)
        y = 0
        for r in range(world):
            frame[y:y + rows[r]] += strips[r].numpy().reshape(rows[r], W, 4)
            y += rows[r]
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_block_gather(built, tmp_path):
    import torch.multiprocessing as mp

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.distributed import block_of, naive_rows

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frame = np.load(out)
    W, H, spp, B = 40, 30, 2, 4
    sc = scenes.SCENES["cornell-diffuse"](W / H)
    seeds = scenes.make_seeds(spp, B)
    rows = naive_rows(2, H)
    expect = np.zeros((H, W, 3), np.float32)
    orc = ob.Oracle("oracle")
    for r in range(2):
        by, bh = block_of(r, rows)
        a, _, _ = orc.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
        expect[by:by + bh] = a[by:by + bh, :, :3]
    assert np.array_equal(frame[..., :3].view(np.uint32), expect.view(np.uint32))
