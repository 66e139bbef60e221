"""What the per-frame control message of PeerExchange costs a rank (CPU only; run under torch.distributed.run):
    python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node 8 tests/tools/exchange_overhead.py shm|gloo
Every rank runs the frame loop of bench.py around a stand-in for the Trace that takes exactly 1.8 ms (an 8-GPU headline block) without
holding the GIL-free time hostage (sleep + a short spin), with a do-nothing tracer port; prints the wall time per frame and what post()
and finish() took.  (On a gpurun box keep it to 4 ranks: every process that imports torch counts as a user of the GPU there, and the box
admits six.)"""
import datetime
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch.distributed as dist  # noqa: E402

from polaris_amd.distributed import PeerExchange  # noqa: E402


class Port:
    n = 0

    def export(self, depth): return b"x"
    def open(self, blob): return 1
    def close(self, p): pass
    def slot(self): return self.n % 3
    def begin_frame(self): pass
    def merge_peer(self, p, slot, y, h): pass
    def merge_self(self, slot, y, h): pass
    def end_frame(self): pass


def main():
    os.dup2(2, 1)   # (gloo's connection messages)
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    rank, world = dist.get_rank(), dist.get_world_size()
    px = PeerExchange(dist, rank, world, 512, 512, Port(), scheduler="naive", control=sys.argv[1])
    assert px.setup() and px.control == sys.argv[1]
    pending, tp, tf, N, t_start = [], 0.0, 0.0, 400, None
    for f in range(N + 20):
        if f == 20:
            t_start = time.perf_counter()
        rows = px.next_rows()
        t_end = time.perf_counter() + 0.0018
        time.sleep(0.0016)
        while time.perf_counter() < t_end:
            pass
        px.port.n += 1
        t0 = time.perf_counter()
        while pending:
            px.finish(pending.pop(0))
        t1 = time.perf_counter()
        pending.append(px.post(rows, 1.8))
        t2 = time.perf_counter()
        if f >= 20:
            tf += t1 - t0
            tp += t2 - t1
    while pending:
        px.finish(pending.pop(0))
    wall = (time.perf_counter() - t_start) / N * 1e6
    dist.barrier()
    if rank in (0, world - 1):
        print(f"{sys.argv[1]:5s} {world} ranks, rank {rank}: {wall:.0f} us per frame around a 1800 us Trace; post {tp / N * 1e6:.0f} us, finish {tf / N * 1e6:.0f} us "
              f"(finish includes waiting for the slowest rank's message)", file=sys.stderr, flush=True)
    px.close()
    dist.destroy_process_group()


main()
