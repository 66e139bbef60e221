"""Soak run (GPU): many Trace calls with changing frame sizes, blocks, scenes and options on ONE
handle; device memory must return to where it started and every call must succeed.
    python tests/tools/soak.py [iterations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes  # noqa: E402

from conftest import make_hip_tracer  # noqa: E402
from oracle import pybind as ob  # noqa: E402
from polaris_amd import scenes  # noqa: E402
from polaris_amd.tracer import ChangeType, UpdateMode  # noqa: E402


_hip = None


def free_mb():
    """hipMemGetInfo through the HIP runtime the tracer library itself is linked against (no torch here:
    a second HIP runtime in the process would not see the device)."""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert _hip.hipDeviceSynchronize() == 0
    assert _hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value / 2**20


def run(iters=200):
    rng = np.random.default_rng(1)
    names = ["cornell", "sphere", "cubes", "materials", "transformed", "material-ball-small", "terrain-small", "instanced-small"]
    built = {n: scenes.SCENES[n]() for n in names}
    tr = make_hip_tracer(built["cornell"], 64, 64)
    start = None
    worst = 0.0
    for it in range(iters):
        n = names[int(rng.integers(0, len(names)))]
        W, H = int(rng.integers(8, 300)), int(rng.integers(8, 200))
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, built[n])
        tr.UpdateState(UpdateMode.Asynchronous, ChangeType.CameraData, built[n])
        tr.set_option("overlap", int(rng.integers(1, 9)))
        tr.set_option("exact_accumulate", int(rng.integers(0, 4) == 0))
        spp, B = int(rng.integers(1, 40)), int(rng.integers(1, 7))
        by = int(rng.integers(0, H))
        bh = int(rng.integers(1, H - by + 1))
        req = ob.make_request(W, H, spp=spp, bounces=B, rr=int(rng.integers(1, B + 2)), block_y=by, block_h=bh)
        ring = int(rng.integers(0, 3)) == 0          # every third iteration: the trace accumulator as a ring (what a peer process maps)
        if ring:
            depth = int(rng.integers(1, 5))
            blob = tr.ipc_export(depth)
            assert len(blob) == 576
        tr.Trace(req, scenes.make_seeds(spp, B, base=it))
        if ring and int(rng.integers(0, 2)):         # a second Trace moves on to the next slot; the first frame's slot stays readable
            first = tr.trace_slot()
            tr.Trace(req, scenes.make_seeds(spp, B, base=it + 1))
            assert tr.trace_slot() == (first + 1) % depth
        if ring:
            tr.reset_frame()
            tr.merge_slot(tr, tr.trace_slot(), req)
        else:
            tr.MergeOutput(tr, req)
        tr.SyncFramebuffer(req)
        acc = tr.read_accumulator(0)
        assert np.isfinite(acc).all(), (it, n)
        if ring:
            frame = tr.read_accumulator(1)
            assert np.array_equal(frame[by:by + bh, :, :3], acc[by:by + bh, :, :3]), (it, n)   # the merged rows are the newest slot's
        if it == 20:
            start = free_mb()
        if start is not None:
            worst = max(worst, start - free_mb())
    end = free_mb()
    tr.Close()
    print(f"soak: {iters} iterations ok; free memory {start:.0f} MiB after warm-up, {end:.0f} MiB at the end, largest dip {worst:.0f} MiB")
    assert start - end < 512, "device memory keeps growing"
    return start, end, worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
