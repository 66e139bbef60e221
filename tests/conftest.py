"""pytest configuration: the `gpu` marker, library builds and shared helpers.

`-m "not gpu"`: oracle vs golden vectors (and vs the compiled reference where /root/reference
exists), host logic, C-ABI surface.  `-m gpu`: parity of the HIP path through the C ABI.
Nothing here reads /root/reference except the optional compiled-reference build.
"""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from polaris_amd.hostinfo import size_openmp  # noqa: E402

size_openmp()  # before any OpenMP library (the oracle, the C++ host layer) is loaded: one thread per CPU we may really use


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build the product library and the CPU oracle once per session (no-op when up to date)."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import pybind as ob

    return ob.Oracle("oracle")


@pytest.fixture(scope="session")
def ref_pm(built):
    from oracle import pybind as ob

    if not ob.available("ref_pm"):
        pytest.skip("compiled reference (oracle/_ref) not built: /root/reference is absent on this machine")
    return ob.Oracle("ref_pm")


def golden_files():
    return sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))


def load_golden(path):
    from oracle import pybind as ob
    from polaris_amd import scene_io

    d = np.load(path)
    sc = scene_io.scene_from_dict(d)
    W, H, spp, B, rr, by, bh = [int(v) for v in d["req"]]
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=rr, block_y=by, block_h=bh)
    return d, sc, req


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def make_hip_tracer(sc, W, H, device=0, **options):
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    tr = HipTracer("test", device)
    tr.Init()
    for k, v in options.items():  # before the upload: max_leaf_tris shapes the scene layout
        tr.set_option(k, v)
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
    return tr
