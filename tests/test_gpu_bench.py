"""bench.py end to end on the GPU box: the single-GPU line carries the contract's fields, and the
N>1 flow (row blocks, gather to the primary, merge, tonemap) completes under torch.distributed --
two ranks sharing the one GPU over gloo, which exercises the same code path the driver runs over
RCCL on 8 GPUs.  Both run as subprocesses under a timeout so a collective mismatch cannot hang the
suite."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


SMALL = ["--width", "128", "--height", "97", "--spp", "8", "--steps", "2", "--warmup", "1"]  # 97 rows: uneven blocks [49, 48]


def test_bench_single_gpu_line(built):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *SMALL], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_complete(built):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL, "--backend", "gloo", "--same-device",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "row blocks [49, 48]" in d["config"]["workload"]
