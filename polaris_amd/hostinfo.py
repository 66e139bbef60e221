"""How many CPUs this process may actually use.

`os.cpu_count()` reports every logical CPU of the host; a container is usually granted a share of them through the
cgroup CPU controller (the GPU boxes show 256 CPUs and grant 16).  OpenMP code that starts one thread per visible CPU
then runs 256 spinning threads on 16 CPUs' worth of time.  Used by bench.py (CPU-baseline leg) and the test suite to size
`OMP_NUM_THREADS` before any OpenMP runtime is loaded.
"""
from __future__ import annotations

import math
import os


def effective_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.floor(quota)))
    return max(1, n)


def size_openmp() -> int:
    """Set OMP_NUM_THREADS (unless the caller's environment already does) to the CPUs this process may use, and make idle
    OpenMP threads sleep instead of spinning.  Call before the first OpenMP library is loaded.  Returns the thread count."""
    cpus = effective_cpus()
    raw = os.environ.setdefault("OMP_NUM_THREADS", str(cpus))
    try:  # OpenMP allows a list ("8,2": one entry per nesting level); the outermost level is what sizes the team here
        n = int(raw.split(",")[0].strip())
    except ValueError:
        n = 0
    if n < 1:
        n = cpus
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    return n
