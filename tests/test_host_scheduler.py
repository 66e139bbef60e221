"""Host layer above the C ABI (C++): the reference's scheduler known answers
(tracer/scheduler_test.go:16-20,48-55) through polaris_amd/host/scheduler.cpp, and the Python
restatement used by bench.py's row partition."""
import pytest


@pytest.fixture(scope="module")
def host(built):
    import subprocess, os
    from conftest import ROOT

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "polaris_amd", "host")])
    from polaris_amd import host_api

    return host_api


@pytest.mark.parametrize("speeds,frame_h,rows", [((1, 2), 10, [4, 6]), ((2, 1), 10, [7, 3]), ((1, 1000), 10, [1, 9])])
def test_naive_scheduler_known_answers(host, speeds, frame_h, rows):
    s = host.Scheduler(host.NAIVE, speeds)
    assert s.schedule(frame_h) == rows
    assert s.schedule(frame_h) == rows          # the naive assignment is computed once and kept (scheduler.go:24-30)
    from polaris_amd.distributed import naive_rows

    assert naive_rows(len(speeds), frame_h, speeds) == rows


def test_perfect_scheduler_known_answers(host):
    """scheduler_test.go:42-80: first call = naive (5,5); then times (1,5) -> (9,1); then (5,1) -> (7,3)."""
    s = host.Scheduler(host.PERFECT, (1, 1))
    rows = s.schedule(10, block_h=[0, 0], render_ns=[1, 5])
    assert rows == [5, 5]
    rows = s.schedule(10, block_h=rows, render_ns=[1, 5])
    assert rows == [9, 1]
    rows = s.schedule(10, block_h=rows, render_ns=[5, 1])
    assert rows == [7, 3]


def test_equal_speed_partitions():
    from polaris_amd.distributed import block_of, naive_rows

    assert naive_rows(8, 512) == [64] * 8
    assert naive_rows(8, 1080) == [135] * 8
    rows = naive_rows(3, 512)
    assert rows == [172, 170, 170] and sum(rows) == 512      # remainder to tracer 0
    assert [block_of(r, rows) for r in range(3)] == [(0, 172), (172, 170), (342, 170)]
    assert naive_rows(1, 7) == [7]
