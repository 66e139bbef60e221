"""The host frame loop under the THREAD SANITIZER: polaris_amd/host/renderer.cpp (one worker thread per
tracer, Trace -> primary.MergeOutput(tracer) from the workers, SyncFramebuffer on the main thread: renderer/default.go:106-196),
hip_tracer.cpp (change buffer, seed draws) and scheduler.cpp, compiled with -fsanitize=thread and linked against a CPU mock of the C
ABI that keeps the library's locking protocol (tests/tools/mock_polaris_hip.cpp: reset epochs, merges under the merge lock only,
random sleeps in Trace) -- plus the same loop over plain mock tracers (the reference's own pattern, tracer/scheduler_test.go:82-123),
which takes renderer.cpp's other ordering branch.  500 frames x 8 tracers x both block schedulers each, 20 progressive frames, camera
updates queued asynchronously from a second thread: zero sanitizer reports, every frame's accumulator exactly right."""
import os
import subprocess

from conftest import ROOT


def _build():
    host = os.path.join(ROOT, "polaris_amd", "host")
    tools = os.path.join(ROOT, "tests", "tools")
    out = os.path.join(ROOT, "tests", "_build", "renderer_tsan")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(host, f) for f in ("scheduler.cpp", "hip_tracer.cpp", "renderer.cpp")] + [os.path.join(tools, f) for f in ("mock_polaris_hip.cpp", "renderer_tsan.cpp")]
    deps = srcs + [os.path.join(host, f) for f in ("renderer.hpp", "hip_tracer.hpp", "tracer.hpp")] + [os.path.join(ROOT, "include", "polaris_hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-pthread", "-Wall", "-Wextra",
                               "-I" + os.path.join(ROOT, "include"), "-I" + host, *srcs, "-lz", "-o", out])
    return out


def test_frame_loop_has_no_data_race_and_every_frame_is_right():
    exe = _build()
    env = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=0 second_deadlock_stack=1")
    # the sanitizer is alive in this binary: an unsynchronised counter must be reported (exit code 66)
    canary = subprocess.run([exe, "--canary"], capture_output=True, text=True, timeout=120, env=env)
    assert canary.returncode == 66 and "ThreadSanitizer: data race" in canary.stderr
    out = subprocess.run([exe, "500", "8", "61"], capture_output=True, text=True, timeout=300, env=env)
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])
    assert "every frame right" in out.stdout
    # another shape: three tracers, a frame their count does not divide, more frames (the perfect scheduler's rows keep moving)
    out = subprocess.run([exe, "800", "3", "31"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]
