"""ctypes binding of polaris_amd/lib/libpolaris_host.so: the C++ host layer (tracer.Tracer mirror,
Naive/Perfect schedulers of tracer/scheduler.go, frame loop of renderer/default.go)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import ctypes_api as T

LIB_PATH = os.path.join(T.LIB_DIR, "libpolaris_host.so")
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        T.load_library()  # dependency first (same directory, rpath $ORIGIN)
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.polaris_host_scheduler_new.restype = vp
        lib.polaris_host_scheduler_new.argtypes = [C.c_int, vp, C.c_uint32]
        lib.polaris_host_scheduler_schedule.argtypes = [vp, vp, vp, C.c_uint32, vp]
        lib.polaris_host_scheduler_schedule.restype = None
        lib.polaris_host_scheduler_free.argtypes = [vp]
        lib.polaris_host_scheduler_free.restype = None
        lib.polaris_host_renderer_new.restype = vp
        lib.polaris_host_renderer_new.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(T.SceneView), vp, vp, C.c_uint32,
                                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_char_p]
        lib.polaris_host_renderer_render.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_double)]
        lib.polaris_host_renderer_read.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        lib.polaris_host_renderer_error.argtypes = [vp]
        lib.polaris_host_renderer_error.restype = C.c_char_p
        lib.polaris_host_renderer_free.argtypes = [vp]
        lib.polaris_host_renderer_free.restype = None
        _lib = lib
    return _lib


NAIVE, PERFECT = 0, 1


class Scheduler:
    """tracer.NaiveScheduler() / tracer.PerfectScheduler() over mock tracers with the given speeds."""

    def __init__(self, kind: int, speeds):
        self._lib = load()
        self._speeds = np.ascontiguousarray(speeds, dtype=np.uint32)
        self._h = self._lib.polaris_host_scheduler_new(kind, self._speeds.ctypes.data, len(self._speeds))

    def schedule(self, frame_h: int, block_h=None, render_ns=None) -> list[int]:
        n = len(self._speeds)
        out = np.zeros(n, dtype=np.uint32)
        bh = None if block_h is None else np.ascontiguousarray(block_h, dtype=np.uint32)
        rt = None if render_ns is None else np.ascontiguousarray(render_ns, dtype=np.int64)
        self._lib.polaris_host_scheduler_schedule(self._h, None if bh is None else bh.ctypes.data, None if rt is None else rt.ctypes.data,
                                                  frame_h, out.ctypes.data)
        return [int(v) for v in out]

    def close(self):
        if self._h:
            self._lib.polaris_host_scheduler_free(self._h)
            self._h = None

    __del__ = close


class Renderer:
    """renderer.NewDefault over HipTracers (device_indices may repeat one GPU)."""

    def __init__(self, scene, device_indices, *, primary=0, scheduler=NAIVE, width=64, height=64, spp=4, bounces=5, min_rr=3,
                 exposure=1.2, seed=1):
        self._lib = load()
        self._scene = scene
        self._view = T.scene_view(scene)
        self.W, self.H, self.n = width, height, len(device_indices)
        dev = np.ascontiguousarray(device_indices, dtype=np.int32)
        eye = np.ascontiguousarray(scene.eye, dtype=np.float32)
        fr = np.ascontiguousarray(scene.frustum, dtype=np.float32)
        err = C.create_string_buffer(256)
        self._h = self._lib.polaris_host_renderer_new(dev.ctypes.data, self.n, primary, scheduler, C.byref(self._view), eye.ctypes.data,
                                                      fr.ctypes.data, width, height, spp, bounces, min_rr, exposure, seed, err)
        if not self._h:
            raise RuntimeError(f"renderer: {err.value.decode()}")

    def render(self, accumulated=0):
        rows = np.zeros(self.n, dtype=np.uint32)
        ms = C.c_double()
        rc = self._lib.polaris_host_renderer_render(self._h, accumulated, rows.ctypes.data, C.byref(ms))
        if rc:
            raise RuntimeError(f"render failed ({rc}): {self._lib.polaris_host_renderer_error(self._h).decode()}")
        return [int(v) for v in rows], ms.value

    def read(self):
        fb = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        acc = np.zeros((self.H, self.W, 4), dtype=np.float32)
        rc = self._lib.polaris_host_renderer_read(self._h, fb.ctypes.data, fb.size, acc.ctypes.data, acc.size)
        if rc:
            raise RuntimeError(f"read failed ({rc})")
        return fb, acc

    def close(self):
        if self._h:
            self._lib.polaris_host_renderer_free(self._h)
            self._h = None

    __del__ = close
