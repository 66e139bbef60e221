"""ctypes binding for the CPU checkers under oracle/ (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It binds oracle/oracle_api.h for

* ``Oracle("oracle")``   -> oracle/_build/libpolaris_oracle.so   (CPU restatement, travels)
* ``Oracle("ref_pm")``   -> oracle/_ref/libpolaris_ref_pm.so     (reference CL compiled for host,
                                                                   built-ins = polaris_math.h)
* ``Oracle("ref_libm")`` -> oracle/_ref/libpolaris_ref_libm.so   (same, built-ins = glibc libm)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from polaris_amd import ctypes_api as T

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

FIX_EMITTER_INDEX = 1
SERIAL = 2
PARALLEL_SAMPLES = 4


class Taps(C.Structure):
    _fields_ = [("tap_sample", C.c_uint32), ("primary_rays", C.c_void_p), ("primary_hit", C.c_void_p),
                ("primary_wuvt", C.c_void_p), ("primary_tri", C.c_void_p), ("throughput0", C.c_void_p),
                ("num_rays", C.c_void_p)]


_PATHS = {
    "oracle": (os.path.join(HERE, "_build", "libpolaris_oracle.so"), "polaris_oracle"),
    "ref_pm": (os.path.join(HERE, "_ref", "libpolaris_ref_pm.so"), "polaris_ref"),
    "ref_libm": (os.path.join(HERE, "_ref", "libpolaris_ref_libm.so"), "polaris_ref"),
}


def build_oracle(force=False):
    """Compile the CPU restatement (oracle/Makefile).  g++ only; works on the GPU box too."""
    out = _PATHS["oracle"][0]
    if force or not os.path.exists(out):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return out


def build_ref():
    """Compile the reference's OpenCL C in place (oracle/refbuild/Makefile); needs /root/reference."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "refbuild")])


def available(kind: str) -> bool:
    return os.path.exists(_PATHS[kind][0])


class Oracle:
    def __init__(self, kind="oracle"):
        path, prefix = _PATHS[kind]
        if kind == "oracle" and not os.path.exists(path):
            build_oracle()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.kind = kind
        self.lib = C.CDLL(path)
        f = lambda n: getattr(self.lib, f"{prefix}_{n}")
        self._trace, self._tonemap, self._random = f("trace"), f("tonemap"), f("random")
        self._bxdf, self._tex, self._emissive, self._describe = f("bxdf_probe"), f("tex_probe"), f("emissive_probe"), f("describe")
        self._intersect = f("intersect_probe")
        self._material = f("material_probe")
        self._material.argtypes = [C.POINTER(T.SceneView), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        self._material.restype = None
        self._intersect.argtypes = [C.POINTER(T.SceneView), C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        vp = C.c_void_p
        self._trace.argtypes = [C.POINTER(T.SceneView), vp, vp, C.POINTER(T.BlockRequest), vp, C.c_size_t, vp,
                                C.POINTER(T.TraceStats), C.POINTER(Taps), C.c_uint32]
        self._tonemap.argtypes = [vp, C.c_uint32, C.c_float, C.c_float, vp]
        self._random.argtypes = [vp, vp]
        self._random.restype = None
        self._bxdf.argtypes = [vp] * 9
        self._bxdf.restype = None
        self._tex.argtypes = [vp, vp, C.c_int32, vp, vp]
        self._tex.restype = None
        self._emissive.argtypes = [C.POINTER(T.SceneView), C.c_uint32, vp, vp, vp, vp, vp]
        self._emissive.restype = None
        self._describe.restype = C.c_char_p

    def describe(self) -> str:
        return self._describe().decode()

    # ---- whole-trace -------------------------------------------------------------------
    def trace(self, scene, req: T.BlockRequest, seeds: np.ndarray, *, flags=FIX_EMITTER_INDEX, tap_sample=None):
        """Returns (trace_accum (H,W,4) float32, TraceStats, taps dict or None)."""
        W, H, N = req.frame_w, req.frame_h, req.frame_w * req.block_h
        B = req.num_bounces
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        acc = np.zeros((H, W, 4), dtype=np.float32)
        stats = T.TraceStats()
        view = T.scene_view(scene)
        eye = np.ascontiguousarray(scene.eye, dtype=np.float32)
        fr = np.ascontiguousarray(scene.frustum, dtype=np.float32)
        taps, tp = None, None
        if tap_sample is not None:
            taps = {"primary_rays": np.zeros((N, 8), np.float32), "primary_hit": np.zeros(N, np.int32),
                    "primary_wuvt": np.zeros((N, 4), np.float32), "primary_tri": np.full((N, 2), -1, np.int32),
                    "throughput0": np.zeros((N, 4), np.float32), "num_rays": np.zeros(2 * max(B, 1), np.int32)}
            tp = Taps(tap_sample, *[taps[k].ctypes.data for k in ("primary_rays", "primary_hit", "primary_wuvt",
                                                                     "primary_tri", "throughput0", "num_rays")])
        rc = self._trace(C.byref(view), eye.ctypes.data, fr.ctypes.data, C.byref(req), seeds.ctypes.data, seeds.size,
                         acc.ctypes.data, C.byref(stats), C.byref(tp) if tp is not None else None, flags)
        if rc != 0:
            raise RuntimeError(f"{self.kind}: trace failed with code {rc}")
        return acc, stats, taps

    def tonemap(self, accum: np.ndarray, sample_weight: float, exposure: float) -> np.ndarray:
        a = np.ascontiguousarray(accum, dtype=np.float32).reshape(-1, 4)
        out = np.zeros((a.shape[0], 4), dtype=np.uint8)
        self._tonemap(a.ctypes.data, a.shape[0], sample_weight, exposure, out.ctypes.data)
        return out

    def random(self, state):
        st = np.array(state, dtype=np.uint32)
        out = np.zeros(2, dtype=np.float32)
        self._random(st.ctypes.data, out.ctypes.data)
        return st, out

    # ---- function-level probes ---------------------------------------------------------
    def intersect(self, scene, rays, any_hit=False):
        """rayIntersectionQuery / rayIntersectionTest over arbitrary rays (n, 8) = origin, maxDist, dir, unused.
        Returns (hit (n,) int32, wuvt (n,4) float32, inst_tri (n,2) int32); the last two only for closest hits."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.zeros(n, np.int32)
        wuvt = np.zeros((n, 4), np.float32)
        it = np.full((n, 2), -1, np.int32)
        view = T.scene_view(scene)
        rc = self._intersect(C.byref(view), rays.ctypes.data, n, int(any_hit), hit.ctypes.data, wuvt.ctypes.data, it.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"{self.kind}: traversal stack overflow")
        return hit, wuvt, it

    def bxdf_probe(self, node, tex_meta, tex_data, normal, uv, in_dir, sample, eval_dir):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        node = np.ascontiguousarray(node)
        n, u, i, s, e = f(normal), f(uv), f(in_dir), f(sample), f(eval_dir)
        out = np.zeros(11, dtype=np.float32)
        self._bxdf(node.ctypes.data, T._ptr(tex_meta), T._ptr(tex_data), n.ctypes.data, u.ctypes.data, i.ctypes.data,
                   s.ctypes.data, e.ctypes.data, out.ctypes.data)
        return out

    def material_probe(self, scene, root, normal, uv, rng_state, path_flags):
        """matSelectNode from material node `root`: 18 floats (oracle_api.h), integer fields as bit patterns."""
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        n, u = f(normal), f(uv)
        st = np.ascontiguousarray(rng_state, dtype=np.uint32)
        out = np.zeros(18, dtype=np.float32)
        view = T.scene_view(scene)
        self._material(C.byref(view), int(root), n.ctypes.data, u.ctypes.data, st.ctypes.data, int(path_flags), out.ctypes.data)
        return out

    def tex_probe(self, tex_meta, tex_data, tex_index, uv):
        u = np.ascontiguousarray(uv, dtype=np.float32)
        out = np.zeros(7, dtype=np.float32)
        self._tex(T._ptr(tex_meta), T._ptr(tex_data), tex_index, u.ctypes.data, out.ctypes.data)
        return out

    def emissive_probe(self, scene, index, point, normal, sample, pdf_dir):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        p, n, s, d = f(point), f(normal), f(sample), f(pdf_dir)
        out = np.zeros(9, dtype=np.float32)
        view = T.scene_view(scene)
        self._emissive(C.byref(view), index, p.ctypes.data, n.ctypes.data, s.ctypes.data, d.ctypes.data, out.ctypes.data)
        return out


def make_request(w, h, *, spp=1, bounces=5, rr=3, block_y=0, block_h=None, exposure=1.2, accumulated=0) -> T.BlockRequest:
    """BlockRequest as renderer/default.go:107-117 builds it (BlockW = FrameW); the CLI turns
    "rr disabled" into MinBouncesForRR = NumBounces + 1 (cmd/render.go:42-45)."""
    r = T.BlockRequest()
    r.frame_w, r.frame_h = w, h
    r.block_x, r.block_y, r.block_w, r.block_h = 0, block_y, w, (h if block_h is None else block_h)
    r.samples_per_pixel, r.num_bounces, r.min_bounces_for_rr = spp, bounces, rr
    r.exposure, r.seed, r.accumulated_samples = exposure, 0, accumulated
    return r
