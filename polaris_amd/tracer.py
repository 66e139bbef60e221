"""Host-side mirror of the reference's tracer.Tracer interface over the C ABI (ctypes).

Same method names, argument meaning and error behaviour as tracer/tracer.go:80-111 and its
OpenCL implementation tracer/opencl/tracer.go, so tests and bench.py drive the HIP backend the way
renderer/default.go drives a tracer.  This module is plumbing only: every method is one call into
libpolaris_hip.so (include/polaris_hip.h); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import enum
import time
from dataclasses import dataclass

import numpy as np

from . import ctypes_api as T


class Flag(enum.IntFlag):          # tracer/tracer.go:49-61
    Local = 1
    Remote = 2
    CpuDevice = 4


class UpdateMode(enum.IntEnum):    # tracer/tracer.go:63-69
    Synchronous = 0
    Asynchronous = 1


class ChangeType(enum.IntEnum):    # tracer/tracer.go:71-78
    FrameDimensions = 0
    SceneData = 1
    CameraData = 2


@dataclass
class Stats:                       # tracer/tracer.go:37-47 (durations in seconds)
    BlockW: int = 0
    BlockH: int = 0
    UpdateTime: float = 0.0
    RenderTime: float = 0.0


class TracerError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"polaris_hip error {code}: {msg}")
        self.code = code


class ErrNoSceneData(TracerError):  # tracer/opencl/errors.go
    pass


E_NO_SCENE_DATA = 1


class HipTracer:
    """tracer.Tracer implemented on one MI355X through the C ABI."""

    def __init__(self, id: str, device_index: int = 0, lib_path: str | None = None):
        self._lib = T.load_library(lib_path)
        self._id = id
        self._device = device_index
        self._h = C.c_void_p()
        self._stats = Stats()
        self._changes: dict[ChangeType, object] = {}
        self._name = ""
        self._speed = 0
        self.last_trace_stats: T.TraceStats | None = None
        self._keepalive = None

    # ---- tracer.Tracer ---------------------------------------------------------------------
    def Id(self) -> str:
        return self._id

    def Flags(self) -> Flag:
        return Flag.Local

    def Speed(self) -> int:
        """compute units * MHz / 1000 (tracer/opencl/device/device.go:219)."""
        return self._speed

    def Init(self) -> None:
        T.check_load_order(self._lib)
        name = C.create_string_buffer(256)
        cus, mhz, mem = C.c_uint32(), C.c_uint32(), C.c_uint64()
        self._check(self._lib.polaris_hip_device_info(self._device, name, C.byref(cus), C.byref(mhz), C.byref(mem)), None)
        self._name = name.value.decode()
        self._speed = int(cus.value * mhz.value // 1000)
        self.device_cus = int(cus.value)   # compute units of the device (bench.py: the issue roofline's peak)
        self._check(self._lib.polaris_hip_create(self._device, C.byref(self._h)), None)

    def Close(self) -> None:
        if self._h:
            self._lib.polaris_hip_destroy(self._h)
            self._h = C.c_void_p()

    def Stats(self) -> Stats:
        return self._stats

    def UpdateState(self, mode: UpdateMode, change: ChangeType, data) -> float:
        """Queue a state change; Synchronous commits at once, Asynchronous at the next Trace
        (tracer/opencl/tracer.go:150-192)."""
        self._changes[ChangeType(change)] = data
        if mode == UpdateMode.Synchronous:
            return self._commit()
        return 0.0

    def Trace(self, req: T.BlockRequest, seeds: np.ndarray | None = None) -> float:
        """tracer/opencl/tracer.go:194-247.  `seeds` stands in for the Go math/rand draws (one
        per sample + one per bounce); if omitted they are drawn from numpy's global RNG the way
        the reference draws from math/rand."""
        t0 = time.perf_counter()
        self._commit()
        n = req.samples_per_pixel * (1 + req.num_bounces)
        if seeds is None:
            seeds = np.random.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        st = T.TraceStats()
        rc = self._lib.polaris_hip_trace(self._h, C.byref(req), seeds.ctypes.data_as(C.POINTER(C.c_uint32)), seeds.size, C.byref(st))
        self._check(rc, self._h)
        self.last_trace_stats = st
        req.accumulated_samples += req.samples_per_pixel  # tracer.go:240 (on the caller's copy)
        self._stats.BlockW, self._stats.BlockH = req.block_w, req.block_h
        self._stats.RenderTime = time.perf_counter() - t0
        return self._stats.RenderTime

    def MergeOutput(self, other: "HipTracer", req: T.BlockRequest) -> float:
        t0 = time.perf_counter()
        if not isinstance(other, HipTracer):  # tracer.go:280-283
            raise TracerError(6, "merge failed: unsupported tracer instance")
        self._check(self._lib.polaris_hip_merge(self._h, other._h, C.byref(req)), self._h)
        return time.perf_counter() - t0

    def SyncFramebuffer(self, req: T.BlockRequest) -> float:
        t0 = time.perf_counter()
        self._check(self._lib.polaris_hip_sync_framebuffer(self._h, C.byref(req)), self._h)
        return time.perf_counter() - t0

    # ---- extras used by tests / bench --------------------------------------------------------
    def set_option(self, key: str, value: int) -> None:
        self._check(self._lib.polaris_hip_set_option(self._h, key.encode(), int(value)), self._h)

    def read_accumulator(self, which: int = 0) -> np.ndarray:
        out = np.zeros((self._H, self._W, 4), dtype=np.float32)
        self._check(self._lib.polaris_hip_read_accumulator(self._h, which, out.ctypes.data, out.size), self._h)
        return out

    def read_framebuffer(self) -> np.ndarray:
        out = np.zeros((self._H, self._W, 4), dtype=np.uint8)
        self._check(self._lib.polaris_hip_read_framebuffer(self._h, out.ctypes.data, out.size), self._h)
        return out

    def tap_primary(self, req: T.BlockRequest, seed: int) -> dict:
        n = req.frame_w * req.block_h
        taps = {"primary_rays": np.zeros((n, 8), np.float32), "primary_hit": np.zeros(n, np.int32),
                "primary_wuvt": np.zeros((n, 4), np.float32), "primary_tri": np.full((n, 2), -1, np.int32)}
        self._check(self._lib.polaris_hip_tap_primary(self._h, C.byref(req), seed, taps["primary_rays"].ctypes.data,
                                                      taps["primary_hit"].ctypes.data, taps["primary_wuvt"].ctypes.data,
                                                      taps["primary_tri"].ctypes.data), self._h)
        return taps

    PROBE_BXDF, PROBE_TEXTURE, PROBE_EMISSIVE, PROBE_MATERIAL = 0, 1, 2, 3
    _PROBE_SHAPES = {0: (13, 11), 1: (2, 7), 2: (11, 9), 3: (8, 18)}

    def probe(self, kind: int, index: int, inputs) -> np.ndarray:
        """Function-level test tap (polaris_hip_probe): rows of `inputs` through the device-side BxDF / texture / light
        functions of the uploaded scene; returns one row of outputs per probe."""
        n_in, n_out = self._PROBE_SHAPES[kind]
        a = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, n_in)
        out = np.zeros((a.shape[0], n_out), np.float32)
        self._check(self._lib.polaris_hip_probe(self._h, kind, index, a.shape[0], a.ctypes.data, out.ctypes.data), self._h)
        return out

    def probe_intersect(self, rays, any_hit: bool = False):
        """Arbitrary rays (n, 8) = origin, maxDist, dir, unused through the selected traversal kernel
        (polaris_hip_probe_intersect).  Returns (hit (n,), wuvt (n, 4), triangle (n,))."""
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = r.shape[0]
        hit, wuvt, tri = np.zeros(n, np.int32), np.zeros((n, 4), np.float32), np.full(n, -1, np.int32)
        self._check(self._lib.polaris_hip_probe_intersect(self._h, r.ctypes.data, n, int(any_hit), hit.ctypes.data,
                                                          wuvt.ctypes.data, tri.ctypes.data), self._h)
        return hit, wuvt, tri

    def selftest_rcp(self, lo: float, hi: float) -> tuple[int, int, int]:
        """polaris_hip_selftest_rcp: the triangle tests' 3-instruction reciprocal against 1.0f / x over all 2^32 floats.
        Returns (mismatches with lo <= |x| <= hi, mismatches outside, bit pattern of one mismatch inside)."""
        a, b, smp = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._check(self._lib.polaris_hip_selftest_rcp(self._h, lo, hi, C.byref(a), C.byref(b), C.byref(smp)), self._h)
        return a.value, b.value, smp.value

    def kernel_ms(self, name: str) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.polaris_hip_kernel_ms(self._h, name.encode(), C.byref(ms), C.byref(n)), self._h)
        return ms.value, n.value

    def kernel_symbol(self, name: str) -> str:
        """The kernel symbol the named timer last bracketed (polaris_hip_kernel_symbol)."""
        buf = C.create_string_buffer(128)
        self._check(self._lib.polaris_hip_kernel_symbol(self._h, name.encode(), buf), self._h)
        return buf.value.decode()

    SHADE_TIMERS = ("shade_first", "shade_sort", "shade_plain", "shade_wave")

    def shade_counts(self, bounces: int) -> list[dict]:
        """Per bounce of the last Trace: shaded hits / misses / emitter hits and the shade timer the step ran under."""
        a = (C.c_uint64 * (4 * T.MAX_BOUNCES))()
        self._check(self._lib.polaris_hip_shade_counts(self._h, a, len(a)), self._h)
        return [{"hits": int(a[4 * b]), "misses": int(a[4 * b + 1]), "emitters": int(a[4 * b + 2]), "timer": self.SHADE_TIMERS[int(a[4 * b + 3])]}
                for b in range(bounces)]

    def export_block(self, req: T.BlockRequest, device_ptr: int) -> None:
        self._check(self._lib.polaris_hip_export_block(self._h, C.byref(req), C.c_void_p(device_ptr)), self._h)

    def reset_frame(self) -> None:
        """The pipeline's Reset stage on its own (tracer/opencl/tracer.go:208-213)."""
        self._check(self._lib.polaris_hip_reset_frame(self._h), self._h)

    def merge_device(self, device_ptr: int, req: T.BlockRequest) -> None:
        self._check(self._lib.polaris_hip_merge_device(self._h, C.c_void_p(device_ptr), C.byref(req)), self._h)

    # ---- cross-process merge: peer reads over HIP IPC (include/polaris_hip.h) ------------------------
    def ipc_export(self, depth: int = 3) -> bytes:
        """The trace accumulator becomes a ring of `depth` buffers; returns the PolarisIpcExport blob a peer PROCESS opens."""
        x = T.IpcExport()
        self._check(self._lib.polaris_hip_ipc_export(self._h, depth, C.byref(x)), self._h)
        return bytes(x)

    def ipc_open(self, blob: bytes) -> int:
        """Map another process's ring (its ipc_export blob) on this tracer's device; returns the peer handle."""
        x = T.IpcExport.from_buffer_copy(blob)
        p = C.c_void_p()
        self._check(self._lib.polaris_hip_ipc_open(self._h, C.byref(x), C.byref(p)), self._h)
        return p.value

    def ipc_close(self, peer: int) -> None:
        self._check(self._lib.polaris_hip_ipc_close(self._h, C.c_void_p(peer)), self._h)

    def merge_ipc(self, peer: int, slot: int, req: T.BlockRequest) -> None:
        """MergeOutput from a peer process: rows of `req` of the peer's ring slot, read through the IPC mapping."""
        self._check(self._lib.polaris_hip_merge_ipc(self._h, C.c_void_p(peer), slot, C.byref(req)), self._h)

    def merge_slot(self, other: "HipTracer", slot: int, req: T.BlockRequest) -> None:
        """MergeOutput from ring slot `slot` of a tracer of this process (the primary's own block, merged one frame late)."""
        self._check(self._lib.polaris_hip_merge_slot(self._h, other._h, slot, C.byref(req)), self._h)

    def peer_info(self, peer: int) -> dict:
        """What a mapped peer ring really is (polaris_hip_peer_info): the exporter's GPU by PCI bus id, whether it is this tracer's own
        GPU, and the merge branch that follows from it ("ipc-local" / "ipc-peer" / "ipc-unknown"; "ipc-staged" where the merges go
        through the staging strip: no peer access to the ring's GPU, or option ipc_staged)."""
        i = T.PeerInfo()
        self._check(self._lib.polaris_hip_peer_info(C.c_void_p(peer), C.byref(i)), None)
        same = int(i.same_device)
        return {"pid": int(i.pid), "exporter_device": int(i.exporter_device), "pci_bus_id": i.pci_bus_id.decode(errors="replace"),
                "local_device": int(i.local_device), "same_device": same, "can_access_peer": int(i.can_access_peer), "depth": int(i.depth),
                "has_events": int(i.has_events), "staged": int(i.staged),
                "branch": "ipc-staged" if i.staged else ("ipc-local" if same == 1 else ("ipc-peer" if same == 0 else "ipc-unknown"))}

    def merge_counts(self) -> dict:
        """How many merges onto this tracer took which branch since it was created (polaris_hip_merge_counts)."""
        a = (C.c_uint64 * len(T.MERGE_BRANCHES))()
        self._check(self._lib.polaris_hip_merge_counts(self._h, a), self._h)
        return {name: int(a[k]) for k, name in enumerate(T.MERGE_BRANCHES)}

    def device_identity(self) -> dict:
        """Which physical GPU this tracer runs on (polaris_hip_device_identity of its device index)."""
        return T.device_identity(self._device)

    def trace_slot(self) -> int:
        s = C.c_uint32()
        self._check(self._lib.polaris_hip_trace_slot(self._h, C.byref(s)), self._h)
        return int(s.value)

    @property
    def device_name(self) -> str:
        return self._name

    # ---- internals ---------------------------------------------------------------------------
    def _commit(self) -> float:
        if not self._changes:
            return 0.0
        t0 = time.perf_counter()
        for change, data in list(self._changes.items()):
            if change == ChangeType.FrameDimensions:
                self._W, self._H = int(data[0]), int(data[1])
                self._check(self._lib.polaris_hip_resize(self._h, self._W, self._H), self._h)
            elif change == ChangeType.SceneData:
                view = T.scene_view(data)
                self._check(self._lib.polaris_hip_upload_scene(self._h, C.byref(view)), self._h)
            elif change == ChangeType.CameraData:
                eye = np.ascontiguousarray(data.eye, dtype=np.float32)
                fr = np.ascontiguousarray(data.frustum, dtype=np.float32)
                self._check(self._lib.polaris_hip_set_camera(self._h, eye.ctypes.data_as(C.POINTER(C.c_float)),
                                                             fr.ctypes.data_as(C.POINTER(C.c_float))), self._h)
            else:
                raise TracerError(2, f"unsupported change type {change}")
        self._changes.clear()
        self._stats.UpdateTime = time.perf_counter() - t0
        return self._stats.UpdateTime

    def _check(self, rc: int, h) -> None:
        if rc == 0:
            return
        msg = self._lib.polaris_hip_last_error(h if h else None)
        msg = msg.decode() if msg else ""
        if rc == E_NO_SCENE_DATA:
            raise ErrNoSceneData(rc, msg)
        raise TracerError(rc, msg)
