"""ctypes mirrors of include/polaris_types.h and the loader for the C-ABI library.

The numpy dtypes below are byte-for-byte the structs of include/polaris_types.h, which are in
turn the reference's device layouts (asset/scene/optimized_scene.go:25-190, CL/types.cl:4-187).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpolaris_hip.so")

MAX_BOUNCES = 32

# ---- numpy structured dtypes (scene arrays) -------------------------------------------------
BVH_NODE = np.dtype([("min", "<f4", 3), ("ldata", "<i4"), ("max", "<f4", 3), ("rdata", "<i4")])
MESH_INSTANCE = np.dtype([("mesh_index", "<u4"), ("bvh_root", "<u4"), ("reserved", "<u4", 2),
                          ("inv_transform", "<f4", 16)])
MATERIAL_NODE = np.dtype([("type", "<u4"), ("left_child", "<u4"), ("right_child", "<i4"), ("tex", "<i4"),
                          ("k", "<f4", 4), ("t", "<f4", 4), ("int_ior", "<f4"), ("ext_ior", "<f4"),
                          ("scale", "<f4"), ("roughness_tex", "<i4")])
EMISSIVE = np.dtype([("transform", "<f4", 16), ("area", "<f4"), ("tri_index", "<u4"),
                     ("mat_node_index", "<u4"), ("type", "<u4")])
TEXTURE_META = np.dtype([("format", "<u4"), ("width", "<u4"), ("height", "<u4"), ("data_offset", "<u4")])
assert BVH_NODE.itemsize == 32 and MESH_INSTANCE.itemsize == 80 and MATERIAL_NODE.itemsize == 64
assert EMISSIVE.itemsize == 80 and TEXTURE_META.itemsize == 16

BXDF_INVALID, BXDF_EMISSIVE, BXDF_DIFFUSE, BXDF_CONDUCTOR = 0, 2, 4, 8
BXDF_ROUGH_CONDUCTOR, BXDF_DIELECTRIC, BXDF_ROUGH_DIELECTRIC = 16, 32, 64
OP_MIX, OP_MIX_MAP, OP_BUMP_MAP, OP_NORMAL_MAP, OP_DISPERSE = 10001, 10002, 10003, 10004, 10005
TEX_L8, TEX_L32F, TEX_RGBA8, TEX_RGBA32F = 0, 1, 2, 3
EMISSIVE_AREA, EMISSIVE_ENVIRONMENT = 0, 1


# ---- ctypes structs -------------------------------------------------------------------------
class SceneView(C.Structure):
    _fields_ = [
        ("bvh_nodes", C.c_void_p), ("num_bvh_nodes", C.c_uint32),
        ("mesh_instances", C.c_void_p), ("num_mesh_instances", C.c_uint32),
        ("material_nodes", C.c_void_p), ("num_material_nodes", C.c_uint32),
        ("emissives", C.c_void_p), ("num_emissives", C.c_uint32),
        ("texture_data", C.c_void_p), ("texture_data_bytes", C.c_uint32),
        ("texture_meta", C.c_void_p), ("num_textures", C.c_uint32),
        ("vertices", C.c_void_p), ("normals", C.c_void_p), ("uvs", C.c_void_p),
        ("material_index", C.c_void_p), ("num_triangles", C.c_uint32),
        ("scene_diffuse_mat_index", C.c_int32), ("scene_emissive_mat_index", C.c_int32),
    ]


class BlockRequest(C.Structure):
    """tracer.BlockRequest (tracer/tracer.go:6-34)."""
    _fields_ = [
        ("frame_w", C.c_uint32), ("frame_h", C.c_uint32),
        ("block_x", C.c_uint32), ("block_y", C.c_uint32), ("block_w", C.c_uint32), ("block_h", C.c_uint32),
        ("samples_per_pixel", C.c_uint32), ("num_bounces", C.c_uint32), ("min_bounces_for_rr", C.c_uint32),
        ("exposure", C.c_float), ("seed", C.c_uint32), ("accumulated_samples", C.c_uint32),
    ]


class TraceStats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64), ("indirect_rays", C.c_uint64), ("occlusion_rays", C.c_uint64),
        ("shaded_hits", C.c_uint64), ("shaded_misses", C.c_uint64), ("emitter_hits", C.c_uint64),
        ("unoccluded", C.c_uint64),
        ("rays_per_bounce", C.c_uint64 * MAX_BOUNCES), ("occl_per_bounce", C.c_uint64 * MAX_BOUNCES),
        ("device_ms", C.c_double),
    ]

    def total_rays(self) -> int:
        """BASELINE.md section 3 counting rule: primary + indirect + occlusion rays traced."""
        return int(self.primary_rays + self.indirect_rays + self.occlusion_rays)

    def algorithmic_bytes(self, pixels: int, traces: int = 1, frames: int = 1) -> int:
        """compulsory stream bytes of the wavefront formulation."""
        return int(112 * self.primary_rays + 68 * self.shaded_hits + 92 * self.indirect_rays
                   + 80 * self.occlusion_rays + 44 * self.unoccluded + 60 * self.shaded_misses
                   + 24 * self.emitter_hits + 48 * pixels * traces + 16 * pixels * frames)

    def as_dict(self) -> dict:
        d = {k: int(getattr(self, k)) for k in ("primary_rays", "indirect_rays", "occlusion_rays", "shaded_hits",
                                                  "shaded_misses", "emitter_hits", "unoccluded")}
        d["rays_per_bounce"] = [int(v) for v in self.rays_per_bounce]
        d["occl_per_bounce"] = [int(v) for v in self.occl_per_bounce]
        d["device_ms"] = float(self.device_ms)
        return d


IPC_MAX_DEPTH = 4


class IpcExport(C.Structure):
    """PolarisIpcExport (include/polaris_hip.h): what a tracer publishes once so that another PROCESS can map its trace
    accumulator ring.  Plain bytes: bytes(export) travels over any channel, IpcExport.from_buffer_copy(b) restores it."""
    _fields_ = [
        ("abi_version", C.c_uint32), ("depth", C.c_uint32), ("frame_w", C.c_uint32), ("frame_h", C.c_uint32),
        ("device", C.c_int32), ("pid", C.c_uint32), ("has_event", C.c_uint32), ("reserved", C.c_uint32),
        ("mem", (C.c_uint8 * 64) * IPC_MAX_DEPTH), ("event", (C.c_uint8 * 64) * IPC_MAX_DEPTH),
        ("pci_bus_id", C.c_char * 32),
    ]


assert C.sizeof(IpcExport) == 32 + 64 * IPC_MAX_DEPTH + 64 * IPC_MAX_DEPTH + 32


class DeviceIdentity(C.Structure):
    """PolarisDeviceIdentity (include/polaris_hip.h): which physical GPU a HIP device index of this process is."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("hip_index", C.c_int32), ("pci_bus_id", C.c_char * 32), ("uuid", C.c_uint8 * 16),
        ("compute_units", C.c_uint32), ("clock_mhz", C.c_uint32), ("global_mem_bytes", C.c_uint64),
        ("name", C.c_char * 64), ("gcn_arch", C.c_char * 32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)

    def as_dict(self) -> dict:
        return {"hip_index": int(self.hip_index), "pci_bus_id": self.pci_bus_id.decode(errors="replace"), "uuid": bytes(self.uuid).hex(),
                "cus": int(self.compute_units), "clock_mhz": int(self.clock_mhz), "global_mem_bytes": int(self.global_mem_bytes),
                "name": self.name.decode(errors="replace"), "gcn_arch": self.gcn_arch.decode(errors="replace")}


class PeerInfo(C.Structure):
    """PolarisPeerInfo (include/polaris_hip.h): what a mapped peer ring really is."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("pid", C.c_uint32), ("exporter_device", C.c_int32), ("local_device", C.c_int32),
        ("same_device", C.c_int32), ("can_access_peer", C.c_int32), ("depth", C.c_uint32), ("has_events", C.c_uint32),
        ("pci_bus_id", C.c_char * 32), ("staged", C.c_int32), ("reserved", C.c_int32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)


assert C.sizeof(DeviceIdentity) == 168 and C.sizeof(PeerInfo) == 72
MERGE_BRANCHES = ("local", "peer-access", "staged", "ipc-local", "ipc-peer", "ipc-unknown", "device-strip", "ipc-staged")   # POLARIS_MERGE_* (include/polaris_hip.h)


def device_identity(index: int) -> dict:
    """polaris_hip_device_identity as a dict (hip_index, pci_bus_id, uuid hex, cus, clock_mhz, global_mem_bytes, name, gcn_arch)."""
    lib = load_library()
    d = DeviceIdentity()
    rc = lib.polaris_hip_device_identity(int(index), C.byref(d))
    if rc:
        msg = lib.polaris_hip_last_error(None)
        raise RuntimeError(f"polaris_hip_device_identity({index}) failed with {rc}: {msg.decode() if msg else ''}")
    return d.as_dict()


def can_access_peer(device: int, peer: int) -> int:
    """hipDeviceCanAccessPeer(device, peer) for two device indices of this process (0 for device == peer)."""
    lib = load_library()
    can = C.c_int(0)
    rc = lib.polaris_hip_can_access_peer(int(device), int(peer), C.byref(can))
    if rc:
        msg = lib.polaris_hip_last_error(None)
        raise RuntimeError(f"polaris_hip_can_access_peer({device}, {peer}) failed with {rc}: {msg.decode() if msg else ''}")
    return int(can.value)


class BvhBuildInput(C.Structure):
    """PolarisBvhBuildInput (include/polaris_hip.h): what polaris_hip_build_bvh builds the two-level BVH from."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("vertices", C.c_void_p), ("num_triangles", C.c_uint32),
        ("mesh_first_tri", C.c_void_p), ("mesh_num_tris", C.c_void_p), ("num_meshes", C.c_uint32),
        ("instance_boxes", C.c_void_p), ("instance_mesh", C.c_void_p), ("num_instances", C.c_uint32),
        ("max_leaf_tris", C.c_uint32), ("algorithm", C.c_uint32),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)   # (the library refuses another layout: ABI 5)


BVH_SAH, BVH_LBVH = 0, 1   # PolarisBvhBuildInput.algorithm


def _ptr(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def scene_view(scene) -> SceneView:
    """Borrow a polaris_amd.scenes.Scene's arrays as a PolarisSceneView (arrays must stay alive)."""
    v = SceneView()
    v.bvh_nodes, v.num_bvh_nodes = _ptr(scene.bvh_nodes), len(scene.bvh_nodes)
    v.mesh_instances, v.num_mesh_instances = _ptr(scene.mesh_instances), len(scene.mesh_instances)
    v.material_nodes, v.num_material_nodes = _ptr(scene.material_nodes), len(scene.material_nodes)
    v.emissives, v.num_emissives = _ptr(scene.emissives), len(scene.emissives)
    v.texture_data, v.texture_data_bytes = _ptr(scene.texture_data), scene.texture_data.size
    v.texture_meta, v.num_textures = _ptr(scene.texture_meta), len(scene.texture_meta)
    v.vertices, v.normals, v.uvs = _ptr(scene.vertices), _ptr(scene.normals), _ptr(scene.uvs)
    v.material_index, v.num_triangles = _ptr(scene.material_index), len(scene.material_index)
    v.scene_diffuse_mat_index = int(scene.scene_diffuse_mat_index)
    v.scene_emissive_mat_index = int(scene.scene_emissive_mat_index)
    return v


# Every symbol include/polaris_hip.h declares; tests check the built library exports all of them.
C_ABI_SYMBOLS = [
    "polaris_hip_device_count", "polaris_hip_device_info", "polaris_hip_create", "polaris_hip_destroy",
    "polaris_hip_last_error", "polaris_hip_resize", "polaris_hip_upload_scene", "polaris_hip_set_camera",
    "polaris_hip_set_option", "polaris_hip_trace", "polaris_hip_merge", "polaris_hip_export_block",
    "polaris_hip_merge_device", "polaris_hip_sync_framebuffer", "polaris_hip_read_framebuffer",
    "polaris_hip_read_accumulator", "polaris_hip_tap_primary", "polaris_hip_abi_version",
    "polaris_hip_kernel_ms", "polaris_hip_reset_frame", "polaris_hip_probe", "polaris_hip_probe_intersect",
    "polaris_hip_selftest_rcp", "polaris_hip_reset_epoch", "polaris_hip_wait_reset",
    "polaris_hip_kernel_symbol", "polaris_hip_shade_counts",
    "polaris_hip_ipc_export", "polaris_hip_ipc_open", "polaris_hip_ipc_close", "polaris_hip_merge_ipc",
    "polaris_hip_trace_slot", "polaris_hip_merge_slot", "polaris_hip_build_bvh", "polaris_hip_build_bvh_error",
    "polaris_hip_device_identity", "polaris_hip_can_access_peer", "polaris_hip_peer_info", "polaris_hip_merge_counts",
]

_lib = None


def _torch_first() -> None:
    """The load-order rule of INTEGRATION.md section 4, enforced: this image's torch wheel bundles its own libamdhip64, and a
    process in which libpolaris_hip.so (system libamdhip64) is loaded BEFORE torch has created its GPU context ends up with
    polaris_hip_device_count() == 0.  If torch is already imported, make it touch the GPU now, before the library loads."""
    import sys

    torch = sys.modules.get("torch")
    if torch is None:
        return
    try:
        if torch.cuda.is_available() and not torch.cuda.is_initialized():
            torch.cuda.init()
    except Exception as e:  # a broken torch install must not be reported as a tracer problem later
        raise RuntimeError(f"polaris_amd: torch is imported but its GPU context cannot be created ({e}); "
                           f"libpolaris_hip.so must be loaded after torch has touched the GPU (INTEGRATION.md section 4)") from e


def check_load_order(lib) -> None:
    """Fail loudly on the other half of the trap: the library was loaded first, torch came later and sees GPUs the library
    does not.  Called by HipTracer.Init before it reports 'no device'."""
    import sys

    torch = sys.modules.get("torch")
    if torch is None or lib.polaris_hip_device_count() > 0:
        return
    try:
        seen = torch.cuda.device_count()
    except Exception:
        seen = 0
    if seen > 0:
        raise RuntimeError(f"libpolaris_hip.so sees no HIP device while torch sees {seen}: the library was loaded before torch "
                           f"created its GPU context (this image's torch bundles its own libamdhip64).  Import torch and call "
                           f"torch.cuda.init() before the first polaris_amd.ctypes_api.load_library() / HipTracer() "
                           f"(INTEGRATION.md section 4)")


def load_library(path: str | None = None) -> C.CDLL:
    """Load libpolaris_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly if absent:
    there is no CPU fallback for the product path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("POLARIS_HIP_LIB") or LIB_PATH  # env override: A/B builds while tuning
    if not os.path.exists(p):
        raise RuntimeError(f"{p} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                           f"g.build()'); the tracer has no CPU fallback")
    _torch_first()
    lib = C.CDLL(p)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    lib.polaris_hip_abi_version.restype = i32
    lib.polaris_hip_device_count.restype = i32
    lib.polaris_hip_device_info.argtypes = [i32, C.c_char_p, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_uint64)]
    lib.polaris_hip_create.argtypes = [i32, C.POINTER(vp)]
    lib.polaris_hip_destroy.argtypes = [vp]
    lib.polaris_hip_destroy.restype = None
    lib.polaris_hip_last_error.argtypes = [vp]
    lib.polaris_hip_last_error.restype = C.c_char_p
    lib.polaris_hip_resize.argtypes = [vp, u32, u32]
    lib.polaris_hip_upload_scene.argtypes = [vp, C.POINTER(SceneView)]
    lib.polaris_hip_set_camera.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.polaris_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.polaris_hip_trace.argtypes = [vp, C.POINTER(BlockRequest), C.POINTER(u32), C.c_size_t, C.POINTER(TraceStats)]
    lib.polaris_hip_merge.argtypes = [vp, vp, C.POINTER(BlockRequest)]
    lib.polaris_hip_export_block.argtypes = [vp, C.POINTER(BlockRequest), vp]
    lib.polaris_hip_merge_device.argtypes = [vp, vp, C.POINTER(BlockRequest)]
    lib.polaris_hip_reset_frame.argtypes = [vp]
    lib.polaris_hip_sync_framebuffer.argtypes = [vp, C.POINTER(BlockRequest)]
    lib.polaris_hip_read_framebuffer.argtypes = [vp, vp, C.c_size_t]
    lib.polaris_hip_read_accumulator.argtypes = [vp, i32, vp, C.c_size_t]
    lib.polaris_hip_tap_primary.argtypes = [vp, C.POINTER(BlockRequest), u32, vp, vp, vp, vp]
    lib.polaris_hip_probe.argtypes = [vp, i32, u32, u32, vp, vp]
    lib.polaris_hip_probe_intersect.argtypes = [vp, vp, u32, i32, vp, vp, vp]
    lib.polaris_hip_selftest_rcp.argtypes = [vp, C.c_float, C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(u32)]
    lib.polaris_hip_reset_epoch.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.polaris_hip_wait_reset.argtypes = [vp, C.c_uint64]
    lib.polaris_hip_kernel_symbol.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.polaris_hip_shade_counts.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
    lib.polaris_hip_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.polaris_hip_ipc_export.argtypes = [vp, u32, C.POINTER(IpcExport)]
    lib.polaris_hip_ipc_open.argtypes = [vp, C.POINTER(IpcExport), C.POINTER(vp)]
    lib.polaris_hip_ipc_close.argtypes = [vp, vp]
    lib.polaris_hip_merge_ipc.argtypes = [vp, vp, u32, C.POINTER(BlockRequest)]
    lib.polaris_hip_trace_slot.argtypes = [vp, C.POINTER(u32)]
    lib.polaris_hip_merge_slot.argtypes = [vp, vp, u32, C.POINTER(BlockRequest)]
    lib.polaris_hip_build_bvh.argtypes = [i32, C.POINTER(BvhBuildInput), vp, u32, C.POINTER(u32), vp, vp, C.POINTER(C.c_double)]
    lib.polaris_hip_build_bvh_error.restype = C.c_char_p
    lib.polaris_hip_device_identity.argtypes = [i32, C.POINTER(DeviceIdentity)]
    lib.polaris_hip_can_access_peer.argtypes = [i32, i32, C.POINTER(i32)]
    lib.polaris_hip_peer_info.argtypes = [vp, C.POINTER(PeerInfo)]
    lib.polaris_hip_merge_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    for name in C_ABI_SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("polaris_hip_device_count", "polaris_hip_abi_version"):
            pass
    if path is None:
        _lib = lib
    return lib
