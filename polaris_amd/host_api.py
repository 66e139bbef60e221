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
        lib.polaris_host_renderer_push_seeds.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
        lib.polaris_host_renderer_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
        lib.polaris_host_renderer_read.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
        lib.polaris_host_renderer_merge_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.polaris_host_renderer_tracer_stats.argtypes = [vp, C.c_uint32, C.POINTER(T.TraceStats), C.POINTER(C.c_double)]
        lib.polaris_host_renderer_save.argtypes = [vp, C.c_char_p]
        lib.polaris_host_write_png.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
        lib.polaris_host_renderer_error.argtypes = [vp]
        lib.polaris_host_renderer_error.restype = C.c_char_p
        lib.polaris_host_renderer_free.argtypes = [vp]
        lib.polaris_host_renderer_free.restype = None
        lib.polaris_host_bvh_build.restype = C.c_uint32
        lib.polaris_host_bvh_build.argtypes = [vp, C.c_uint32, C.c_int, vp, C.c_uint32, vp, vp]
        lib.polaris_host_compile_scene.restype = vp
        lib.polaris_host_compile_scene.argtypes = [vp, vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, vp, C.c_uint32, vp, C.c_uint32, vp,
                                                   C.c_uint32, vp, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_char_p]
        lib.polaris_host_compiled_view.restype = C.POINTER(T.SceneView)
        lib.polaris_host_compiled_view.argtypes = [vp]
        lib.polaris_host_compiled_free.argtypes = [vp]
        lib.polaris_host_compiled_free.restype = None
        lib.polaris_host_material_check.argtypes = [C.c_char_p, C.c_char_p]
        lib.polaris_host_material_ior.argtypes = [C.c_char_p, C.POINTER(C.c_float)]
        lib.polaris_host_read_scene.restype = vp
        lib.polaris_host_read_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p]
        lib.polaris_host_compiled_camera.argtypes = [vp, C.c_float, C.c_int, vp, vp, vp]
        lib.polaris_host_compiled_camera.restype = None
        lib.polaris_host_compiled_warnings.argtypes = [vp, C.c_char_p, C.c_size_t]
        lib.polaris_host_compiled_warnings.restype = C.c_size_t
        lib.polaris_host_parse_obj.argtypes = [C.c_char_p, vp, vp, vp, C.c_uint32, C.c_char_p, C.c_size_t, C.c_char_p]
        lib.polaris_host_parse_mtl.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p]
        lib.polaris_host_select_face_index.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_char_p]
        lib.polaris_host_texture_load.argtypes = [C.c_char_p, vp, vp, C.c_size_t, C.c_char_p]
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

    def push_seeds(self, tracer_index: int, seeds) -> None:
        """Test hook: tracer `tracer_index` takes its next host PRNG draws (one per sample + one per bounce,
        tracer.go:222, pipeline.go:146) from this list."""
        s = np.ascontiguousarray(seeds, dtype=np.uint32)
        if self._lib.polaris_host_renderer_push_seeds(self._h, tracer_index, s.ctypes.data, s.size):
            raise RuntimeError("push_seeds: bad tracer index")

    def set_option(self, key: str, value: int) -> None:
        if self._lib.polaris_host_renderer_set_option(self._h, key.encode(), int(value)):
            raise RuntimeError(f"set_option failed: {self._lib.polaris_host_renderer_error(self._h).decode()}")

    def render(self, accumulated=0):
        rows = np.zeros(self.n, dtype=np.uint32)
        ms = C.c_double()
        rc = self._lib.polaris_host_renderer_render(self._h, accumulated, rows.ctypes.data, C.byref(ms))
        if rc:
            raise RuntimeError(f"render failed ({rc}): {self._lib.polaris_host_renderer_error(self._h).decode()}")
        return [int(v) for v in rows], ms.value

    def tracer_stats(self, tracer_index: int):
        """(TraceStats, wall milliseconds) of tracer `tracer_index`'s last Trace."""
        st, ms = T.TraceStats(), C.c_double()
        if self._lib.polaris_host_renderer_tracer_stats(self._h, tracer_index, C.byref(st), C.byref(ms)):
            raise RuntimeError("tracer_stats: bad tracer index")
        return st, ms.value

    def merge_counts(self) -> dict:
        """Which branch the merges onto the primary took so far (polaris_hip_merge_counts)."""
        a = (C.c_uint64 * len(T.MERGE_BRANCHES))()
        if self._lib.polaris_host_renderer_merge_counts(self._h, a):
            raise RuntimeError("merge_counts failed")
        return {name: int(a[k]) for k, name in enumerate(T.MERGE_BRANCHES)}

    def read(self):
        fb = np.zeros((self.H, self.W, 4), dtype=np.uint8)
        acc = np.zeros((self.H, self.W, 4), dtype=np.float32)
        rc = self._lib.polaris_host_renderer_read(self._h, fb.ctypes.data, fb.size, acc.ctypes.data, acc.size)
        if rc:
            raise RuntimeError(f"read failed ({rc})")
        return fb, acc

    def save(self, path: str):
        """The SaveFrameBuffer post-process stage: the primary's RGBA8 frame buffer as a PNG."""
        rc = self._lib.polaris_host_renderer_save(self._h, path.encode())
        if rc:
            raise RuntimeError(f"save failed ({rc}): {self._lib.polaris_host_renderer_error(self._h).decode()}")

    def close(self):
        if self._h:
            self._lib.polaris_host_renderer_free(self._h)
            self._h = None

    __del__ = close


def bvh_build(boxes, min_leaf: int):
    """bvh.Build (asset/compiler/bvh/bvh_builder.go:100-124) over boxes (n, 6) = min.xyz, max.xyz.
    Returns (nodes as T.BVH_NODE array, list of leaf sizes in callback order)."""
    lib = load()
    b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    cap = 2 * len(b) + 1
    nodes = np.zeros(cap, dtype=T.BVH_NODE)
    sizes = np.zeros(cap, dtype=np.uint32)
    nl = C.c_uint32()
    n = lib.polaris_host_bvh_build(b.ctypes.data, len(b), min_leaf, nodes.ctypes.data, cap, sizes.ctypes.data, C.byref(nl))
    return nodes[:n].copy(), [int(v) for v in sizes[: nl.value]]


def _scene_from_handle(lib, h, name):
    """Copy the arrays of a compiled scene (CompiledBox handle) into a polaris_amd.scenes.Scene."""
    from .scenes import Scene

    v = lib.polaris_host_compiled_view(h).contents

    def grab(ptr, count, dtype):
        if not ptr or count == 0:
            return np.zeros(0, dtype=dtype)
        nbytes = count * np.dtype(dtype).itemsize
        return np.frombuffer(C.string_at(ptr, nbytes), dtype=dtype).copy()

    nt = v.num_triangles
    return Scene(
        bvh_nodes=grab(v.bvh_nodes, v.num_bvh_nodes, T.BVH_NODE), mesh_instances=grab(v.mesh_instances, v.num_mesh_instances, T.MESH_INSTANCE),
        material_nodes=grab(v.material_nodes, v.num_material_nodes, T.MATERIAL_NODE), emissives=grab(v.emissives, v.num_emissives, T.EMISSIVE),
        texture_data=grab(v.texture_data, v.texture_data_bytes, np.uint8), texture_meta=grab(v.texture_meta, v.num_textures, T.TEXTURE_META),
        vertices=grab(v.vertices, nt * 12, np.float32).reshape(-1, 4), normals=grab(v.normals, nt * 12, np.float32).reshape(-1, 4),
        uvs=grab(v.uvs, nt * 6, np.float32).reshape(-1, 2), material_index=grab(v.material_index, nt, np.uint32),
        scene_diffuse_mat_index=int(v.scene_diffuse_mat_index), scene_emissive_mat_index=int(v.scene_emissive_mat_index), name=name)


def compile_scene(meshes, instances, mats, *, scene_diffuse=-1, scene_emissive=-1, min_leaf=10, name="compiled"):
    """The C++ scene compiler (polaris_amd/host/scene_compiler.cpp = compiler.go partitionGeometry) on
    the same inputs polaris_amd.scenes.compile_scene takes: list[scenes.Mesh], list[(mesh index, 4x4
    world matrix)], scenes.MaterialTable (per-triangle `mat` values are material root nodes).
    Returns a polaris_amd.scenes.Scene holding copies of the compiled arrays."""
    from .scenes import Scene

    lib = load()
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    V = f32(np.concatenate([m.verts for m in meshes]))
    Nn = f32(np.concatenate([m.normals for m in meshes]))
    U = f32(np.concatenate([m.uvs for m in meshes]))
    roots = sorted({int(r) for m in meshes for r in np.asarray(m.mat)})
    root_to_mat = {r: i for i, r in enumerate(roots)}
    M = np.ascontiguousarray([root_to_mat[int(r)] for m in meshes for r in np.asarray(m.mat)], dtype=np.int32)
    offs = np.zeros(len(meshes) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(m.verts) for m in meshes])
    inst_mesh = np.ascontiguousarray([i for i, _ in instances], dtype=np.uint32)
    inst_xf = f32(np.stack([np.asarray(x, dtype=np.float64).T.reshape(-1) for _, x in instances]))  # column major
    nodes, tex_meta, tex_blob = mats.arrays()
    mroots = np.ascontiguousarray(roots, dtype=np.int32)
    err = C.create_string_buffer(256)
    h = lib.polaris_host_compile_scene(V.ctypes.data, Nn.ctypes.data, U.ctypes.data, M.ctypes.data, offs.ctypes.data, len(meshes),
                                       inst_mesh.ctypes.data, inst_xf.ctypes.data, len(instances), nodes.ctypes.data, len(nodes),
                                       mroots.ctypes.data, len(mroots), T._ptr(tex_meta), len(tex_meta), T._ptr(tex_blob), tex_blob.size,
                                       scene_diffuse, scene_emissive, min_leaf, err)
    if not h:
        raise RuntimeError(f"compile_scene: {err.value.decode()}")
    try:
        sc = _scene_from_handle(lib, h, name)
    finally:
        lib.polaris_host_compiled_free(h)
    return sc


# ---- scene front-end: Wavefront OBJ/MTL reader, material expressions, textures ------------------
def material_check(expr: str):
    """(status, message): 0 valid, 1 parse error, 2 semantic error (material.ParseExpression + Validate)."""
    err = C.create_string_buffer(512)
    rc = load().polaris_host_material_check(expr.encode(), err)
    return rc, err.value.decode()


def material_ior(name: str):
    out = C.c_float()
    return None if load().polaris_host_material_ior(name.encode(), C.byref(out)) else float(out.value)


def read_scene(path=None, *, content=None, name="embedded", aspect=1.0, invert_y=False, min_leaf=0):
    """reader.ReadScene: parse a Wavefront .obj (+ .mtl, textures) and compile it.  Returns a
    polaris_amd.scenes.Scene whose eye/frustum come from the file's camera_* statements for a frame of
    the given aspect; `.camera` = dict(fov, eye, look, up), `.warnings` = list of strings."""
    lib = load()
    err = C.create_string_buffer(1024)
    h = lib.polaris_host_read_scene(path.encode() if path is not None else None, name.encode(),
                                    content.encode() if content is not None else None, min_leaf, err)
    if not h:
        raise RuntimeError(err.value.decode())
    try:
        sc = _scene_from_handle(lib, h, os.path.basename(path) if path else name)
        params = np.zeros(10, np.float32)
        eye = np.zeros(3, np.float32)
        fr = np.zeros((4, 4), np.float32)
        lib.polaris_host_compiled_camera(h, float(aspect), int(invert_y), params.ctypes.data, eye.ctypes.data, fr.ctypes.data)
        sc.eye, sc.frustum = eye, fr
        sc.camera = {"fov": float(params[0]), "eye": params[1:4].copy(), "look": params[4:7].copy(), "up": params[7:10].copy()}
        n = lib.polaris_host_compiled_warnings(h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.polaris_host_compiled_warnings(h, buf, n + 1)
        sc.warnings = [w for w in buf.value.decode().split("\n") if w]
    finally:
        lib.polaris_host_compiled_free(h)
    return sc


def parse_obj(content: str, max_instances=64):
    """Parse-level view of an .obj: dict(counts, transforms [n][4][4] (row = matrix row), boxes, materials)."""
    counts = np.zeros(4, np.uint32)
    xf = np.zeros((max_instances, 16), np.float32)
    boxes = np.zeros((max_instances, 9), np.float32)
    mats = C.create_string_buffer(1 << 16)
    err = C.create_string_buffer(1024)
    if load().polaris_host_parse_obj(content.encode(), counts.ctypes.data, xf.ctypes.data, boxes.ctypes.data, max_instances, mats, len(mats), err):
        raise RuntimeError(err.value.decode())
    n = int(counts[1])
    materials = [tuple(l.split("\t")) for l in mats.value.decode().split("\n") if l]
    return {"meshes": int(counts[0]), "instances": n, "materials": materials, "mesh0_primitives": int(counts[3]),
            "transforms": xf[:n].reshape(n, 4, 4).transpose(0, 2, 1).copy(), "bbox": boxes[:n, :6].reshape(n, 2, 3).copy(), "center": boxes[:n, 6:].copy()}


def parse_mtl(content: str):
    """[(name, generated material expression)] of a material library (parseMaterials + GetExpression)."""
    mats = C.create_string_buffer(1 << 16)
    err = C.create_string_buffer(1024)
    if load().polaris_host_parse_mtl(content.encode(), mats, len(mats), err):
        raise RuntimeError(err.value.decode())
    return [tuple(l.split("\t")) for l in mats.value.decode().split("\n") if l]


def select_face_index(token: str, list_len: int, rel_offset: int = 0):
    out = C.c_int(-1)
    err = C.create_string_buffer(256)
    if load().polaris_host_select_face_index(token.encode(), list_len, rel_offset, C.byref(out), err):
        raise ValueError(err.value.decode())
    return out.value


def texture_load(path: str):
    """(format, width, height, texel bytes) of an image file as the scene compiler bakes it."""
    meta = np.zeros(4, np.uint32)
    err = C.create_string_buffer(512)
    lib = load()
    if lib.polaris_host_texture_load(path.encode(), meta.ctypes.data, None, 0, err):
        raise RuntimeError(err.value.decode())
    data = np.zeros(int(meta[3]), np.uint8)
    lib.polaris_host_texture_load(path.encode(), meta.ctypes.data, data.ctypes.data, data.size, err)
    return int(meta[0]), int(meta[1]), int(meta[2]), data


def write_png(path: str, rgba: np.ndarray):
    """renderer::WritePNG: (h, w, 4) uint8 -> PNG file (the encoder behind Renderer.save)."""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = a.shape[:2]
    if load().polaris_host_write_png(path.encode(), a.ctypes.data, w, h):
        raise RuntimeError(f"write_png: could not write {path}")
