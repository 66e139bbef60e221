"""Synthetic scene producers emitting the reference's compiled-scene arrays.

The reference ships no scene files (its example scenes live in a separate repository) and its
scene compiler is Go, which is absent here, so bench.py and the tests feed the tracer with scenes
built by this module.  What is emitted is exactly the GPU data contract of
asset/scene/optimized_scene.go:167-190 as produced by asset/compiler/compiler.go:81-231:

* a two-level BVH in one node array: top-level tree over mesh instances first (root = node 0,
  every instance in its own leaf), then one bottom-level tree per mesh, child indices absolute,
  nodes in pre-order, leaf triangles numbered in depth-first leaf order
  (compiler.go:88-180, bvh/bvh_builder.go:100-210);
* MeshInstance.inv_transform = inverse instance matrix, column major (compiler.go:185-192);
* one emissive primitive per (instance, emissive triangle) carrying the instance's inverse
  matrix (compiler.go:200-211, a reference quirk: harmless for identity/translation-free
  emitters only) and an environment emissive for `scene_emissive_material` (compiler.go:214-220);
* material trees flattened children-first (compiler.go:330-438).

This is host-side input generation ("the step before the path"), numpy
only; it is not part of the timed path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import ctypes_api as T

F32 = np.float32


# --------------------------------------------------------------------------------------------
# camera (asset/scene/camera.go:97-141, types/matrix.go Perspective4 / LookAtV)
# --------------------------------------------------------------------------------------------
def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def camera_frustum(eye, look, up, fov, aspect, invert_y=False):
    """Frustum corner rays TL, TR, BL, BR (float4 each) as camera.go:125-141 derives them:
    clip-space corners on the near plane through inv(proj*view), perspective divide, minus eye.
    `fov` is passed to the projection un-converted (the reference treats the OBJ camera_fov
    value as radians, types/matrix.go:156-160)."""
    eye = np.asarray(eye, dtype=np.float64)
    f = _normalize(np.asarray(look, dtype=np.float64) - eye)
    s = _normalize(np.cross(f, _normalize(up)))
    u = np.cross(s, f)
    view = np.eye(4)
    view[0, :3], view[1, :3], view[2, :3] = s, u, -f
    view[:3, 3] = -view[:3, :3] @ eye
    near, far = 1.0, 1000.0
    fy = 1.0 / math.tan(fov / 2.0)
    proj = np.zeros((4, 4))
    proj[0, 0] = fy / aspect
    proj[1, 1] = fy
    proj[2, 2] = (near + far) / (near - far)
    proj[2, 3] = 2.0 * far * near / (near - far)
    proj[3, 2] = -1.0
    inv = np.linalg.inv(proj @ view)
    yup = -1.0 if invert_y else 1.0
    out = np.zeros((4, 4), dtype=F32)
    for i, (cx, cy) in enumerate([(-1, yup), (1, yup), (-1, -yup), (1, -yup)]):
        v = inv @ np.array([cx, cy, -1.0, 1.0])
        out[i, :3] = (v[:3] / v[3] - eye).astype(F32)
    return out


# --------------------------------------------------------------------------------------------
# scene container
# --------------------------------------------------------------------------------------------
@dataclass
class Scene:
    bvh_nodes: np.ndarray
    mesh_instances: np.ndarray
    material_nodes: np.ndarray
    emissives: np.ndarray
    texture_data: np.ndarray
    texture_meta: np.ndarray
    vertices: np.ndarray        # (3T, 4) f32
    normals: np.ndarray         # (3T, 4) f32
    uvs: np.ndarray             # (3T, 2) f32
    material_index: np.ndarray  # (T,) u32
    scene_diffuse_mat_index: int = -1
    scene_emissive_mat_index: int = -1
    eye: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=F32))
    frustum: np.ndarray = field(default_factory=lambda: np.zeros((4, 4), dtype=F32))
    name: str = "scene"
    bvh_max_depth: int = 0      # deepest traversal stack use (instance push included)

    @property
    def num_triangles(self) -> int:
        return len(self.material_index)

    def set_camera(self, eye, look, up=(0, 1, 0), fov=0.6, aspect=1.0):
        self.eye = np.asarray(eye, dtype=F32)
        self.frustum = camera_frustum(eye, look, up, fov, aspect)
        return self

    def nbytes(self) -> int:
        return sum(a.nbytes for a in (self.bvh_nodes, self.mesh_instances, self.material_nodes, self.emissives,
                                      self.texture_data, self.texture_meta, self.vertices, self.normals, self.uvs,
                                      self.material_index))


# --------------------------------------------------------------------------------------------
# materials (asset/compiler/compiler.go:330-438; defaults asset/material/defaults.go)
# --------------------------------------------------------------------------------------------
class MaterialTable:
    """Builds the flat material node list and the texture blob."""

    GLASS_IOR = 1.5   # material.DefaultIntIOR ("Glass"), asset/material/ior.go
    AIR_IOR = 1.0     # material.DefaultExtIOR ("Air"); exact table values are inputs, not path logic

    def __init__(self):
        self.nodes = []
        self.tex_meta = []
        self.tex_blob = bytearray()

    def _node(self, type_, **kw):
        n = np.zeros((), dtype=T.MATERIAL_NODE)
        n["type"] = type_
        n["left_child"] = np.uint32(0xFFFFFFFF)  # compiler initialises Union1 = {0,-1,-1,-1}
        n["right_child"] = -1
        n["tex"] = -1
        n["roughness_tex"] = -1
        n["int_ior"] = self.GLASS_IOR
        n["ext_ior"] = self.AIR_IOR
        for k, v in kw.items():
            if k in ("k", "t"):
                vv = np.zeros(4, dtype=F32)
                vv[:len(v)] = v
                n[k] = vv
            else:
                n[k] = v
        self.nodes.append(n)
        return len(self.nodes) - 1

    def diffuse(self, kd=(0.2, 0.2, 0.2), tex=-1):
        return self._node(T.BXDF_DIFFUSE, k=kd, tex=tex)

    def emissive(self, radiance=(1, 1, 1), scale=1.0, tex=-1):
        return self._node(T.BXDF_EMISSIVE, k=radiance, scale=scale, tex=tex)

    def conductor(self, ks=(1, 1, 1), int_ior=0.0, ext_ior=1.0, tex=-1):
        return self._node(T.BXDF_CONDUCTOR, k=ks, int_ior=int_ior, ext_ior=ext_ior, tex=tex)

    def rough_conductor(self, ks=(1, 1, 1), roughness=0.1, int_ior=0.0, ext_ior=1.0, tex=-1, roughness_tex=-1):
        return self._node(T.BXDF_ROUGH_CONDUCTOR, k=ks, scale=roughness, int_ior=int_ior, ext_ior=ext_ior, tex=tex,
                          roughness_tex=roughness_tex)

    def dielectric(self, ks=(1, 1, 1), tf=(1, 1, 1), int_ior=1.5, ext_ior=1.0, tex=-1, trans_tex=-1):
        return self._node(T.BXDF_DIELECTRIC, k=ks, t=tf, int_ior=int_ior, ext_ior=ext_ior, tex=tex, right_child=trans_tex)

    def rough_dielectric(self, ks=(1, 1, 1), tf=(1, 1, 1), roughness=0.1, int_ior=1.5, ext_ior=1.0, tex=-1,
                         trans_tex=-1, roughness_tex=-1):
        return self._node(T.BXDF_ROUGH_DIELECTRIC, k=ks, t=tf, scale=roughness, int_ior=int_ior, ext_ior=ext_ior,
                          tex=tex, right_child=trans_tex, roughness_tex=roughness_tex)

    def mix(self, left, right, weight):
        return self._node(T.OP_MIX, left_child=left, right_child=right, k=(weight,))

    def mix_map(self, left, right, tex):
        return self._node(T.OP_MIX_MAP, left_child=left, right_child=right, tex=tex)

    def bump_map(self, child, tex):
        return self._node(T.OP_BUMP_MAP, left_child=child, tex=tex)

    def normal_map(self, child, tex):
        return self._node(T.OP_NORMAL_MAP, left_child=child, tex=tex)

    def disperse(self, child, int_iors, ext_iors):
        return self._node(T.OP_DISPERSE, left_child=child, k=int_iors, t=ext_iors)

    def texture(self, fmt, pixels: np.ndarray):
        """Append a texture; `pixels` is (h, w[, 4]) uint8 or float32 matching fmt.  Data is kept
        dword aligned (compiler.go bakeTexture)."""
        pixels = np.ascontiguousarray(pixels)
        h, w = pixels.shape[:2]
        while len(self.tex_blob) % 4:
            self.tex_blob.append(0)
        m = np.zeros((), dtype=T.TEXTURE_META)
        m["format"], m["width"], m["height"], m["data_offset"] = fmt, w, h, len(self.tex_blob)
        self.tex_blob += pixels.tobytes()
        self.tex_meta.append(m)
        return len(self.tex_meta) - 1

    def find_emissive(self, root):
        """compiler.go:246-268 findMaterialNodeByBxdf(root, BxdfEmissive)."""
        n = self.nodes[root]
        t = int(n["type"])
        if t < T.OP_MIX:
            return root if t == T.BXDF_EMISSIVE else -1
        out = self.find_emissive(int(n["left_child"]))
        if out != -1:
            return out
        if t == T.OP_MIX:
            return self.find_emissive(int(n["right_child"]))
        return -1

    def arrays(self):
        nodes = np.array(self.nodes, dtype=T.MATERIAL_NODE) if self.nodes else np.zeros(0, dtype=T.MATERIAL_NODE)
        meta = np.array(self.tex_meta, dtype=T.TEXTURE_META) if self.tex_meta else np.zeros(0, dtype=T.TEXTURE_META)
        blob = np.frombuffer(bytes(self.tex_blob), dtype=np.uint8).copy() if self.tex_blob else np.zeros(0, dtype=np.uint8)
        return nodes, meta, blob


# --------------------------------------------------------------------------------------------
# BVH construction
# --------------------------------------------------------------------------------------------
def _build_bvh(bmin, bmax, max_leaf, leaf_cb):
    """Binned-SAH builder over boxes (bmin, bmax: (n,3) float32).  Emits nodes in pre-order
    with leaves created depth-first left-first, like bvh_builder.go:100-210 does, and calls
    leaf_cb(node, item_indices) for every leaf.  Returns (nodes, max_depth).  16 bins per axis on
    the centroid bounds, bin boxes by sort + reduceat (vectorised: a 1 M-triangle mesh builds in
    about a minute); falls back to a median split when no bin boundary separates the items, so
    depth stays O(log n)."""
    n = len(bmin)
    bmin64 = bmin.astype(np.float64)
    bmax64 = bmax.astype(np.float64)
    cent = 0.5 * (bmin64 + bmax64)
    nodes = []
    max_depth = [0]
    NB = 16

    def half_area(lo, hi):
        d = np.maximum(hi - lo, 0.0)
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 0] * d[..., 2]

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))

    def rec(idx, depth):
        max_depth[0] = max(max_depth[0], depth)
        node = np.zeros((), dtype=T.BVH_NODE)
        node["min"], node["max"] = bmin[idx].min(axis=0), bmax[idx].max(axis=0)
        me = len(nodes)
        nodes.append(node)
        m = len(idx)
        if m <= max_leaf:
            leaf_cb(node, idx)
            return me
        c = cent[idx]
        clo, chi = c.min(axis=0), c.max(axis=0)
        best_cost, best_mask = None, None
        lo_i, hi_i = bmin64[idx], bmax64[idx]
        for ax in range(3):
            ext = chi[ax] - clo[ax]
            if ext <= 1e-12:
                continue
            b = np.minimum(((c[:, ax] - clo[ax]) / ext * NB).astype(np.int64), NB - 1)
            order = np.argsort(b, kind="stable")
            counts = np.bincount(b, minlength=NB)
            used = np.nonzero(counts)[0]
            if len(used) < 2:
                continue
            starts = np.concatenate(([0], np.cumsum(counts)))[used]
            blo = np.minimum.reduceat(lo_i[order], starts, axis=0)
            bhi = np.maximum.reduceat(hi_i[order], starts, axis=0)
            cnt_u = counts[used]
            # sweep: left = bins[:k], right = bins[k:] for k = 1..len(used)-1
            llo = np.minimum.accumulate(blo, axis=0)[:-1]
            lhi = np.maximum.accumulate(bhi, axis=0)[:-1]
            rlo = np.minimum.accumulate(blo[::-1], axis=0)[::-1][1:]
            rhi = np.maximum.accumulate(bhi[::-1], axis=0)[::-1][1:]
            nl = np.cumsum(cnt_u)[:-1]
            cost = nl * half_area(llo, lhi) + (m - nl) * half_area(rlo, rhi)
            k = int(np.argmin(cost))
            if best_cost is None or cost[k] < best_cost:
                best_cost = float(cost[k])
                best_mask = b <= used[k]
        if best_mask is None:
            # all centroids coincide (or a degenerate bin layout): median split by position
            ax = int(np.argmax(chi - clo))
            order = np.argsort(c[:, ax], kind="stable")
            best_mask = np.zeros(m, dtype=bool)
            best_mask[order[: m // 2]] = True
        l = rec(idx[best_mask], depth + 1)
        r = rec(idx[~best_mask], depth + 1)
        nodes[me]["ldata"], nodes[me]["rdata"] = l, r
        return me

    rec(np.arange(n), 0)
    return nodes, max_depth[0]


@dataclass
class Mesh:
    """Triangle soup: verts (T,3,3), normals (T,3,3), uvs (T,3,2), material root per triangle (T,)."""
    verts: np.ndarray
    normals: np.ndarray
    uvs: np.ndarray
    mat: np.ndarray


def _flat_normals(verts):
    n = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    return np.repeat(n[:, None, :], 3, axis=1)


def quad(p0, p1, p2, p3, mat, uv=((0, 0), (1, 0), (1, 1), (0, 1))):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with flat normals."""
    p = np.array([p0, p1, p2, p3], dtype=np.float64)
    verts = np.array([[p[0], p[1], p[2]], [p[0], p[2], p[3]]])
    u = np.array(uv, dtype=np.float64)
    uvs = np.array([[u[0], u[1], u[2]], [u[0], u[2], u[3]]])
    return Mesh(verts, _flat_normals(verts), uvs, np.array([mat, mat]))


def box(lo, hi, mat, rot_y=0.0, center=None):
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [
        ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # +z
        ((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # -z
        ((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # +x
        ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # -x
        ((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # +y
        ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),  # -y
    ]
    m = merge([quad(*f, mat=mat) for f in faces])
    if rot_y != 0.0:
        c = np.asarray(center if center is not None else 0.5 * (lo + hi))
        cs, sn = math.cos(rot_y), math.sin(rot_y)
        R = np.array([[cs, 0, sn], [0, 1, 0], [-sn, 0, cs]])
        m.verts = (m.verts - c) @ R.T + c
        m.normals = _flat_normals(m.verts)
    return m


def uv_sphere(center, radius, mat, n_lat=16, n_lon=24, smooth=True):
    c = np.asarray(center, dtype=np.float64)
    tris, nors, uvs = [], [], []

    def P(i, j):
        th = math.pi * i / n_lat
        ph = 2 * math.pi * j / n_lon
        d = np.array([math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)])
        return c + radius * d, d, (j / n_lon, i / n_lat)

    for i in range(n_lat):
        for j in range(n_lon):
            a, b, cc, d = P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1)
            if i != 0:
                tris.append([a[0], d[0], cc[0]]); nors.append([a[1], d[1], cc[1]]); uvs.append([a[2], d[2], cc[2]])
            if i != n_lat - 1:
                tris.append([a[0], cc[0], b[0]]); nors.append([a[1], cc[1], b[1]]); uvs.append([a[2], cc[2], b[2]])
    verts = np.array(tris)
    normals = np.array(nors) if smooth else _flat_normals(verts)
    return Mesh(verts, normals, np.array(uvs), np.full(len(verts), mat))


def merge(meshes):
    return Mesh(np.concatenate([m.verts for m in meshes]), np.concatenate([m.normals for m in meshes]),
                np.concatenate([m.uvs for m in meshes]), np.concatenate([m.mat for m in meshes]))


def translation(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def compile_scene(meshes, instances, mats: MaterialTable, *, max_leaf=4, scene_diffuse=-1, scene_emissive=-1,
                  name="scene") -> Scene:
    """meshes: list[Mesh]; instances: list[(mesh_index, 4x4 world matrix)].  Mirrors
    compiler.go:81-231 partitionGeometry (see module docstring)."""
    nodes_all = []
    inst_boxes_lo, inst_boxes_hi = [], []
    mesh_nodes, mesh_depth = [], []
    V, Nn, U, M = [], [], [], []
    tri_off = 0
    mesh_emissives = []  # (mesh index, tri index, area, emissive node)
    for mi, m in enumerate(meshes):
        v32 = m.verts.astype(F32)
        bmin, bmax = v32.min(axis=1), v32.max(axis=1)
        order = []

        def leaf_cb(node, idx, order=order, base=tri_off):
            node["ldata"] = -(base + len(order))
            node["rdata"] = len(idx)
            order.extend(int(i) for i in idx)

        nodes, depth = _build_bvh(bmin, bmax, max_leaf, leaf_cb)
        order = np.array(order, dtype=np.int64)
        V.append(v32[order]); Nn.append(m.normals.astype(F32)[order]); U.append(m.uvs.astype(F32)[order])
        M.append(np.asarray(m.mat)[order])
        for local, src in enumerate(order):
            root = int(m.mat[src])
            en = mats.find_emissive(root)
            if en != -1:
                p = m.verts[src]
                area = 0.5 * np.linalg.norm(np.cross(p[2] - p[0], p[2] - p[1]))  # compiler.go:151
                mesh_emissives.append((mi, tri_off + local, F32(area), en))
        mesh_nodes.append(nodes)
        mesh_depth.append(depth)
        tri_off += len(order)

    # top-level tree over instance world boxes, one instance per leaf (compiler.go:88-103)
    for (mi, xf) in instances:
        v = meshes[mi].verts.reshape(-1, 3)
        w = v @ np.asarray(xf)[:3, :3].T + np.asarray(xf)[:3, 3]
        inst_boxes_lo.append(w.min(0)); inst_boxes_hi.append(w.max(0))
    lo = np.array(inst_boxes_lo, dtype=F32)
    hi = np.array(inst_boxes_hi, dtype=F32)

    def top_leaf(node, idx):
        node["ldata"] = -int(idx[0])
        node["rdata"] = 0

    top_nodes, top_depth = _build_bvh(lo, hi, 1, top_leaf)
    nodes_all.extend(top_nodes)
    roots = []
    for nodes in mesh_nodes:
        off = len(nodes_all)
        roots.append(off)
        for nd in nodes:
            if nd["ldata"] > 0:  # inner node: offset children (optimized_scene.go:66-74)
                nd["ldata"] += off
                nd["rdata"] += off
        nodes_all.extend(nodes)
    # the root of a bottom tree is itself a node that may be a leaf with ldata == 0 etc.: fine.

    inst = np.zeros(len(instances), dtype=T.MESH_INSTANCE)
    for i, (mi, xf) in enumerate(instances):
        inst[i]["mesh_index"] = mi
        inst[i]["bvh_root"] = roots[mi]
        inv = np.linalg.inv(np.asarray(xf, dtype=np.float64))
        inst[i]["inv_transform"] = inv.T.reshape(-1).astype(F32)  # column major

    ems = []
    for i in range(len(instances)):
        for (mi, tri, area, en) in mesh_emissives:
            if instances[i][0] != mi:
                continue
            e = np.zeros((), dtype=T.EMISSIVE)
            e["transform"] = inst[i]["inv_transform"]
            e["area"], e["tri_index"], e["mat_node_index"], e["type"] = area, tri, en, T.EMISSIVE_AREA
            ems.append(e)
    if scene_emissive != -1 and mats.find_emissive(scene_emissive) != -1:
        e = np.zeros((), dtype=T.EMISSIVE)
        e["mat_node_index"], e["type"] = mats.find_emissive(scene_emissive), T.EMISSIVE_ENVIRONMENT
        ems.append(e)

    mat_nodes, tex_meta, tex_blob = mats.arrays()
    verts = np.concatenate(V).reshape(-1, 3)
    norms = np.concatenate(Nn).reshape(-1, 3)
    v4 = np.zeros((len(verts), 4), dtype=F32); v4[:, :3] = verts
    n4 = np.zeros((len(norms), 4), dtype=F32); n4[:, :3] = norms
    sc = Scene(
        bvh_nodes=np.array(nodes_all, dtype=T.BVH_NODE),
        mesh_instances=inst, material_nodes=mat_nodes,
        emissives=np.array(ems, dtype=T.EMISSIVE) if ems else np.zeros(0, dtype=T.EMISSIVE),
        texture_data=tex_blob, texture_meta=tex_meta,
        vertices=v4, normals=n4, uvs=np.ascontiguousarray(np.concatenate(U).reshape(-1, 2), dtype=F32),
        material_index=np.concatenate(M).astype(np.uint32),
        scene_diffuse_mat_index=scene_diffuse, scene_emissive_mat_index=scene_emissive, name=name,
        bvh_max_depth=top_depth + 1 + max(mesh_depth),
    )
    return sc


# --------------------------------------------------------------------------------------------
# the scenes of BASELINE.json.configs
# --------------------------------------------------------------------------------------------
def checker_rgba8(n=64, cells=8, a=(230, 230, 230), b=(60, 60, 70)):
    yy, xx = np.mgrid[0:n, 0:n]
    m = (((xx * cells) // n + (yy * cells) // n) % 2).astype(bool)
    img = np.zeros((n, n, 4), dtype=np.uint8)
    img[m] = (*a, 255)
    img[~m] = (*b, 255)
    return img


def cornell_box(variant="layered", aspect=1.0, compiler="numpy") -> Scene:
    """Cornell box from the classic measured geometry (dimensions of the Cornell Program of
    Computer Graphics data set, scaled by 1/555 so the box spans ~[0,1]^3).

    variant "diffuse": five diffuse walls, two diffuse blocks, a 2-triangle ceiling light.
    variant "layered" (BASELINE.json configs[2] / headline): the tall block is
    mix(diffuse, roughConductor, 0.5), the short block is replaced by a dielectric sphere over a
    rough-dielectric pedestal, the floor carries an RGBA8 checker texture, the back wall a
    bump-mapped diffuse -- every BxDF family and the MIS / RR machinery is exercised.
    """
    s = 1.0 / 555.0
    mt = MaterialTable()
    white = mt.diffuse((0.725, 0.71, 0.68))
    red = mt.diffuse((0.63, 0.065, 0.05))
    green = mt.diffuse((0.14, 0.45, 0.091))
    light = mt.emissive((17.0, 12.0, 4.0), 1.0)
    floor_mat, back_mat, tall_mat = white, white, white
    if variant == "layered":
        tex = mt.texture(T.TEX_RGBA8, checker_rgba8())
        floor_mat = mt.diffuse((0.725, 0.71, 0.68), tex=tex)
        yy, xx = np.mgrid[0:32, 0:32]
        bump = (0.5 + 0.5 * np.sin(xx * 0.8) * np.cos(yy * 0.6)).astype(F32)
        btex = mt.texture(T.TEX_L32F, bump)
        back_mat = mt.bump_map(mt.diffuse((0.725, 0.71, 0.68)), btex)
        gold = mt.rough_conductor((1.0, 0.78, 0.34), roughness=0.35)
        tall_mat = mt.mix(mt.diffuse((0.725, 0.71, 0.68)), gold, 0.5)
    X, Y, Z = 556.0 * s, 548.8 * s, 559.2 * s
    parts = [
        quad((X, 0, 0), (0, 0, 0), (0, 0, Z), (X, 0, Z), floor_mat, uv=((0, 0), (4, 0), (4, 4), (0, 4))),   # floor
        quad((X, Y, 0), (X, Y, Z), (0, Y, Z), (0, Y, 0), white),                                            # ceiling
        quad((X, 0, Z), (0, 0, Z), (0, Y, Z), (X, Y, Z), back_mat, uv=((0, 0), (6, 0), (6, 6), (0, 6))),    # back
        quad((0, 0, Z), (0, 0, 0), (0, Y, 0), (0, Y, Z), green),                                            # right (x=0)
        quad((X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0), red),                                              # left
        quad((343 * s, Y - 1e-3, 227 * s), (343 * s, Y - 1e-3, 332 * s), (213 * s, Y - 1e-3, 332 * s),
             (213 * s, Y - 1e-3, 227 * s), light),                                                          # light
        box((265 * s, 0, 296 * s), (265 * s + 165 * s, 330 * s, 296 * s + 165 * s), tall_mat, rot_y=-0.29),   # tall block
    ]
    if variant == "layered":
        glass = mt.dielectric((1, 1, 1), (0.95, 0.98, 1.0), int_ior=1.5)
        frosted = mt.rough_dielectric((1, 1, 1), (0.9, 0.95, 0.9), roughness=0.3, int_ior=1.45)
        mirror = mt.conductor((0.9, 0.9, 0.95))
        parts.append(box((100 * s, 0, 90 * s), (240 * s, 60 * s, 230 * s), frosted, rot_y=0.3))
        parts.append(uv_sphere((170 * s, 60 * s + 75 * s, 160 * s), 75 * s, glass, n_lat=14, n_lon=20))
        parts.append(uv_sphere((440 * s, 40 * s, 130 * s), 40 * s, mirror, n_lat=10, n_lon=14))
    else:
        parts.append(box((130 * s, 0, 65 * s), (130 * s + 165 * s, 165 * s, 65 * s + 165 * s), white, rot_y=0.29))
    mesh = merge(parts)
    if compiler == "reference":
        # the C++ restatement of the reference's own compiler (bvh_builder.go SAH on up to 1024
        # candidate planes per axis, leaves of <= 10 triangles): the BVH `polaris render` would upload
        from . import host_api

        sc = host_api.compile_scene([mesh], [(0, np.eye(4))], mt, min_leaf=10, name=f"cornell-{variant}-refbvh")
    else:
        sc = compile_scene([mesh], [(0, np.eye(4))], mt, name=f"cornell-{variant}")
    sc.set_camera(eye=(278 * s, 273 * s, -800 * s), look=(278 * s, 273 * s, 0), fov=0.6911, aspect=aspect)
    return sc


def sphere_scene(aspect=1.0, n_lat=20, n_lon=19) -> Scene:
    """BASELINE.json configs[0]/[1]: a ~760-triangle UV sphere on a ground quad, diffuse only,
    lit by a constant environment emissive (scene_emissive_material) with a
    scene_diffuse_material background (docs/cli.md:68-89 describes the reference's sphere.obj as
    ~760 triangles / 3 material nodes / 1 emissive)."""
    mt = MaterialTable()
    kd = mt.diffuse((0.75, 0.35, 0.25))
    ground = mt.diffuse((0.6, 0.6, 0.6))
    bg = mt.diffuse((0.55, 0.7, 0.95))
    env = mt.emissive((1.0, 1.0, 1.0), 1.5)
    sp = uv_sphere((0, 1.0, 0), 1.0, kd, n_lat=n_lat, n_lon=n_lon)
    gr = quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), ground)
    sc = compile_scene([merge([sp, gr])], [(0, np.eye(4))], mt, scene_diffuse=bg, scene_emissive=env, name="sphere")
    sc.set_camera(eye=(0, 1.6, 4.5), look=(0, 0.9, 0), fov=0.75, aspect=aspect)
    return sc


def instanced_cubes(n_side=3, aspect=1.0, spacing=2.5) -> Scene:
    """A 12-triangle cube mesh (the shape of tracer/opencl/fixtures/cube.obj) instanced on an
    n x n grid with translation-only transforms, one emissive ceiling quad mesh, env background."""
    mt = MaterialTable()
    cube_mat = mt.diffuse((0.3, 0.6, 0.8))
    metal = mt.rough_conductor((0.9, 0.6, 0.3), roughness=0.4)
    mixm = mt.mix(cube_mat, metal, 0.7)
    floor = mt.diffuse((0.7, 0.7, 0.7))
    light = mt.emissive((10, 10, 9), 1.0)
    bg = mt.diffuse((0.1, 0.12, 0.2))
    cube = box((-0.5, 0, -0.5), (0.5, 1, 0.5), mixm)
    ext = spacing * n_side
    room = merge([quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), floor),
                  quad((-1.5, 4.0, -1.5), (1.5, 4.0, -1.5), (1.5, 4.0, 1.5), (-1.5, 4.0, 1.5), light)])
    insts = [(1, np.eye(4))]
    for i in range(n_side):
        for j in range(n_side):
            insts.append((0, translation(((i - (n_side - 1) / 2) * spacing, 0, (j - (n_side - 1) / 2) * spacing))))
    sc = compile_scene([cube, room], insts, mt, scene_diffuse=bg, name=f"cubes-{n_side}x{n_side}")
    sc.set_camera(eye=(0, 5.0, 2.2 * ext), look=(0, 0.5, 0), fov=0.7, aspect=aspect)
    return sc


def rotation_y(a):
    c, s_ = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s_, -s_, c
    return m


def scaling(sx, sy, sz):
    return np.diag([sx, sy, sz, 1.0])


def transformed_instances(aspect=1.0) -> Scene:
    """Instances with rotation and NON-UNIFORM SCALE, one of them carrying an emissive triangle
    pair: exercises the ray transform on instance entry / exit (intersect.cl:239-252, 330-335), the
    comparison of hit distances across differently scaled instance spaces and reference quirk
    a-9(4) (emissive normals and pdf edges go through the point transform of the INVERSE matrix)."""
    mt = MaterialTable()
    a = mt.diffuse((0.7, 0.5, 0.2))
    b = mt.mix(mt.diffuse((0.2, 0.4, 0.7)), mt.conductor((0.9, 0.9, 0.9)), 0.5)
    floor = mt.diffuse((0.6, 0.6, 0.6))
    light = mt.emissive((6, 6, 5), 1.5)
    bg = mt.diffuse((0.2, 0.25, 0.35))
    cube = box((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5), a)
    ball = uv_sphere((0, 0, 0), 0.5, b, n_lat=8, n_lon=10)
    panel = merge([quad((-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5), light)])
    ground = quad((-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6), floor)
    insts = [
        (3, np.eye(4)),
        (0, translation((-1.5, 0.8, 0)) @ rotation_y(0.6) @ scaling(1.0, 1.6, 0.7)),
        (0, translation((1.4, 0.4, -0.8)) @ rotation_y(-1.1) @ scaling(0.8, 0.8, 0.8)),
        (1, translation((0.1, 0.9, 0.6)) @ scaling(1.8, 0.9, 1.2)),
        (1, translation((-0.4, 0.35, 2.0)) @ rotation_y(2.0) @ scaling(0.7, 0.7, 0.7)),
        (2, translation((0.0, 3.5, 0.0)) @ rotation_y(0.4) @ scaling(2.5, 1.0, 1.5)),
    ]
    sc = compile_scene([cube, ball, panel, ground], insts, mt, scene_diffuse=bg, name="transformed-instances")
    sc.set_camera(eye=(0, 2.4, 6.0), look=(0, 0.8, 0), fov=0.75, aspect=aspect)
    return sc


def textured_materials_scene(aspect=1.0) -> Scene:
    """A small scene touching every material operator and texture format (mix, mixMap, bumpMap,
    normalMap, disperse; L8, L32F, RGBA8, RGBA32F) -- parity coverage, not a benchmark."""
    rng = np.random.default_rng(7)
    mt = MaterialTable()
    l8 = mt.texture(T.TEX_L8, (rng.random((16, 16)) * 255).astype(np.uint8))
    l32 = mt.texture(T.TEX_L32F, rng.random((8, 16)).astype(F32))
    rgba8 = mt.texture(T.TEX_RGBA8, checker_rgba8(32, 4, (250, 40, 40), (40, 40, 250)))
    rgba32 = mt.texture(T.TEX_RGBA32F, rng.random((8, 8, 4)).astype(F32))
    nmap = np.zeros((16, 16, 4), dtype=np.uint8)
    nmap[..., 0] = 128 + (40 * np.sin(np.arange(16) * 0.9)).astype(np.int32)[None, :]
    nmap[..., 1] = 128 + (40 * np.cos(np.arange(16) * 0.7)).astype(np.int32)[:, None]
    nmap[..., 2] = 230
    nmap[..., 3] = 255
    ntex = mt.texture(T.TEX_RGBA8, nmap)
    d_tex = mt.diffuse(tex=rgba8)
    d_f32 = mt.diffuse(tex=rgba32)
    m_mixmap = mt.mix_map(mt.diffuse((0.8, 0.8, 0.1)), mt.conductor((0.9, 0.9, 0.9)), l8)
    m_bump = mt.bump_map(mt.rough_conductor((0.95, 0.64, 0.54), roughness=0.5, roughness_tex=l32), l32)
    m_normal = mt.normal_map(mt.diffuse((0.2, 0.7, 0.3)), ntex)
    m_disp = mt.disperse(mt.dielectric((1, 1, 1), (1, 1, 1), int_ior=1.5), (1.45, 1.5, 1.55), (1.0, 1.0, 1.0))
    m_rd = mt.rough_dielectric((1, 1, 1), (0.9, 0.9, 1.0), roughness=0.25, tex=rgba8, trans_tex=rgba32)
    light = mt.emissive((8, 8, 8), 2.0, tex=-1)
    bg = mt.diffuse((0.3, 0.3, 0.35), tex=-1)
    env = mt.emissive((0.6, 0.7, 0.9), 0.8)
    parts = [
        quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), d_tex, uv=((0, 0), (3, 0), (3, 3), (0, 3))),
        quad((-4, 0, -4), (4, 0, -4), (4, 4, -4), (-4, 4, -4), d_f32, uv=((0, 0), (2, 0), (2, 2), (0, 2))),
        box((-3.2, 0, -1), (-2.2, 1.2, 0), m_mixmap),
        box((-1.8, 0, -1), (-0.8, 1.2, 0), m_bump),
        box((-0.4, 0, -1), (0.6, 1.2, 0), m_normal),
        uv_sphere((1.6, 0.7, -0.4), 0.7, m_disp, n_lat=10, n_lon=14),
        uv_sphere((3.0, 0.6, 0.4), 0.6, m_rd, n_lat=10, n_lon=14),
        quad((-1, 3.9, -1), (1, 3.9, -1), (1, 3.9, 1), (-1, 3.9, 1), light),
    ]
    sc = compile_scene([merge(parts)], [(0, np.eye(4))], mt, scene_diffuse=bg, scene_emissive=env, name="materials")
    sc.set_camera(eye=(0, 2.2, 6.5), look=(0, 0.8, 0), fov=0.8, aspect=aspect)
    return sc


def many_materials_scene(aspect=1.0) -> Scene:
    """More than 15 distinct material-tree shapes (every pair of the five BxDF families mixed, some under a bump map or a
    dispersion node) on a grid of small boxes under an area light and an environment light: more shading classes than the
    HIP backend's sort key holds (classes beyond the 15th share one) -- parity coverage, not a benchmark."""
    import itertools

    rng = np.random.default_rng(19)
    mt = MaterialTable()
    bump = mt.texture(T.TEX_L32F, rng.random((8, 8)).astype(F32))

    def leaf(kind, tint):
        if kind == 0:
            return mt.diffuse(tint)
        if kind == 1:
            return mt.conductor(tint)
        if kind == 2:
            return mt.rough_conductor(tint, roughness=0.3)
        if kind == 3:
            return mt.dielectric((1, 1, 1), tint, int_ior=1.5)
        return mt.rough_dielectric((1, 1, 1), tint, roughness=0.25, int_ior=1.45)

    mats = [leaf(k, (0.8, 0.6, 0.4)) for k in range(5)]
    for a, b in itertools.combinations(range(5), 2):  # 10 two-family mixes
        mats.append(mt.mix(leaf(a, (0.7, 0.7, 0.3)), leaf(b, (0.3, 0.6, 0.8)), 0.5))
    for k in range(5):                                 # each family under a bump map, and two under dispersion
        mats.append(mt.bump_map(leaf(k, (0.6, 0.8, 0.6)), bump))
    mats.append(mt.disperse(leaf(3, (1, 1, 1)), (1.45, 1.5, 1.55), (1.0, 1.0, 1.0)))
    mats.append(mt.disperse(leaf(4, (1, 1, 1)), (1.4, 1.5, 1.6), (1.0, 1.0, 1.0)))
    floor = mt.diffuse((0.6, 0.6, 0.6))
    light = mt.emissive((9, 9, 8), 1.5)
    bg = mt.diffuse((0.25, 0.3, 0.4))
    env = mt.emissive((0.7, 0.8, 1.0), 0.5)
    side = 5
    parts = [quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), floor),
             quad((-1.5, 3.5, -1.5), (1.5, 3.5, -1.5), (1.5, 3.5, 1.5), (-1.5, 3.5, 1.5), light)]
    for i, m in enumerate(mats):
        x, z = (i % side - (side - 1) / 2) * 1.4, (i // side - 2) * 1.4
        parts.append(box((x - 0.45, 0, z - 0.45), (x + 0.45, 0.6 + 0.05 * (i % 4), z + 0.45), m, rot_y=0.1 * i))
    sc = compile_scene([merge(parts)], [(0, np.eye(4))], mt, scene_diffuse=bg, scene_emissive=env, name="many-materials")
    sc.set_camera(eye=(0, 4.5, 7.0), look=(0, 0.3, 0), fov=0.8, aspect=aspect)
    return sc


def material_ball(aspect=1.0, detail=1.0) -> Scene:
    """Stand-in for BASELINE.json configs[3] ("Mitsuba scene", 1920x1080, 512 spp): the Mitsuba material
    preview ball is not redistributable and there is no network, so this is a procedural scene of the same
    character -- ~60 k triangles in one mesh (a finely tessellated coated ball on a pedestal, a rough glass
    ball, a polished ring of small spheres), layered materials (mix of rough conductor over textured diffuse,
    bump map, rough dielectric), a checker floor, two area lights and an environment light."""
    mt = MaterialTable()
    chk = mt.texture(T.TEX_RGBA8, checker_rgba8(64, 8, (235, 235, 225), (70, 75, 90)))
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:32, 0:32]
    bump = mt.texture(T.TEX_L8, (127 + 90 * np.sin(xx * 0.7) * np.sin(yy * 0.7) + rng.integers(-8, 8, (32, 32))).clip(0, 255).astype(np.uint8))
    coat = mt.mix(mt.rough_conductor((0.95, 0.78, 0.45), roughness=0.2), mt.diffuse((0.55, 0.12, 0.1)), 0.35)
    ball_mat = mt.bump_map(coat, bump)
    glass = mt.rough_dielectric((1, 1, 1), (0.92, 0.97, 0.95), roughness=0.15, int_ior=1.5)
    steel = mt.rough_conductor((0.8, 0.82, 0.85), roughness=0.3)
    floor = mt.diffuse(tex=chk)
    stand = mt.diffuse((0.25, 0.25, 0.28))
    key = mt.emissive((14, 13, 11), 1.0)
    fill = mt.emissive((3, 4, 6), 1.0)
    bg = mt.diffuse((0.35, 0.4, 0.5))
    env = mt.emissive((0.5, 0.6, 0.8), 0.6)
    lat, lon = max(6, int(120 * detail)), max(8, int(200 * detail))
    parts = [
        uv_sphere((0, 1.35, 0), 1.0, ball_mat, n_lat=lat, n_lon=lon),                                   # 48 k triangles at detail 1
        uv_sphere((1.9, 0.55, 0.9), 0.55, glass, n_lat=max(5, int(50 * detail)), n_lon=max(6, int(60 * detail))),  # 6 k
        box((-0.45, 0, -0.45), (0.45, 0.4, 0.45), stand),
        quad((-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8), floor, uv=((0, 0), (6, 0), (6, 6), (0, 6))),
        quad((-2.5, 4.5, 1.0), (-0.5, 4.5, 1.0), (-0.5, 4.5, 3.0), (-2.5, 4.5, 3.0), key),
        quad((3.5, 0.5, -2.0), (3.5, 2.5, -2.0), (3.5, 2.5, 0.0), (3.5, 0.5, 0.0), fill),
    ]
    n_ring = max(4, int(24 * detail))
    for i in range(n_ring):  # 24 small spheres of 10x12x2 = 240 triangles
        a = 2 * math.pi * i / n_ring
        parts.append(uv_sphere((1.45 * math.cos(a), 0.16, 1.45 * math.sin(a)), 0.16, steel, n_lat=max(4, int(10 * detail)), n_lon=max(5, int(12 * detail))))
    sc = compile_scene([merge(parts)], [(0, np.eye(4))], mt, scene_diffuse=bg, scene_emissive=env, name="material-ball")
    sc.set_camera(eye=(2.2, 2.6, 5.2), look=(0.2, 1.0, 0), fov=0.62, aspect=aspect)
    return sc


def displaced_grid(n=64, size=10.0, amp=0.6, mat=0, seed=3):
    """(n x n) quads = 2 n^2 triangles of a smooth random height field (flat normals)."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-size / 2, size / 2, n + 1)
    X, Z = np.meshgrid(g, g, indexing="ij")
    Y = np.zeros_like(X)
    for _ in range(6):
        fx, fz, ph = rng.uniform(0.3, 2.5), rng.uniform(0.3, 2.5), rng.uniform(0, 6.28)
        Y += amp / 3.0 * np.sin(fx * X + ph) * np.cos(fz * Z - ph)
    P = np.stack([X, Y, Z], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, c, b], axis=2).reshape(-1, 3, 3)   # wound so the normal points +y
    t2 = np.stack([a, d, c], axis=2).reshape(-1, 3, 3)
    verts = np.concatenate([t1, t2])
    u = (verts[..., [0, 2]] / size + 0.5)
    return Mesh(verts, _flat_normals(verts), u, np.full(len(verts), mat))


def instanced_stress(n_side=32, base_lat=23, base_lon=24, aspect=1.0) -> Scene:
    """BASELINE.json configs[4] stand-in: a ~1 k-triangle base mesh instanced n_side^2 times on a
    jittered grid (translation-only transforms, so reference quirk a-9(4) stays harmless) over a
    ground plane, one area light + environment light.  n_side=32 -> 1024 instances, >= 1 M
    instanced triangles."""
    rng = np.random.default_rng(11)
    mt = MaterialTable()
    blob = mt.mix(mt.diffuse((0.7, 0.3, 0.25)), mt.rough_conductor((0.9, 0.85, 0.6), roughness=0.3), 0.6)
    ground = mt.diffuse((0.55, 0.55, 0.5))
    light = mt.emissive((12, 11, 9), 1.0)
    bg = mt.diffuse((0.35, 0.45, 0.7))
    env = mt.emissive((0.8, 0.9, 1.0), 0.6)
    base = uv_sphere((0, 0.5, 0), 0.5, blob, n_lat=base_lat, n_lon=base_lon)
    spacing = 1.4
    ext = spacing * n_side / 2 + 2
    room = merge([quad((-ext, 0, -ext), (-ext, 0, ext), (ext, 0, ext), (ext, 0, -ext), ground),
                  quad((-4, 9.0, -4), (4, 9.0, -4), (4, 9.0, 4), (-4, 9.0, 4), light)])
    insts = [(1, np.eye(4))]
    for i in range(n_side):
        for j in range(n_side):
            jx, jz = rng.uniform(-0.2, 0.2, size=2)
            insts.append((0, translation(((i - (n_side - 1) / 2) * spacing + jx, rng.uniform(0, 0.3), (j - (n_side - 1) / 2) * spacing + jz))))
    sc = compile_scene([base, room], insts, mt, scene_diffuse=bg, scene_emissive=env, name=f"instanced-{n_side * n_side}x{len(base.verts)}")
    sc.set_camera(eye=(0, 0.35 * ext + 3, 1.15 * ext), look=(0, 0.0, 0), fov=0.75, aspect=aspect)
    return sc


def unique_stress(n=512, aspect=1.0) -> Scene:
    """A 2 n^2-triangle height field with unique geometry (n=708 -> 1.0 M triangles): unlike the
    instanced scene its BVH and triangles do not fit the caches, which is what exercises HBM."""
    mt = MaterialTable()
    tex = mt.texture(T.TEX_RGBA8, checker_rgba8(128, 16, (200, 190, 170), (90, 110, 90)))
    terrain = mt.diffuse((0.6, 0.6, 0.6), tex=tex)
    light = mt.emissive((15, 14, 12), 1.0)
    bg = mt.diffuse((0.4, 0.5, 0.8))
    env = mt.emissive((0.9, 0.95, 1.0), 0.7)
    mesh = merge([displaced_grid(n, 20.0, 1.2, terrain), quad((-3, 8.0, -3), (3, 8.0, -3), (3, 8.0, 3), (-3, 8.0, 3), light)])
    sc = compile_scene([mesh], [(0, np.eye(4))], mt, scene_diffuse=bg, scene_emissive=env, name=f"terrain-{len(mesh.verts)}")
    sc.set_camera(eye=(0, 7.0, 16.0), look=(0, 0.0, 0), fov=0.8, aspect=aspect)
    return sc


def make_seeds(spp: int, bounces: int, base: int = 0xC0FFEE) -> np.ndarray:
    """seeds[s][k] = splitmix32(base + s*(1+B) + k); layout [s][0] =
    camera seed, [s][1+b] = shade seed of bounce b.  Stands in for Go's math/rand draws
    (tracer/opencl/tracer.go:222, pipeline.go:146)."""
    i = (np.arange(spp * (1 + bounces), dtype=np.uint64) + np.uint64(base)) & np.uint64(0xFFFFFFFF)
    z = (i + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    z = ((z ^ (z >> np.uint64(16))) * np.uint64(0x21F0AAAD)) & np.uint64(0xFFFFFFFF)
    z = ((z ^ (z >> np.uint64(15))) * np.uint64(0x735A2D97)) & np.uint64(0xFFFFFFFF)
    z = z ^ (z >> np.uint64(15))
    return z.astype(np.uint32)


SCENES = {
    "cornell": lambda aspect=1.0: cornell_box("layered", aspect),
    "cornell-diffuse": lambda aspect=1.0: cornell_box("diffuse", aspect),
    "cornell-refbvh": lambda aspect=1.0: cornell_box("layered", aspect, compiler="reference"),
    "sphere": sphere_scene,
    "cubes": lambda aspect=1.0: instanced_cubes(3, aspect),
    "materials": textured_materials_scene,
    "many-materials": many_materials_scene,
    "transformed": transformed_instances,
    "instanced": lambda aspect=1.0: instanced_stress(32, aspect=aspect),
    "instanced-small": lambda aspect=1.0: instanced_stress(6, 9, 10, aspect=aspect),
    "material-ball": material_ball,
    "material-ball-small": lambda aspect=1.0: material_ball(aspect, detail=0.12),
    "terrain": lambda aspect=1.0: unique_stress(708, aspect),
    "terrain-small": lambda aspect=1.0: unique_stress(48, aspect),
}
