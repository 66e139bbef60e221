"""The scene's two-level BVH rebuilt ON THE DEVICE (polaris_hip_build_bvh, include/polaris_hip.h) -- an alternative to the CPU
producers (polaris_amd/scenes.py's numpy builder, polaris_amd/host/scene_compiler.cpp's restatement of the reference's
asset/compiler/bvh/bvh_builder.go:100-308).  Plumbing only: the tree is built by the HIP library; this module derives the
builder's inputs from a compiled scene, puts the triangle arrays in the order the new leaves name, and returns a Scene with
exactly the arrays a PolarisSceneView carries.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import dataclasses

import numpy as np

from . import ctypes_api as T


def mesh_ranges_and_instance_boxes(scene):
    """From a compiled scene: per distinct mesh BVH root (ascending) the triangle range its leaves cover, per instance its mesh
    ordinal and the world box of its top-level leaf (the scene reader's box, kept as it is: SURVEY.md 8a-9 (4))."""
    nodes = scene.bvh_nodes
    roots = np.unique(scene.mesh_instances["bvh_root"]).astype(np.int64)
    first, count = [], []
    for k, r in enumerate(roots):
        end = int(roots[k + 1]) if k + 1 < len(roots) else len(nodes)
        sub = nodes[int(r):end]
        leaf = (sub["ldata"] <= 0) & (sub["rdata"] > 0)
        if not leaf.any():
            raise ValueError(f"mesh BVH at node {r} has no triangle leaf")
        f = int((-sub["ldata"][leaf].astype(np.int64)).min())
        n = int(sub["rdata"][leaf].astype(np.int64).sum())
        first.append(f)
        count.append(n)
    order = np.argsort(first, kind="stable")
    roots, first, count = roots[order], np.asarray(first)[order], np.asarray(count)[order]
    if first[0] != 0 or not np.array_equal(first[1:], np.cumsum(count)[:-1]) or int(count.sum()) != len(scene.material_index):
        raise ValueError("the meshes' triangle ranges do not tile the triangle arrays")
    ordinal = {int(r): k for k, r in enumerate(roots)}
    top = nodes[: int(roots.min())]
    ni = len(scene.mesh_instances)
    boxes = np.zeros((ni, 6), np.float32)
    seen = np.zeros(ni, bool)
    for nd in top[(top["ldata"] <= 0) & (top["rdata"] == 0)]:
        i = int(-int(nd["ldata"]))
        boxes[i, :3], boxes[i, 3:], seen[i] = nd["min"], nd["max"], True
    if not seen.all():
        raise ValueError("an instance has no top-level leaf")
    inst_mesh = np.array([ordinal[int(r)] for r in scene.mesh_instances["bvh_root"]], np.uint32)
    return first.astype(np.uint32), count.astype(np.uint32), boxes, inst_mesh


def rebuild_on_device(scene, max_leaf_tris: int = 4, device: int = 0, algorithm: str = "sah"):
    """Returns (scene with the BVH built by polaris_hip_build_bvh, {"device_ms": .., "num_nodes": ..}).  algorithm: "sah" (binned
    surface-area heuristic, level by level: the default) or "lbvh" (linear BVH: faster to build, slower to trace)."""
    lib = T.load_library()
    first, count, boxes, inst_mesh = mesh_ranges_and_instance_boxes(scene)
    nt, ni = len(scene.material_index), len(scene.mesh_instances)
    verts = np.ascontiguousarray(scene.vertices, np.float32)
    inp = T.BvhBuildInput()
    inp.vertices, inp.num_triangles = verts.ctypes.data, nt
    inp.mesh_first_tri, inp.mesh_num_tris, inp.num_meshes = first.ctypes.data, count.ctypes.data, len(first)
    inp.instance_boxes, inp.instance_mesh, inp.num_instances = boxes.ctypes.data, inst_mesh.ctypes.data, ni
    inp.max_leaf_tris = max_leaf_tris
    inp.algorithm = {"sah": T.BVH_SAH, "lbvh": T.BVH_LBVH}[algorithm]
    cap = 2 * (nt + ni)
    nodes = np.zeros(cap, T.BVH_NODE)
    order = np.zeros(nt, np.uint32)
    roots = np.zeros(len(first), np.uint32)
    n_nodes, ms = C.c_uint32(), C.c_double()
    rc = lib.polaris_hip_build_bvh(device, C.byref(inp), nodes.ctypes.data, cap, C.byref(n_nodes), order.ctypes.data, roots.ctypes.data, C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"polaris_hip_build_bvh failed ({rc}): {lib.polaris_hip_build_bvh_error().decode()}")
    # the triangle arrays in the order the new leaves name; emissive triangles follow the permutation
    o = order.astype(np.int64)
    v3 = (o[:, None] * 3 + np.arange(3)[None, :]).reshape(-1)
    inverse = np.empty(nt, np.int64)
    inverse[o] = np.arange(nt)
    inst = scene.mesh_instances.copy()
    inst["bvh_root"] = roots[inst_mesh]
    ems = scene.emissives.copy()
    area = ems["type"] == T.EMISSIVE_AREA
    ems["tri_index"][area] = inverse[ems["tri_index"][area].astype(np.int64)]
    out = dataclasses.replace(scene, bvh_nodes=nodes[: n_nodes.value].copy(), mesh_instances=inst, emissives=ems,
                              vertices=np.ascontiguousarray(scene.vertices[v3]), normals=np.ascontiguousarray(scene.normals[v3]),
                              uvs=np.ascontiguousarray(scene.uvs[v3]), material_index=np.ascontiguousarray(scene.material_index[o]),
                              name=scene.name + "-gpubvh")
    return out, {"device_ms": ms.value, "num_nodes": int(n_nodes.value), "algorithm": algorithm}
