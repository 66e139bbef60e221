"""Save / load a compiled scene (polaris_amd.scenes.Scene) as a flat .npz -- used for the golden
fixtures under tests/golden/ (inputs + expected outputs travel together, so a fixture stays valid
even if the procedural scene builders change)."""
from __future__ import annotations

import numpy as np

from . import ctypes_api as T
from .scenes import Scene

_ARRAYS = {"bvh_nodes": T.BVH_NODE, "mesh_instances": T.MESH_INSTANCE, "material_nodes": T.MATERIAL_NODE,
           "emissives": T.EMISSIVE, "texture_meta": T.TEXTURE_META}
_PLAIN = ("texture_data", "vertices", "normals", "uvs", "material_index", "eye", "frustum")


def scene_to_dict(sc: Scene, prefix="scene_") -> dict:
    d = {}
    for k in _ARRAYS:
        d[prefix + k] = np.frombuffer(getattr(sc, k).tobytes(), dtype=np.uint8)
    for k in _PLAIN:
        d[prefix + k] = getattr(sc, k)
    d[prefix + "ints"] = np.array([sc.scene_diffuse_mat_index, sc.scene_emissive_mat_index, sc.bvh_max_depth], dtype=np.int64)
    d[prefix + "name"] = np.array(sc.name)
    return d


def scene_from_dict(d, prefix="scene_") -> Scene:
    kw = {}
    for k, dt in _ARRAYS.items():
        kw[k] = np.frombuffer(d[prefix + k].tobytes(), dtype=dt).copy()
    for k in _PLAIN:
        kw[k] = np.ascontiguousarray(d[prefix + k])
    ints = d[prefix + "ints"]
    return Scene(scene_diffuse_mat_index=int(ints[0]), scene_emissive_mat_index=int(ints[1]), bvh_max_depth=int(ints[2]),
                 name=str(d[prefix + "name"]), **kw)
