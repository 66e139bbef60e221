"""polaris_hip_build_bvh (the stretch): the scene's two-level BVH built on the device -- an ALTERNATIVE producer of
the reference's node arrays (optimized_scene.go:14-64), not a copy of its builder (asset/compiler/bvh/bvh_builder.go:100-308).
It cannot produce the reference compiler's tree, so what is pinned is what any producer must satisfy: the tree is valid under
the upload's rules, every triangle sits in exactly one leaf of its mesh, every box contains what is below it, the instance
leaves keep the reader's boxes -- and a scene on the tree built here traces to the CPU oracle's result for the same arrays BIT
FOR BIT, and to (statistically) the image of the original tree."""
import numpy as np
import pytest

from conftest import bits, make_hip_tracer

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "cubes", "transformed", "materials", "material-ball-small", "terrain-small", "instanced-small"]


def check_tree(sc, old):
    """Structural validity of the rebuilt scene (numpy / Python: the trees of the test scenes are small)."""
    nodes = sc.bvh_nodes
    nt = sc.num_triangles
    seen = np.zeros(nt, np.int32)
    inst_seen = np.zeros(len(sc.mesh_instances), np.int32)
    verts = sc.vertices[:, :3].reshape(nt, 3, 3)

    def walk(root, top):
        depth_max = 0
        stack = [(int(root), 0)]
        visited = 0
        while stack:
            i, d = stack.pop()
            visited += 1
            depth_max = max(depth_max, d)
            nd = nodes[i]
            if nd["ldata"] > 0:                                   # inner node: both children inside its box
                assert nd["rdata"] > 0
                for c in (int(nd["ldata"]), int(nd["rdata"])):
                    assert 0 < c < len(nodes)
                    assert (nodes[c]["min"] >= nd["min"]).all() and (nodes[c]["max"] <= nd["max"]).all(), (i, c)
                    stack.append((c, d + 1))
            elif top:
                assert nd["rdata"] == 0
                inst_seen[-int(nd["ldata"])] += 1
            else:
                f, n = -int(nd["ldata"]), int(nd["rdata"])
                assert n >= 1 and f + n <= nt
                seen[f:f + n] += 1
                tv = verts[f:f + n].reshape(-1, 3)
                assert (tv >= nd["min"]).all() and (tv <= nd["max"]).all(), i   # the leaf's box contains its triangles
        return depth_max, visited

    top_depth, _ = walk(0, True)
    assert (inst_seen == 1).all()
    depths = {}
    for r in np.unique(sc.mesh_instances["bvh_root"]):
        assert r != 0
        depths[int(r)], _ = walk(int(r), False)
    assert (seen == 1).all(), "every triangle in exactly one leaf"
    # the same triangles as before, permuted within their mesh (compare as multisets of vertex triples + material)
    key_new = np.concatenate([verts.reshape(nt, 9), sc.material_index[:, None].astype(np.float32)], axis=1)
    ov = old.vertices[:, :3].reshape(nt, 9)
    key_old = np.concatenate([ov, old.material_index[:, None].astype(np.float32)], axis=1)
    assert np.array_equal(key_new[np.lexsort(key_new.T[::-1])], key_old[np.lexsort(key_old.T[::-1])])
    # instance leaves keep the boxes the scene's producer gave the instances
    return top_depth + 1 + max(depths.values())


@pytest.mark.parametrize("name", SCENES)
def test_device_built_bvh_is_valid_and_traces_like_the_oracle(built, oracle, name):
    from oracle import pybind as ob
    from polaris_amd import bvh_build, scenes

    old = scenes.SCENES[name]()
    for max_leaf in (1, 4):
        sc, info = bvh_build.rebuild_on_device(old, max_leaf_tris=max_leaf)
        assert info["num_nodes"] == len(sc.bvh_nodes) <= 2 * (sc.num_triangles + len(sc.mesh_instances))
        leaves = sc.bvh_nodes[(sc.bvh_nodes["ldata"] <= 0) & (sc.bvh_nodes["rdata"] > 0)]
        assert leaves["rdata"].max() <= max_leaf
        stack_need = check_tree(sc, old)
        assert stack_need < 32, stack_need                     # the upload's (and the reference's) 32-entry traversal stack
        W, H, spp, B = 80, 60, 3, 4
        seeds = scenes.make_seeds(spp, B, base=77)
        req = ob.make_request(W, H, spp=spp, bounces=B)
        want, wst, _ = oracle.trace(sc, req, seeds)
        tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
        try:
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B), seeds)
            got, st = tr.read_accumulator(0), tr.last_trace_stats
        finally:
            tr.Close()
        assert list(st.rays_per_bounce[:B]) == list(wst.rays_per_bounce[:B]) and list(st.occl_per_bounce[:B]) == list(wst.occl_per_bounce[:B])
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), (name, max_leaf)
        # the same picture as on the original tree: a tree decides no hit except exact ties (and, through them, a few paths)
        ref, _, _ = oracle.trace(old, req, seeds)
        diff = np.abs(want[..., :3] - ref[..., :3]).reshape(-1, 3).max(axis=1)
        assert (diff > 1e-6).mean() < 0.02, (name, float((diff > 1e-6).mean()))


def test_the_build_is_deterministic(built):
    """Two builds give the same arrays, byte for byte (the sort keys are unique, node places come from a prefix sum)."""
    from polaris_amd import bvh_build, scenes

    old = scenes.SCENES["material-ball-small"]()
    a, _ = bvh_build.rebuild_on_device(old, max_leaf_tris=2)
    b, _ = bvh_build.rebuild_on_device(old, max_leaf_tris=2)
    assert a.bvh_nodes.tobytes() == b.bvh_nodes.tobytes() and np.array_equal(a.material_index, b.material_index) and np.array_equal(a.vertices, b.vertices)


def test_build_refuses_malformed_input(built):
    import ctypes as C

    from polaris_amd import ctypes_api as T

    lib = T.load_library()
    inp = T.BvhBuildInput()
    n = C.c_uint32()
    assert lib.polaris_hip_build_bvh(0, C.byref(inp), None, 0, C.byref(n), None, None, None) == 2
    assert b"null" in lib.polaris_hip_build_bvh_error()
    verts = np.zeros((6, 4), np.float32)
    first, count = np.array([0], np.uint32), np.array([3], np.uint32)       # 3 triangles claimed, 2 present
    boxes, im = np.zeros((1, 6), np.float32), np.zeros(1, np.uint32)
    inp.vertices, inp.num_triangles = verts.ctypes.data, 2
    inp.mesh_first_tri, inp.mesh_num_tris, inp.num_meshes = first.ctypes.data, count.ctypes.data, 1
    inp.instance_boxes, inp.instance_mesh, inp.num_instances, inp.max_leaf_tris = boxes.ctypes.data, im.ctypes.data, 1, 4
    nodes, order, roots = np.zeros(8, T.BVH_NODE), np.zeros(2, np.uint32), np.zeros(1, np.uint32)
    assert lib.polaris_hip_build_bvh(0, C.byref(inp), nodes.ctypes.data, 8, C.byref(n), order.ctypes.data, roots.ctypes.data, None) == 2
    assert b"range" in lib.polaris_hip_build_bvh_error()
    count[0] = 2
    inp.max_leaf_tris = 99
    assert lib.polaris_hip_build_bvh(0, C.byref(inp), nodes.ctypes.data, 8, C.byref(n), order.ctypes.data, roots.ctypes.data, None) == 2
    inp.max_leaf_tris = 4
    assert lib.polaris_hip_build_bvh(0, C.byref(inp), nodes.ctypes.data, 2, C.byref(n), order.ctypes.data, roots.ctypes.data, None) == 2   # capacity
    assert lib.polaris_hip_build_bvh(0, C.byref(inp), nodes.ctypes.data, 8, C.byref(n), order.ctypes.data, roots.ctypes.data, None) == 0
    assert n.value == 2 and nodes[0]["rdata"] == 0 and nodes[1]["rdata"] == 2   # one instance leaf + one leaf of two (degenerate) triangles
