"""CPU tests of the upload-time scene re-layout (polaris_amd/csrc/scene_layout.h).

The HIP backend subdivides big triangle leaves when a scene is uploaded (option max_leaf_tris).
tests/tools/layout_check.cpp walks the resulting layout on the CPU with the kernels' traversal
rules; here we check that

* the hit record of every ray (triangle, instance, bits of t/u/v) is identical whatever the leaf
  size -- the added boxes only cull inside a leaf the reference traversal had reached anyway;
* primary hits agree with the CPU oracle's taps (reference traversal order) bit for bit;
* the subdivision actually lowers the triangle tests per ray on a big-leaf BVH.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from polaris_amd import ctypes_api as T
from polaris_amd import scenes

BUILD = os.path.join(ROOT, "tests", "_build")
SRC = os.path.join(ROOT, "tests", "tools", "layout_check.cpp")
LIB = os.path.join(BUILD, "liblayout_check.so")


@pytest.fixture(scope="module")
def harness():
    os.makedirs(BUILD, exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "polaris_amd", "csrc", "scene_layout.h"), os.path.join(ROOT, "include", "polaris_math.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "polaris_amd", "csrc"), SRC, "-o", LIB])
    lib = C.CDLL(LIB)
    for fn in (lib.layout_check_traverse,):
        fn.argtypes = [C.POINTER(T.SceneView), C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    return lib


def traverse(lib, sc, rays, max_leaf, any_hit=False):
    """Walk the uploaded layout on the CPU with k_trace's rules."""
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    hit = np.zeros((rays.shape[0], 6), np.int32)
    cnt = np.zeros(7, np.uint64)
    err = C.create_string_buffer(256)
    view = T.scene_view(sc)
    fn = lib.layout_check_traverse
    rc = fn(C.byref(view), max_leaf, rays.ctypes.data, rays.shape[0], int(any_hit), hit.ctypes.data, cnt.ctypes.data, err, 256)
    assert rc == 0, err.value.decode()
    return hit, cnt


def camera_and_bounce_rays(oracle, sc, W=48, H=36, seed=7):
    """Primary rays of one sample (from the oracle's tap) + random rays leaving the primary hit points."""
    from oracle import pybind as ob

    req = ob.make_request(W, H, spp=1, bounces=1, rr=2)
    _, _, taps = oracle.trace(sc, req, scenes.make_seeds(1, 1, base=seed), tap_sample=0)
    prim = taps["primary_rays"].copy()
    rng = np.random.default_rng(seed)
    hitm = taps["primary_hit"] != 0
    t = taps["primary_wuvt"][hitm, 3:4]
    p = prim[hitm, 0:3] + prim[hitm, 4:7] * t
    d = rng.normal(size=p.shape).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sec = np.zeros((p.shape[0], 8), np.float32)
    sec[:, 0:3] = p + d * np.float32(1e-3)
    sec[:, 3] = np.float32(3.402823466e+38)
    sec[:, 4:7] = d
    # short shadow-like rays with a finite maxDist
    sh = sec.copy()
    sh[:, 3] = rng.uniform(0.05, 2.0, size=sh.shape[0]).astype(np.float32)
    return prim, taps, np.concatenate([prim, sec, sh], axis=0)


SCENE_MAKERS = {
    "cornell-refbvh": lambda: scenes.cornell_box(compiler="reference"),
    "cornell": lambda: scenes.cornell_box(),
    "sphere": lambda: scenes.sphere_scene(),
    "cubes": lambda: scenes.instanced_cubes(),
    "transformed": lambda: scenes.transformed_instances(),
    "terrain-small": lambda: scenes.SCENES["terrain-small"](),
}


@pytest.mark.parametrize("name", list(SCENE_MAKERS))
def test_leaf_subdivision_never_changes_a_hit(harness, oracle, name):
    sc = SCENE_MAKERS[name]()
    prim, taps, rays = camera_and_bounce_rays(oracle, sc)
    base, cnt0 = traverse(harness, sc, rays, 0)
    base_any, _ = traverse(harness, sc, rays, 0, any_hit=True)
    assert base[:, 5].sum() > 0
    # primary hits against the oracle's reference-order traversal
    n = prim.shape[0]
    hm = taps["primary_hit"] != 0
    assert np.array_equal(base[:n, 5] != 0, hm)
    assert np.array_equal(base[:n][hm][:, [1, 0]], taps["primary_tri"][hm])
    assert np.array_equal(base[:n][hm][:, 2], taps["primary_wuvt"][hm][:, 3].view(np.int32))
    for max_leaf in (1, 2, 3, 4, 8):
        got, cnt = traverse(harness, sc, rays, max_leaf)
        assert np.array_equal(got, base), f"{name}: closest hits differ with max_leaf_tris={max_leaf}"
        got_any, _ = traverse(harness, sc, rays, max_leaf, any_hit=True)
        assert np.array_equal(got_any[:, 5], base_any[:, 5]), f"{name}: occlusion differs with max_leaf_tris={max_leaf}"
        assert cnt[5] <= 32  # traversal stack


def _obj_room(tmp):
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import obj_fixtures
    from polaris_amd import host_api

    return host_api.read_scene(obj_fixtures.write_cornell(str(tmp)), aspect=4 / 3)


RANDOM_RAY_SCENES = {
    "obj-room": _obj_room,  # rotated + scaled instances: the reader's instance boxes do NOT bound them (wavefront.go:514-519)
    "transformed": lambda tmp: scenes.transformed_instances(),
    "cubes": lambda tmp: scenes.instanced_cubes(),
    "cornell-refbvh": lambda tmp: scenes.cornell_box(compiler="reference"),
    "materials": lambda tmp: scenes.textured_materials_scene(),
    "material-ball-small": lambda tmp: scenes.SCENES["material-ball-small"](),
}


@pytest.mark.parametrize("name", list(RANDOM_RAY_SCENES))
def test_layout_traversal_equals_the_reference_query_on_random_rays(harness, oracle, name, tmp_path, request):
    """The kernels' traversal rules (near child first, distance culling where a box really bounds
    its subtree, DFS-rank tie break) over the uploaded layout vs rayIntersectionQuery /
    rayIntersectionTest of the oracle -- and of the compiled reference where it exists -- on
    rays the camera never produces: hit flags, instance, triangle and the bits of t, u, v."""
    from oracle import pybind as ob

    sc = RANDOM_RAY_SCENES[name](tmp_path)
    box_lo = np.minimum(sc.vertices[:, :3].min(axis=0), -1.0) - 0.5
    box_hi = np.maximum(sc.vertices[:, :3].max(axis=0), 1.0) + 0.5
    rng = np.random.default_rng(5)
    n = 60000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(box_lo, box_hi, size=(n, 3))
    d = rng.normal(size=(n, 3))
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3] = np.float32(3.402823466e+38)
    rays[: n // 8, 4 + (np.arange(n // 8) % 3)] = 0.0  # axis-parallel components: infinities in the slab test
    shadow = rays.copy()
    shadow[:, 3] = rng.uniform(0.05, 4.0, size=n)
    ref = ob.Oracle("ref_pm") if ob.available("ref_pm") else None
    h, wuvt, it = oracle.intersect(sc, rays)
    ha, _, _ = oracle.intersect(sc, shadow, any_hit=True)
    assert 0 < h.sum() < n
    if ref is not None:
        rh, rw, rit = ref.intersect(sc, rays)
        rha, _, _ = ref.intersect(sc, shadow, any_hit=True)
        assert np.array_equal(rh, h) and np.array_equal(rha, ha)
        assert np.array_equal(rw.view(np.uint32)[h != 0], wuvt.view(np.uint32)[h != 0]) and np.array_equal(rit[h != 0], it[h != 0])
    hm = h != 0
    for max_leaf in (0, 2):
        got, cnt = traverse(harness, sc, rays, max_leaf)
        assert np.array_equal(got[:, 5] != 0, hm), f"{name}: hit flags differ (max_leaf_tris={max_leaf})"
        assert np.array_equal(got[hm][:, [1, 0]], it[hm])
        assert np.array_equal(got[hm][:, 2], wuvt[hm][:, 3].view(np.int32))
        assert np.array_equal(got[hm][:, 3:5], wuvt[hm][:, 1:3].view(np.int32))
        got_any, _ = traverse(harness, sc, shadow, max_leaf, any_hit=True)
        assert np.array_equal(got_any[:, 5] != 0, ha != 0)
        if name == "obj-room":
            assert cnt[6] > 0  # the non-bounding instance boxes were detected (and are never culled by distance)
        if name == "cornell-refbvh":
            assert cnt[6] == 0


def face_rays(sc, rng, n):
    """Rays that sit exactly ON box faces and run exactly parallel to them: origins take coordinates of BVH box bounds (node
    mins / maxes are vertex coordinates, so these are also the planes of axis-aligned walls), directions have one or two
    exact zeros.  0 * inf = NaN in the slab test is the one place where the test is not monotone in the box bounds."""
    nodes = sc.bvh_nodes
    rays = np.zeros((n, 8), np.float32)
    pick = rng.integers(0, len(nodes), size=n)
    lo, hi = nodes["min"][pick], nodes["max"][pick]
    o = rng.uniform(lo - 0.25, hi + 0.25).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    for k in range(n):
        a = k % 3
        o[k, a] = (lo if (k // 3) % 2 else hi)[k, a]     # on a face of some box
        d[k, a] = 0.0 if (k // 6) % 2 == 0 else -0.0     # parallel to it (both signs of zero: inv = +inf / -inf)
        if (k // 12) % 3 == 0:                           # ... and on a second face too
            b = (a + 1) % 3
            o[k, b] = (lo if (k // 36) % 2 else hi)[k, b]
            d[k, b] = 0.0
    nz = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nz > 0, d / np.maximum(nz, 1e-20), np.float32([0, 0, 1])).astype(np.float32)
    rays[:, 0:3], rays[:, 4:7] = o, d
    rays[:, 3] = np.float32(3.402823466e+38)
    return rays


QUAD_SCENES = dict(RANDOM_RAY_SCENES)
QUAD_SCENES.update({"cornell": lambda tmp: scenes.cornell_box(), "terrain-small": lambda tmp: scenes.SCENES["terrain-small"](),
                    "instanced-small": lambda tmp: scenes.SCENES["instanced-small"](), "sphere": lambda tmp: scenes.sphere_scene()})


@pytest.mark.parametrize("name", ["cornell", "sphere", "cubes", "cornell-refbvh"])
def test_small_scenes_order_their_triangle_slots_by_leaf_area_and_pack_one_word_per_slot(harness, name, tmp_path):
    """scene_layout.h, for scenes of < 2 047 triangle slots (the tiny-scene traversal mode keeps a prefix of the slots in LDS):
    leaves take their slots in descending order of their box's surface area, and every slot carries
    rank << 19 | shading class << 11 | triangle -- which decodes back to the (rank, triangle | class << 24) the other modes use."""
    sc = QUAD_SCENES[name](tmp_path)
    cap = 4096
    area = np.zeros(cap, np.float32)
    word, rank, orig = (np.zeros(cap, np.uint32) for _ in range(3))
    err = C.create_string_buffer(256)
    view = T.scene_view(sc)
    harness.layout_check_slots.argtypes = [C.POINTER(T.SceneView), C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    for max_leaf in (0, 2):
        n = harness.layout_check_slots(C.byref(view), max_leaf, cap, area.ctypes.data, word.ctypes.data, rank.ctypes.data, orig.ctypes.data, err, 256)
        assert n > 0, err.value.decode()
        a = area[:n].copy()
        a[a < 0] = np.float32(3.0e38)                           # (a mesh tree that is ONE leaf has no box: it is always reached, and comes first)
        assert (a[1:] <= a[:-1]).all(), "slots are not in descending order of their leaf's box area"
        w, r, o = word[:n].astype(np.uint64), rank[:n].astype(np.uint64), orig[:n].astype(np.uint64)
        assert ((w >> np.uint64(19)) == r).all() and (w >> np.uint64(30) == 0).all()
        assert (((w & np.uint64(0x7FF)) | (((w >> np.uint64(11)) & np.uint64(0xFF)) << np.uint64(24))) == o).all()
        assert len(set(o.tolist())) == len(set((o & np.uint64(0xFFFFFF)).tolist()))   # (one class per triangle)


def test_subdivision_cuts_triangle_tests_on_big_leaves(harness, oracle):
    sc = scenes.cornell_box(compiler="reference")  # leaves of up to 10 triangles, as `polaris render` compiles them
    _, _, rays = camera_and_bounce_rays(oracle, sc)
    _, c0 = traverse(harness, sc, rays, 0)
    _, c2 = traverse(harness, sc, rays, 2)
    assert c2[1] < 0.8 * c0[1]
    assert c2[4] == c0[4]  # same triangles, only regrouped


# ---- corrupted scenes must be rejected (or be harmless), never walked out of bounds ----------------------
@pytest.fixture(scope="module")
def asan_harness():
    """The same harness under AddressSanitizer + UBSan, in a child process (the sanitizer runtime must be
    the first library of its process)."""
    out = os.path.join(BUILD, "layout_check_asan")
    drv = os.path.join(ROOT, "tests", "tools", "layout_corrupt.cpp")
    deps = [SRC, drv, os.path.join(ROOT, "polaris_amd", "csrc", "scene_layout.h")]
    os.makedirs(BUILD, exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "polaris_amd", "csrc"), SRC, drv, "-o", out])
    return out


@pytest.mark.parametrize("name", ["cubes", "cornell-refbvh", "materials"])
def test_corrupted_scenes_are_rejected_not_followed(asan_harness, name, tmp_path):
    """Every index the GPU would dereference is checked at upload (polaris_amd/csrc/scene_layout.h).
    tests/tools/layout_corrupt.cpp overwrites random words of every scene array with hostile values,
    builds the layout and, when the layout is accepted, traverses it: under ASan/UBSan nothing may be
    read out of bounds and no traversal may run away."""
    sc = RANDOM_RAY_SCENES[name](tmp_path) if name in RANDOM_RAY_SCENES else scenes.SCENES[name]()
    raw = str(tmp_path / "scene.bin")
    # flat binary for the C++ driver: counts then arrays
    with open(raw, "wb") as f:
        arrs = [sc.bvh_nodes, sc.mesh_instances, sc.material_nodes, sc.emissives, sc.texture_meta, sc.texture_data, sc.vertices, sc.normals, sc.uvs,
                sc.material_index]
        hdr = np.array([len(a) for a in arrs] + [sc.scene_diffuse_mat_index, sc.scene_emissive_mat_index], dtype=np.int64)
        f.write(hdr.tobytes())
        for a in arrs:
            f.write(np.ascontiguousarray(a).tobytes())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([asan_harness, raw, "400", "7"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
    accepted, rejected = (int(t.split("=")[1]) for t in p.stdout.split()[:2])
    assert rejected > 50 and accepted + rejected == 400


def _layout_error(lib, sc, max_leaf=2):
    hit = np.zeros((1, 6), np.int32)
    cnt = np.zeros(7, np.uint64)
    rays = np.zeros((1, 8), np.float32)
    rays[0, 4] = 1.0
    err = C.create_string_buffer(256)
    view = T.scene_view(sc)
    rc = lib.layout_check_traverse(C.byref(view), max_leaf, rays.ctypes.data, 1, 0, hit.ctypes.data, cnt.ctypes.data, err, 256)
    return rc, err.value.decode()


@pytest.mark.parametrize("depth", [6, 20, 60])
def test_a_mesh_bvh_that_is_a_dag_is_rejected_at_once(harness, depth):
    """A chain of inner nodes whose two children are the SAME next node passes every index check, but
    every walk of it -- the upload's and the GPU traversal's -- costs 2^depth steps.  A node reached
    twice inside one tree is rejected (and quickly: 60 levels used to hang the upload)."""
    import time

    sc = scenes.cornell_box()
    nodes = sc.bvh_nodes.copy()
    roots = sorted({int(r) for r in sc.mesh_instances["bvh_root"]})
    root = roots[0]
    assert nodes[root]["ldata"] > 0, "mesh root is an inner node"
    first_new = len(nodes)
    chain = np.zeros(depth, dtype=nodes.dtype)
    for k in range(depth):
        chain[k]["min"], chain[k]["max"] = nodes[root]["min"], nodes[root]["max"]
        nxt = first_new + k + 1 if k + 1 < depth else int(nodes[root]["ldata"])
        chain[k]["ldata"], chain[k]["rdata"] = nxt, nxt          # both children -> the same node
    nodes = np.concatenate([nodes, chain])
    nodes[root]["ldata"] = first_new                              # hang the chain under the mesh root
    sc.bvh_nodes = nodes
    t = time.perf_counter()
    rc, msg = _layout_error(harness, sc)
    assert rc != 0 and ("reachable twice" in msg or (depth > 32 and "too deep" in msg)), msg  # left-first: a chain deeper than the stack trips the depth limit first
    assert time.perf_counter() - t < 5.0


def test_texture_index_of_mix_and_disperse_nodes_is_range_checked(harness):
    """scene_diffuse_mat_index / an emissive's material may point at ANY node, and the shading code
    reads that node's texture: an operator node's unused `tex` must be -1 or a real texture."""
    sc = scenes.cornell_box("layered")
    ops = np.nonzero(sc.material_nodes["type"] == T.OP_MIX)[0]
    assert len(ops) > 0
    assert _layout_error(harness, sc)[0] == 0
    mn = sc.material_nodes.copy()
    mn[ops[0]]["tex"] = 12345
    sc.material_nodes = mn
    rc, msg = _layout_error(harness, sc)
    assert rc != 0 and "texture out of range" in msg, msg


def test_multi_byte_texels_must_be_dword_aligned(harness):
    """The shade kernels fetch texels as whole dwords (the reference reads them through uchar4 / float / float4 pointers,
    texture_sampler.cl:42-91): RGBA8 / L32F / RGBA32F data at an unaligned offset is rejected at upload; 1-byte L8 texels
    may start anywhere."""
    sc = scenes.textured_materials_scene()
    assert _layout_error(harness, sc)[0] == 0
    for fmt, ok in ((T.TEX_RGBA8, False), (T.TEX_L32F, False), (T.TEX_RGBA32F, False), (T.TEX_L8, True)):
        s2 = scenes.textured_materials_scene()
        t = int(np.nonzero(s2.texture_meta["format"] == fmt)[0][0])
        meta = s2.texture_meta.copy()
        s2.texture_data = np.concatenate([s2.texture_data, np.zeros(1, np.uint8), s2.texture_data])  # a copy of the blob, one byte off
        meta[t]["data_offset"] = int(meta[t]["data_offset"]) + len(sc.texture_data) + 1
        s2.texture_meta = meta
        rc, msg = _layout_error(harness, s2)
        assert (rc == 0) == ok and (ok or "not dword aligned" in msg), (fmt, rc, msg)


def test_coordinates_beyond_the_supported_magnitude_are_rejected(harness):
    """The triangle tests take 1 / det from v_rcp_f32 + one Newton step, which is the correctly rounded quotient only for
    2^-126 <= |det| < 2^126 (kernels.h rcp_det; swept on the GPU by test_gpu_probes.py): the upload bounds what det is made
    of -- vertex coordinates <= 2^40, instance matrix entries <= 2^30, both finite."""
    assert _layout_error(harness, scenes.SCENES["cubes"]())[0] == 0
    for bad in (np.float32(2.0 ** 41), np.float32(np.inf), np.float32(np.nan), np.float32(-3e38)):
        sc = scenes.SCENES["cubes"]()
        sc.vertices = sc.vertices.copy()
        sc.vertices.reshape(-1, 4)[7, 1] = bad
        rc, msg = _layout_error(harness, sc)
        assert rc != 0 and "vertex coordinate" in msg, (bad, rc, msg)
        sc = scenes.SCENES["cubes"]()
        sc.mesh_instances = sc.mesh_instances.copy()
        sc.mesh_instances[1]["inv_transform"][5] = bad
        rc, msg = _layout_error(harness, sc)
        assert rc != 0 and "matrix entry" in msg, (bad, rc, msg)
    sc = scenes.SCENES["cubes"]()                       # the bound itself is accepted
    sc.vertices = sc.vertices.copy()
    sc.vertices.reshape(-1, 4)[7, 1] = np.float32(2.0 ** 40)
    assert _layout_error(harness, sc)[0] == 0
