"""Seeded random scenes for the parity fuzz tests (TEST INFRASTRUCTURE).

`random_case(seed)` draws a small scene out of everything the data contract admits -- 1-4 meshes of boxes, spheres and loose
quads, 1-7 instances under translations / rotations / non-uniform scales, every BxDF family with random parameters, material
trees of mix / mixMap / bumpMap / normalMap / disperse nodes up to three levels deep, textures of the four formats in odd
sizes (1 x 1 included), area lights, an environment light with or without a map, a background colour or none -- plus a frame
(odd widths, sometimes a partial row block), sample count, bounce count and Russian-roulette threshold.  The CPU oracle, the
compiled reference and the HIP path must agree bit for bit on every one of them (tests/test_gpu_fuzz_parity.py,
tests/test_oracle_vs_reference.py).

Inputs that make the REFERENCE produce NaNs (singular transforms, zero-area lights) are not drawn: the sign and payload of a
NaN differ between an x86 host and the GPU, and that is not a property of the path.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from polaris_amd import ctypes_api as T  # noqa: E402
from polaris_amd import scenes as S  # noqa: E402

F32 = np.float32


def _texture(mt, rng, fmt=None):
    fmt = fmt if fmt is not None else int(rng.choice([T.TEX_L8, T.TEX_L32F, T.TEX_RGBA8, T.TEX_RGBA32F]))
    h, w = (int(v) for v in rng.choice([1, 2, 3, 5, 8, 13, 16, 31], size=2))
    if fmt == T.TEX_L8:
        px = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif fmt == T.TEX_L32F:
        px = rng.random((h, w)).astype(F32)
    elif fmt == T.TEX_RGBA8:
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    else:
        px = (rng.random((h, w, 4)) * rng.choice([1.0, 1.0, 3.0])).astype(F32)
    return mt.texture(fmt, px)


def _tint(rng, lo=0.05, hi=1.0):
    return tuple(float(v) for v in rng.uniform(lo, hi, 3))


def _leaf(mt, rng, textures, allow_emissive=False):
    def tex(p=0.3):
        return int(rng.choice(textures)) if textures and rng.random() < p else -1

    kind = int(rng.integers(0, 6 if allow_emissive else 5))
    if kind == 0:
        return mt.diffuse(_tint(rng), tex=tex())
    if kind == 1:
        return mt.conductor(_tint(rng, 0.3), int_ior=float(rng.choice([0.0, 0.2, 1.5, 2.4])), ext_ior=1.0, tex=tex())
    if kind == 2:
        return mt.rough_conductor(_tint(rng, 0.3), roughness=float(rng.choice([0.01, 0.1, 0.3, 0.7, 1.0])), int_ior=float(rng.choice([0.0, 0.6, 1.8])),
                                  tex=tex(), roughness_tex=tex(0.2))
    if kind == 3:
        return mt.dielectric(_tint(rng, 0.5), _tint(rng, 0.5), int_ior=float(rng.uniform(1.05, 2.4)), ext_ior=float(rng.choice([1.0, 1.0, 1.33])),
                             tex=tex(0.15), trans_tex=tex(0.15))
    if kind == 4:
        return mt.rough_dielectric(_tint(rng, 0.5), _tint(rng, 0.5), roughness=float(rng.choice([0.05, 0.2, 0.5, 0.9])), int_ior=float(rng.uniform(1.1, 2.0)),
                                   tex=tex(0.15), trans_tex=tex(0.15), roughness_tex=tex(0.2))
    return mt.emissive(_tint(rng, 0.5, 4.0), float(rng.uniform(0.5, 3.0)), tex=tex(0.2))


def _tree(mt, rng, textures, depth):
    """A material tree (asset/material/material_expr.y grammar: operators over BxDF leaves)."""
    if depth == 0 or rng.random() < 0.35:
        return _leaf(mt, rng, textures)
    op = int(rng.integers(0, 5))
    if op == 0:
        return mt.mix(_tree(mt, rng, textures, depth - 1), _tree(mt, rng, textures, depth - 1), float(rng.choice([0.0, 0.25, 0.5, 0.8, 1.0])))
    if op == 1 and textures:
        return mt.mix_map(_tree(mt, rng, textures, depth - 1), _tree(mt, rng, textures, depth - 1), int(rng.choice(textures)))
    if op == 2 and textures:
        return mt.bump_map(_tree(mt, rng, textures, depth - 1), int(rng.choice(textures)))
    if op == 3 and textures:
        return mt.normal_map(_tree(mt, rng, textures, depth - 1), int(rng.choice(textures)))
    if op == 4:
        d = mt.dielectric(_tint(rng, 0.6), _tint(rng, 0.6), int_ior=1.5) if rng.random() < 0.5 else mt.rough_dielectric(_tint(rng, 0.6), _tint(rng, 0.6), roughness=0.2)
        base = float(rng.uniform(1.2, 1.9))
        return mt.disperse(d, (base - 0.05, base, base + 0.05), (1.0, 1.0, 1.0))
    return _leaf(mt, rng, textures)


def _rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return m


def _object(rng, mat):
    kind = int(rng.integers(0, 4))
    c = rng.uniform(-1.0, 1.0, 3)
    if kind == 0:
        half = rng.uniform(0.15, 0.7, 3)
        return S.box(c - half, c + half, mat, rot_y=float(rng.uniform(-1.5, 1.5)) if rng.random() < 0.5 else 0.0)
    if kind == 1:
        return S.uv_sphere(c, float(rng.uniform(0.2, 0.7)), mat, n_lat=int(rng.integers(3, 9)), n_lon=int(rng.integers(3, 11)), smooth=bool(rng.random() < 0.7))
    if kind == 2:  # a loose quad, any orientation, with uv coordinates beyond [0, 1] (wrap)
        a, b = rng.normal(size=3), rng.normal(size=3)
        a /= np.linalg.norm(a)
        b -= a * (a @ b)
        b /= np.linalg.norm(b)
        a, b = a * rng.uniform(0.3, 1.2), b * rng.uniform(0.3, 1.2)
        k = float(rng.choice([1.0, 2.5, -1.5]))
        return S.quad(c - a - b, c + a - b, c + a + b, c - a + b, mat, uv=((0, 0), (k, 0), (k, k), (0, k)))
    # a fan of thin triangles round a point: shared edges and a shared apex (ties between neighbouring triangles)
    n = int(rng.integers(3, 8))
    r = float(rng.uniform(0.3, 0.9))
    tilt = _rot(int(rng.integers(0, 3)), float(rng.uniform(0, math.pi)))[:3, :3]
    rim = [c + tilt @ np.array([r * math.cos(2 * math.pi * i / n), 0.0, r * math.sin(2 * math.pi * i / n)]) for i in range(n)]
    apex = c + tilt @ np.array([0.0, float(rng.uniform(0.0, 0.6)), 0.0])
    verts = np.array([[apex, rim[i], rim[(i + 1) % n]] for i in range(n)])
    uvs = rng.uniform(-1, 2, (n, 3, 2))
    return S.Mesh(verts, S._flat_normals(verts), uvs, np.full(n, mat))


def _baked(mesh, xf):
    """The mesh with the instance transform applied to its vertices (normals by the inverse transpose)."""
    m3 = np.asarray(xf)[:3, :3]
    verts = mesh.verts @ m3.T + np.asarray(xf)[:3, 3]
    nrm = mesh.normals @ np.linalg.inv(m3)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-30)
    return S.Mesh(verts, nrm, mesh.uvs, mesh.mat)


def random_case(seed: int, big: bool = False, single: bool = False, wide: bool = False, refbvh: bool = False):
    """-> (scene, dict(W, H, spp, bounces, rr, block_y, block_h)).  `big`: the same scene plus a height field of 300-4 600 triangles and / or
    a swarm of 20-150 instances (drawn from a second stream, so the plain cases keep their scenes): trees that do not fit LDS, deep
    top-level trees -- the general traversal kernels instead of the tiny-scene ones.  `single`: all of it baked into ONE mesh under the
    identity transform (the kernels specialised for single-instance scenes).  `wide`: the request's edges instead of the small frame --
    rows of 255-1025 pixels, 1-40 rows, up to 9 samples, 0 to 32 bounces.  `refbvh`: the arrays come out of the C++ scene compiler
    (polaris_amd/host/scene_compiler.cpp, the reference's builder restated: what `polaris render` would upload) instead of scenes.py's."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    rng2 = np.random.default_rng(0xB160000 + seed)
    mt = S.MaterialTable()
    textures = [_texture(mt, rng) for _ in range(int(rng.integers(0, 5)))]
    mats = [_tree(mt, rng, textures, int(rng.integers(0, 4))) for _ in range(int(rng.integers(2, 9)))]
    floor = mt.diffuse(_tint(rng, 0.3, 0.8), tex=int(rng.choice(textures)) if textures and rng.random() < 0.4 else -1)

    n_mesh = int(rng.integers(1, 5))
    meshes = []
    for _ in range(n_mesh):
        parts = [_object(rng, int(rng.choice(mats))) for _ in range(int(rng.integers(1, 5)))]
        meshes.append(S.merge(parts))
    insts = []
    for mi in range(n_mesh):
        for _ in range(int(rng.integers(1, 3))):
            xf = np.eye(4)
            if rng.random() < 0.75:
                sc = rng.uniform(0.5, 1.8, 3) if rng.random() < 0.5 else np.full(3, rng.uniform(0.5, 1.5))
                xf = S.translation(rng.uniform(-2.0, 2.0, 3) * (1, 0.4, 1) + (0, 1.0, 0)) @ _rot(1, rng.uniform(-3, 3)) @ _rot(int(rng.integers(0, 3)), rng.uniform(-0.6, 0.6)) @ S.scaling(*sc)
            else:
                xf = S.translation((0, 1.0, 0))
            insts.append((mi, xf))
    # the room: a floor always, sometimes walls and a ceiling (a closed box makes every path run its full length)
    room = [S.quad((-4, 0, -4), (-4, 0, 4), (4, 0, 4), (4, 0, -4), floor, uv=((0, 0), (4, 0), (4, 4), (0, 4)))]
    closed = rng.random() < 0.4
    if closed:
        wall = mt.diffuse(_tint(rng, 0.3, 0.8))
        room += [S.quad((-4, 0, -4), (4, 0, -4), (4, 5, -4), (-4, 5, -4), wall), S.quad((-4, 0, 4), (-4, 0, -4), (-4, 5, -4), (-4, 5, 4), wall),
                 S.quad((4, 0, -4), (4, 0, 4), (4, 5, 4), (4, 5, -4), wall), S.quad((-4, 5, -4), (4, 5, -4), (4, 5, 4), (-4, 5, 4), wall)]
    # lights: area lights (possibly a mix with an emissive side: compiler.go:246-268 finds it), an environment light, or both
    n_area = int(rng.integers(0, 3))
    env = -1
    if n_area == 0 or rng.random() < 0.5:
        env = mt.emissive(_tint(rng, 0.3, 1.2), float(rng.uniform(0.4, 1.5)), tex=int(rng.choice(textures)) if textures and rng.random() < 0.5 else -1)
    for _ in range(n_area):
        e = mt.emissive(_tint(rng, 2.0, 9.0), float(rng.uniform(0.8, 2.5)), tex=int(rng.choice(textures)) if textures and rng.random() < 0.2 else -1)
        if rng.random() < 0.25:
            e = mt.mix(e, mt.diffuse(_tint(rng)), 0.5)
        c = rng.uniform(-2.0, 2.0, 3) * (1, 0, 1) + (0, rng.uniform(2.8, 4.6), 0)
        hx, hz = rng.uniform(0.3, 1.2, 2)
        room.append(S.quad(c + (-hx, 0, -hz), c + (hx, 0, -hz), c + (hx, 0, hz), c + (-hx, 0, hz), e))
    if big:
        kind = int(rng2.integers(0, 3))
        if kind in (0, 2):   # a height field, as its own mesh under a transform of its own
            n = int(rng2.integers(12, 49))
            meshes.append(S.displaced_grid(n, float(rng2.uniform(3.0, 7.0)), float(rng2.uniform(0.2, 0.9)), int(rng2.choice(mats)), seed=int(rng2.integers(0, 1 << 30))))
            insts.append((len(meshes) - 1, S.translation((0, float(rng2.uniform(0.05, 0.6)), 0)) @ _rot(1, rng2.uniform(-3, 3)) @ S.scaling(*rng2.uniform(0.7, 1.3, 3))))
        if kind in (1, 2):   # a swarm: many instances of one small mesh
            meshes.append(S.uv_sphere((0, 0, 0), 0.5, int(rng2.choice(mats)), n_lat=int(rng2.integers(4, 11)), n_lon=int(rng2.integers(4, 13))))
            mi = len(meshes) - 1
            for _ in range(int(rng2.integers(20, 151))):
                xf = S.translation(rng2.uniform(-3.5, 3.5, 3) * (1, 0.5, 1) + (0, 2.0, 0)) @ _rot(1, rng2.uniform(-3, 3)) @ _rot(0, rng2.uniform(-1, 1)) @ S.scaling(*rng2.uniform(0.15, 0.6, 3))
                insts.append((mi, xf))
    meshes.append(S.merge(room))
    insts.append((len(meshes) - 1, np.eye(4)))
    order = rng.permutation(len(insts))
    insts = [insts[i] for i in order]
    if single:   # the same geometry as ONE mesh under the identity: the single-instance kernels (the headline's `k_trace<*, 16, 2, true>`)
        meshes, insts = [S.merge([_baked(meshes[mi], xf) for mi, xf in insts])], [(0, np.eye(4))]
    bg = mt.diffuse(_tint(rng, 0.05, 0.5)) if rng.random() < 0.6 else -1
    name = f"random-{seed}{'-big' if big else ''}{'-single' if single else ''}{'-refbvh' if refbvh else ''}"
    max_leaf = int(rng.choice([1, 2, 4]))
    if refbvh:   # the C++ restatement of the reference's own compiler (asset/compiler: SAH sweep, leaves of up to `min_leaf` triangles) builds the trees
        from polaris_amd import host_api

        scn = host_api.compile_scene(meshes, insts, mt, scene_diffuse=bg, scene_emissive=env, min_leaf=int(np.random.default_rng(0x1EAF + seed).choice([1, 3, 10])), name=name)
    else:
        scn = S.compile_scene(meshes, insts, mt, max_leaf=max_leaf, scene_diffuse=bg, scene_emissive=env, name=name)
    W, H = int(rng.choice([17, 33, 64, 97, 130])), int(rng.choice([9, 24, 40, 71]))
    th, el, dist = rng.uniform(0, 2 * math.pi), rng.uniform(0.15, 0.9), rng.uniform(3.0, 3.9 if closed else 7.0)
    eye = (dist * math.cos(el) * math.cos(th), 0.3 + dist * math.sin(el), dist * math.cos(el) * math.sin(th))
    scn.set_camera(eye=eye, look=tuple(rng.uniform(-0.5, 0.5, 3) + (0, 1.0, 0)), fov=float(rng.uniform(0.5, 1.1)), aspect=W / H)
    B = int(rng.integers(1, 7))
    by, bh = 0, H
    if rng.random() < 0.3:
        by = int(rng.integers(0, H - 1))
        bh = int(rng.integers(1, H - by + 1))
    case = dict(W=W, H=H, spp=int(rng.integers(1, 5)), bounces=B, rr=int(rng.integers(0, B + 2)), block_y=by, block_h=bh)
    if wide:   # the edges of the request: rows wider than a workgroup / a chunk / several, single rows, many samples, 0 and up to 32 bounces
        r3 = np.random.default_rng(0x51DE0000 + seed)
        W, H = int(r3.choice([255, 256, 257, 300, 511, 513, 1025])), int(r3.choice([1, 2, 3, 7, 16, 40]))
        scn.set_camera(eye=eye, look=(0.0, 1.0, 0.0), fov=float(r3.uniform(0.5, 1.1)), aspect=W / H)
        B = int(r3.choice([0, 1, 2, 7, 12, 20, 32]))
        by = int(r3.integers(0, H))
        bh = int(r3.integers(1, H - by + 1)) if r3.random() < 0.5 else H - by
        case = dict(W=W, H=H, spp=int(r3.choice([1, 2, 5, 9])), bounces=B, rr=int(r3.integers(0, B + 2)), block_y=by if bh < H else 0, block_h=bh)
    return scn, case
