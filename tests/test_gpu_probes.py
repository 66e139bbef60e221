"""Function-level parity on the GPU: the device-side samplers, BxDFs, light functions and traversal kernels
of the HIP library, one function at a time, against the CPU oracle on the same inputs -- through the C ABI
(polaris_hip_probe / polaris_hip_probe_intersect).  Bar: bit-exact (NaN where the oracle has NaN).

Reference functions covered: bxdfGetSample / bxdfGetPdf / bxdfEval (bxdf/bxdf.cl:31-105 and the five bxdf/*.cl behind
them), matSelectNode (samplers/material_sampler.cl:21-95, every operator), texGetSample3f / texGetSample1f /
texGetBumpSample3f (samplers/texture_sampler.cl:14-252, all four formats, wrap and clamp edges), emissiveGetSample / emissiveGetPdf (samplers/emissive_sampler.cl:176-223, area and environment lights),
rayIntersectionQuery / rayIntersectionTest (kernels/intersect.cl:26-347) on arbitrary rays through every traversal kernel.
"""
import numpy as np
import pytest

from conftest import bits, make_hip_tracer

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(np.all((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))))


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def test_bxdf_probes_equal_the_oracle(built, oracle):
    """Every BxDF leaf of the materials scene (textured and untextured parameters, all five BxDF families): sample, pdf
    and eval on random and on degenerate geometry (grazing / back-facing / normal incidence, samples at 0 and next to 1)."""
    from polaris_amd import ctypes_api as T, scenes

    sc = scenes.textured_materials_scene()
    rng = np.random.default_rng(21)
    leaves = [i for i, n in enumerate(sc.material_nodes) if 4 <= int(n["type"]) < T.OP_MIX]
    assert {int(sc.material_nodes[i]["type"]) for i in leaves} == {4, 8, 16, 32, 64}
    n = 600
    tr = make_hip_tracer(sc, 8, 8)
    checked = 0
    try:
        for stage_lds in (1, 0):  # LDS-staged tables (what the shade kernels run for small scenes) and global tables
            tr.set_option("stage_lds", stage_lds)
            for leaf in leaves:
                nrm, wi, wo = unit(rng.normal(size=(n, 3))), unit(rng.normal(size=(n, 3))), unit(rng.normal(size=(n, 3)))
                uv = rng.uniform(-2, 3, size=(n, 2)).astype(np.float32)
                xi = rng.random((n, 2)).astype(np.float32)
                # degenerate rows
                wi[0] = nrm[0]; wi[1] = -nrm[1]                                  # normal incidence from either side
                t = unit(np.cross(nrm[2:6], rng.normal(size=(4, 3))))
                wi[2:6] = t                                                      # grazing: in_dir perpendicular to the normal
                wo[6] = (2.0 * np.dot(wi[6], nrm[6]) * nrm[6] - wi[6]).astype(np.float32)  # the mirror direction (conductor pdf / eval)
                xi[7] = (0.0, 0.0); xi[8] = (np.float32(1.0) - np.float32(2.0 ** -24), 0.5); xi[9] = (0.5, 0.0)
                uv[10] = (0.0, 0.0); uv[11] = (1.0, 1.0); uv[12] = (-0.25, 1.75)
                rows = np.concatenate([nrm, uv, wi, xi, wo], axis=1)
                got = tr.probe(tr.PROBE_BXDF, leaf, rows)
                node = sc.material_nodes[leaf:leaf + 1]
                for r in range(n):
                    want = oracle.bxdf_probe(node, sc.texture_meta, sc.texture_data, nrm[r], uv[r], wi[r], xi[r], wo[r])
                    assert same_bits(got[r], want), (stage_lds, leaf, int(node["type"][0]), r, got[r], want)
                    checked += 1
    finally:
        tr.Close()
    assert checked >= 2 * 5 * n


@pytest.mark.parametrize("name", ["materials", "cornell"])
def test_material_walk_probes_equal_the_oracle(built, oracle, name):
    """matSelectNode (samplers/material_sampler.cl:21-95) from every material node of two scenes -- mixMap, bumpMap,
    normalMap, disperse (materials), mix (Cornell box) and plain leaves: leaf chosen, normal after the maps, tint,
    dispersion flags, IOR override, PRNG state left."""
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    rng = np.random.default_rng(23)
    n = 500
    ops = {int(t) for t in sc.material_nodes["type"]}
    assert ({10002, 10003, 10004, 10005} <= ops) if name == "materials" else (10001 in ops)
    tr = make_hip_tracer(sc, 8, 8)
    try:
        for stage_lds in (1, 0):
            tr.set_option("stage_lds", stage_lds)
            for root in range(len(sc.material_nodes)):
                nrm = unit(rng.normal(size=(n, 3)))
                uv = rng.uniform(-2, 3, size=(n, 2)).astype(np.float32)
                st = rng.integers(0, 2 ** 32, size=(n, 2), dtype=np.uint64).astype(np.uint32)
                flags = rng.choice([0, 0, 1, 2, 4], size=n).astype(np.uint32)
                rows = np.concatenate([nrm, uv, st.view(np.float32), flags.view(np.float32)[:, None]], axis=1)
                got = tr.probe(tr.PROBE_MATERIAL, root, rows)
                for r in range(n):
                    want = oracle.material_probe(sc, root, nrm[r], uv[r], st[r], int(flags[r]))
                    assert same_bits(got[r], want), (stage_lds, root, int(sc.material_nodes[root]["type"]), r, got[r], want)
    finally:
        tr.Close()


def _with_odd_l8(sc):
    """The materials scene plus one more L8 texture whose data starts at an ODD byte offset (legal for 1-byte texels: the
    fetch reads the aligned dword around each texel) and is the LAST thing in the blob (the fetch's spare dwords fall into
    the upload's padding)."""
    from polaris_amd import ctypes_api as T

    rng = np.random.default_rng(3)
    w, h = 7, 5
    px = (rng.random(w * h) * 255).astype(np.uint8)
    blob = np.concatenate([sc.texture_data, np.zeros((-len(sc.texture_data)) % 4 + 1, np.uint8)])
    off = len(blob)
    assert off % 4 == 1
    sc.texture_data = np.concatenate([blob, px])
    m = np.zeros(1, dtype=T.TEXTURE_META)
    m["format"], m["width"], m["height"], m["data_offset"] = T.TEX_L8, w, h, off
    sc.texture_meta = np.concatenate([sc.texture_meta, m])
    return sc


def test_texture_probes_equal_the_oracle(built, oracle):
    """All four texel formats, three fetch flavours, texture borders (clamped +1 neighbour), wrapped and negative uv."""
    from polaris_amd import scenes

    sc = _with_odd_l8(scenes.textured_materials_scene())
    rng = np.random.default_rng(5)
    edge = [(0.0, 0.0), (1.0, 1.0), (0.999999, 0.5), (-0.25, 1.75), (3.0, -2.0), (0.5, 0.0), (0.0, 0.999999), (0.9999999, 0.9999999),
            (-1e-7, -1e-7), (1e-8, 123.456), (-0.0, 0.5)]
    tr = make_hip_tracer(sc, 8, 8)
    try:
        assert {int(m["format"]) for m in sc.texture_meta} == {0, 1, 2, 3}
        for stage_lds in (1, 0):
            tr.set_option("stage_lds", stage_lds)
            for t in range(len(sc.texture_meta)):
                w, h = int(sc.texture_meta[t]["width"]), int(sc.texture_meta[t]["height"])
                grid = [((x + fx) / w, (y + fy) / h) for x in (0, w - 2, w - 1) for y in (0, h - 2, h - 1) for fx, fy in ((0.0, 0.0), (0.5, 0.5), (0.99, 0.01))]
                uvs = np.array(edge + grid + [tuple(rng.uniform(-2, 3, size=2)) for _ in range(1500)], dtype=np.float32)
                got = tr.probe(tr.PROBE_TEXTURE, t, uvs)
                for r, uv in enumerate(uvs):
                    want = oracle.tex_probe(sc.texture_meta, sc.texture_data, t, uv)
                    assert same_bits(got[r], want), (stage_lds, t, int(sc.texture_meta[t]["format"]), uv, got[r], want)
    finally:
        tr.Close()


@pytest.mark.parametrize("name", ["cornell", "materials", "sphere", "cubes", "transformed"])
def test_emissive_probes_equal_the_oracle(built, oracle, name):
    """Area lights (incl. the ones of rotated / non-uniformly scaled instances: quirk a-9(4)) and environment lights."""
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    rng = np.random.default_rng(9)
    n = 400
    tr = make_hip_tracer(sc, 8, 8)
    try:
        for stage_lds in (1, 0):
            tr.set_option("stage_lds", stage_lds)
            for e in range(len(sc.emissives)):
                p = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
                nrm, d = unit(rng.normal(size=(n, 3))), unit(rng.normal(size=(n, 3)))
                xi = rng.random((n, 2)).astype(np.float32)
                xi[0] = (0.0, 0.0); xi[1] = (np.float32(1.0) - np.float32(2.0 ** -24), np.float32(1.0) - np.float32(2.0 ** -24))
                got = tr.probe(tr.PROBE_EMISSIVE, e, np.concatenate([p, nrm, xi, d], axis=1))
                for r in range(n):
                    want = oracle.emissive_probe(sc, e, p[r], nrm[r], xi[r], d[r])
                    assert same_bits(got[r], want), (name, stage_lds, e, r, got[r], want)
    finally:
        tr.Close()


def _random_rays(sc, n, rng):
    """Rays that start inside, on and outside the scene's box, towards points of the box; a tenth with a short maxDist."""
    v = sc.vertices[:, :3]
    lo, hi = v.min(axis=0), v.max(axis=0)
    # instanced scenes: the vertices are in mesh space; use the union of a generous box around the origin too
    lo, hi = np.minimum(lo, -1.0), np.maximum(hi, 1.0)
    ext = hi - lo
    o = (lo - 0.5 * ext + rng.random((n, 3)) * 2.0 * ext).astype(np.float32)
    tgt = (lo + rng.random((n, 3)) * ext).astype(np.float32)
    d = unit(tgt - o)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 4:7] = o, d
    rays[:, 3] = np.float32(3.402823466e+38)
    short = rng.random(n) < 0.1
    rays[short, 3] = (rng.random(short.sum()) * np.linalg.norm(ext)).astype(np.float32)
    # axis-parallel directions (infinite reciprocals in the slab test)
    k = min(n, 64)
    rays[:k, 4:7] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1)).astype(np.float32)
    # one direction component at the edge of / below the normal range: denormal, 2^-127, 2^-126 and its neighbours, both signs (the
    # kernels' three-instruction reciprocal is only taken for magnitudes in [2^-126, 2^126): kernels.h rcp_dir)
    tiny = np.array([1e-39, 2.0 ** -127, 2.0 ** -126, np.nextafter(np.float32(2.0 ** -126), np.float32(0)), np.nextafter(np.float32(2.0 ** -126), np.float32(1)), 1e-45, 0.0],
                    np.float32)
    for j in range(k, min(n, k + 96)):
        rays[j, 4 + j % 3] = tiny[j % len(tiny)] * (-1.0 if (j // 7) % 2 else 1.0)
    return rays


@pytest.mark.parametrize("name", ["cornell", "cubes", "transformed", "material-ball-small", "instanced-small", "terrain-small", "many-materials"])
def test_arbitrary_rays_through_every_traversal_kernel(built, oracle, name):
    """rayIntersectionQuery / rayIntersectionTest on 100 000 arbitrary rays: persistent refill kernel in each node mode
    the scene admits, one-ray-per-lane kernel, wave-packet kernel -- hit flag,
    triangle and (w,u,v,t) bit-equal to the oracle."""
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    rng = np.random.default_rng(17)
    rays = _random_rays(sc, 100_000, rng)
    w_hit, w_wuvt, w_it = oracle.intersect(sc, rays, any_hit=False)
    w_occ, _, _ = oracle.intersect(sc, rays, any_hit=True)
    assert 0.2 < w_hit.mean() < 1.0 and 0.0 < w_occ.mean()
    variants = [dict(traversal=1, node_mode=m) for m in (0, 1, 2)] + [dict(traversal=0), dict(packet_primary=1, packet_shadow=32)]
    variants += [dict(node_mode=2, lds_tris=0), dict(node_mode=2, lds_tris=37)]  # tiny mode: no / a few triangle records in LDS (both fetch paths in one wave)
    variants += [dict(node_mode=2, tiny_one=0), dict(node_mode=2, tiny_one=0, lds_tris=37)]  # ... and its general variant where the single-instance one would run
    for opts in variants:
        tr = make_hip_tracer(sc, 8, 8, **opts)
        try:
            hit, wuvt, tri = tr.probe_intersect(rays, any_hit=False)
            occ, _, _ = tr.probe_intersect(rays, any_hit=True)
        finally:
            tr.Close()
        assert np.array_equal(hit, w_hit), (name, opts, int((hit != w_hit).sum()))
        h = w_hit != 0
        assert np.array_equal(tri[h], w_it[h, 1]), (name, opts)
        assert np.array_equal(bits(wuvt[h]), bits(w_wuvt[h])), (name, opts)
        assert np.array_equal(occ, w_occ), (name, opts, int((occ != w_occ).sum()))


def test_the_triangle_tests_reciprocal_is_the_correctly_rounded_one(built):
    """kernels.h rcp_det: 1 / det as v_rcp_f32 + one Newton step (3 instructions) instead of the 11 of the correctly
    rounded division the rest of the code is built with.  polaris_hip_selftest_rcp sweeps ALL 2^32 float bit patterns
    on the device: no pattern with 2^-126 <= |x| < 2^126 may differ from 1.0f / x.  (Outside it does differ -- zeros,
    infinities, denormal arguments and denormal results: 3 * 2^24 patterns -- which is why the upload bounds the scene's
    coordinates and the triangle tests ignore the quotient below INTERSECTION_EPSILON; the count is asserted so that the
    sweep is known to discriminate.)"""
    from polaris_amd import scenes

    tr = make_hip_tracer(scenes.SCENES["cubes"](), 16, 16)
    try:
        below_2_126 = float(np.nextafter(np.float32(2.0 ** 126), np.float32(0)))
        inside, outside, sample = tr.selftest_rcp(2.0 ** -126, below_2_126)
        assert inside == 0, hex(sample)
        assert 0 < outside <= 3 * 2 ** 24
        everywhere, _, _ = tr.selftest_rcp(0.0, float("inf"))
        assert everywhere == outside            # every mismatch lies outside the interval
    finally:
        tr.Close()


def test_probe_rays_outside_the_supported_magnitudes_are_refused(built):
    from polaris_amd import scenes
    from polaris_amd.tracer import TracerError

    tr = make_hip_tracer(scenes.SCENES["cubes"](), 16, 16)
    try:
        rays = np.zeros((4, 8), np.float32)
        rays[:, 3] = 1e30
        rays[:, 4] = 1.0
        tr.probe_intersect(rays)
        for col, bad in ((5, 2048.0), (5, np.nan), (1, 2.0 ** 41), (2, np.inf)):
            r = rays.copy()
            r[2, col] = bad
            with pytest.raises(TracerError, match="ray 2"):
                tr.probe_intersect(r)
    finally:
        tr.Close()
