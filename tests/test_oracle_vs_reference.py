"""CPU oracle against the compiled reference itself (only where /root/reference exists).

The reference's OpenCL C is compiled in place for the host (oracle/refbuild); the restatement
must agree with it BIT FOR BIT at whole-trace level on every synthetic scene and at function
level on randomised BxDF / texture / light probes.  A second build of the reference with glibc's
libm behind the built-ins gives the statistical cross-check.
"""
import numpy as np
import pytest

from conftest import bits


SCENES = ["cornell-diffuse", "cornell", "sphere", "cubes", "materials", "many-materials", "transformed"]


@pytest.mark.parametrize("name", SCENES)
def test_trace_bit_exact(oracle, ref_pm, name):
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    W, H, spp, B = 40, 28, 3, 5
    seeds = scenes.make_seeds(spp, B, base=77)
    for by, bh in ((0, H), (5, 9)):
        req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
        a, sa, ta = ref_pm.trace(sc, req, seeds, tap_sample=1)
        req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
        b, sb, tb = oracle.trace(sc, req, seeds, tap_sample=1)
        assert np.array_equal(bits(a[..., :3]), bits(b[..., :3]))
        assert list(sa.rays_per_bounce[:B]) == list(sb.rays_per_bounce[:B])
        assert list(sa.occl_per_bounce[:B]) == list(sb.occl_per_bounce[:B])
        assert sa.unoccluded == sb.unoccluded and sa.shaded_hits == sb.shaded_hits and sa.shaded_misses == sb.shaded_misses
        for k in ("primary_rays", "primary_hit", "primary_wuvt", "primary_tri"):
            assert np.array_equal(bits(ta[k]), bits(tb[k])), k
        assert np.array_equal(bits(ta["throughput0"][:, :3]), bits(tb["throughput0"][:, :3]))


@pytest.mark.parametrize("first,family", [(0, "plain"), (24, "plain"), (48, "plain"), (72, "plain"), (0, "big"), (24, "big"), (0, "single"), (24, "single"), (0, "wide"), (0, "refbvh")])
def test_random_scenes_bit_exact(oracle, ref_pm, first, family):
    """Seeded random scenes (tests/tools/random_scenes.py: every BxDF family, material trees of all five operators, the four texture
    formats, instances under non-uniform scales, area / environment lights, odd frames, partial blocks, 1-6 bounces, any RR threshold):
    the restatement equals the compiled reference bit for bit on each.  The same seeds run HIP against the oracle on the GPU
    (tests/test_gpu_fuzz_parity.py)."""
    import os
    import sys

    from oracle import pybind as ob
    from polaris_amd import scenes

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from random_scenes import random_case

    compared = 0
    for seed in range(first, first + 24):
        # (big: plus a height field and / or a swarm of instances -- trees beyond the tiny-scene kernels; single: all of it as ONE mesh)
        sc, c = random_case(seed, big="big" in family, single="single" in family, wide="wide" in family, refbvh="refbvh" in family)   # (refbvh: trees from the C++ scene compiler; wide: rows of up to 1025 pixels, 0-32 bounces)
        B = c["bounces"]
        seeds = scenes.make_seeds(c["spp"], B, base=1000 + seed)

        def request():
            return ob.make_request(c["W"], c["H"], spp=c["spp"], bounces=B, rr=c["rr"], block_y=c["block_y"], block_h=c["block_h"])

        a, sa, _ = ref_pm.trace(sc, request(), seeds)
        b, sb, _ = oracle.trace(sc, request(), seeds)
        assert np.array_equal(bits(a[..., :3]), bits(b[..., :3])), (seed, family, c)
        assert list(sa.rays_per_bounce[:B]) == list(sb.rays_per_bounce[:B]) and list(sa.occl_per_bounce[:B]) == list(sb.occl_per_bounce[:B]), (seed, c)
        assert (sa.unoccluded, sa.shaded_hits, sa.shaded_misses) == (sb.unoccluded, sb.shaded_hits, sb.shaded_misses), (seed, c)
        compared += 1
    assert compared == 24


def test_reference_emitter_index_quirk(oracle, ref_pm):
    """Without the fix the reference stores direct emitter hits at the block-local index
    (pt_integrator.cl:106); both checkers reproduce that too when asked."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cornell-diffuse"]()
    W, H, spp, B = 32, 32, 2, 3
    seeds = scenes.make_seeds(spp, B)
    req = ob.make_request(W, H, spp=spp, bounces=B, block_y=0, block_h=12)  # the light is visible in the top rows
    a, _, _ = ref_pm.trace(sc, req, seeds, flags=0)
    req = ob.make_request(W, H, spp=spp, bounces=B, block_y=0, block_h=12)
    b, _, _ = oracle.trace(sc, req, seeds, flags=0)
    assert np.array_equal(bits(a[..., :3]), bits(b[..., :3]))


def test_tonemap_bit_exact(oracle, ref_pm):
    rng = np.random.default_rng(3)
    acc = (rng.random((64, 64, 4)) * rng.choice([0.01, 1.0, 50.0], size=(64, 64, 1))).astype(np.float32)
    acc[0, 0] = 0.0
    assert np.array_equal(oracle.tonemap(acc, 1.0 / 16, 1.2), ref_pm.tonemap(acc, 1.0 / 16, 1.2))


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


def test_bxdf_probes_bit_exact(oracle, ref_pm):
    from polaris_amd import scenes

    sc = scenes.textured_materials_scene()
    rng = np.random.default_rng(11)
    leaves = [i for i, n in enumerate(sc.material_nodes) if int(n["type"]) < 10001]
    checked = 0
    for i in leaves:
        node = sc.material_nodes[i:i + 1]
        for _ in range(40):
            n = _unit(rng.normal(size=3))
            wi = _unit(rng.normal(size=3))
            wo = _unit(rng.normal(size=3))
            uv = rng.uniform(-2, 3, size=2).astype(np.float32)
            xi = rng.random(2).astype(np.float32)
            a = ref_pm.bxdf_probe(node, sc.texture_meta, sc.texture_data, n, uv, wi, xi, wo)
            b = oracle.bxdf_probe(node, sc.texture_meta, sc.texture_data, n, uv, wi, xi, wo)
            assert np.array_equal(bits(a), bits(b)), (i, int(node["type"][0]), a, b)
            checked += 1
    assert checked >= 200


def test_texture_probes_bit_exact(oracle, ref_pm):
    from polaris_amd import scenes

    sc = scenes.textured_materials_scene()
    rng = np.random.default_rng(5)
    edge = [(0.0, 0.0), (1.0, 1.0), (0.999999, 0.5), (-0.25, 1.75), (3.0, -2.0), (0.5, 0.0)]
    for t in range(len(sc.texture_meta)):
        uvs = edge + [tuple(rng.uniform(-2, 3, size=2)) for _ in range(50)]
        for uv in uvs:
            a = ref_pm.tex_probe(sc.texture_meta, sc.texture_data, t, uv)
            b = oracle.tex_probe(sc.texture_meta, sc.texture_data, t, uv)
            assert np.array_equal(bits(a), bits(b)), (t, uv)


def test_material_walk_probes_bit_exact(oracle, ref_pm):
    """matSelectNode (material_sampler.cl:21-95) from every material node of the scene that touches every operator: which
    leaf is chosen, the bumped / normal-mapped normal, tint, dispersion flags, IOR override and the PRNG state it leaves."""
    from polaris_amd import scenes

    rng = np.random.default_rng(13)
    checked = 0
    for sc in (scenes.textured_materials_scene(), scenes.cornell_box("layered")):  # every operator incl. plain mix (Cornell)
      for root in range(len(sc.material_nodes)):
        for _ in range(40):
            n = _unit(rng.normal(size=3))
            uv = rng.uniform(-2, 3, size=2).astype(np.float32)
            st = rng.integers(0, 2 ** 32, size=2, dtype=np.uint64).astype(np.uint32)
            flags = int(rng.choice([0, 0, 1, 2, 4]))
            a = ref_pm.material_probe(sc, root, n, uv, st, flags)
            b = oracle.material_probe(sc, root, n, uv, st, flags)
            assert np.array_equal(bits(a), bits(b)), (root, int(sc.material_nodes[root]["type"]), a, b)
            checked += 1
    assert checked >= 400


def test_emissive_probes_bit_exact(oracle, ref_pm):
    from polaris_amd import scenes

    rng = np.random.default_rng(9)
    for name in ("cornell", "materials", "sphere", "cubes"):
        sc = scenes.SCENES[name]()
        for e in range(len(sc.emissives)):
            for _ in range(30):
                p = rng.uniform(-1, 1, size=3).astype(np.float32)
                n = _unit(rng.normal(size=3))
                xi = rng.random(2).astype(np.float32)
                d = _unit(rng.normal(size=3))
                a = ref_pm.emissive_probe(sc, e, p, n, xi, d)
                b = oracle.emissive_probe(sc, e, p, n, xi, d)
                assert np.array_equal(bits(a), bits(b)), (name, e)


@pytest.mark.parametrize("name", SCENES)
def test_libm_build_agrees_pixel_by_pixel(oracle, built, name):
    """The same reference kernels with glibc's libm behind sin / cos / atan / atan2 / acos / pow / sqrt instead of
    polaris_math.h -- the one place where a second, independent implementation of the built-ins meets the whole path.
    Last-bit differences of the built-ins move a path by ~1e-7 (images agree to 1e-5 of the mean, pixel by pixel) unless
    they flip a discrete decision (Fresnel choice, Russian roulette, mix-node selection, a hit at an edge): the layered
    scenes may flip a fraction of their paths, which moves single pixels by ~1/spp but not the image."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    if not ob.available("ref_libm"):
        pytest.skip("compiled reference kernels not built (needs /root/reference)")
    lib = ob.Oracle("ref_libm")
    sc = scenes.SCENES[name]()
    W = H = 32
    spp, B = 64, 5
    seeds = scenes.make_seeds(spp, B)
    a, sa, _ = lib.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    b, sb, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    a, b = a[..., :3] / spp, b[..., :3] / spp
    mean = float(b.mean())
    d = np.abs(a - b).max(axis=2)
    assert abs(float(a.mean()) - mean) / mean < 0.005
    assert abs(sa.total_rays() - sb.total_rays()) / sb.total_rays() < 0.002
    flipped = float((d > 1e-5 * mean).mean())               # pixels in which at least one path took another branch
    assert flipped <= (0.40 if name in ("cornell", "many-materials") else 0.03), (name, flipped)  # scenes full of mix nodes and dielectrics: many discrete decisions per path
    assert float(np.sqrt((d ** 2).mean())) / mean < 0.05    # per-pixel RMS deviation, relative to the image mean
    assert float(d.max()) / mean < 1.0                       # no pixel moves by more than the image mean (a few paths of 64)
