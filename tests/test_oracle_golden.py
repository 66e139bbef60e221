"""CPU oracle (oracle/polaris_oracle.cpp) against the committed golden vectors.

The vectors were produced by the reference's own OpenCL C compiled for the host
(tests/tools/make_golden.py -> oracle/_ref/libpolaris_ref_pm.so); the bar is BIT equality of the
radiance accumulator, every ray counter and the primary hit tables.
"""
import numpy as np
import pytest

from conftest import bits, golden_files, load_golden


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_reproduces_golden(oracle, path):
    d, sc, req = load_golden(path)
    B = req.num_bounces
    acc, st, taps = oracle.trace(sc, req, d["seeds"], tap_sample=0)
    assert np.array_equal(bits(acc[..., :3]), bits(d["accum"])), "trace accumulator differs from the reference"
    assert list(st.rays_per_bounce[:B]) == list(d["rays_per_bounce"])
    assert list(st.occl_per_bounce[:B]) == list(d["occl_per_bounce"])
    got = [st.primary_rays, st.indirect_rays, st.occlusion_rays, st.shaded_hits, st.shaded_misses, st.unoccluded]
    assert got == list(d["counters"])
    assert np.array_equal(bits(taps["primary_rays"]), bits(d["primary_rays"]))
    assert np.array_equal(taps["primary_hit"], d["primary_hit"])
    assert np.array_equal(bits(taps["primary_wuvt"]), bits(d["primary_wuvt"]))
    assert np.array_equal(taps["primary_tri"], d["primary_tri"])
    assert np.array_equal(bits(taps["throughput0"][:, :3]), bits(d["throughput0"]))
    spp = req.samples_per_pixel
    assert np.array_equal(oracle.tonemap(acc, 1.0 / spp, 1.2), d["framebuffer"])


def test_golden_fixtures_present():
    assert len(golden_files()) >= 6


def test_prng_known_answer(oracle):
    """SURVEY.md 8c known answer for randomGetSample2f (samplers/random_sampler.cl:7-16)."""
    st, out = oracle.random([12345, 7])
    assert list(st) == [2471975582, 38620935]
    assert out[0] == np.float32(0.745002508) and out[1] == np.float32(0.76711452)
    st, out = oracle.random(st)
    assert list(st) == [718108755, 3787548295]
    assert out[0] == np.float32(0.299206108) and out[1] == np.float32(0.546144664)
    st, out = oracle.random(st)
    assert list(st) == [2186590168, 3055192711]
    assert out[0] == np.float32(0.425114244) and out[1] == np.float32(0.263019592)


def test_prng_matches_numpy_restatement(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        sx, sy = (int(v) for v in rng.integers(0, 2 ** 32, size=2))
        x = (sx * 17 + sy * 13123) & 0xFFFFFFFF
        nsx = ((x << 13) ^ x) & 0xFFFFFFFF
        nsy = (sy ^ (x << 7)) & 0xFFFFFFFF
        a = (x * ((x * x * 15731 + 74323) & 0xFFFFFFFF) + 871483) & 0xFFFFFFFF
        b = (x * ((x * x * 13734 + 37828) & 0xFFFFFFFF) + 234234) & 0xFFFFFFFF
        st, out = oracle.random([sx, sy])
        assert list(st) == [nsx, nsy]
        assert out[0] == np.float32(np.float32(a) * np.float32(1.0 / 4294967296.0))
        assert out[1] == np.float32(np.float32(b) * np.float32(1.0 / 4294967296.0))


def test_sample_parallel_mode_matches_reference_order(oracle):
    """The CPU-baseline mode of the restatement (threads take whole samples) traces the same paths:
    identical counters, radiance equal up to the association of per-pixel sums."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cornell"]()
    W, H, spp, B = 48, 40, 12, 5
    seeds = scenes.make_seeds(spp, B)
    a, sa, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    b, sb, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds, flags=ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES)
    assert list(sa.rays_per_bounce[:B]) == list(sb.rays_per_bounce[:B]) and list(sa.occl_per_bounce[:B]) == list(sb.occl_per_bounce[:B])
    assert sa.unoccluded == sb.unoccluded and sa.emitter_hits == sb.emitter_hits
    assert float(np.sqrt(np.mean((a[..., :3] / spp - b[..., :3] / spp) ** 2))) <= 1e-6
