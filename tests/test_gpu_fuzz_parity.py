"""Parity fuzz on a real MI355X: seeded random scenes (tests/tools/random_scenes.py -- every BxDF family, material trees of all five
operators, the four texture formats in odd sizes and at unaligned offsets, instances under non-uniform scales, area and environment
lights, odd frame sizes, partial row blocks, 1-6 bounces, any Russian-roulette threshold) traced by the HIP path through the C ABI under
randomly drawn tracer options, against the CPU oracle.

Bars: exact mode -- trace accumulator and every ray counter BIT-IDENTICAL to the oracle; default (batched) mode -- identical ray
counters (=> identical paths) and per-pixel RMSE <= 1e-6 of the frame's brightest value (the batches' sums are associated differently).
The same seeds are checked oracle == compiled reference on the CPU (tests/test_oracle_vs_reference.py).
"""
import os
import sys

import numpy as np
import pytest

from conftest import bits, make_hip_tracer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu

PER_TEST = 12


def counters(st, B):
    return (list(st.rays_per_bounce[:B]), list(st.occl_per_bounce[:B]), st.primary_rays, st.indirect_rays, st.occlusion_rays,
            st.shaded_hits, st.shaded_misses, st.unoccluded, st.emitter_hits)


def draw_options(rng):
    """A random point of the tracer's option space (every value is one some test in test_gpu_parity.py uses on the fixed scenes)."""
    o = {}
    for key, values in (("node_mode", (0, 1, 2)), ("tiny_one", (0, 1)), ("lds_tris", (0, 7, 29)), ("packet_primary", (0, 1)), ("shade_sort", (1, 2, 32)),
                        ("shade_wave", (0, 1)), ("shade_wave_from", (0, 1, 3)), ("stage_lds", (0, 1)), ("max_leaf_tris", (1, 2, 4)), ("traversal", (0, 1))):
        if rng.random() < 0.3:
            o[key] = int(rng.choice(values))
    return o


FAMILIES = ([(f, "plain") for f in range(0, 8 * PER_TEST, PER_TEST)] + [(f, "big") for f in range(0, 4 * PER_TEST, PER_TEST)]
            + [(f, "single") for f in range(0, 4 * PER_TEST, PER_TEST)] + [(0, "big-single"), (0, "wide"), (PER_TEST, "wide-single"), (0, "refbvh"), (PER_TEST, "refbvh-big")])


@pytest.mark.parametrize("first,family", FAMILIES)
def test_random_scenes_against_the_oracle(built, oracle, first, family):
    """`big`: the same generator plus a height field of up to 4 600 triangles and / or a swarm of up to 150 instances -- trees that do not fit
    LDS and deep top-level trees, i.e. the general traversal kernels (`k_trace<*, 24, 0 / 1, false>`) instead of the tiny-scene ones.
    `single`: everything baked into ONE mesh under the identity transform -- the single-instance kernels (`k_trace<*, *, *, true>`, the
    headline's) and, for camera rays, the wave-packet kernel.
    `wide`: the request's edges -- rows of 255-1025 pixels, 1-40 rows, up to 9 samples, 0 to 32 bounces.
    `refbvh`: the arrays come out of the C++ scene compiler (the reference's builder restated: leaves of up to 10 triangles, which the upload
    subdivides) -- the trees `polaris render` would hand the tracer."""
    big, single, wide, refbvh = "big" in family, "single" in family, "wide" in family, "refbvh" in family
    from oracle import pybind as ob
    from polaris_amd import scenes
    from random_scenes import random_case

    for seed in range(first, first + PER_TEST):
        sc, c = random_case(seed, big=big, single=single, wide=wide, refbvh=refbvh)
        B, spp = c["bounces"], c["spp"]
        seeds = scenes.make_seeds(spp, B, base=1000 + seed)

        def request():
            return ob.make_request(c["W"], c["H"], spp=spp, bounces=B, rr=c["rr"], block_y=c["block_y"], block_h=c["block_h"])

        want, ws, _ = oracle.trace(sc, request(), seeds)
        assert not np.isnan(want[..., :3]).any(), seed      # (the generator's promise: NaN signs differ between hosts and are not fuzzed)
        rng = np.random.default_rng(0xF00D + seed)
        opts = draw_options(rng)
        batched = dict(opts, samples_per_batch=int(rng.integers(1, spp + 1)), overlap=int(rng.integers(1, 4)))
        for options, exact in ((dict(opts, exact_accumulate=1), True), (batched, False)):
            tr = make_hip_tracer(sc, c["W"], c["H"], **options)
            try:
                tr.Trace(request(), seeds)
                got, gs = tr.read_accumulator(0), tr.last_trace_stats
            finally:
                tr.Close()
            what = (seed, family, c, options)
            assert counters(gs, B) == counters(ws, B), what
            by, bh = c["block_y"], c["block_h"]
            if exact:
                assert np.array_equal(bits(got[by:by + bh, :, :3]), bits(want[by:by + bh, :, :3])), what
            else:
                scale = max(1.0, float(np.abs(want[by:by + bh, :, :3]).max()) / spp)
                err = float(np.sqrt(np.mean((got[by:by + bh, :, :3] / spp - want[by:by + bh, :, :3] / spp) ** 2)))
                assert err <= 1e-6 * scale, (what, err)
