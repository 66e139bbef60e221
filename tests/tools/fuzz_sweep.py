"""A long run of the parity fuzz (GPU): seeds [first, last) of tests/tools/random_scenes.py, HIP (exact mode + a batched run, random
tracer options) against the CPU oracle -- the loop of tests/test_gpu_fuzz_parity.py without pytest, reporting every seed that differs.
    python tests/tools/fuzz_sweep.py 96 3000
    python tests/tools/fuzz_sweep.py 0 4096 ref      (CPU, build container: the oracle against the compiled reference instead)
    ... big / single                                  (random_case(seed, big=True / single=True))
    ... devbvh                                        (GPU: the tree rebuilt by polaris_hip_build_bvh first)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

from conftest import bits, make_hip_tracer  # noqa: E402
from oracle import pybind as ob  # noqa: E402
from polaris_amd import scenes  # noqa: E402
from random_scenes import random_case  # noqa: E402
from test_gpu_fuzz_parity import counters, draw_options  # noqa: E402


BIG = "big" in sys.argv[3:]     # the generator's second family: plus a height field and / or a swarm of instances
SINGLE = "single" in sys.argv[3:]   # ... and its third: everything baked into one mesh under the identity
WIDE = "wide" in sys.argv[3:]   # ... the request's edges: rows of 255-1025 pixels, 1-40 rows, up to 9 samples, 0-32 bounces
REFBVH = "refbvh" in sys.argv[3:]   # ... the trees built by the C++ scene compiler (the reference's builder restated)
DEVBVH = "devbvh" in sys.argv[3:]   # the scene's tree rebuilt by polaris_hip_build_bvh first (SAH or linear, leaves of 1-4 triangles), validated, then traced


def against_the_reference(first, last):
    oracle, ref = ob.Oracle("oracle"), ob.Oracle("ref_pm")
    bad, rays, t0 = [], 0, time.time()
    for seed in range(first, last):
        sc, c = random_case(seed, big=BIG, single=SINGLE, wide=WIDE, refbvh=REFBVH)
        B = c["bounces"]
        seeds = scenes.make_seeds(c["spp"], B, base=1000 + seed)

        def request():
            return ob.make_request(c["W"], c["H"], spp=c["spp"], bounces=B, rr=c["rr"], block_y=c["block_y"], block_h=c["block_h"])

        a, sa, _ = ref.trace(sc, request(), seeds)
        b, sb, _ = oracle.trace(sc, request(), seeds)
        rays += sb.total_rays()
        same = (np.array_equal(bits(a[..., :3]), bits(b[..., :3])) and list(sa.rays_per_bounce[:B]) == list(sb.rays_per_bounce[:B])
                and list(sa.occl_per_bounce[:B]) == list(sb.occl_per_bounce[:B]) and (sa.unoccluded, sa.shaded_hits, sa.shaded_misses) == (sb.unoccluded, sb.shaded_hits, sb.shaded_misses))
        if not same:
            bad.append(seed)
            print(f"seed {seed} MISMATCH oracle vs compiled reference; case {c}", flush=True)
    print(f"fuzz sweep{' (big)' if BIG else ''}{' (single)' if SINGLE else ''}{' (wide)' if WIDE else ''}{' (refbvh)' if REFBVH else ''}, oracle vs compiled reference: seeds [{first}, {last}): {last - first - len(bad)} equal, {len(bad)} differing {bad[:20]}, "
          f"{rays} rays traced, {time.time() - t0:.0f} s")
    return 1 if bad else 0


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    if "ref" in sys.argv[3:]:
        return against_the_reference(first, last)
    oracle = ob.Oracle("oracle")
    bad, rays, t0 = [], 0, time.time()
    symbols = {}
    for seed in range(first, last):
        sc, c = random_case(seed, big=BIG, single=SINGLE, wide=WIDE, refbvh=REFBVH)
        B, spp = c["bounces"], c["spp"]
        seeds = scenes.make_seeds(spp, B, base=1000 + seed)
        if DEVBVH:
            from polaris_amd import bvh_build
            from test_gpu_bvh_build import check_tree

            r2 = np.random.default_rng(0xB0B + seed)
            algorithm, max_leaf = ("sah", "lbvh")[int(r2.integers(0, 2))], int(r2.integers(1, 5))
            old = sc
            sc, _ = bvh_build.rebuild_on_device(old, max_leaf_tris=max_leaf, algorithm=algorithm)
            try:
                need = check_tree(sc, old)
                assert need < 32, f"stack need {need}"
            except AssertionError as e:
                bad.append(seed)
                print(f"seed {seed} INVALID TREE ({algorithm}, leaves of {max_leaf}): {e}", flush=True)
                continue

        def request():
            return ob.make_request(c["W"], c["H"], spp=spp, bounces=B, rr=c["rr"], block_y=c["block_y"], block_h=c["block_h"])

        want, ws, _ = oracle.trace(sc, request(), seeds)
        rays += ws.total_rays()
        if np.isnan(want[..., :3]).any():
            print(f"seed {seed}: NaN in the oracle's frame (not compared)", flush=True)
            continue
        rng = np.random.default_rng(0xF00D + seed)
        opts = draw_options(rng)
        batched = dict(opts, samples_per_batch=int(rng.integers(1, spp + 1)), overlap=int(rng.integers(1, 4)))
        for options, exact in ((dict(opts, exact_accumulate=1), True), (batched, False)):
            tr = make_hip_tracer(sc, c["W"], c["H"], **options)
            try:
                tr.set_option("time_kernels", 1)    # (so that the library remembers which kernel symbol every step ran)
                tr.Trace(request(), seeds)
                got, gs = tr.read_accumulator(0), tr.last_trace_stats
                for timer in ("intersect_packet", "intersect", "occlusion", *tr.SHADE_TIMERS):
                    sym = tr.kernel_symbol(timer)
                    if sym:
                        symbols[sym] = symbols.get(sym, 0) + 1
            finally:
                tr.Close()
            by, bh = c["block_y"], c["block_h"]
            why = ""
            if counters(gs, B) != counters(ws, B):
                why = f"counters {counters(gs, B)} != {counters(ws, B)}"
            elif exact and not np.array_equal(bits(got[by:by + bh, :, :3]), bits(want[by:by + bh, :, :3])):
                why = f"{int((bits(got[by:by + bh, :, :3]) != bits(want[by:by + bh, :, :3])).sum())} accumulator words differ"
            elif not exact:
                scale = max(1.0, float(np.abs(want[by:by + bh, :, :3]).max()) / spp)
                err = float(np.sqrt(np.mean((got[by:by + bh, :, :3] / spp - want[by:by + bh, :, :3] / spp) ** 2)))
                if err > 1e-6 * scale:
                    why = f"rmse {err:.3e} > 1e-6 x {scale:.3g}"
            if why:
                bad.append(seed)
                print(f"seed {seed} MISMATCH ({'exact' if exact else 'batched'}): {why}; case {c}; options {options}", flush=True)
        if (seed - first) % 250 == 249:
            print(f"... {seed + 1 - first} seeds, {len(bad)} differing, {rays} rays, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz sweep{' (big)' if BIG else ''}{' (single)' if SINGLE else ''}{' (wide)' if WIDE else ''}{' (refbvh)' if REFBVH else ''}{' (tree built on the device)' if DEVBVH else ''}: seeds [{first}, {last}): {last - first - len(set(bad))} equal, {len(set(bad))} differing {sorted(set(bad))[:20]}, {rays} rays traced, {time.time() - t0:.0f} s")
    print("kernel symbols the traces ran (traces that used each):")
    for sym, n in sorted(symbols.items(), key=lambda kv: -kv[1]):
        print(f"  {n:6d}  {sym}")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
