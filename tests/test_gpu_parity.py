"""Parity of the HIP path (through the C ABI) on a real MI355X.

Bars (stated per test):
* exact_accumulate=1 : the trace accumulator, every ray counter and the primary hit tables are
  BIT-IDENTICAL to the CPU oracle and to the committed golden vectors of the compiled reference.
* default (batched)  : identical ray counters (=> identical paths); radiance differs only by the
  association of the per-sample sums: per-pixel RMSE <= 1e-6 (north_star bar: 1e-4).
"""
import numpy as np
import pytest

from conftest import bits, golden_files, load_golden, make_hip_tracer

pytestmark = pytest.mark.gpu

SCENES = ["cornell-diffuse", "cornell", "sphere", "cubes", "materials", "many-materials", "transformed", "material-ball-small"]


def rmse(a, b, spp):
    return float(np.sqrt(np.mean((a[..., :3] / spp - b[..., :3] / spp) ** 2)))


def counters(st, B):
    return (list(st.rays_per_bounce[:B]), list(st.occl_per_bounce[:B]), st.primary_rays, st.indirect_rays, st.occlusion_rays,
            st.shaded_hits, st.shaded_misses, st.unoccluded)


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1][:-4])
def test_hip_reproduces_reference_golden(built, path):
    """HIP vs the compiled reference's golden vectors: bit-exact (exact mode)."""
    d, sc, req = load_golden(path)
    B, spp = req.num_bounces, req.samples_per_pixel
    tr = make_hip_tracer(sc, req.frame_w, req.frame_h, exact_accumulate=1)
    try:
        taps = tr.tap_primary(req, int(d["seeds"][0]))          # wave-packet kernel (default)
        tr.set_option("packet_primary", 0)
        taps_ray = tr.tap_primary(req, int(d["seeds"][0]))      # one-ray-per-lane kernel
        tr.set_option("packet_primary", 1)
        for k in taps:
            assert np.array_equal(bits(taps[k]), bits(taps_ray[k])), k
        tr.Trace(req, d["seeds"])
        acc, st = tr.read_accumulator(0), tr.last_trace_stats
        tr.MergeOutput(tr, req)
        full = load_golden(path)[2]
        full.block_y, full.block_h = 0, req.frame_h
        tr.SyncFramebuffer(full)
        fb = tr.read_framebuffer()
    finally:
        tr.Close()
    assert np.array_equal(bits(acc[..., :3]), bits(d["accum"]))
    assert list(st.rays_per_bounce[:B]) == list(d["rays_per_bounce"])
    assert list(st.occl_per_bounce[:B]) == list(d["occl_per_bounce"])
    assert [st.primary_rays, st.indirect_rays, st.occlusion_rays, st.shaded_hits, st.shaded_misses, st.unoccluded] == list(d["counters"])
    for k in ("primary_rays", "primary_hit", "primary_wuvt", "primary_tri"):
        assert np.array_equal(bits(taps[k]), bits(d[k])), k
    by, bh = req.block_y, req.block_h
    assert np.array_equal(fb[by:by + bh], d["framebuffer"].reshape(req.frame_h, req.frame_w, 4)[by:by + bh])


@pytest.mark.parametrize("name", SCENES)
def test_hip_vs_oracle_exact_and_batched(built, oracle, name):
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    W, H, spp, B = 96, 80, 6, 5   # 96 columns: workgroups straddle rows; 7680 rays = 30 workgroups
    seeds = scenes.make_seeds(spp, B, base=1234)
    want, ws, wt = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds, tap_sample=0)
    variants = (({"exact_accumulate": 1}, True), ({"exact_accumulate": 1, "traversal": 0, "packet_primary": 0}, True),
                ({"exact_accumulate": 1, "packet_primary": 0}, True), ({"exact_accumulate": 1, "packet_primary": 1}, True), ({}, False),
                ({"packet_primary": 1, "samples_per_batch": 3}, False),   # (the default sends camera rays through the packet kernel in single-instance scenes only)
                ({"samples_per_batch": 4}, False), ({"samples_per_batch": 1, "overlap": 3}, False),
                ({"samples_per_batch": 2, "overlap": 1, "traversal": 0}, False),
                # where k_trace reads its node records: global memory / top of the tree in LDS / (tiny scenes) whole tree in LDS
                ({"exact_accumulate": 1, "node_mode": 0}, True), ({"exact_accumulate": 1, "node_mode": 1}, True),
                ({"exact_accumulate": 1, "node_mode": 2}, True), ({"node_mode": 1, "samples_per_batch": 3}, False),
                # tiny scenes: the general kernel where the single-instance variant would run; no / few triangle records in LDS
                ({"exact_accumulate": 1, "tiny_one": 0}, True), ({"exact_accumulate": 1, "lds_tris": 0}, True), ({"tiny_one": 0, "lds_tris": 29, "samples_per_batch": 3}, False),
                ({"exact_accumulate": 1, "shade_wave": 0}, True), ({"exact_accumulate": 1, "shade_wave_from": 0}, True),
                ({"shade_wave": 0, "stage_lds": 0, "samples_per_batch": 3}, False), ({"shade_wave_from": 1, "samples_per_batch": 2}, False),
                # shading order: never sorted by class / sorted from bounce 2 / sorted with the tables in global memory
                ({"exact_accumulate": 1, "shade_sort": 32}, True), ({"exact_accumulate": 1, "shade_sort": 2}, True),
                ({"exact_accumulate": 1, "stage_lds": 0}, True), ({"shade_sort": 32, "shade_wave": 0, "samples_per_batch": 3}, False))
    for opts, exact in variants:
        tr = make_hip_tracer(sc, W, H, **opts)
        try:
            req = ob.make_request(W, H, spp=spp, bounces=B)
            tr.Trace(req, seeds)
            got, gs = tr.read_accumulator(0), tr.last_trace_stats
            assert req.accumulated_samples == spp    # tracer.go:240
        finally:
            tr.Close()
        assert counters(gs, B) == counters(ws, B), (name, opts)
        assert gs.emitter_hits == ws.emitter_hits
        if exact:
            assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), name
        else:
            assert rmse(got, want, spp) <= 1e-6, name


@pytest.mark.parametrize("max_leaf", [0, 1, 2, 5])
def test_leaf_subdivision_is_invisible(built, oracle, max_leaf):
    """The reference compiler's BVH (leaves of up to 10 triangles) traced with the leaves kept
    (0) or subdivided at upload: bit-identical radiance and counters vs the CPU oracle."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.cornell_box(compiler="reference")
    W, H, spp, B = 64, 48, 3, 5
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=3)
    seeds = scenes.make_seeds(spp, B, base=5)
    want, wst, _ = oracle.trace(sc, req, seeds)
    tr = make_hip_tracer(sc, W, H, exact_accumulate=1, max_leaf_tris=max_leaf)
    try:
        tr.Trace(req, seeds)
        got, st = tr.read_accumulator(0), tr.last_trace_stats
    finally:
        tr.Close()
    assert counters(st, B) == counters(wst, B)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))


def test_obj_scene_read_by_the_front_end(built, oracle, tmp_path):
    """A Wavefront scene (quads, mat_expr, textures, rotated/scaled instances, refractive prism)
    through reader -> compiler -> HIP, against the CPU oracle on the same compiled arrays: bit-exact."""
    import os
    import sys

    from oracle import pybind as ob
    from polaris_amd import host_api, scenes

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "tools"))
    import obj_fixtures

    W, H, spp, B = 64, 48, 4, 5
    sc = host_api.read_scene(obj_fixtures.write_cornell(str(tmp_path)), aspect=W / H)
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=3)
    seeds = scenes.make_seeds(spp, B, base=11)
    want, wst, _ = oracle.trace(sc, req, seeds)
    tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
    try:
        tr.Trace(req, seeds)
        got, st = tr.read_accumulator(0), tr.last_trace_stats
        tr.set_option("exact_accumulate", 0)
        tr.Trace(req, seeds)
        batched, bst = tr.read_accumulator(0), tr.last_trace_stats
    finally:
        tr.Close()
    assert wst.shaded_hits > 0 and wst.emitter_hits >= 0
    assert counters(st, B) == counters(wst, B) == counters(bst, B)
    assert np.array_equal(bits(got[..., :3]), bits(want[..., :3]))
    assert rmse(batched, want, spp) <= 1e-6


def test_row_blocks_merge_to_the_full_frame(built, oracle):
    """Two tracers on one GPU render row blocks [0,h0) and [h0,H) and merge into the primary
    (renderer/default.go:127-136,188-191): equals the oracle run block by block."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cornell"]()
    W, H, spp, B = 64, 48, 3, 5
    seeds = scenes.make_seeds(spp, B)
    blocks = [(0, 29), (29, 19)]
    trs = [make_hip_tracer(sc, W, H, exact_accumulate=1) for _ in blocks]
    try:
        expect = np.zeros((H, W, 3), np.float32)
        for tr, (by, bh) in zip(trs, blocks):
            req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
            tr.Trace(req, seeds)
            trs[0].MergeOutput(tr, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh))
            o, _, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
            expect[by:by + bh] = o[by:by + bh, :, :3]
        full = ob.make_request(W, H, spp=spp, bounces=B)
        trs[0].SyncFramebuffer(full)
        frame = trs[0].read_accumulator(1)
        fb = trs[0].read_framebuffer()
    finally:
        for tr in trs:
            tr.Close()
    assert np.array_equal(bits(frame[..., :3]), bits(expect))
    e4 = np.zeros((H, W, 4), np.float32)
    e4[..., :3] = expect
    assert np.array_equal(fb, oracle.tonemap(e4, 1.0 / spp, 1.2).reshape(H, W, 4))


def test_merge_from_a_tracer_on_another_gpu(built, oracle):
    """The in-process multi-GPU path the Go renderer uses (renderer/default.go:188-191,
    tracer/opencl/resources.go:108-124): the primary on device 0 merges the row block traced on
    device 1 -- polaris_hip_merge's peer branch (xGMI peer access, or the staged hipMemcpyPeerAsync).
    Needs two visible GPUs: skips itself on a one-GPU lease."""
    from oracle import pybind as ob
    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes

    if T.load_library().polaris_hip_device_count() < 2:
        pytest.skip("needs 2 HIP devices (the in-process peer merge cannot run on a 1-GPU lease)")
    sc = scenes.SCENES["cornell"]()
    W, H, spp, B = 64, 48, 3, 5
    seeds = scenes.make_seeds(spp, B)
    blocks = [(0, 29), (29, 19)]
    trs = [make_hip_tracer(sc, W, H, device=d, exact_accumulate=1) for d in (0, 1)]
    try:
        expect = np.zeros((H, W, 3), np.float32)
        for tr, (by, bh) in zip(trs, blocks):
            req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
            tr.Trace(req, seeds)
            trs[0].MergeOutput(tr, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh))
            o, _, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
            expect[by:by + bh] = o[by:by + bh, :, :3]
        trs[0].SyncFramebuffer(ob.make_request(W, H, spp=spp, bounces=B))
        frame = trs[0].read_accumulator(1)
    finally:
        for tr in trs:
            tr.Close()
    assert np.array_equal(bits(frame[..., :3]), bits(expect))


def test_asynchronous_camera_update_applies_at_the_next_trace(built, oracle):
    """UpdateState(Asynchronous, CameraData) queues the change; the NEXT Trace commits it before tracing
    (tracer/opencl/tracer.go:150-158,161-192 -- how the interactive renderer moves the camera, renderer/opengl.go:294-303).
    Nothing changes until then: a SyncFramebuffer in between still shows the old frame; the second Trace equals the oracle with
    the NEW camera bit for bit; a second queued update replaces the first (latest wins)."""
    import copy

    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.tracer import ChangeType, UpdateMode

    sc = scenes.SCENES["cornell"]()
    W, H, spp, B = 64, 48, 3, 4
    seeds = scenes.make_seeds(spp, B, base=77)
    moved, moved2 = copy.copy(sc), copy.copy(sc)
    moved.eye = np.asarray(sc.eye, np.float32) + np.float32([0.15, -0.1, 0.2])
    moved.frustum = np.asarray(sc.frustum, np.float32).copy()
    moved.frustum.reshape(4, 4)[:, 0] += np.float32(0.05)         # a different view direction per corner ray
    moved2.eye = np.asarray(sc.eye, np.float32) + np.float32([-0.2, 0.05, 0.1])
    moved2.frustum = moved.frustum
    tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
    try:
        req = ob.make_request(W, H, spp=spp, bounces=B)
        tr.Trace(req, seeds)
        first = tr.read_accumulator(0)
        took = tr.UpdateState(UpdateMode.Asynchronous, ChangeType.CameraData, moved2)
        assert took == 0.0
        tr.UpdateState(UpdateMode.Asynchronous, ChangeType.CameraData, moved)        # latest wins
        assert np.array_equal(bits(tr.read_accumulator(0)), bits(first))             # nothing happened yet
        tr.Trace(ob.make_request(W, H, spp=spp, bounces=B), seeds)
        second = tr.read_accumulator(0)
    finally:
        tr.Close()
    want_first, _, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    want_second, _, _ = oracle.trace(moved, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    assert np.array_equal(bits(first[..., :3]), bits(want_first[..., :3]))
    assert np.array_equal(bits(second[..., :3]), bits(want_second[..., :3]))
    assert not np.array_equal(bits(first[..., :3]), bits(second[..., :3]))


def test_merge_does_not_wait_for_a_trace_on_the_destination(built, oracle):
    """MergeOutput is queued on the destination's merge stream under a lock of its own (include/polaris_hip.h): a secondary's
    merge called while the primary is inside a long Trace returns at once, and -- ordered behind the primary's Reset stage
    through the reset epoch -- its rows are in the frame when SyncFramebuffer completes.  (renderer/default.go:188-191 merges
    from the secondaries' goroutines while the primary still traces.)"""
    import threading
    import time

    from oracle import pybind as ob
    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    import ctypes as C

    sc = scenes.SCENES["cornell"]()
    W, H, B = 256, 256, 5
    lib = T.load_library()
    prim, sec = make_hip_tracer(sc, W, H, exact_accumulate=1), make_hip_tracer(sc, W, H, exact_accumulate=1)
    try:
        # the secondary's block first (small), its Trace is over before the primary starts
        sreq = ob.make_request(W, H, spp=2, bounces=B, block_y=128, block_h=128)
        sseeds = scenes.make_seeds(2, B, base=5)
        sec.Trace(sreq, sseeds)
        epoch = C.c_uint64()
        assert lib.polaris_hip_reset_epoch(prim._h, C.byref(epoch)) == 0
        preq = ob.make_request(W, H, spp=96, bounces=B, block_y=0, block_h=128)  # long: exact mode traces one sample per batch
        pseeds = scenes.make_seeds(96, B, base=6)
        t_trace = {}

        def run_primary():
            t0 = time.perf_counter()
            prim.Trace(preq, pseeds)
            t_trace["s"] = time.perf_counter() - t0

        th = threading.Thread(target=run_primary)
        th.start()
        assert lib.polaris_hip_wait_reset(prim._h, epoch.value) == 0       # the primary's Reset is queued: merges land behind it
        t0 = time.perf_counter()
        prim.MergeOutput(sec, ob.make_request(W, H, spp=2, bounces=B, block_y=128, block_h=128))
        t_merge = time.perf_counter() - t0
        still_tracing = th.is_alive()
        th.join()
        prim.MergeOutput(prim, ob.make_request(W, H, spp=96, bounces=B, block_y=0, block_h=128))
        prim.SyncFramebuffer(ob.make_request(W, H, spp=96, bounces=B))
        frame = prim.read_accumulator(1)
    finally:
        prim.Close()
        sec.Close()
    assert still_tracing and t_merge < 0.5 * t_trace["s"], (t_merge, t_trace)   # the merge did not sit out the Trace
    a, _, _ = oracle.trace(sc, sreq, sseeds)
    assert np.array_equal(bits(frame[128:, :, :3]), bits(a[128:, :, :3]))     # the early merge survived the primary's Reset
    assert frame[:128, :, :3].sum() > 0


def test_merges_are_ordered_against_the_next_trace_without_a_sync(built, oracle):
    """MergeOutput is asynchronous (Exec1DNoWait, resources.go:119) on the destination's merge stream; the source's NEXT Trace
    clears and rewrites the rows the merge reads.  The library orders the two on the device (self-merge: the main stream joins
    the merge stream; another handle: the event its merge stream records behind the read), so a host may go
    Trace -> MergeOutput -> Trace -> MergeOutput ... with ONE SyncFramebuffer at the end -- the progressive loop -- on a frame
    big enough that the merge of one pass is still queued when the next Trace starts."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["sphere"]()
    W, H, B, spp, passes = 512, 384, 2, 1, 4
    lists = [scenes.make_seeds(spp, B, base=900 + i) for i in range(passes)]
    total = np.zeros((H, W, 3), np.float32)
    for sd in lists:
        o, _, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), sd)
        total = total + o[..., :3]
    for two_handles in (False, True):
        prim = make_hip_tracer(sc, W, H, exact_accumulate=1)
        src = make_hip_tracer(sc, W, H, exact_accumulate=1) if two_handles else prim
        try:
            prim.reset_frame()
            for i, sd in enumerate(lists):
                src.Trace(ob.make_request(W, H, spp=spp, bounces=B, accumulated=1), sd)   # (accumulated > 0: no Reset stage)
                prim.MergeOutput(src, ob.make_request(W, H, spp=spp, bounces=B, accumulated=1))
            prim.SyncFramebuffer(ob.make_request(W, H, spp=spp, bounces=B, accumulated=passes - spp))
            frame = prim.read_accumulator(1)
        finally:
            if two_handles:
                src.Close()
            prim.Close()
        assert np.array_equal(bits(frame[..., :3]), bits(total)), two_handles


def test_trace_accumulator_ring_and_merge_from_a_slot(built, oracle):
    """polaris_hip_ipc_export turns the trace accumulator into a ring: Trace i writes slot (i mod depth), older frames stay
    readable (what a peer process relies on while this tracer is already tracing the next frame), polaris_hip_merge_slot adds
    a given slot; opening one's own export is refused (HIP cannot), a resize drops the ring."""
    from oracle import pybind as ob
    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    from polaris_amd.tracer import TracerError

    sc = scenes.SCENES["cornell"]()
    W, H, B, spp = 64, 48, 4, 2
    tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
    try:
        blob = tr.ipc_export(3)
        x = T.IpcExport.from_buffer_copy(blob)
        assert (x.depth, x.frame_w, x.frame_h, x.abi_version) == (3, W, H, 5) and x.pci_bus_id == tr.device_identity()["pci_bus_id"].encode() and x.pid == __import__("os").getpid()
        with pytest.raises(TracerError, match="this process"):
            tr.ipc_open(blob)
        want, slots = [], []
        for i in range(4):
            sd = scenes.make_seeds(spp, B, base=40 + i)
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B), sd)
            slots.append(tr.trace_slot())
            want.append(oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B), sd)[0][..., :3])
            assert np.array_equal(bits(tr.read_accumulator(0)[..., :3]), bits(want[-1]))   # the "trace accumulator" is the newest slot
        assert slots == [1, 2, 0, 1]
        # frames 2 and 3 (slots 0 and 1) and frame 1 (slot 2) are all still there
        for frame_i, slot in ((1, 2), (2, 0), (3, 1)):
            tr.reset_frame()
            tr.merge_slot(tr, slot, ob.make_request(W, H, spp=spp, bounces=B, block_y=5, block_h=30))
            tr.SyncFramebuffer(ob.make_request(W, H, spp=spp, bounces=B))
            got = tr.read_accumulator(1)[..., :3]
            assert np.array_equal(bits(got[5:35]), bits(want[frame_i][5:35])) and not got[:5].any() and not got[35:].any()
        with pytest.raises(TracerError, match="slot"):
            tr.merge_slot(tr, 3, ob.make_request(W, H, spp=spp, bounces=B))
        from polaris_amd.tracer import ChangeType, UpdateMode
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))   # resize: back to the single buffer
        tr.Trace(ob.make_request(W, H, spp=spp, bounces=B), scenes.make_seeds(spp, B, base=40))
        assert tr.trace_slot() == 0
        assert np.array_equal(bits(tr.read_accumulator(0)[..., :3]), bits(want[0]))
    finally:
        tr.Close()


def test_progressive_accumulation(built, oracle):
    """accumulated_samples > 0 keeps the frame accumulator (tracer.go:208-213) and the tone-map
    weight is 1/(accumulated+spp) (resources.go:347)."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["sphere"]()
    W, H, B = 48, 32, 3
    tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
    try:
        total = np.zeros((H, W, 3), np.float32)
        acc = 0
        for frame_i in range(3):
            seeds = scenes.make_seeds(2, B, base=100 + frame_i)
            req = ob.make_request(W, H, spp=2, bounces=B, accumulated=acc)
            tr.Trace(req, seeds)
            tr.MergeOutput(tr, ob.make_request(W, H, spp=2, bounces=B, accumulated=acc))
            o, _, _ = oracle.trace(sc, ob.make_request(W, H, spp=2, bounces=B), seeds)
            total = total + o[..., :3]
            sync = ob.make_request(W, H, spp=2, bounces=B, accumulated=acc)
            tr.SyncFramebuffer(sync)
            acc += 2
        frame = tr.read_accumulator(1)
        fb = tr.read_framebuffer()
    finally:
        tr.Close()
    assert np.array_equal(bits(frame[..., :3]), bits(total))
    e4 = np.zeros((H, W, 4), np.float32)
    e4[..., :3] = total
    assert np.array_equal(fb, oracle.tonemap(e4, 1.0 / 6, 1.2).reshape(H, W, 4))


def test_edge_shapes(built, oracle):
    """Single-row block, width not a multiple of the wavefront, one bounce, zero samples."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cubes"]()
    for (W, H, by, bh, spp, B) in [(70, 9, 4, 1, 3, 4), (1, 5, 0, 5, 4, 2), (257, 3, 1, 2, 2, 1), (33, 7, 0, 7, 0, 3)]:
        seeds = scenes.make_seeds(max(spp, 1), B)
        tr = make_hip_tracer(sc, W, H, exact_accumulate=1)
        try:
            req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
            tr.Trace(req, seeds)
            got, gs = tr.read_accumulator(0), tr.last_trace_stats
        finally:
            tr.Close()
        want, ws, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), (W, H, by, bh, spp, B)
        assert counters(gs, B) == counters(ws, B)


def test_edge_shapes_in_batched_mode(built, oracle):
    """The shapes of test_edge_shapes through the default (batched) path -- k_fold_nee / k_resolve on chunks and pixel blocks that are not full,
    batches of one sample, several batches in flight, a single bounce: counters equal, per-pixel RMSE <= 1e-6."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cubes"]()
    for (W, H, by, bh, spp, B, opts) in [(70, 9, 4, 1, 3, 4, {}), (1, 5, 0, 5, 4, 2, {"samples_per_batch": 1}), (257, 3, 1, 2, 7, 1, {"samples_per_batch": 2, "overlap": 3}),
                                         (300, 2, 0, 2, 5, 3, {"samples_per_batch": 5}), (33, 7, 0, 7, 0, 3, {})]:
        seeds = scenes.make_seeds(max(spp, 1), B)
        tr = make_hip_tracer(sc, W, H, **opts)
        try:
            req = ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh)
            tr.Trace(req, seeds)
            got, gs = tr.read_accumulator(0), tr.last_trace_stats
        finally:
            tr.Close()
        want, ws, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, block_y=by, block_h=bh), seeds)
        assert counters(gs, B) == counters(ws, B), (W, H, by, bh, spp, B)
        assert rmse(got, want, max(spp, 1)) <= 1e-6, (W, H, by, bh, spp, B)


def test_error_behaviour(built):
    """ErrNoSceneData before an upload (tracer.go:203-205), bad requests and bad scenes are status
    codes with a message."""
    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.tracer import ChangeType, ErrNoSceneData, HipTracer, TracerError, UpdateMode

    tr = HipTracer("err", 0)
    tr.Init()
    try:
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (16, 16))
        with pytest.raises(ErrNoSceneData):
            tr.Trace(ob.make_request(16, 16, spp=1), scenes.make_seeds(1, 5))
        with pytest.raises(ErrNoSceneData):
            tr.SyncFramebuffer(ob.make_request(16, 16, spp=1))
        sc = scenes.SCENES["cubes"]()
        bad = scenes.SCENES["cubes"]()
        bad.bvh_nodes = bad.bvh_nodes.copy()
        inner = [i for i, n in enumerate(bad.bvh_nodes) if n["ldata"] > 0][0]
        bad.bvh_nodes[inner]["rdata"] = 10 ** 6
        with pytest.raises(TracerError, match="out of range"):
            tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, bad)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)
        with pytest.raises(TracerError, match="outside the frame"):
            tr.Trace(ob.make_request(16, 16, spp=1, block_y=10, block_h=10), scenes.make_seeds(1, 5))
        with pytest.raises(TracerError, match="seed list"):
            tr.Trace(ob.make_request(16, 16, spp=4), scenes.make_seeds(1, 5))
        with pytest.raises(TracerError, match="does not match"):
            tr.Trace(ob.make_request(32, 16, spp=1), scenes.make_seeds(1, 5))
        assert tr.Speed() > 0 and tr.Id() == "err"
    finally:
        tr.Close()


def test_full_size_invariants(built):
    """BASELINE.json headline size (512x512, 16 of the 128 spp to keep the suite short): properties
    that do not need the oracle -- run-to-run bit determinism, independence from the batch size up
    to summation order, energy conservation of the closed diffuse box, sample linearity."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES["cornell"]()
    W = H = 512
    spp, B = 16, 5
    seeds = scenes.make_seeds(spp, B)
    outs, cnts = [], []
    for opts in ({}, {}, {"samples_per_batch": 3}):
        tr = make_hip_tracer(sc, W, H, **opts)
        try:
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B), seeds)
            outs.append(tr.read_accumulator(0))
            cnts.append(counters(tr.last_trace_stats, B))
        finally:
            tr.Close()
    assert np.array_equal(bits(outs[0]), bits(outs[1])), "two identical runs differ"
    assert cnts[0] == cnts[1] == cnts[2]
    assert rmse(outs[0], outs[2], spp) <= 1e-6
    assert np.isfinite(outs[0]).all() and (outs[0][..., :3] >= 0).all()
    # linearity in the sample set: tracing the two halves of the seed list separately adds up
    tr = make_hip_tracer(sc, W, H)
    try:
        half = spp // 2
        tr.Trace(ob.make_request(W, H, spp=half, bounces=B), seeds[: half * (1 + B)])
        a = tr.read_accumulator(0)
        tr.Trace(ob.make_request(W, H, spp=spp - half, bounces=B), seeds[half * (1 + B):])
        b = tr.read_accumulator(0)
    finally:
        tr.Close()
    assert rmse(a + b, outs[0], spp) <= 1e-6


@pytest.mark.parametrize("name,W,H,spp", [("cornell", 1024, 1024, 256),        # C3 at its full size: 268 M primary samples
                                          ("material-ball", 1920, 1080, 6),  # C4's frame (of 512 spp)
                                          ("instanced", 2048, 2048, 4)])     # C5's frame (of 1024 spp): 4.19 M pixels, 1.09 M instanced triangles
def test_full_frame_sizes_of_the_other_configs(built, name, W, H, spp):
    """BASELINE.json configs[2..4] at their FULL frame sizes (slot / pixel index arithmetic at up to 4.19 M pixels x K
    samples per batch): run-to-run bit determinism, every ray counter independent of batch size and overlap depth, radiance
    independent of them up to the association of the per-sample sums, finite and non-negative."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name](W / H)
    B = 5
    seeds = scenes.make_seeds(spp, B)
    outs, cnts = [], []
    for opts in ({}, {}, {"samples_per_batch": max(1, spp // 3), "overlap": 2}, {"overlap": 1}):
        tr = make_hip_tracer(sc, W, H, **opts)
        try:
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B, rr=3), seeds)
            outs.append(tr.read_accumulator(0))
            cnts.append(counters(tr.last_trace_stats, B))
            if len(outs) == 1:
                st = tr.last_trace_stats
                assert st.primary_rays == W * H * spp and st.shaded_hits > 0 and st.occlusion_rays > 0
        finally:
            tr.Close()
    assert np.array_equal(bits(outs[0]), bits(outs[1])), "two identical runs differ"
    assert cnts[0] == cnts[1] == cnts[2] == cnts[3]
    assert rmse(outs[0], outs[2], spp) <= 1e-6 and rmse(outs[0], outs[3], spp) <= 1e-6
    assert np.isfinite(outs[0]).all() and (outs[0][..., :3] >= 0).all()
    lit = outs[0][..., :3].reshape(H, -1, 3).mean(axis=(1, 2)) > 0   # rows that received radiance: the whole height is traced
    assert lit[H // 2] and lit[: H // 4].any() and lit[-(H // 4):].any() and lit.mean() > 0.5


@pytest.mark.parametrize("name,W,H,spp", [("material-ball", 1920, 1080, 512),   # C4 at its full sample count: 1.06 G primary samples
                                          ("instanced", 2048, 2048, 1024)])    # C5: EXACTLY 2^32 primary samples
def test_full_sample_counts_of_c4_and_c5(built, name, W, H, spp):
    """BASELINE.json configs[3] and [4] at their FULL sample counts, one Trace each.  C5's 2048 x 2048 x 1024 spp is exactly 2^32
    primary rays: the boundary SURVEY.md 8 flags -- the reference's ray counters are u32 and the path index travels as a float
    (util/ray.cl:9-11); here every counter is 64-bit, a batch holds at most 2^25 path slots and a block at most 2^24 pixels, so
    nothing wraps: primary_rays == W * H * spp exactly, every per-bounce counter identical between two runs and between 1 and
    4 batches in flight, the accumulators bit-identical (same batch shapes => same order of every float sum), finite and
    non-negative."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name](W / H)
    B = 5
    seeds = scenes.make_seeds(spp, B)
    outs, cnts, ms = [], [], []
    for opts in ({}, {}, {"overlap": 1}):
        tr = make_hip_tracer(sc, W, H, **opts)
        try:
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B, rr=3), seeds)
            st = tr.last_trace_stats
            outs.append(tr.read_accumulator(0))
            cnts.append(counters(st, B))
            ms.append(st.device_ms)
            assert st.primary_rays == W * H * spp and st.rays_per_bounce[0] == W * H * spp
            assert st.primary_rays + st.indirect_rays + st.occlusion_rays == st.total_rays() > 2 * W * H * spp
            assert sum(st.rays_per_bounce[1:B]) == st.indirect_rays and sum(st.occl_per_bounce[:B]) == st.occlusion_rays
            assert 0 < st.shaded_hits and st.shaded_hits + st.shaded_misses <= st.primary_rays + st.indirect_rays   # (misses count only where a background exists)
        finally:
            tr.Close()
    if name == "instanced":
        assert cnts[0][2] == 2 ** 32
    assert cnts[0] == cnts[1] == cnts[2]
    assert np.array_equal(bits(outs[0]), bits(outs[1])) and np.array_equal(bits(outs[0]), bits(outs[2]))
    assert np.isfinite(outs[0]).all() and (outs[0][..., :3] >= 0).all() and outs[0][..., :3].mean() > 0
    print(f"{name} {W}x{H}x{spp}spp: {cnts[0][2] + cnts[0][3] + cnts[0][4]} rays, device ms per Trace {[round(v, 1) for v in ms]}")


@pytest.mark.parametrize("name", ["cornell", "sphere"])
def test_headline_frame_against_the_oracle(built, oracle, name):
    """BASELINE.json's headline workload (and configs[1], the diffuse sphere) at FULL size -- layered Cornell box, 512x512, 128 spp, 5
    bounces, RR from bounce 3, 163.5 M rays -- default (batched, overlapped) mode against the CPU
    oracle run over the same seeds (sample-parallel OpenMP mode: a few seconds on the GPU box's host
    cores): every ray counter identical (=> identical paths), per-pixel RMSE <= 1e-6 (the two differ
    only in the association of the per-sample sums; north_star bar: 1e-4)."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    W = H = 512
    spp, B = 128, 5
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=3)
    seeds = scenes.make_seeds(spp, B)
    want, wst, _ = oracle.trace(sc, req, seeds, flags=ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES)
    tr = make_hip_tracer(sc, W, H)
    try:
        tr.Trace(req, seeds)
        got, st = tr.read_accumulator(0), tr.last_trace_stats
    finally:
        tr.Close()
    assert st.primary_rays == W * H * spp
    assert counters(st, B) == counters(wst, B)
    if name == "cornell":
        assert st.primary_rays + st.indirect_rays + st.occlusion_rays == 163525398  # the frame bench.py reports
    assert rmse(got, want, spp) <= 1e-6


@pytest.mark.parametrize("name,W,H,spp", [("cornell", 1024, 1024, 256),       # C3 at its full size: 1.3 G rays
                                          ("material-ball", 1920, 1080, 32),  # C4's frame, 1/16 of its samples: 170 M rays
                                          ("instanced", 2048, 2048, 4)])      # C5's frame, 4 of its 1024 samples
def test_frames_of_the_other_configs_against_the_oracle(built, oracle, name, W, H, spp):
    """The remaining BASELINE.json configurations at their FULL frame size in the default (batched, overlapped) mode against
    the CPU oracle over the same seeds (C3 at its full sample count; C4 / C5 with as many samples as the host cores trace in
    about half a minute): every ray counter identical (=> identical paths through 4 M-pixel blocks, 33 M-slot batches, the
    global-memory node modes and the 24- / 32-entry stacks), per-pixel RMSE <= 1e-6."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name](W / H)
    B = 5
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=3)
    seeds = scenes.make_seeds(spp, B)
    want, wst, _ = oracle.trace(sc, req, seeds, flags=ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES)
    tr = make_hip_tracer(sc, W, H)
    try:
        tr.Trace(req, seeds)
        got, st = tr.read_accumulator(0), tr.last_trace_stats
    finally:
        tr.Close()
    assert st.primary_rays == W * H * spp
    assert counters(st, B) == counters(wst, B)
    assert rmse(got, want, spp) <= 1e-6


@pytest.mark.parametrize("name", ["material-ball", "instanced"])
def test_big_scene_kernel_variants_agree_with_the_oracle(built, oracle, name):
    """Scenes that select the other kernel variants -- the one-lane-per-ray kernel with its 24-entry stack and no LDS tree top,
    leaves of up to 4 triangles, (for > 256 K triangles) per-ray camera traversal -- traced in exact mode with each
    upload / traversal option flipped: all bit-identical to the CPU oracle."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name](4 / 3)
    W, H, spp, B = 96, 72, 2, 4
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=2)
    seeds = scenes.make_seeds(spp, B, base=21)
    want, wst, _ = oracle.trace(sc, req, seeds)
    assert wst.shaded_hits > 0
    for opts in ({}, {"max_leaf_tris": 0}, {"max_leaf_tris": 1}, {"max_leaf_tris": 2}, {"packet_primary": 0}, {"traversal": 0}, {"stage_lds": 0}, {"node_mode": 0},
                 {"node_mode": 1}, {"node_mode": 2}):
        tr = make_hip_tracer(sc, W, H, exact_accumulate=1, **opts)
        try:
            tr.Trace(req, seeds)
            got, st = tr.read_accumulator(0), tr.last_trace_stats
        finally:
            tr.Close()
        assert counters(st, B) == counters(wst, B), opts
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), opts


def _nested_shells_scene(n=22):
    """A tree whose rays NEED a deep stack: n camera-facing triangles at z = -0.05 x 1.6^k, each 1.6^k big (every camera ray's line
    crosses every one of their boxes), under a HAND-MADE BVH in the reference's encoding (optimized_scene.go:14-64): a chain that splits
    the outermost triangle off per level -- node j = { leaf of triangle n-1-j, node j+1 }.  Both children of every node lie on a camera
    ray; the traversal takes the nearer one (the rest of the chain) first and pushes the far leaf: one stack entry per level, n - 1 = 21
    of them, well beyond the 16 the spill variant keeps in LDS.  (Any tree whose boxes bound their contents is a valid input.)  A light
    behind the camera faces the shells: the innermost one is lit, its shadow rays walk the same chain."""
    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes

    tris = []
    for k in range(n):
        sz, z = 0.15 * 1.6 ** k, -0.05 * 1.6 ** k
        tris.append([[-sz, -sz, z], [sz, -sz, z], [0.0, 1.5 * sz, z]])
    tris.append([[-2.0, -2.0, 3.0], [0.0, 3.0, 3.0], [2.0, -2.0, 3.0]])      # the light: behind the camera, facing the shells (-z)
    n += 1
    verts = np.asarray(tris, np.float32)
    mt = scenes.MaterialTable()
    d, e = mt.diffuse((0.6, 0.7, 0.8)), mt.emissive((3, 3, 3))
    mat = np.full(n, d, np.uint32)
    mat[-1] = e
    mesh = scenes.Mesh(verts, scenes._flat_normals(verts), np.zeros((n, 3, 2), np.float32), mat)
    sc = scenes.compile_scene([mesh], [(0, np.eye(4))], mt, max_leaf=1, name="nested-shells")
    sc.set_camera(eye=(0.0, 0.0, 1.0), look=(0.0, 0.0, -1.0), fov=0.4, aspect=4 / 3)
    v = sc.vertices[:, :3].reshape(n, 3, 3)
    order = np.argsort(-v[:, 0, 2], kind="stable")   # compiled triangle indices, nearest (largest z) first
    nodes = np.zeros(2 * n, T.BVH_NODE)
    lo, hi = v.min(axis=1), v.max(axis=1)
    nodes[0] = (lo.min(axis=0), 0, hi.max(axis=0), 0)                         # the top-level tree: one leaf = instance 0
    for k in range(n):                                                        # leaf of the k-th nearest triangle: node n + k
        t = int(order[k])
        nodes[n + k] = (lo[t], -t, hi[t], 1)
    for j in range(n - 1):                                                    # inner node j: node 1 + j over the j-th .. nearest triangles
        members = order[: n - j]
        far_leaf = n + (n - 1 - j)
        rest = 1 + j + 1 if j + 1 < n - 1 else n + 0                          # the last inner node holds the two nearest leaves
        nodes[1 + j] = (lo[members].min(axis=0), far_leaf, hi[members].max(axis=0), rest)
    sc.bvh_nodes = nodes
    sc.mesh_instances["bvh_root"] = 1
    return sc


def test_rays_that_hold_twenty_stack_entries(built, oracle):
    """The deep end of the per-lane node stack: on the nested-shells chain every camera ray holds one entry per level, 20 and more --
    the 24-entry LDS variant of k_trace (max_stack is computed exactly at upload and picks it), closest hit and any hit, exact and
    batched: the oracle's frame bit for bit / within 1e-6.  (Also the regression scene of the round-6 experiment that kept only 16
    entries in LDS: polaris_amd/csrc/experiments/trace_spill.patch.)"""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = _nested_shells_scene()
    W, H, spp, B = 64, 48, 3, 4
    seeds = scenes.make_seeds(spp, B, base=5)
    want, wst, _ = oracle.trace(sc, ob.make_request(W, H, spp=spp, bounces=B, rr=2), seeds)
    assert wst.shaded_hits > 0 and want[..., :3].sum() > 0 and wst.occlusion_rays > 0
    for opts in ({"exact_accumulate": 1}, {"exact_accumulate": 0}, {"exact_accumulate": 1, "traversal": 0}):
        tr = make_hip_tracer(sc, W, H, time_kernels=1, node_mode=0, max_leaf_tris=0, **opts)
        try:
            tr.Trace(ob.make_request(W, H, spp=spp, bounces=B, rr=2), seeds)
            got, st = tr.read_accumulator(0), tr.last_trace_stats
            syms = (tr.kernel_symbol("intersect"), tr.kernel_symbol("occlusion"))
        finally:
            tr.Close()
        if opts.get("traversal", 1):
            assert syms == ("pol::k_trace<false, 24, 0, false>", "pol::k_trace<true, 24, 0, false>"), syms   # 17 .. 24 entries needed
        assert counters(st, B) == counters(wst, B), opts
        if opts["exact_accumulate"]:
            assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), opts
        else:
            assert rmse(got, want, spp) <= 1e-6


@pytest.mark.parametrize("name,one", [("cornell", True), ("sphere", True), ("cubes", False), ("transformed", False)])
def test_tiny_scene_kernel_variants_are_selected_as_documented(built, oracle, name, one):
    """Tiny-scene mode (whole tree + triangle records in LDS): scenes that ARE one instance with bounding boxes run the variant
    compiled for them (k_trace<.., ONE = true>), scenes of several instances the general one; option tiny_one = 0 forces the
    general one; every combination with the triangle records in LDS or not traces the oracle's frame bit for bit."""
    from oracle import pybind as ob
    from polaris_amd import scenes

    sc = scenes.SCENES[name]()
    W, H, spp, B = 64, 48, 2, 4
    req = ob.make_request(W, H, spp=spp, bounces=B, rr=2)
    seeds = scenes.make_seeds(spp, B, base=77)
    want, wst, _ = oracle.trace(sc, req, seeds)
    for opts in ({}, {"tiny_one": 0}, {"lds_tris": 0}, {"tiny_one": 0, "lds_tris": 0}, {"lds_tris": 5}):
        tr = make_hip_tracer(sc, W, H, exact_accumulate=1, time_kernels=1, **opts)
        try:
            tr.Trace(req, seeds)
            got, st = tr.read_accumulator(0), tr.last_trace_stats
            symbols = [tr.kernel_symbol("intersect"), tr.kernel_symbol("occlusion")]
        finally:
            tr.Close()
        expect_one = one and opts.get("tiny_one", 1) != 0
        for sym in symbols:
            assert sym.startswith("pol::k_trace<") and ", 16, 2, " in sym, sym                      # the tiny-scene mode
            assert sym.endswith(", true>" if expect_one else ", false>"), (name, opts, sym)
        assert counters(st, B) == counters(wst, B), (name, opts)
        assert np.array_equal(bits(got[..., :3]), bits(want[..., :3])), (name, opts)


def test_soak_one_handle_many_shapes_scenes_and_options(built):
    """150 Trace calls on one handle with random frame sizes, row blocks, scenes, bounce counts,
    overlap depths and accumulation modes (tests/tools/soak.py): every call succeeds, radiance stays
    finite, device memory does not creep."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "tools"))
    import soak

    start, end, worst = soak.run(150)
    assert start - end < 512


def test_results_do_not_depend_on_timing(built):
    """The shade kernels append a chunk's rays in whatever order their waves finish (the reference order is carried as
    data: parent's canonical index + emit masks): 25 traces of the same frame per scene must be bit-identical -- radiance
    and every counter (tests/tools/determinism_stress.py; 150 frames per scene were run once by hand)."""
    import subprocess
    import sys
    import os

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "determinism_stress.py")
    p = subprocess.run([sys.executable, tool, "25"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("bit-identical") == 4, p.stdout
