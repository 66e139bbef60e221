"""Random CALL SEQUENCES on the C ABI against a model (GPU).

The parity tests call the tracer in a handful of fixed orders.  Here up to three tracers of one process receive seeded random sequences
of everything `tracer.Tracer` offers -- Trace over any row block with or without accumulated samples, MergeOutput between any two of
them (their own trace accumulator included), merges from a named ring slot, SyncFramebuffer over any rows / sample weight / exposure,
the Reset stage, synchronous and asynchronous camera and scene updates, resizes, ring exports of depth 1-4 -- and after every step a
model built from the reference's documented semantics and the CPU oracle says what the three buffers of every tracer must hold:

    Trace          tracer/opencl/tracer.go:194-247   pending changes commit; AccumulatedSamples == 0 clears the frame accumulator;
                                                      the WHOLE trace accumulator is cleared, then the block's rows are traced
    MergeOutput    resources.go:108-124              dst.frame[rows] += src.trace[rows]
    SyncFramebuffer resources.go:344-360             framebuffer[rows] = tonemap(frame[rows] / (accumulated + spp), exposure)
    UpdateState    tracer.go:150-192                 Synchronous: at once (with everything queued before); Asynchronous: at the next commit

Checked bit for bit (exact_accumulate = 1): trace accumulator, frame accumulator, RGBA8 frame buffer, at random points and at the end.
"""
import copy
import os
import sys

import numpy as np
import pytest

from conftest import bits, make_hip_tracer

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))

pytestmark = pytest.mark.gpu

DIMS = [(33, 17), (64, 24), (20, 31), (97, 9)]


class Model:
    """What one tracer must hold, from the documented semantics + the oracle."""

    def __init__(self, oracle, scene, W, H):
        self.oracle = oracle
        self.pending = []                         # queued (kind, payload), in order
        self.scene, self.camera = scene, scene    # committed scene arrays / committed camera (an object with eye + frustum)
        self.resize(W, H)

    def resize(self, W, H):
        self.W, self.H = W, H
        self.ring = [np.zeros((H, W, 3), np.float32)]
        self.pos = 0
        self.frame = np.zeros((H, W, 3), np.float32)
        self.fb = np.zeros((H, W, 4), np.uint8)

    @property
    def trace(self):
        return self.ring[self.pos]

    def commit(self):
        for kind, payload in self.pending:
            if kind == "dims":
                self.resize(*payload)
            elif kind == "scene":
                self.scene = payload
            else:
                self.camera = payload
        self.pending = []

    def export(self, depth):
        H, W = self.H, self.W
        self.ring = [self.ring[i] if i < len(self.ring) else np.zeros((H, W, 3), np.float32) for i in range(depth)]
        self.pos %= depth

    def do_trace(self, req, seeds):
        from oracle import pybind as ob

        self.commit()
        if req["accumulated"] == 0:
            self.frame[:] = 0
        if len(self.ring) > 1:
            self.pos = (self.pos + 1) % len(self.ring)
        sc = copy.copy(self.scene)
        sc.eye, sc.frustum = self.camera.eye, self.camera.frustum
        want, st, _ = self.oracle.trace(sc, ob.make_request(self.W, self.H, spp=req["spp"], bounces=req["bounces"], rr=req["rr"], block_y=req["by"], block_h=req["bh"]), seeds)
        self.ring[self.pos] = want[..., :3].copy()
        return st

    def merge(self, src_rows, by, bh):
        self.frame[by:by + bh] = self.frame[by:by + bh] + src_rows[by:by + bh]

    def sync(self, by, bh, accumulated, spp, exposure):
        a4 = np.zeros((bh, self.W, 4), np.float32)
        a4[..., :3] = self.frame[by:by + bh]
        w = 1.0 / (accumulated + spp)     # (the library divides in double and rounds to float once, resources.go:347; ctypes rounds the same way)
        self.fb[by:by + bh] = self.oracle.tonemap(a4, w, exposure).reshape(bh, self.W, 4)


def dict_order_queue(pending, kind, payload):
    """HipTracer keeps queued changes in a dict keyed by change type: re-queuing a type replaces the payload but keeps the position."""
    for i, (k, _) in enumerate(pending):
        if k == kind:
            pending[i] = (kind, payload)
            return pending
    return pending + [(kind, payload)]


def moved_camera(scene, rng):
    cam = copy.copy(scene)
    cam.eye = (np.asarray(scene.eye, np.float32) + rng.uniform(-0.3, 0.3, 3).astype(np.float32)).astype(np.float32)
    fr = np.asarray(scene.frustum, np.float32).copy()
    fr.reshape(4, 4)[:, :3] += rng.uniform(-0.05, 0.05, 3).astype(np.float32)
    cam.frustum = fr
    return cam


@pytest.mark.parametrize("seed", range(12))
def test_random_call_sequences_against_the_model(built, oracle, seed):
    from oracle import pybind as ob
    from polaris_amd import scenes
    from polaris_amd.tracer import ChangeType, TracerError, UpdateMode
    from random_scenes import random_case

    rng = np.random.default_rng(0xCA11 + seed)
    pool = [scenes.SCENES["cornell"](), scenes.SCENES["cubes"](), random_case(100 + seed)[0], random_case(200 + seed, single=True)[0]]
    n_tr = int(rng.integers(1, 4))
    W, H = DIMS[int(rng.integers(0, len(DIMS)))]
    trs, models, acc = [], [], []
    log = []
    try:
        for t in range(n_tr):
            sc = pool[int(rng.integers(0, len(pool)))]
            trs.append(make_hip_tracer(sc, W, H, exact_accumulate=1))
            models.append(Model(oracle, sc, W, H))
            acc.append(0)

        def check(t, what=("trace", "frame", "fb")):
            m, tr = models[t], trs[t]
            if "trace" in what:
                assert np.array_equal(bits(tr.read_accumulator(0)[..., :3]), bits(m.trace)), (seed, t, "trace accumulator", log[-6:])
            if "frame" in what or "fb" in what:
                # (reading the frame accumulator / frame buffer completes the pending merges first, like SyncFramebuffer's wait does)
                assert np.array_equal(bits(tr.read_accumulator(1)[..., :3]), bits(m.frame)), (seed, t, "frame accumulator", log[-6:])
            if "fb" in what:
                assert np.array_equal(tr.read_framebuffer(), m.fb), (seed, t, "frame buffer", log[-6:])

        for step in range(int(rng.integers(25, 45))):
            t = int(rng.integers(0, n_tr))
            tr, m = trs[t], models[t]
            op = rng.choice(["trace", "trace", "trace", "merge", "merge", "merge_slot", "sync", "sync", "reset", "camera", "scene", "resize", "export", "check"])
            if op == "trace":
                m_dims = dict(m.pending).get("dims", (m.W, m.H))
                w, h = m_dims
                by = int(rng.integers(0, h))
                bh = int(rng.integers(1, h - by + 1)) if rng.random() < 0.6 else h - by
                spp, B = int(rng.choice([0, 1, 1, 2, 3])), int(rng.integers(1, 5))
                a = acc[t] if rng.random() < 0.4 else 0
                rq = dict(by=by, bh=bh, spp=spp, bounces=B, rr=int(rng.integers(0, B + 2)), accumulated=a)
                seeds = scenes.make_seeds(max(spp, 1), B, base=int(rng.integers(0, 1 << 30)))
                log.append((step, t, "trace", rq))
                req = ob.make_request(w, h, spp=spp, bounces=B, rr=rq["rr"], block_y=by, block_h=bh, accumulated=a)
                tr.Trace(req, seeds)
                st = m.do_trace(rq, seeds)
                acc[t] = a + spp
                gs = tr.last_trace_stats
                assert (gs.primary_rays, gs.indirect_rays, gs.occlusion_rays, gs.shaded_hits) == (st.primary_rays, st.indirect_rays, st.occlusion_rays, st.shaded_hits), (seed, log[-3:])
                check(t, ("trace",))
            elif op in ("merge", "merge_slot"):
                s = int(rng.integers(0, n_tr))
                src, sm = trs[s], models[s]
                by = int(rng.integers(0, m.H))
                bh = int(rng.integers(1, m.H - by + 1))
                req = ob.make_request(m.W, m.H, spp=1, bounces=1, block_y=by, block_h=bh)
                slot = int(rng.integers(0, len(sm.ring))) if op == "merge_slot" else None
                log.append((step, t, op, dict(src=s, by=by, bh=bh, slot=slot)))
                if (sm.W, sm.H) != (m.W, m.H):
                    with pytest.raises(TracerError):
                        tr.MergeOutput(src, req) if slot is None else tr.merge_slot(src, slot, req)
                    continue
                if slot is None:
                    tr.MergeOutput(src, req)
                    m.merge(sm.trace, by, bh)
                else:
                    tr.merge_slot(src, slot, req)
                    m.merge(sm.ring[slot], by, bh)
            elif op == "sync":
                by = int(rng.integers(0, m.H))
                bh = int(rng.integers(1, m.H - by + 1)) if rng.random() < 0.5 else m.H - by
                a, spp, exposure = int(rng.choice([0, 0, 2, 7])), int(rng.integers(1, 5)), float(np.float32(rng.choice([0.6, 1.0, 1.2, 2.5])))
                req = ob.make_request(m.W, m.H, spp=spp, bounces=1, block_y=by, block_h=bh, exposure=exposure, accumulated=a)
                log.append((step, t, "sync", dict(by=by, bh=bh, a=a, spp=spp, exposure=exposure)))
                tr.SyncFramebuffer(req)
                m.sync(by, bh, a, spp, exposure)
                if rng.random() < 0.5:
                    check(t, ("frame", "fb"))
            elif op == "reset":
                log.append((step, t, "reset"))
                tr.reset_frame()
                m.frame[:] = 0
            elif op == "camera":
                cam = moved_camera(m.scene, rng)
                sync = bool(rng.random() < 0.5)
                log.append((step, t, "camera", sync))
                m.pending = dict_order_queue(m.pending, "camera", cam)
                tr.UpdateState(UpdateMode.Synchronous if sync else UpdateMode.Asynchronous, ChangeType.CameraData, cam)
                if sync:
                    m.commit()
            elif op == "scene":
                sc = pool[int(rng.integers(0, len(pool)))]
                sync = bool(rng.random() < 0.5)
                log.append((step, t, "scene", sc.name, sync))
                m.pending = dict_order_queue(dict_order_queue(m.pending, "scene", sc), "camera", sc)   # (a new scene comes with its camera, as the renderer sends them)
                tr.UpdateState(UpdateMode.Asynchronous, ChangeType.SceneData, sc)
                tr.UpdateState(UpdateMode.Synchronous if sync else UpdateMode.Asynchronous, ChangeType.CameraData, sc)
                if sync:
                    m.commit()
            elif op == "resize":
                dims = DIMS[int(rng.integers(0, len(DIMS)))]
                sync = bool(rng.random() < 0.6)
                log.append((step, t, "resize", dims, sync))
                m.pending = dict_order_queue(m.pending, "dims", dims)
                tr.UpdateState(UpdateMode.Synchronous if sync else UpdateMode.Asynchronous, ChangeType.FrameDimensions, dims)
                if sync:
                    m.commit()
                acc[t] = 0
            elif op == "export":
                depth = int(rng.integers(1, 5))
                log.append((step, t, "export", depth))
                tr.ipc_export(depth)
                m.export(depth)
                assert tr.trace_slot() == m.pos
            else:
                check(t)
            if rng.random() < 0.35:      # (not after every step: a read-back completes the queued merges, and sequences without one are wanted too)
                check(int(rng.integers(0, n_tr)), ("frame",))
        for t in range(n_tr):
            trs[t].SyncFramebuffer(ob.make_request(models[t].W, models[t].H, spp=1, bounces=1))
            models[t].sync(0, models[t].H, 0, 1, 1.2)
            check(t)
    finally:
        for tr in trs:
            tr.Close()
