#!/usr/bin/env python3
"""scripts/emulate_ranks.py <tag> [config ...] -- on ONE MI355X (inside gpurun): what a rank of an N-GPU frame costs, for the
naive scheduler's equal blocks (what `polaris render` passes, cmd/render.go:65) and for the rows the reference's perfect
scheduler (tracer/scheduler.go:50-80, restated in polaris_amd/host/scheduler.cpp) settles on when it is fed the measured
times.  Every rank's block is traced alone, the way bench.py drives a rank (Trace -> MergeOutput -> SyncFramebuffer of the
block): an N-GPU frame takes what its slowest rank takes; the exchange (peer reads of 0.5 - 8 MiB per rank) runs one frame
behind the tracing.  An ESTIMATE of the scaling curve -> gpurun_out/<tag>_emulated_ranks_<config>.json.

configs: headline (Cornell 512^2 x 128), C3 (Cornell 1024^2 x 256), C4 (material-ball 1920 x 1080 x 512, 135 rows per rank at 8),
C5 (instanced 2048^2, 256 of its 1024 spp: 256 rows per rank at 8 -- a rank's batches are full either way).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROUNDS = int(os.environ.get("EMULATE_ROUNDS", "3"))  # round 1 = naive rows, rounds 2.. = the perfect scheduler fed with the previous round's times
CONFIGS = {
    "headline": ("cornell", 512, 512, 128, 10, 3),
    "C3": ("cornell", 1024, 1024, 256, 3, 1),
    "C4": ("material-ball", 1920, 1080, 512, 2, 1),
    "C5": ("instanced", 2048, 2048, 256, 2, 1),
}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    which = sys.argv[2:] or ["headline"]
    from polaris_amd import ctypes_api as T
    from polaris_amd import host_api, scenes
    from polaris_amd.distributed import block_of, naive_rows
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    for cfg in which:
        scene, W, H, spp, steps, warmup = CONFIGS[cfg]
        B, rr = 5, 3
        sc = scenes.SCENES[scene](W / H)
        seeds = scenes.make_seeds(spp, B)
        tr = HipTracer("emu", 0)
        tr.Init()
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
        tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)

        def rank_ms(r, rows):
            by, bh = block_of(r, rows)
            req = T.BlockRequest()
            req.frame_w, req.frame_h, req.block_x, req.block_y, req.block_w, req.block_h = W, H, 0, by, W, bh
            req.samples_per_pixel, req.num_bounces, req.min_bounces_for_rr = spp, B, rr
            req.exposure, req.seed = 1.2, 0

            def frame():
                req.accumulated_samples = 0
                tr.Trace(req, seeds)
                tr.MergeOutput(tr, req)
                tr.SyncFramebuffer(req)

            for _ in range(warmup):
                frame()
            t = time.perf_counter()
            for _ in range(steps):
                frame()
            return (time.perf_counter() - t) / steps * 1e3

        def primary_merge_ms(rows):
            """What the PRIMARY pays per frame on top of its own Trace: the Reset stage, one MergeOutput (k_aggregate) per block and the
            tone-map of the whole frame with its synchronisation (renderer/default.go:159-161,191) -- bench.py's `own_work` of rank 0
            inside PeerExchange.merge().  Every block is read from this tracer's own accumulator here (a peer read over xGMI moves the
            same bytes: 16 B per pixel of the block)."""
            full = T.BlockRequest()
            full.frame_w, full.frame_h, full.block_x, full.block_y, full.block_w, full.block_h = W, H, 0, 0, W, H
            full.samples_per_pixel, full.num_bounces, full.min_bounces_for_rr, full.exposure = spp, B, rr, 1.2
            reqs = []
            for r in range(len(rows)):
                by, bh = block_of(r, rows)
                q = T.BlockRequest()
                q.frame_w, q.frame_h, q.block_x, q.block_y, q.block_w, q.block_h = W, H, 0, by, W, bh
                q.samples_per_pixel, q.num_bounces, q.min_bounces_for_rr, q.exposure = spp, B, rr, 1.2
                reqs.append(q)

            def once():
                tr.reset_frame()
                for q in reqs:
                    tr.MergeOutput(tr, q)
                tr.SyncFramebuffer(full)

            for _ in range(3):
                once()
            t = time.perf_counter()
            for _ in range(20):
                once()
            return (time.perf_counter() - t) / 20 * 1e3

        one = rank_ms(0, [H])
        out = {"what": f"{sc.name} {W}x{H}x{spp}spp, one rank's block at a time on one MI355X ({steps} timed frames after {warmup})", "one_gpu_ms": round(one, 3), "N": {}}
        print(cfg, "1 GPU", round(one, 3), flush=True)
        for n in (2, 4, 8):
            rows = naive_rows(n, H)
            sched = host_api.Scheduler(host_api.PERFECT, [1] * n)
            sched.schedule(H)  # its first frame is the naive split (scheduler.go:52-56)
            rounds = []
            for it in range(ROUNDS):
                ms = [rank_ms(r, rows) for r in range(n)]
                merge_ms = primary_merge_ms(rows)
                rounds.append({"scheduler": "naive" if it == 0 else f"perfect, frame {it + 1}", "rows": rows, "ms_per_rank": [round(v, 3) for v in ms],
                               "slowest_ms": round(max(ms), 3), "speedup_vs_1gpu": round(one / max(ms), 2), "efficiency": round(one / max(ms) / n, 3),
                               "n_times_slowest_over_1gpu": round(n * max(ms) / one, 3), "sum_of_ranks_over_1gpu": round(sum(ms) / one, 3),
                               # the primary's per-frame merge work (reset + n k_aggregate + tone-map + sync).  In bench.py it runs one frame behind the
                               # tracing on rank 0's host thread, i.e. it ADDS to rank 0's frame: the column below is the frame time with it
                               "primary_merge_ms": round(merge_ms, 3), "slowest_ms_with_primary_merge": round(max([ms[0] + merge_ms] + ms[1:]), 3),
                               "speedup_vs_1gpu_with_primary_merge": round(one / max([ms[0] + merge_ms] + ms[1:]), 2)})
                print(cfg, n, rounds[-1], flush=True)
                rows = sched.schedule(H, block_h=rows, render_ns=[int(v * 1e6) for v in ms])  # the next frame's rows, from this frame's times
            out["N"][str(n)] = rounds
        tr.Close()
        path = os.path.join(ROOT, "gpurun_out", f"{tag}_emulated_ranks_{cfg}.json")
        json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
