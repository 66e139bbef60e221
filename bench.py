#!/usr/bin/env python3
"""bench.py -- Mrays/s + ms/frame of the polaris tracer hot path on MI355X.

A "step" is one frame of BASELINE.json's headline workload: the (layered) Cornell box at 512x512,
128 spp, 5 bounces, Russian roulette from bounce 3, exposure 1.2 (CLI defaults, main.go:76-100),
rendered the way renderer/default.go:106-171 renders a frame:

    Schedule (naive, equal speeds -> FrameH/N rows each, remainder to tracer 0)
    -> every tracer Trace()s its row block -> the primary MergeOutput()s every block
    -> primary SyncFramebuffer() (tone-map)

One process per GPU (torch.distributed, backend nccl = RCCL): rank r owns row block r; the single
exchange step of the path -- the gather of the blocks' accumulator strips to the primary
(renderer/default.go:191, tracer/opencl/resources.go:108-124) -- is a dist.gather of
block_h*frame_w float4 per rank (512 KiB per peer at 8 GPUs).  The frame is fixed, so scaling is
STRONG.  Scene and seeds are synthetic (polaris_amd/scenes.py); inputs are resident in HBM before
the timed region starts.

Prints ONE JSON line (rank 0).  `value` = rays traced by all ranks in the K timed frames / wall
time (max over ranks); rays = primary + indirect + occlusion rays handed to an intersection kernel
(BASELINE.md section 3).  `roofline` prices the dominant kernel against HBM: algorithmic bytes
(SURVEY.md 8d split per kernel) / its HIP-event time.  `cpu_baseline` times the CPU
restatement under oracle/ (the checker, never the product) on a bounded sample of the same
workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

KERNELS = ("generate", "intersect_packet", "intersect", "shade", "scan", "occlusion", "resolve", "aggregate", "tonemap")
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def kernel_algorithmic_bytes(st: dict) -> dict:
    """SURVEY.md 8d per-unit stream bytes, attributed to the kernel that moves them
    (intersect_packet = k_trace_packet on camera rays, intersect = k_trace<closest> on bounce rays)."""
    return {
        # camera rays: SURVEY's 52 + 60 B minus what is constant for a camera ray and therefore neither stored nor read
        # (origin | max distance: 16 B written + 16 B read; throughput: 16 B written)
        "generate": 20 * st["primary_rays"],
        "intersect_packet": 44 * st["primary_rays"],
        "intersect": 60 * st["indirect_rays"],
        "shade": 68 * st["shaded_hits"] + 32 * st["indirect_rays"] + 44 * st["occlusion_rays"]
        + 60 * st["shaded_misses"] + 24 * st["emitter_hits"],
        "occlusion": 36 * st["occlusion_rays"] + 44 * st["unoccluded"],
    }


def host_cpu() -> str:
    """'<n> x <model name>' of the host the CPU baseline runs on (state the cores)."""
    try:
        models = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        return f"{len(models)} x {models[0]}" if models else f"{os.cpu_count()} logical CPUs"
    except OSError:
        return f"{os.cpu_count()} logical CPUs"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="cornell")
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--rr", type=int, default=3)
    ap.add_argument("--samples-per-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--save-accumulator", default="", help="rank 0 writes its frame accumulator of the last frame as .npy (tests)")
    ap.add_argument("--opt", action="append", default=[], help="tracer option key=value (polaris_hip_set_option)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to test the N>1 flow on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="testing aid: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--emulate-rank", default="", help="R/N: on ONE GPU, trace only the row block rank R of N would own "
                    "(tuning aid for the strong-scaling path; not a valid bench line)")
    args = ap.parse_args()

    from polaris_amd.hostinfo import size_openmp

    cpu_threads = size_openmp()  # the CPU-baseline leg: one OpenMP thread per CPU this process may really use (cgroup quota)

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Launched bare (`python bench.py --gpus N`): start the N ranks ourselves, one process per GPU, BEFORE anything in
        # this process touches HIP (torch.cuda.device_count() does not initialise the GPU on this image), relay the
        # child's output (rank 0 prints the JSON line) and exit with its status.  renderer/default.go:127-156 is the
        # reference's equivalent: one worker per device inside one Render().
        if not args.same_device:
            import torch

            visible = torch.cuda.device_count()
            if visible < args.gpus:
                raise SystemExit(f"bench.py --gpus {args.gpus}: only {visible} HIP device(s) visible on this host")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: they must agree")

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracer has no CPU fallback")
    dev = torch.device("cuda", local_rank)

    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    from polaris_amd.distributed import StripExchange, block_of, naive_rows
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    W, H, spp, B = args.width, args.height, args.spp, args.bounces
    sc = scenes.SCENES[args.scene](W / H)
    seeds = scenes.make_seeds(spp, B)
    rows = naive_rows(world, H)                      # tracer/scheduler.go:83-106, equal speeds
    block_y, block_h = block_of(rank, rows)
    if args.emulate_rank:
        er, en = (int(v) for v in args.emulate_rank.split("/"))
        block_y, block_h = block_of(er, naive_rows(en, H))
        rows = [block_h]

    tr = HipTracer(f"hip-{rank}", local_rank)
    tr.Init()
    if args.samples_per_batch:
        tr.set_option("samples_per_batch", args.samples_per_batch)
    tr.set_option("time_kernels", 0)  # no event pairs inside the timed region; per-kernel times come from one extra frame afterwards
    for kv in args.opt:  # before the upload: some options shape the scene layout
        k, v = kv.split("=")
        tr.set_option(k, int(v))
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.FrameDimensions, (W, H))
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.SceneData, sc)
    tr.UpdateState(UpdateMode.Synchronous, ChangeType.CameraData, sc)

    def make_req(by, bh):
        r = T.BlockRequest()
        r.frame_w, r.frame_h, r.block_x, r.block_y, r.block_w, r.block_h = W, H, 0, by, W, bh
        r.samples_per_pixel, r.num_bounces, r.min_bounces_for_rr = spp, B, args.rr
        r.exposure, r.seed, r.accumulated_samples = 1.2, 0, 0
        return r

    # The path's one exchange step (renderer/default.go:191): gather of the blocks' strips to the primary.  It runs one
    # frame behind the tracing (polaris_amd/distributed.py): frame i's gather / merge / tone-map overlap frame i+1's Trace.
    ex = StripExchange(dist, rank, rows, W, dev, via_host=args.backend != "nccl") if world > 1 else None
    totals = {k: 0 for k in ("primary_rays", "indirect_rays", "occlusion_rays", "shaded_hits", "shaded_misses", "emitter_hits", "unoccluded")}
    pending = []

    def finish_frame(ticket):
        parts = ex.wait(ticket)
        if rank == 0:
            tr.reset_frame()                       # the Reset stage: this frame's blocks land on a cleared accumulator
            for y, h, t in parts:                  # ONE merge over the whole frame when the blocks are equally tall
                tr.merge_device(t.data_ptr(), make_req(y, h))
            tr.SyncFramebuffer(make_req(0, H))     # default.go:159-161

    def flush():
        while pending:
            finish_frame(pending.pop(0))

    def frame(count: bool):
        req = make_req(block_y, block_h)
        tr.Trace(req, seeds)                       # Trace (tracer.go:194-247)
        if count:
            st = tr.last_trace_stats
            for k in totals:
                totals[k] += int(getattr(st, k))
        if world == 1:
            tr.MergeOutput(tr, req)                # primary merges its own block (default.go:191)
            tr.SyncFramebuffer(make_req(0, H) if not args.emulate_rank else req)
        else:
            ticket = ex.post(lambda strip: tr.export_block(req, strip.data_ptr()))
            flush()                                # the PREVIOUS frame: its gather ran beside this frame's Trace
            pending.append(ticket)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame(False)
    flush()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame(True)
    flush()                                        # all K frames merged and tone-mapped inside the timed region
    fence()
    elapsed = time.perf_counter() - t0
    if args.save_accumulator and rank == 0:
        np.save(args.save_accumulator, tr.read_accumulator(1))  # the primary's frame accumulator of the last frame (tests)

    rdev = dev if args.backend == "nccl" else torch.device("cpu")
    el = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    cnt = torch.tensor([totals[k] for k in sorted(totals)], dtype=torch.int64, device=rdev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    tot = {k: int(v) for k, v in zip(sorted(totals), cnt.tolist())}
    rays = tot["primary_rays"] + tot["indirect_rays"] + tot["occlusion_rays"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/s (primary+indirect+occlusion rays traced / wall), Cornell box 512x512x128spp",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_frame": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{sc.name} {W}x{H} {spp}spp {B} bounces rr>={args.rr}, {sc.num_triangles} tris, "
                                   f"row blocks {rows} (naive scheduler), gather-to-primary + tonemap",
                       "ranks": world, "exchange": "none (1 GPU)" if world == 1 else f"{args.backend} gather of the row-block strips to rank 0, one frame behind the tracing",
                       "frame": [W, H], "spp": spp, "bounces": B, "rays_per_frame": rays // args.steps,
                       "paths_per_s": W * H * spp * args.steps / elapsed},
        }
        # ---- roofline of the dominant kernel ---------------------------------------------------
        # Kernel durations are HIP events on the tracer's own streams (the library brackets every launch
        # when time_kernels=1).  They are NOT taken inside the timed region (event pairs around ~88 launches
        # per frame, and up to four batches share the GPU there): ONE extra frame is traced afterwards with
        # overlap=1 -- same kernels, same inputs, one batch at a time -- and the roofline uses those times.
        if not args.no_kernel_timers:
            mine = {k: totals[k] // args.steps for k in totals}     # per-frame counters of rank 0
            tr.set_option("overlap", 1)
            tr.set_option("time_kernels", 1)
            for name in KERNELS:
                tr.kernel_ms(name)
            tr.Trace(make_req(block_y, block_h), seeds)   # local to rank 0: no collective in here
            iso = {name: tr.kernel_ms(name) for name in KERNELS[:-2]}
            alg = kernel_algorithmic_bytes(mine)
            dom = max(("generate", "intersect_packet", "intersect", "shade", "occlusion"), key=lambda k: iso[k][0])
            ms, n = iso[dom]
            achieved = alg[dom] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            # HBM traffic per launch of that kernel: PMC data (FETCH_SIZE x2 + WRITE_SIZE, separate passes: scripts/traffic.sh)
            # cannot be collected inside this run; it is read from the newest committed profile of THIS workload, and
            # traffic_source says which file that was (null + null when there is none for the workload).
            traffic, traffic_source = None, None
            if (W, H, spp, B, args.scene, world) == (512, 512, 128, 5, "cornell", 1):
                import glob

                for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
                    tj = json.load(open(tpath))
                    ent = tj.get("timers", {}).get(dom)
                    if ent is None:  # round-1 file: keyed by kernel name
                        kn = {"intersect": ("pol::k_trace<false",), "occlusion": ("pol::k_trace<true",), "shade": ("pol::k_shade<", "pol::k_shade_wave<"),
                              "generate": ("pol::k_generate",), "intersect_packet": ("pol::k_trace_packet<false>",)}[dom]
                        tks = [v for k, v in tj["kernels"].items() if k.startswith(kn)]
                        if tks:
                            ent = {"hbm_bytes": sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in tks), "launches": sum(v["launches"] for v in tks)}
                    if ent and ent.get("launches"):
                        traffic = ent["hbm_bytes"] / ent["launches"]
                        traffic_source = "profiles/" + os.path.basename(tpath) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1`, " \
                            "collected on the builder's MI355X lease; not re-measured in this run)"
                        break
            names = {"intersect": "k_trace<false,...> (closest hit, bounce rays)", "intersect_packet": "k_trace_packet<false> (camera rays)",
                     "occlusion": "k_trace<true,...> (any hit + NEE accumulate)", "shade": "k_shade + k_shade_wave (shadeHits, miss shading, compaction)",
                     "generate": "k_generate"}
            out["roofline"] = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                               "algorithmic_bytes_per_launch": alg[dom] / max(n, 1), "avg_launch_ms": ms / max(n, 1), "launches": n,
                               "measured": "HIP events, one extra frame with overlap=1 after the timed region"}
            # the same object for every traversal / shading kernel (the dominant one is repeated above as `roofline`)
            per_kernel = {}
            for k in ("generate", "intersect_packet", "intersect", "shade", "occlusion"):
                kms, kn_ = iso[k]
                if kms > 0 and kn_:
                    ach = alg[k] / (kms * 1e-3) / 1e9
                    per_kernel[k] = {"kernel": names[k], "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(kms / kn_, 4),
                                     "launches": kn_, "algorithmic_bytes_per_launch": round(alg[k] / kn_)}
            out["roofline_per_kernel"] = per_kernel
            out["kernels_isolated_ms_per_frame"] = {k: round(v[0], 3) for k, v in iso.items()}
            out["kernels_isolated_GBps_algorithmic"] = {k: round(alg[k] / (iso[k][0] * 1e-3) / 1e9, 1) for k in alg if iso[k][0] > 0}
            whole = (112 * mine["primary_rays"] + 68 * mine["shaded_hits"] + 92 * mine["indirect_rays"] + 80 * mine["occlusion_rays"]
                     + 44 * mine["unoccluded"] + 60 * mine["shaded_misses"] + 24 * mine["emitter_hits"] + (48 + 16) * rows[0] * W)
            out["whole_path_algorithmic_GBps_rank0"] = whole / (elapsed / args.steps) / 1e9
        # ---- CPU baseline: the oracle (checker) on a bounded sample of the same workload ----
        # Sample-parallel mode of the restatement (threads take whole samples; same paths as the
        # reference order, per-pixel sums re-associated): the fastest way to run this path on CPU cores.
        if not args.no_cpu_baseline and world == 1:
            try:
                from oracle import pybind as ob

                orc = ob.Oracle("oracle")
                cores = cpu_threads
                flags = ob.FIX_EMITTER_INDEX | ob.PARALLEL_SAMPLES
                probe_spp = cores                        # one sample per thread
                max_spp = 64 * cores
                cseeds = scenes.make_seeds(max(max_spp, spp), B)
                t = time.perf_counter()
                _, cs, _ = orc.trace(sc, ob.make_request(W, H, spp=probe_spp, bounces=B, rr=args.rr), cseeds[: probe_spp * (1 + B)], flags=flags)
                dt = time.perf_counter() - t
                cpu_rays, cpu_dt, cpu_n = cs.total_rays(), dt, probe_spp
                if dt < 6.0:                             # scale the sample towards ~12 s of wall time
                    more = int(min(max_spp, max(probe_spp, probe_spp * 12.0 / max(dt, 1e-3)))) // cores * cores
                    if more > probe_spp:
                        t = time.perf_counter()
                        _, cs2, _ = orc.trace(sc, ob.make_request(W, H, spp=more, bounces=B, rr=args.rr), cseeds[: more * (1 + B)], flags=flags)
                        cpu_rays, cpu_dt, cpu_n = cs2.total_rays(), time.perf_counter() - t, more
                out["cpu_baseline"] = {"value": cpu_rays / cpu_dt / 1e6, "unit": "Mrays/s", "cores": min(cores, cpu_n), "kind": "port",
                                       "sample": f"same scene/frame/options, {cpu_n} samples per pixel ({cpu_rays} rays, {cpu_dt:.1f} s wall), "
                                                 f"oracle/polaris_oracle.cpp, OpenMP over samples on {min(cores, cpu_n)} threads of {host_cpu()}",
                                       "ms_per_frame_extrapolated": cpu_dt / cpu_n * spp * 1e3}
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": cpu_threads, "kind": "port", "sample": f"failed: {e}"}
        if args.save_png:
            from PIL import Image

            Image.fromarray(tr.read_framebuffer()[..., :3]).save(args.save_png)
        print(json.dumps(out))
    tr.Close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
