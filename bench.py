#!/usr/bin/env python3
"""bench.py -- Mrays/s + ms/frame of the polaris tracer hot path on MI355X.

A "step" is one frame of BASELINE.json's headline workload: the (layered) Cornell box at 512x512,
128 spp, 5 bounces, Russian roulette from bounce 3, exposure 1.2 (CLI defaults, main.go:76-100),
rendered the way renderer/default.go:106-171 renders a frame:

    Schedule (naive, equal speeds -> FrameH/N rows each, remainder to tracer 0)
    -> every tracer Trace()s its row block -> the primary MergeOutput()s every block
    -> primary SyncFramebuffer() (tone-map)

One process per GPU: rank r owns row block r; the single exchange step of the path -- the primary adds
every block's rows of the trace accumulators into its frame accumulator (renderer/default.go:191,
tracer/opencl/resources.go:108-124) -- is a PEER READ, as in the reference (whose devices share one
OpenCL context, renderer/default.go:225-229): every rank publishes HIP-IPC handles of its trace
accumulator ring once, rank 0 maps them and its merge stream runs the aggregate kernel straight over the
peer-mapped rows (polaris_hip_merge_ipc; xGMI).  No RCCL on the data path ("independent tiles, so no
RCCL"): control only travels otherwise -- the handles at set-up over torch.distributed (gloo), 32 bytes per rank and
frame through a shared-memory mailbox (ranks of one host; a gloo all_gather where they cannot map one).  If a mapping cannot be opened the run falls back, in the same processes, to point-to-point strip
transfers (`--exchange strips`: backend nccl = RCCL); `config.exchange` names which ran.  The block
scheduler for N > 1 is `naive` (what `polaris render` passes, cmd/render.go:65); the same run then times
the perfect scheduler (tracer/scheduler.go:50-80) as a second region -> `config.perfect_scheduler`.
The frame is fixed, so scaling is STRONG.  Scene and seeds are synthetic (polaris_amd/scenes.py); inputs
are resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `value` = rays traced by all ranks in the K timed frames / wall
time (max over ranks); rays = primary + indirect + occlusion rays handed to an intersection kernel
(BASELINE.md section 3).  `roofline` prices ONE kernel symbol -- the one with the largest isolated
time -- against HBM: algorithmic bytes (SURVEY.md 8d split per kernel) / its HIP-event
time; `roofline_per_kernel` holds the same object for every symbol that moves ray streams.
`roofline.traffic`, `lane_util` and `roofline_issue` come from PMC counters collected LIVE at the end of the
run (three rocprofv3 child runs of `bench.py --steps 1` on the same workload: FETCH_SIZE, WRITE_SIZE, one SQ pass;
~10 s; `--no-live-counters` or a failing profiler falls back to the newest committed profile, and
`counters_source` says which).  `cpu_baseline` times the CPU restatement under oracle/ (the checker, never the
product) on a bounded sample of the same workload on the host cores.

`--inproc` runs the same frame the way the reference itself does: ONE process, one worker thread per
GPU (polaris_amd/host/renderer.cpp), blocks merged into the primary by polaris_hip_merge over xGMI
peer access, `--scheduler naive|perfect` (tracer/scheduler.go) -- no torch.distributed, no collective.
The driver's launcher (one process per GPU under torch.distributed.run) always gets the default mode.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this, hipIpcGetMemHandle fails across processes.  The images export it already;
# a launcher with a scrubbed environment must not cost the peer-read exchange (read by the runtime when it initialises, i.e. after this line)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

SHADE_TIMERS = ("shade_first", "shade_sort", "shade_plain", "shade_wave")  # one library timer per shade kernel symbol
KERNELS = ("generate", "intersect_packet", "intersect", *SHADE_TIMERS, "scan", "occlusion", "fold", "resolve", "aggregate", "tonemap")
PRICED = ("generate", "intersect_packet", "intersect", *SHADE_TIMERS, "occlusion", "fold")  # the kernels that move ray streams
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def kernel_algorithmic_bytes(st, shade_counts, B: int, packet_camera: bool):
    """Two prices per library timer (= kernel symbol), in bytes per frame:
    `reference` -- SURVEY.md 8d's per-unit stream bytes of the REFERENCE's formulation, attributed to the kernel that moves them;
    `layout`    -- what this design's own ray streams make the kernel move per unit (float4 streams): where they
                   are leaner than the reference's (no origin / throughput for camera rays, hit flag and record in one word ...)
                   pricing a kernel with the reference's bytes would credit it with traffic it never causes.
    A kernel's `frac` uses the SMALLER of the two; both are carried.  st = PolarisTraceStats of one frame, shade_counts =
    HipTracer.shade_counts(B) of the same frame."""
    prim, rays, occl = int(st.primary_rays), [int(v) for v in st.rays_per_bounce], [int(v) for v in st.occl_per_bounce]
    cam_per_ray = 0 if packet_camera else prim       # camera rays traced by k_trace: their common origin is a kernel-side constant
    ref = {
        # camera rays: SURVEY's 52 B minus what is constant for a camera ray and therefore not stored (origin | max distance,
        # throughput) = 20 B -- plus the 16 B per path slot of the batch's per-path radiance, which k_generate zeroes
        "generate": (20 + 16) * prim,
        "intersect_packet": 44 * prim if packet_camera else 0,
        "intersect": 60 * int(st.indirect_rays) + 44 * cam_per_ray,   # (a camera ray: SURVEY's 60 B minus the 16 B origin nobody stores)
        "occlusion": 36 * int(st.occlusion_rays),            # rayIntersectionTest: 32 B ray in, 4 B flag out
        "fold": 44 * int(st.unoccluded),                     # accumulateEmissiveSamples: 4 + 4 + 12 B in, 24 B read-modify-write
    }
    lay = {
        "generate": (16 + 16) * prim,                                      # ray_d + lsum stores
        "intersect_packet": (16 + 12) * prim if packet_camera else 0,      # ray_d load, hit store (12 B: u, v, triangle word -- no t inside a Trace)
        "intersect": (12 + 16 + 12) * int(st.indirect_rays) + (16 + 12) * cam_per_ray,  # ray_o (12 B: no max distance inside a Trace) + ray_d loads, hit store (camera rays: no ray_o)
        "occlusion": 32 * int(st.occlusion_rays) + 4 * int(st.unoccluded),   # occ_o + occ_d loads; an unoccluded ray marks its NEE record (4 B)
        "fold": 16 * int(st.occlusion_rays) + 32 * prim,                     # every NEE record once + the per-path cells read and written once per batch
    }
    for name in SHADE_TIMERS:
        ref[name] = lay[name] = 0
    for b in range(B):
        c = shade_counts[b]
        emitted = rays[b + 1] if b + 1 < B else 0
        # reference: 68 B per shaded hit, 60 per shaded miss, 24 per emitter hit, 32 per emitted bounce ray, 44 per shadow ray
        ref[c["timer"]] += 68 * c["hits"] + 60 * c["misses"] + 24 * c["emitters"] + 32 * emitted + 44 * occl[b]
        # layout: every shaded ray loads ray_d + its 12-byte hit record (+ thr after the first bounce); an emitted bounce ray is 12 + 16 + 16 B of stores
        # (ray_o, ray_d, thr), a shadow ray three (occ_o, occ_d, occ_e); a miss / emitter hit is a 16 B load + 12 B store of its cell
        lay[c["timer"]] += (28 if b == 0 else 44) * (c["hits"] + c["misses"]) + 28 * (c["misses"] + c["emitters"]) + 44 * emitted + 48 * occl[b]
    return ref, lay


def committed_counters(symbol: str, workload_is_headline: bool):
    """The FALLBACK for live_counters() below (which collects the PMC counters of this build on this machine with rocprofv3 child runs
    at the end of every N = 1 run): the newest committed profile of the headline workload (scripts/traffic.sh, scripts/pmc.sh), keyed by
    kernel symbol -- used when the profiler is missing, a pass fails, or --no-live-counters is given.  Returns (hbm bytes per launch,
    lane utilisation, wave-level VALU instructions per launch, description of the source) -- None where there is no committed
    number for this symbol."""
    if not workload_is_headline:
        return None, None, None, None
    import glob

    traffic = lane_util = valu = None
    src = []
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        ent = json.load(open(tpath)).get("kernels", {}).get(symbol)
        if ent and ent.get("launches"):
            traffic = (ent["hbm_read_bytes"] + ent["hbm_write_bytes"]) / ent["launches"]
            src.append("profiles/" + os.path.basename(tpath) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1`)")
            break
    for spath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")), reverse=True):
        ent = json.load(open(spath)).get("kernels", {}).get(symbol)
        if ent and "lane_util" in ent:
            lane_util = ent["lane_util"]
            if ent.get("launches") and ent.get("SQ_INSTS_VALU"):
                valu = ent["SQ_INSTS_VALU"] / ent["launches"]
            src.append("profiles/" + os.path.basename(spath) + " (SQ_INSTS_VALU; SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU)")
            break
    return traffic, lane_util, valu, ("; ".join(src) + "; collected on the builder's MI355X lease, not re-measured in this run") if src else None


def live_counters(argv, timeout_s: float = 150.0, total_timeout_s: float = 240.0):
    """PMC counters of THIS build on THIS machine, collected at the end of the run: three rocprofv3 passes (FETCH_SIZE, WRITE_SIZE, one
    SQ pass -- separate passes with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes and the pool
    requires) over `python3 bench.py --steps 1 --warmup 0` of the same workload, each a CHILD process.  Returns {kernel symbol:
    {"traffic": HBM bytes per launch (FETCH_SIZE is in KiB and counts 64 B per 128-B request of a wide stream on gfx950: read side
    doubled), "lane_util": .., "valu": wave-level VALU instructions per launch}} or None if any pass fails (the committed profiles
    are used then).  Never raises: the bench line must not depend on the profiler.  The program after `--` is THIS interpreter's
    resolved binary (os.path.realpath(sys.executable)), never a name looked up on PATH: a shim or wrapper there would be an exec hop
    behind the profiler's preloaded library, which has already initialised the GPU (the pool forbids that), and a venv's `python3`
    may be another Python altogether.  timeout_s bounds one pass, total_timeout_s all three."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    passes = (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    python = os.path.realpath(sys.executable)
    t_start = time.perf_counter()
    try:
        for ctrs in passes:
            left = total_timeout_s - (time.perf_counter() - t_start)
            if left < 5.0:
                return None, f"the counter passes ran out of their {total_timeout_s:.0f} s budget before {ctrs[0]}"
            d = tempfile.mkdtemp(prefix="polaris_pmc_", dir="/tmp")
            try:
                # (the workload's own arguments first, the overrides LAST: argparse keeps the last occurrence, whatever form -- `--steps N` or
                # `--steps=N` -- the caller used)
                cmd = [exe, "--kernel-trace", "--pmc", *ctrs, "--output-format", "csv", "-d", d, "--", python, os.path.abspath(__file__),
                       *argv, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-kernel-timers", "--no-live-counters"]
                p = subprocess.run(cmd, capture_output=True, text=True, timeout=min(timeout_s, left), env=env, cwd="/tmp")
                found = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
                if p.returncode != 0 or not found:
                    return None, f"rocprofv3 pass {ctrs[0]} failed (exit {p.returncode}): {(p.stderr or '')[-200:]}"
                for r in csv.DictReader(open(found[0])):
                    sym = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    agg[sym][r["Counter_Name"]] += float(r["Counter_Value"])
                    if r["Counter_Name"] == ctrs[0]:
                        launches[(sym, ctrs[0])].add(r["Dispatch_Id"])
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception as e:  # a timeout, a missing tool, an unreadable csv: fall back to the committed profiles
        return None, f"{type(e).__name__}: {e}"
    out = {}
    for sym, c in agg.items():
        n = len(launches.get((sym, "FETCH_SIZE"), ())) or len(launches.get((sym, "SQ_INSTS_VALU"), ()))
        if "pol::" not in sym or not n:
            continue
        out[sym] = {"traffic": (c.get("FETCH_SIZE", 0.0) * 1024 * 2 + c.get("WRITE_SIZE", 0.0) * 1024) / n,
                    "lane_util": (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"])) if c.get("SQ_ACTIVE_INST_VALU") else None,
                    "valu": (c["SQ_INSTS_VALU"] / n) if c.get("SQ_INSTS_VALU") else None}
    return out, "rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE; SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU) of `bench.py --steps 1` on the same workload, run as child processes at the end of THIS run"


# The roof these kernels are actually under: vector-instruction ISSUE.  tests/tools/valu_rate.hip measures 0.86-0.93 ns
# per wave64 VALU instruction per SIMD with 2-8 waves resident (plain v_mul_f32 / v_fma_f32; a packed f32 instruction takes two
# slots): the chip issues at most CUs x 4 SIMDs / 0.9 ns wave-level vector instructions per second.
VALU_NS_PER_INST_PER_SIMD = 0.9


def issue_roofline(valu_per_launch, launches, ms_per_frame, lane_util, cus: int):
    """`roofline_issue` of one kernel symbol: wave-level VALU instructions per second (committed SQ_INSTS_VALU per launch x this
    run's launches / this run's HIP-event time) against the chip's issue rate, and the same weighted by the live lanes per
    instruction -- the fraction of the vector ALUs' lane-slots doing useful work."""
    if not valu_per_launch or not launches or ms_per_frame <= 0:
        return None
    peak = cus * 4 / (VALU_NS_PER_INST_PER_SIMD * 1e-9) / 1e9     # G wave-instructions / s
    ach = valu_per_launch * launches / (ms_per_frame * 1e-3) / 1e9
    return {"bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave64 VALU instructions/s", "frac": ach / peak,
            "frac_useful_lanes": (ach / peak * lane_util) if lane_util else None, "valu_instructions_per_launch": valu_per_launch,
            "peak_from": f"{cus} CUs x 4 SIMDs / {VALU_NS_PER_INST_PER_SIMD} ns per wave64 VALU instruction per SIMD (tests/tools/valu_rate.hip: 0.86-0.93 ns with 2-8 waves resident)",
            "instructions_from": "SQ_INSTS_VALU of the same kernel symbol on the same workload (see counters_source); time: this run's HIP events"}


def ray_counts(st, B: int):
    """PolarisTraceStats of one Trace as a dict: rays per bounce by class, the shading events, and the occluded fraction of the shadow rays."""
    if st is None:
        return None
    occl, un = int(st.occlusion_rays), int(st.unoccluded)
    return {"closest_hit_rays_per_bounce": [int(v) for v in st.rays_per_bounce[:B]], "shadow_rays_per_bounce": [int(v) for v in st.occl_per_bounce[:B]],
            "primary_rays": int(st.primary_rays), "indirect_rays": int(st.indirect_rays), "shadow_rays": occl, "unoccluded_shadow_rays": un,
            "occluded_fraction": (1.0 - un / occl) if occl else None, "shaded_hits": int(st.shaded_hits), "shaded_misses": int(st.shaded_misses),
            "emitter_hits": int(st.emitter_hits)}


def host_cpu() -> str:
    """'<n> x <model name>' of the host the CPU baseline runs on (state the cores)."""
    try:
        models = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        return f"{len(models)} x {models[0]}" if models else f"{os.cpu_count()} logical CPUs"
    except OSError:
        return f"{os.cpu_count()} logical CPUs"


def run_inproc(args) -> None:
    """`--inproc`: the frame loop of renderer/default.go:106-196 as the reference runs it -- one process, one worker thread per
    tracer (polaris_amd/host/renderer.cpp), every tracer Trace()s the rows the scheduler hands it, the workers MergeOutput
    into the primary (tracer 0) through polaris_hip_merge -- xGMI peer access when the block lives on another GPU -- and the
    primary tone-maps.  No collective anywhere ("independent tiles, so no RCCL").  A step = one renderFrame()."""
    from polaris_amd import ctypes_api as T
    from polaris_amd import host_api, scenes

    lib = T.load_library()
    visible = lib.polaris_hip_device_count()
    if visible < 1:
        raise SystemExit("bench.py needs a GPU: the tracer has no CPU fallback")
    devices = [int(v) for v in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit(f"bench.py --inproc --gpus {args.gpus}: --devices names {len(devices)} tracer(s)")
    if max(devices) >= visible or min(devices) < 0:
        raise SystemExit(f"bench.py --inproc --gpus {args.gpus}: only {visible} HIP device(s) visible on this host")
    W, H, spp, B = args.width, args.height, args.spp, args.bounces
    sc = scenes.SCENES[args.scene](W / H)
    identities = [{"tracer": t, **T.device_identity(d)} for t, d in enumerate(devices)]
    peer_row = {str(d): (T.can_access_peer(devices[0], d) if d != devices[0] else None) for d in sorted(set(devices))}
    r = host_api.Renderer(sc, devices, primary=0, scheduler=host_api.PERFECT if args.scheduler == "perfect" else host_api.NAIVE, width=W, height=H,
                          spp=spp, bounces=B, min_rr=args.rr, exposure=1.2, seed=1)
    try:
        if args.samples_per_batch:
            r.set_option("samples_per_batch", args.samples_per_batch)
        for kv in args.opt:
            k, v = kv.split("=")
            r.set_option(k, int(v))
        rows = None
        for _ in range(args.warmup):               # (the perfect scheduler settles on its rows here: it feeds on the last frame's times)
            rows, _ = r.render(0)
        rays = 0
        all_rows = []
        t0 = time.perf_counter()
        for f in range(args.steps):
            if args.test_seeds:
                for t in range(len(devices)):
                    r.push_seeds(t, scenes.make_seeds(spp, B, base=1000 * f + 17 * t))
            rows, _ = r.render(0)                  # synchronous: every Trace, every merge and the primary's tone-map are done
            all_rows.append(rows)
            rays += sum(r.tracer_stats(i)[0].total_rays() for i in range(len(devices)))
        elapsed = time.perf_counter() - t0
        trace_ms = [round(r.tracer_stats(i)[1], 3) for i in range(len(devices))]
        merge_counts = r.merge_counts()
        if args.save_accumulator:
            np.save(args.save_accumulator, r.read()[1])
        if args.save_png:
            r.save(args.save_png)
    finally:
        r.close()
    ms = elapsed / args.steps * 1e3
    print(json.dumps({
        "metric": "Mrays/s (primary+indirect+occlusion rays traced / wall), Cornell box 512x512x128spp",
        "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "ms_per_frame": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{sc.name} {W}x{H} {spp}spp {B} bounces rr>={args.rr}, {sc.num_triangles} tris, row blocks {rows} of the last frame "
                               f"({args.scheduler} scheduler), merge into the primary + tonemap",
                   "mode": "in-process: one worker thread per tracer, polaris_hip_merge (peer access), no collective", "tracers": len(devices), "device_indices": devices,
                   "devices": identities, "distinct_gpus": len({(d["pci_bus_id"], d["uuid"]) for d in identities}),
                   "exchange_detail": {"mode": "inproc", "primary_device": devices[0], "can_access_peer_from_primary": peer_row, "merge_counts": merge_counts},
                   "scheduler": args.scheduler, "rows_first_timed_frame": all_rows[0] if all_rows else None, "rows_last_frame": rows,
                   "trace_ms_last_frame_per_tracer": trace_ms, "frame": [W, H], "spp": spp, "bounces": B, "rays_per_frame": rays // max(args.steps, 1),
                   "paths_per_s": W * H * spp * args.steps / elapsed},
        "roofline": None, "cpu_baseline": None,
        "note": "roofline / cpu_baseline are measured by the default (one process per GPU) mode at N = 1; this mode times the reference's in-process frame loop"}))


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
    ap.add_argument("--bvh", default="scene", choices=("scene", "device", "device-lbvh"), help="`device`: the scene's two-level BVH is rebuilt on the GPU before the upload "
                    "(polaris_hip_build_bvh: binned SAH, level by level; `device-lbvh`: its linear-BVH algorithm -- an alternative producer, not the headline's tree)")
    ap.add_argument("--bvh-max-leaf", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true", help="do not collect the PMC counters (HBM traffic, live lanes, VALU instructions) with rocprofv3 child runs "
                    "at the end (N = 1 only; ~25 s); `roofline.traffic` then comes from the newest committed profile of the headline workload")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--save-accumulator", default="", help="rank 0 writes its frame accumulator of the last frame as .npy (tests)")
    ap.add_argument("--opt", action="append", default=[], help="tracer option key=value (polaris_hip_set_option)")
    ap.add_argument("--exchange", default="ipc", choices=("ipc", "strips"), help="N > 1: how the blocks reach the primary -- `ipc`: peer reads of every rank's "
                    "trace accumulator through HIP IPC mappings (polaris_hip_merge_ipc; falls back to `strips` if a mapping cannot be opened), "
                    "`strips`: point-to-point transfers of the strips on --backend")
    ap.add_argument("--backend", default="nccl", help="--exchange strips: torch.distributed backend of the strip transfers (nccl = RCCL; gloo only to test that flow on one GPU)")
    ap.add_argument("--test-ipc-failure", nargs="?", const="open", default="", choices=("open", "export"), help="testing aid: `open` (the default when the flag is "
                    "given bare): rank 0 refuses to map the peers' rings, as if hipIpcOpenMemHandle had failed; `export`: the last rank's export fails, as if "
                    "hipIpcGetMemHandle had -- either way the run must fall back to the strip transfers inside the same processes, on every rank")
    ap.add_argument("--test-delay-rank", default="", help="testing aid, R:MS -- rank R sleeps MS milliseconds after every Trace, so that the others run as far "
                    "ahead as the exchange protocol lets them (a ring slot reused too early then shows in the assembled frame)")
    ap.add_argument("--test-random-delays", type=int, default=0, help="testing aid, SEED: every rank sleeps a seeded random time (0 - 3 ms, most of them short) after every "
                    "Trace of every frame, so that who runs ahead and who lags changes from frame to frame (the randomised schedule of "
                    "tests/test_distributed_cpu.py, here through the real IPC mappings)")
    ap.add_argument("--no-second-scheduler", action="store_true", help="N > 1: skip the second timed region (the perfect scheduler when --scheduler naive)")
    ap.add_argument("--same-device", action="store_true", help="testing aid: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--control", default="shm", choices=("shm", "gloo"), help="N > 1, --exchange ipc: the per-frame control message (32 bytes per rank) through a "
                    "shared-memory mailbox (ranks of one host; falls back to gloo where it cannot be mapped) or always as a gloo all_gather")
    ap.add_argument("--control-timeout", type=float, default=120.0, help="N > 1: timeout in seconds of the gloo control group (set-up exchange, 32 bytes per rank "
                    "and frame): a rank that dies surfaces as an error on the others within this time")
    ap.add_argument("--test-setup-failure", type=int, default=-1, help="testing aid: rank R fails during set-up (as a tracer that cannot be created would); every "
                    "rank must learn of it and exit non-zero before any collective can hang")
    ap.add_argument("--emulate-rank", default="", help="R/N: on ONE GPU, trace only the row block rank R of N would own "
                    "(tuning aid for the strong-scaling path; not a valid bench line)")
    ap.add_argument("--rows", default="", help="with --emulate-rank: the N block heights to use instead of the naive scheduler's (comma separated)")
    ap.add_argument("--inproc", action="store_true", help="ONE process, one worker thread per GPU, blocks merged into the primary over peer access "
                    "(polaris_hip_merge): the reference renderer's own model (renderer/default.go:106-196); no torch.distributed, no RCCL")
    ap.add_argument("--devices", default="", help="--inproc: comma separated device index per tracer (default 0..N-1; repeating a device, "
                    "e.g. 0,0,0, runs several tracers on one GPU: a testing aid)")
    ap.add_argument("--scheduler", default="naive", choices=("naive", "perfect"), help="block scheduler for N > 1 (tracer/scheduler.go): `naive` = equal "
                    "rows (what `polaris render` passes, cmd/render.go:65: the default), `perfect` = rows from the previous frames' times (scheduler.go:50-80)")
    ap.add_argument("--test-seeds", action="store_true", help="testing aid: fixed host PRNG draws -- --inproc: tracer t of frame f draws from "
                    "make_seeds(base = 1000 f + 17 t); default mode: frame f uses make_seeds(base = 0xC0FFEE + f) instead of one list for every frame")
    args = ap.parse_args()
    if args.inproc:
        return run_inproc(args)

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
        # (--standalone: torchrun picks a free rendezvous port itself -- no bind / close / reuse race on a busy box)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE line on the job's stdout, the JSON line of rank 0: whatever a library prints there (gloo announces every rank's connections on
    # stdout, from every rank) goes to stderr instead -- file descriptor 1 is pointed at 2 before anything is loaded, and the line is
    # written to a duplicate of the original descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: they must agree")

    import torch

    dist = None
    if world > 1:
        import datetime

        import torch.distributed as dist

        # CONTROL plane only (handles and identities at set-up, 32 bytes per rank and frame, the final reductions).  A BOUNDED timeout: a
        # rank that dies must surface as an error on the others within two minutes, not after the default thirty.
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=args.control_timeout))
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the tracer has no CPU fallback")

    from polaris_amd import ctypes_api as T
    from polaris_amd import scenes
    from polaris_amd.distributed import SchedulerFeedback, StripExchange, block_of, device_for_rank, gather_setup_errors, naive_rows

    xdev = strip_group = None
    from polaris_amd.tracer import ChangeType, HipTracer, UpdateMode

    W, H, spp, B = args.width, args.height, args.spp, args.bounces
    # ---- set-up: this rank's device, scene and tracer.  One process per GPU, one tracer per device -- the reference enumerates the
    # devices inside ONE process and gives each a tracer (renderer/default.go:204-256).  Nothing in here is a collective; whatever fails
    # is carried to EVERY rank by the gather below, before the first collective that would otherwise wait for the failed rank.
    tr = sc = bvh_info = identity = None
    dev_index, dev_why, setup_err = 0, "", ""
    try:
        visible = torch.cuda.device_count()
        # a launcher may hand every rank all GPUs (LOCAL_RANK picks) or mask each rank to one (index 0 everywhere): device_for_rank
        dev_index, dev_why = (0, "--same-device (testing aid)") if args.same_device else device_for_rank(local_rank, visible)
        torch.cuda.set_device(dev_index)
        if not torch.cuda.is_available():
            raise RuntimeError("no usable HIP device: the tracer has no CPU fallback")
        if args.test_setup_failure == rank:
            raise RuntimeError("injected set-up failure (--test-setup-failure)")
        sc = scenes.SCENES[args.scene](W / H)
        if args.bvh != "scene":
            from polaris_amd import bvh_build

            t_b = time.perf_counter()
            sc, bvh_info = bvh_build.rebuild_on_device(sc, max_leaf_tris=args.bvh_max_leaf, device=dev_index, algorithm="lbvh" if args.bvh == "device-lbvh" else "sah")
            bvh_info["wall_ms_with_permutation"] = (time.perf_counter() - t_b) * 1e3
        tr = HipTracer(f"hip-{rank}", dev_index)
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
        # which physical GPU this rank traces on: device indices are per process, the PCI bus id and the UUID are not
        identity = {"rank": rank, "local_rank": local_rank, "pid": os.getpid(), **tr.device_identity(), "visible_devices": visible, "device_chosen_by": dev_why,
                    "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES")}
    except Exception as e:  # noqa: BLE001 -- reported below, on every rank
        setup_err = f"{type(e).__name__}: {e}"
    if world > 1:
        errs = gather_setup_errors(dist, rank, world, setup_err)
        if errs:
            if rank == 0:
                print("bench.py: set-up failed, no frame was traced -- " + "; ".join(errs), file=sys.stderr)
            if tr is not None:
                tr.Close()
            dist.destroy_process_group()
            raise SystemExit(1)
        identities = [None] * world
        dist.all_gather_object(identities, identity)
    else:
        if setup_err:
            raise SystemExit(f"bench.py: set-up failed: {setup_err}")
        identities = [identity]
    distinct_gpus = len({(d["pci_bus_id"], d["uuid"]) for d in identities})
    dev = torch.device("cuda", dev_index)
    seeds = scenes.make_seeds(spp, B)
    rows = naive_rows(world, H)                      # tracer/scheduler.go:83-106, equal speeds
    block_y, block_h = block_of(rank, rows)
    if args.emulate_rank:
        er, en = (int(v) for v in args.emulate_rank.split("/"))
        erows = [int(v) for v in args.rows.split(",")] if args.rows else naive_rows(en, H)
        if len(erows) != en or sum(erows) != H or min(erows) < 1:
            raise SystemExit(f"bench.py --rows: need {en} positive block heights that add up to {H}")
        block_y, block_h = block_of(er, erows)
        rows = [block_h]

    def make_req(by, bh):
        r = T.BlockRequest()
        r.frame_w, r.frame_h, r.block_x, r.block_y, r.block_w, r.block_h = W, H, 0, by, W, bh
        r.samples_per_pixel, r.num_bounces, r.min_bounces_for_rr = spp, B, args.rr
        r.exposure, r.seed, r.accumulated_samples = 1.2, 0, 0
        return r

    # The path's one exchange step (renderer/default.go:191): the primary adds every block's rows into its frame accumulator.
    # It runs one frame behind the tracing (polaris_amd/distributed.py): frame i's merge / tone-map overlap frame i+1's Trace.
    # The block scheduler runs identically on every rank from all-gathered (rows, time) pairs, also one frame behind.
    px = ex = fb = None
    exchange = "none (1 GPU)"
    # config.exchange_detail: what the exchange REALLY was -- per peer the branch its block reaches the primary by, what rank 0 sees of the
    # other GPUs, and (at the end of the run) how many merges took which branch inside the library.  A fallback shows here, not in a timeout.
    detail = {"mode": "none (1 GPU)", "peers": []}
    if world > 1:
        if rank == 0:   # the hipDeviceCanAccessPeer row of the primary's device over the devices THIS process sees (one entry under a per-rank mask)
            detail["primary_device"] = dev_index
            detail["can_access_peer_from_primary"] = {str(j): (T.can_access_peer(dev_index, j) if j != dev_index else None) for j in range(torch.cuda.device_count())}
        from polaris_amd.distributed import HipPort, PeerExchange

        if args.exchange == "ipc":
            port = HipPort(tr, make_req)
            if args.test_ipc_failure == "open":
                def refuse(blob):
                    raise RuntimeError("hipIpcOpenMemHandle: refused (--test-ipc-failure)")
                port.open = refuse
            elif args.test_ipc_failure == "export" and rank == world - 1:
                def refuse_export(depth):
                    raise RuntimeError("hipIpcGetMemHandle: refused (--test-ipc-failure export)")
                port.export = refuse_export
            px = PeerExchange(dist, rank, world, W, H, port, scheduler=args.scheduler, control=args.control, control_timeout_s=args.control_timeout)
            if px.setup():
                exchange = (f"hip-ipc: rank 0's merge stream reads every rank's rows through an IPC mapping of its trace accumulator ring "
                            f"(depth {px.depth}), one frame behind the tracing; control = 32 B per rank and frame "
                            f"{'through a shared-memory mailbox' if px.control == 'shm' else 'over gloo'}; no RCCL")
                detail["mode"] = "hip-ipc"
                detail["control"] = px.control      # the 32 bytes per rank and frame: "shm" (a mailbox in /dev/shm: ranks of one host) or "gloo"
                if rank == 0:   # what every mapping really is: the peer's GPU by bus id, local or across xGMI (polaris_hip_peer_info)
                    detail["peers"] = [{"rank": r, **tr.peer_info(p)} for r, p in sorted(px.peers().items())]
            else:
                if rank == 0:
                    print(f"bench.py: HIP IPC mapping failed ({px.why_not}); falling back to strip transfers on {args.backend}", file=sys.stderr)
                exchange = f"fallback after a failed IPC mapping ({px.why_not[:120]}): "
                px = None
        if px is None:
            backend = args.backend
            if backend == "nccl" and distinct_gpus < world:
                # decided over gloo, from the gathered identities, BEFORE any communicator exists: RCCL needs one GPU per rank
                if rank == 0:
                    print(f"bench.py: {world} ranks on {distinct_gpus} distinct GPU(s): RCCL needs one device per rank; the strips travel over gloo through the host", file=sys.stderr)
                detail["rccl"] = f"not tried: {world} ranks share {distinct_gpus} GPU(s)"
                backend = "gloo"
            if backend == "nccl":                      # (RCCL communicators only exist on this path)
                # the last resort must not be able to fail: if RCCL cannot be brought up on EVERY rank (one warm-up all_reduce
                # each, the verdicts gathered over gloo), the strips travel over gloo, staged through the host.  The probe has to RAISE
                # on this rank, not end the process: torch's watchdog would abort on a timed-out collective unless asynchronous error
                # handling is off, and Work.wait(timeout) only blocks (and throws) with blocking wait on -- both are read when the group
                # is created.
                err = ""
                os.environ["TORCH_NCCL_ASYNC_ERROR_HANDLING"] = "0"
                os.environ["TORCH_NCCL_BLOCKING_WAIT"] = "1"
                try:
                    strip_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))  # (a rank whose RCCL bring-up fails must not leave the others waiting for ten minutes)
                    probe = torch.ones(1, device=dev)
                    work = dist.all_reduce(probe, group=strip_group, async_op=True)
                    work.wait(timeout=datetime.timedelta(seconds=60))
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        err = f"rank {rank}: RCCL warm-up all_reduce returned {probe.item()}, expected {world}"
                except Exception as e:  # noqa: BLE001
                    err = f"rank {rank}: {type(e).__name__}: {e}"
                errs = [None] * world
                dist.all_gather_object(errs, err)
                if any(errs):
                    if rank == 0:
                        print(f"bench.py: RCCL is not usable for the strips ({next(e for e in errs if e)[:200]}); using gloo through the host", file=sys.stderr)
                    detail["rccl"] = "bring-up failed: " + next(e for e in errs if e)[:200]
                    if strip_group is not None:     # do not keep a half-initialised communicator around: destroy_process_group at exit may block on it
                        try:
                            dist.destroy_process_group(strip_group)
                        except Exception:  # noqa: BLE001
                            pass
                    backend, strip_group = "gloo", None
                else:
                    detail["rccl"] = "warm-up all_reduce succeeded on every rank"
            xdev = dev if backend == "nccl" else torch.device("cpu")
            ex = StripExchange(dist, rank, world, W, H, dev, via_host=backend != "nccl", group=strip_group)
            fb = SchedulerFeedback(dist, rank, world, H, xdev, kind=args.scheduler, group=strip_group)
            exchange = (exchange if args.exchange == "ipc" else "") + f"{backend} point-to-point transfers of the row-block strips to rank 0, one frame behind the tracing"
            detail["mode"] = "strips-rccl" if backend == "nccl" else "strips-gloo"
            detail["peers"] = [{"rank": r, "branch": detail["mode"]} for r in range(1, world)]
    totals = {k: 0 for k in ("primary_rays", "indirect_rays", "occlusion_rays", "shaded_hits", "shaded_misses", "emitter_hits", "unoccluded")}
    last_counts = [None]                           # PolarisTraceStats of this rank's last timed Trace (config.ray_counts)
    pending = []
    rows_log = []

    own_work = [0.0]                               # seconds of this rank's OWN work in the current frame (what the scheduler balances)
    trace_s = [0.0]                                # seconds this rank spent inside Trace over the timed frames (config.frame_loop)

    def finish_frame(ticket):
        if px is not None:
            # rank 0: Reset stage, one peer-read merge per block, tone-map (default.go:159-161).  Only those seconds are this
            # rank's work; the wait for the slowest rank's message inside finish() is not (as with ex.wait() below): billed to
            # the primary it would make the perfect scheduler shrink rank 0's block frame after frame
            own_work[0] += px.finish(ticket)
            return
        parts = ex.wait(ticket)                    # (waiting for the others' strips is not this rank's work)
        if rank == 0:
            t = time.perf_counter()
            tr.reset_frame()                       # the Reset stage: this frame's blocks land on a cleared accumulator
            for y, h, buf in parts:                # ONE merge over the assembled frame
                tr.merge_device(buf.data_ptr(), make_req(y, h))
            tr.SyncFramebuffer(make_req(0, H))     # default.go:159-161
            own_work[0] += time.perf_counter() - t

    def flush():
        while pending:
            finish_frame(pending.pop(0))

    frame_no = [0]
    delay_rank, delay_s = -1, 0.0
    if args.test_delay_rank:
        dr, dms = args.test_delay_rank.split(":")
        delay_rank, delay_s = int(dr), float(dms) * 1e-3

    jitter = None
    if args.test_random_delays:
        import random

        jitter = random.Random(args.test_random_delays * 7919 + rank)

    def frame(count: bool):
        nonlocal rows, block_y, block_h
        if world > 1 and not args.emulate_rank:
            rows = px.next_rows() if px is not None else fb.next_rows()   # Schedule (default.go:124): the same rows on every rank
            block_y, block_h = block_of(rank, rows)
        req = make_req(block_y, block_h)
        fseeds = seeds
        if args.test_seeds:                        # a different frame every step: a strip merged one frame late or early shows
            fseeds = scenes.make_seeds(spp, B, base=0xC0FFEE + frame_no[0])
            frame_no[0] += 1
        t_own = time.perf_counter()
        tr.Trace(req, fseeds)                      # Trace (tracer.go:194-247)
        if count:
            trace_s[0] += time.perf_counter() - t_own
        if delay_rank == rank:
            time.sleep(delay_s)
        if jitter is not None:
            time.sleep(jitter.choice((0.0, 0.0, 0.0, 0.0002, 0.0005, 0.001, 0.003)))
        if count:
            st = tr.last_trace_stats
            for k in totals:
                totals[k] += int(getattr(st, k))
            rows_log.append(list(rows))
            last_counts[0] = st
        if world == 1:
            tr.MergeOutput(tr, req)                # primary merges its own block (default.go:191)
            tr.SyncFramebuffer(make_req(0, H) if not args.emulate_rank else req)
        elif px is not None:
            own_work[0] = time.perf_counter() - t_own
            flush()                                # the PREVIOUS frame, BEFORE this frame is announced (PeerExchange: why depth 3 is enough)
            pending.append(px.post(rows, own_work[0] * 1e3))
        else:
            ticket = ex.post(lambda strip: tr.export_block(req, strip.data_ptr()), rows)
            own_work[0] = time.perf_counter() - t_own
            flush()                                # the PREVIOUS frame: its transfers ran beside this frame's Trace
            pending.append(ticket)
            # what this rank spent on the frame itself -- its Trace, its strip and, on the primary, assembling and tone-mapping
            # the previous frame -- is what the scheduler balances (the reference feeds Tracer.Stats().RenderTime, scheduler.go:58-62)
            fb.publish(rows, own_work[0] * 1e3)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(steps: int, warmup: int, count: bool):
        for _ in range(warmup):
            frame(False)
        flush()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame(count)
        flush()                                    # all K frames merged and tone-mapped inside the timed region
        fence()
        return time.perf_counter() - t0

    elapsed = timed_region(args.steps, args.warmup, True)
    # rank 0's frame loop: how much of a frame is the synchronous Trace, how much everything around it (merge / exchange / tone-map / Python)
    frame_loop = {"trace_ms_per_frame": trace_s[0] / args.steps * 1e3, "other_ms_per_frame": (elapsed - trace_s[0]) / args.steps * 1e3, "of": "rank 0, first timed region"}
    if fb is not None:
        fb.drain()
    main_rows = list(rows)                         # (the second region below schedules its own)
    if args.save_accumulator and rank == 0:
        np.save(args.save_accumulator, tr.read_accumulator(1))  # the primary's frame accumulator of the last frame (tests)

    # N > 1: the same frames under the OTHER block scheduler, as a second timed region (reported beside the headline, never as it)
    second = None
    if world > 1 and not args.no_second_scheduler and not args.emulate_rank:
        other = "perfect" if args.scheduler == "naive" else "naive"
        if px is not None:
            px.set_scheduler(other)
        else:
            fb = SchedulerFeedback(dist, rank, world, H, xdev, kind=other, group=strip_group)
        keep_totals, keep_rows_log = dict(totals), list(rows_log)
        rows_log.clear()
        el2 = timed_region(args.steps, max(args.warmup, 4), True)   # (the perfect scheduler needs a few frames to settle on its rows)
        if fb is not None:
            fb.drain()
        rays2 = sum(totals[k] - keep_totals[k] for k in ("primary_rays", "indirect_rays", "occlusion_rays"))
        second = (other, el2, rays2, list(rows_log[-1]) if rows_log else None)
        totals.update(keep_totals)
        rows_log[:] = keep_rows_log

    # N > 1: the exchange proves itself on the frame it has just delivered.  Every rank reads the rows of ITS block back from its own trace
    # accumulator (the ring slot its last Trace wrote), rank 0 gathers them over gloo -- a path that shares nothing with the exchange -- and
    # compares them with the same rows of its frame accumulator (Reset + one merge per block: 0 + x, so the floats must be EQUAL, whatever
    # branch carried them: a peer read over xGMI, a staged copy, a strip over RCCL or gloo).  After the timed regions, never inside one.
    if world > 1 and not args.emulate_rank:
        proof = {"frame": "the last frame of the last timed region", "rows": list(rows)}
        try:
            mine = np.ascontiguousarray(tr.read_accumulator(0)[block_y:block_y + block_h])
            got = [None] * world if rank == 0 else None
            dist.gather_object((block_y, block_h, mine), got, dst=0)
            if rank == 0:
                merged = tr.read_accumulator(1)
                equal = [bool(np.array_equal(merged[y:y + h, :, :3], blk[..., :3])) for (y, h, blk) in got]   # (k_aggregate adds .xyz)
                lit = [bool(np.any(blk[..., :3] != 0)) for (_, _, blk) in got]     # (an all-black frame would prove nothing; a single block may well be black)
                proof.update(blocks_equal=equal, blocks_not_black=lit, all_equal=all(equal) and any(lit),
                             covered_rows=sum(h for _, h, _ in got), bytes_compared=int(sum(blk.nbytes for _, _, blk in got)))
                if not proof["all_equal"]:
                    print(f"bench.py: EXCHANGE PROOF FAILED -- the primary's frame does not hold every rank's rows: equal={equal} not_black={lit}", file=sys.stderr)
        except Exception as e:  # noqa: BLE001 -- the proof is evidence, not part of the measurement: report, do not lose the line
            proof["error"] = f"{type(e).__name__}: {e}"[:300]
        detail["proof"] = proof

    el = torch.tensor([elapsed, second[1] if second else 0.0], dtype=torch.float64)       # (CPU tensors: gloo)
    cnt = torch.tensor([totals[k] for k in sorted(totals)] + [second[2] if second else 0], dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(el[0].item())
    tot = {k: int(v) for k, v in zip(sorted(totals), cnt.tolist())}
    if second:
        second = {"scheduler": second[0], "value": int(cnt[-1].item()) / float(el[1].item()) / 1e6, "unit": "Mrays/s", "ms_per_step": float(el[1].item()) / args.steps * 1e3,
                  "steps": args.steps, "rows_last_frame": second[3]}
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
                                   f"row blocks {main_rows} ({'naive' if world == 1 else args.scheduler} scheduler{'' if world == 1 else ', rows of the last frame'}), "
                                   f"strips to the primary + tonemap",
                       "ranks": world, "scheduler": "naive" if world == 1 else args.scheduler,
                       # which physical GPU every rank traced on (PCI bus id + UUID: device indices are per process) and how many DISTINCT ones
                       # took part: n_gpus is the launch's WORLD_SIZE, distinct_gpus is what the hardware says
                       "devices": identities, "distinct_gpus": distinct_gpus,
                       "exchange_detail": {**detail, "merge_counts": tr.merge_counts()}, "frame_loop": frame_loop,
                       "rows_first_timed_frame": rows_log[0] if rows_log else None, "rows_last_frame": rows_log[-1] if rows_log else None,
                       "bvh": "as compiled with the scene" if bvh_info is None else {"built_on_device": bvh_info}, "exchange": exchange, "perfect_scheduler" if (second or {}).get("scheduler") == "perfect" else "second_scheduler": second,
                       "frame": [W, H], "spp": spp, "bounces": B, "rays_per_frame": rays // args.steps,
                       "paths_per_s": W * H * spp * args.steps / elapsed,
                       # the ray population of rank 0's block in the last timed frame, per bounce (PolarisTraceStats): what anyone reasoning about
                       # the any-hit kernel needs first -- how many shadow rays there are and how many of them are blocked
                       "ray_counts": ray_counts(last_counts[0], B)},
        }
        # ---- roofline, per kernel symbol -----------------------------------------------------------
        # Kernel durations are HIP events on the tracer's own streams (the library brackets every launch
        # when time_kernels=1; one timer per kernel symbol).  They are NOT taken inside the timed region (event
        # pairs around ~88 launches per frame, and up to four batches share the GPU there): ONE extra frame is
        # traced afterwards with overlap=1 -- same kernels, same inputs, one batch at a time.  `roofline` is the
        # symbol with the largest isolated time (the top row of `rocprofv3 --kernel-trace --stats` on the same
        # command, profiles/r*_kernel_stats_overlap1.csv); `roofline_per_kernel` holds the same object for every
        # symbol that moves ray streams.
        if not args.no_kernel_timers:
            tr.set_option("overlap", 1)
            tr.set_option("time_kernels", 1)
            for name in KERNELS:
                tr.kernel_ms(name)
            tr.Trace(make_req(block_y, block_h), seeds)   # local to rank 0: no collective in here
            iso = {name: tr.kernel_ms(name) for name in KERNELS}
            symbols = {name: tr.kernel_symbol(name) for name in PRICED}
            fst = tr.last_trace_stats
            alg_ref, alg_lay = kernel_algorithmic_bytes(fst, tr.shade_counts(B), B, packet_camera=iso["intersect_packet"][1] > 0)
            headline = (W, H, spp, B, args.scene, world, args.opt) == (512, 512, 128, 5, "cornell", 1, [])
            per_kernel = {}
            live, live_src = (None, None)
            under_profiler = "rocprof" in (os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD", "")).lower()
            if world == 1 and not args.no_live_counters and not args.emulate_rank and not under_profiler:   # (never a profiler inside a profiler)
                workload = [a for a in sys.argv[1:] if a not in ("--no-cpu-baseline",)]
                # (the child runs repeat the workload's own arguments and append their own --steps 1 --warmup 0: argparse keeps the last
                # occurrence.  What must not be repeated -- output files -- is dropped here in both spellings, `--opt value` and `--opt=value`)
                skip, keep = {"--save-png", "--save-accumulator"}, []
                it = iter(workload)
                for a in it:
                    if a in skip:
                        next(it, None)
                        continue
                    if a.split("=", 1)[0] in skip and "=" in a:
                        continue
                    keep.append(a)
                # a pass traces one frame under the profiler (~6 x slower than plain): bound it by the measured frame time, not by a constant
                live, live_src = live_counters(keep, timeout_s=max(60.0, min(150.0, 40.0 + 60.0 * ms_per_step * 1e-3)))
                if live is None and rank == 0:
                    print(f"bench.py: live PMC counters unavailable ({live_src}); using the committed profiles", file=sys.stderr)
            for k in PRICED:
                kms, kn_ = iso[k]
                if kms <= 0 or not kn_:
                    continue
                # priced with the SMALLER of the reference's stream bytes and this layout's own: a kernel is never
                # credited with bytes its streams do not move (`frac_reference` = the SURVEY price, for comparison across rounds)
                alg = min(alg_ref[k], alg_lay[k])
                ach = alg / (kms * 1e-3) / 1e9
                traffic, lane_util, valu, source = committed_counters(symbols[k], headline)
                if live and symbols[k] in live:   # measured in this run: preferred over the committed numbers
                    e = live[symbols[k]]
                    traffic, lane_util, valu, source = e["traffic"], e["lane_util"], e["valu"], live_src
                per_kernel[symbols[k]] = {"bound": "hbm", "kernel": symbols[k], "timer": k, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": ach / HBM_PEAK_GBS, "priced_by": "layout" if alg_lay[k] < alg_ref[k] else "reference",
                                          "frac_reference": alg_ref[k] / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_layout": alg_lay[k] / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "traffic": traffic, "lane_util": lane_util, "counters_source": source,
                                          "algorithmic_bytes_per_launch": alg / kn_, "algorithmic_bytes_reference": alg_ref[k] / kn_,
                                          "algorithmic_bytes_layout": alg_lay[k] / kn_, "traffic_over_algorithmic": (traffic / (alg / kn_)) if traffic else None,
                                          "avg_launch_ms": kms / kn_, "launches": kn_, "ms_per_frame": kms,
                                          "measured": "HIP events, one extra frame with overlap=1 after the timed region",
                                          "roofline_issue": issue_roofline(valu, kn_, kms, lane_util, tr.device_cus)}
            if per_kernel:
                dom = max(per_kernel.values(), key=lambda e: e["ms_per_frame"])
                out["roofline"] = dict(dom)
                out["roofline_issue"] = dom.get("roofline_issue")   # the roof the dominant kernel is actually under; `roofline` (HBM) stays the headline object
            out["roofline_per_kernel"] = per_kernel
            shade_ms = sum(iso[k][0] for k in SHADE_TIMERS)
            out["kernels_isolated_ms_per_frame"] = {**{k: round(v[0], 3) for k, v in iso.items() if k not in ("aggregate", "tonemap")}, "shade": round(shade_ms, 3)}
            mine = {k: totals[k] // args.steps for k in totals}     # per-frame counters of rank 0
            whole = (112 * mine["primary_rays"] + 68 * mine["shaded_hits"] + 92 * mine["indirect_rays"] + 80 * mine["occlusion_rays"]
                     + 44 * mine["unoccluded"] + 60 * mine["shaded_misses"] + 24 * mine["emitter_hits"] + (48 + 16) * main_rows[0] * W)
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
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()                             # every rank's ring stays mapped until the primary has closed its mappings
    if px is not None:
        px.close()
    if dist is not None:
        dist.barrier()
    tr.Close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
