#!/usr/bin/env python3
"""scripts/emulate_ranks.py [tag] -- on ONE MI355X (inside gpurun): what a rank of an N-GPU frame costs, for the naive scheduler's
equal blocks and for the rows the reference's perfect scheduler (tracer/scheduler.go:50-80, restated in
polaris_amd/host/scheduler.cpp) settles on when it is fed the measured times.  Every rank's block is traced alone
(`bench.py --emulate-rank R/N --rows ...`): an N-GPU frame takes what its slowest rank takes, plus the strip exchange, which
runs one frame behind the tracing.  An ESTIMATE of the scaling curve, written to gpurun_out/<tag>_emulated_ranks.json.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = 512


def rank_ms(r, n, rows):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-kernel-timers",
           "--emulate-rank", f"{r}/{n}", "--rows", ",".join(map(str, rows))]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        raise RuntimeError(p.stderr[-500:])
    return json.loads(line[-1])["ms_per_frame"]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    from polaris_amd import host_api
    from polaris_amd.distributed import naive_rows

    one = rank_ms(0, 1, [H])
    out = {"what": "headline frame (Cornell box 512x512x128spp), one rank's block at a time on one MI355X", "one_gpu_ms": one, "N": {}}
    print("1 GPU", one, flush=True)
    for n in (2, 4, 8):
        rows = naive_rows(n, H)
        sched = host_api.Scheduler(host_api.PERFECT, [1] * n)
        sched.schedule(H)  # its first frame is the naive split (scheduler.go:52-56)
        rounds = []
        for it in range(3):
            ms = [rank_ms(r, n, rows) for r in range(n)]
            rounds.append({"scheduler": "naive" if it == 0 else f"perfect, frame {it + 1}", "rows": rows, "ms_per_rank": [round(v, 3) for v in ms],
                           "slowest_ms": round(max(ms), 3), "speedup_vs_1gpu": round(one / max(ms), 2), "efficiency": round(one / max(ms) / n, 3),
                           "n_times_slowest_over_1gpu": round(n * max(ms) / one, 3)})
            print(n, rounds[-1], flush=True)
            rows = sched.schedule(H, block_h=rows, render_ns=[int(v * 1e6) for v in ms])  # the next frame's rows, from this frame's times
        out["N"][str(n)] = rounds
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_emulated_ranks.json")
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
