"""Render a Wavefront scene to a PNG on the MI355X backend -- an example driver, not a CLI clone.

    python -m polaris_amd.render scene.obj --width 512 --height 512 --spp 128 --out frame.png

Flags and defaults are those of `polaris render` (cmd/render.go:17-60, cmd/main.go): the scene goes
through the C++ reader/compiler (polaris_amd/host), the frame through the C++ DefaultRenderer
(renderer/default.go's loop) over HipTracers, one per requested device.  RR is disabled the way
the reference does it (rr-bounces 0 or >= num-bounces -> num-bounces + 1)."""
import argparse
import sys
import time

from . import host_api


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m polaris_amd.render", description=__doc__.split("\n")[0])
    ap.add_argument("scene", help="scene.obj")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--num-bounces", type=int, default=5)
    ap.add_argument("--rr-bounces", type=int, default=3)
    ap.add_argument("--exposure", type=float, default=1.2)
    ap.add_argument("--out", default="frame.png")
    ap.add_argument("--devices", default="0", help="comma separated HIP device indices (a device may repeat)")
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args(argv)

    rr = a.rr_bounces
    if rr == 0 or rr >= a.num_bounces:
        rr = a.num_bounces + 1
    t0 = time.perf_counter()
    sc = host_api.read_scene(a.scene, aspect=a.width / a.height)
    for w in sc.warnings:
        print("warning:", w, file=sys.stderr)
    t1 = time.perf_counter()
    devs = [int(d) for d in a.devices.split(",")]
    r = host_api.Renderer(sc, devs, width=a.width, height=a.height, spp=a.spp, bounces=a.num_bounces, min_rr=rr,
                          exposure=a.exposure, seed=a.seed)
    try:
        rows, ms = r.render()
        r.save(a.out)  # the SaveFrameBuffer post-process stage (pipeline.go:215-235)
    finally:
        r.close()
    print(f"{a.scene}: {sc.vertices.shape[0] // 3} triangles, {len(sc.mesh_instances)} instances, {len(sc.material_nodes)} material nodes; "
          f"compiled in {1e3 * (t1 - t0):.0f} ms; {a.width}x{a.height} @ {a.spp} spp on {len(devs)} tracer(s) rows={rows}: {ms:.1f} ms -> {a.out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
