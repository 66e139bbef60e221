"""Host layer above the C ABI (C++): the reference's scheduler known answers
(tracer/scheduler_test.go:16-20,48-55) through polaris_amd/host/scheduler.cpp, and the Python
restatement used by bench.py's row partition."""
import pytest


@pytest.fixture(scope="module")
def host(built):
    import subprocess, os
    from conftest import ROOT

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "polaris_amd", "host")])
    from polaris_amd import host_api

    return host_api


@pytest.mark.parametrize("speeds,frame_h,rows", [((1, 2), 10, [4, 6]), ((2, 1), 10, [7, 3]), ((1, 1000), 10, [1, 9])])
def test_naive_scheduler_known_answers(host, speeds, frame_h, rows):
    s = host.Scheduler(host.NAIVE, speeds)
    assert s.schedule(frame_h) == rows
    assert s.schedule(frame_h) == rows          # the naive assignment is computed once and kept (scheduler.go:24-30)
    from polaris_amd.distributed import naive_rows

    assert naive_rows(len(speeds), frame_h, speeds) == rows


def test_perfect_scheduler_known_answers(host):
    """scheduler_test.go:42-80: first call = naive (5,5); then times (1,5) -> (9,1); then (5,1) -> (7,3)."""
    s = host.Scheduler(host.PERFECT, (1, 1))
    rows = s.schedule(10, block_h=[0, 0], render_ns=[1, 5])
    assert rows == [5, 5]
    rows = s.schedule(10, block_h=rows, render_ns=[1, 5])
    assert rows == [9, 1]
    rows = s.schedule(10, block_h=rows, render_ns=[5, 1])
    assert rows == [7, 3]


def test_equal_speed_partitions():
    from polaris_amd.distributed import block_of, naive_rows

    assert naive_rows(8, 512) == [64] * 8
    assert naive_rows(8, 1080) == [135] * 8
    rows = naive_rows(3, 512)
    assert rows == [172, 170, 170] and sum(rows) == 512      # remainder to tracer 0
    assert [block_of(r, rows) for r in range(3)] == [(0, 172), (172, 170), (342, 170)]
    assert naive_rows(1, 7) == [7]


# ---- scene compiler (polaris_amd/host/scene_compiler.cpp) ---------------------------------------
def test_bvh_builder_known_answers(host):
    """asset/compiler/bvh/bvh_builder_test.go:10-68: four separated boxes -> minLeaf 1: 4 leaf
    callbacks of 1 item / 7 nodes; minLeaf 2: 2 callbacks of 2 items / 3 nodes."""
    boxes = [(-2, 0, -2, -1, 1, -1), (1, 0, -2, 2, 1, -1), (-2, 0, 1, -1, 1, 2), (1, 0, 1, 2, 1, 2)]
    nodes, leaves = host.bvh_build(boxes, 1)
    assert leaves == [1, 1, 1, 1] and len(nodes) == 7
    nodes, leaves = host.bvh_build(boxes, 2)
    assert leaves == [2, 2] and len(nodes) == 3
    # pre-order with the left subtree first (bvh_builder.go:196-208): root's children are 1 and 2
    assert int(nodes[0]["ldata"]) == 1 and int(nodes[0]["rdata"]) == 2


def test_compiler_one_triangle_two_instances(host):
    """asset/compiler/compiler_test.go:100-193: 1 triangle x 2 instances -> 3 vertices, 1 material
    index, 4 BVH nodes (top root + 2 instance leaves + mesh leaf), leaves -> instances 0 and 1, both
    BvhRoot == 3."""
    import numpy as np

    from polaris_amd import scenes

    mt = scenes.MaterialTable()
    d = mt.diffuse((1, 1, 1))
    tri = scenes.Mesh(np.array([[[0, 0, 0], [1, 0, 0], [1, 1, 0]]], dtype=np.float64), np.array([[[0, 0, 1]] * 3], dtype=np.float64),
                      np.zeros((1, 3, 2)), np.array([d]))
    sc = host.compile_scene([tri], [(0, scenes.translation((-5, 0, 0))), (0, np.eye(4))], mt)
    assert len(sc.vertices) == 3 and len(sc.normals) == 3 and len(sc.uvs) == 3 and len(sc.material_index) == 1
    assert len(sc.bvh_nodes) == 4 and len(sc.mesh_instances) == 2
    assert -int(sc.bvh_nodes[1]["ldata"]) == 0 and -int(sc.bvh_nodes[2]["ldata"]) == 1
    assert int(sc.mesh_instances[0]["bvh_root"]) == 3 and int(sc.mesh_instances[1]["bvh_root"]) == 3
    inv = sc.mesh_instances[0]["inv_transform"].reshape(4, 4).T          # inverse of the translation
    assert np.allclose(inv[:3, 3], [5, 0, 0])


def test_compiler_matches_python_producer(host, oracle):
    """The C++ compiler and the numpy scene producer build different BVHs over the same geometry;
    the traced image must be the same (identical candidate hits; only exact ties could differ)."""
    import numpy as np

    from oracle import pybind as ob
    from polaris_amd import scenes

    mt = scenes.MaterialTable()
    red, white = mt.diffuse((0.7, 0.1, 0.1)), mt.diffuse((0.7, 0.7, 0.7))
    light = mt.emissive((9, 9, 8), 1.0)
    mix = mt.mix(white, mt.rough_conductor((0.9, 0.8, 0.5), roughness=0.3), 0.5)
    room = scenes.merge([scenes.quad((-3, 0, -3), (-3, 0, 3), (3, 0, 3), (3, 0, -3), white),
                         scenes.quad((-3, 0, -3), (3, 0, -3), (3, 4, -3), (-3, 4, -3), red),
                         scenes.quad((-1, 3.9, -1), (1, 3.9, -1), (1, 3.9, 1), (-1, 3.9, 1), light)])
    ball = scenes.uv_sphere((0, 0, 0), 0.6, mix, n_lat=8, n_lon=10)
    insts = [(0, np.eye(4)), (1, scenes.translation((-1.2, 0.6, 0))), (1, scenes.translation((1.1, 0.6, 0.4)))]
    a = scenes.compile_scene([room, ball], insts, mt, name="py")
    b = host.compile_scene([room, ball], insts, mt, min_leaf=4)
    for sc in (a, b):
        sc.set_camera(eye=(0, 2.0, 6.0), look=(0, 1.0, 0), fov=0.8)
    assert len(a.emissives) == len(b.emissives) == 2 and b.num_triangles == a.num_triangles
    W, H, spp, B = 40, 30, 4, 4
    seeds = scenes.make_seeds(spp, B)
    ia, sa, _ = oracle.trace(a, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    ib, sb, _ = oracle.trace(b, ob.make_request(W, H, spp=spp, bounces=B), seeds)
    assert list(sa.rays_per_bounce[:B]) == list(sb.rays_per_bounce[:B])
    assert float(np.sqrt(np.mean((ia[..., :3] - ib[..., :3]) ** 2))) / spp <= 1e-6


def test_png_writer_round_trips(tmp_path):
    """renderer::WritePNG (the encoder of the SaveFrameBuffer stage, pipeline.go:215-235): what it writes
    is what PIL and this repo's own PNG decoder read back."""
    import numpy as np
    from PIL import Image

    from polaris_amd import host_api as H

    rng = np.random.default_rng(4)
    for shape in ((1, 1, 4), (37, 53, 4), (64, 200, 4)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        path = str(tmp_path / f"{shape[0]}x{shape[1]}.png")
        H.write_png(path, img)
        assert np.array_equal(np.array(Image.open(path)), img)
        fmt, w, h, data = H.texture_load(path)
        assert (w, h) == (shape[1], shape[0]) and np.array_equal(data.reshape(shape), img)
