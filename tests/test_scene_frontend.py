"""CPU tests of the scene front-end: Wavefront OBJ/MTL reader, material
expression language, texture decoding, material-tree flattening and the camera -- the C++
restatement under polaris_amd/host/ of asset/scene/reader/wavefront.go, asset/material/*.go,
asset/texure/texture.go, asset/compiler/compiler.go:233-552 and asset/scene/camera.go.

Known answers come from the reference's own tests (asset/material/material_expr_test.go,
asset/scene/reader/wavefront_test.go): expression lists, face-index specs, instance transform
points, error texts.  Two of those files are stale against the reference's current sources (they
call `mix(a, b, w1, w2)` and a two-argument selectFaceCoordIndex); where they disagree with the
shipped grammar/code, the shipped grammar/code wins and the case is listed below.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import obj_fixtures as F  # noqa: E402

from polaris_amd import ctypes_api as T  # noqa: E402
from polaris_amd import host_api as H  # noqa: E402
from polaris_amd import scenes  # noqa: E402

pytestmark = pytest.mark.usefixtures("built")


# ---- material expressions (material_expr_test.go:5-33, 35-59) -----------------------------------
VALID = [
    'diffuse()',
    'diffuse(reflectance: {0.9, 0.9, 0.9})',
    'diffuse(reflectance: "texture.jpg")',
    'dielectric(specularity: "texture.jpg", intIOR: "gold", extIOR: "air")',
    'dielectric(specularity: "texture.jpEg", transmittance: {.9,.9,.9}, intIOR: 1.33, extIOR: "air")',
    'roughDielectric(specularity: "texture.jpEg", transmittance: {1,1,1}, intIOR: 1.33, extIOR: "air", roughness: 0.2)',
    'conductor(specularity: "texture.jpg")',
    'roughConductor(specularity: {.3,.3,.3}, intIOR: "gold", roughness: 1)',
    'emissive(radiance: {1,1,1}, scale: 10)',
    'bumpMap(conductor(specularity: "texture.jpg"), "foo.jpg")',
    'normalMap(conductor(specularity: "texture.jpg"), "foo.jpg")',
    # the reference test writes mix(..., 0.2, 0.8); material_expr.y:140 takes ONE weight
    'mix(diffuse(reflectance:{0.2, 0.2, 0.2}), conductor(specularity: "texture.jpg"), 0.2)',
    'mixMap(diffuse(), "other material", "mask.png")',
    'disperse(dielectric(), intIOR: {1.50, 1.52, 1.54}, extIOR: {0, 0, 0})',
]
SEMANTICALLY_INVALID = [
    'diffuse(specularity: {0.9, 0.9, 0.9})',
    'diffuse(reflectance: {1.0, 0.9, 0.9})',
    'conductor(roughness: "texture.jpg")',
    'roughConductor(specularity: {.3,.3,.3}, intIOR: "gold!!!", roughness: 1)',
    'roughConductor(specularity: {.3,.3,.3}, intIOR: 1.2, extIOR: "foo", roughness: 1)',
    'dielectric(transmittance: {1.3,.3,.3})',
    'mix(diffuse(), conductor(), 1.5)',  # reference: mix(diffuse(), conductor(), 0.2, 1.0), stale syntax
    'roughConductor(roughness: 1.5)',
    'disperse(dielectric(), intIOR: {0, 0, 0}, extIOR: {0, 0, 0})',
]
SYNTAX_ERRORS = [
    ('mix(diffuse(), conductor(), 0.2, 0.8)', "syntax error"),           # the stale two-weight form
    ('diffuse(reflectance: {-0.1, 0, 0})', 'invalid expression "-"'),      # numbers cannot start with '-'
    ('diffuse(reflectance: "unterminated)', "unterminated string litera"),  # (sic) material_expr.y:277
    ('shiny()', 'invalid expression "shiny"'),
    ('"just a name"', "syntax error"),                                      # a bare reference is not a material_def
    ('diffuse(reflectance: {1e, 0, 0})', 'invalid float value "1e"'),
    ('diffuse() diffuse()', "syntax error"),
]


@pytest.mark.parametrize("expr", VALID)
def test_valid_material_expressions(expr):
    assert H.material_check(expr) == (0, "")


@pytest.mark.parametrize("expr", SEMANTICALLY_INVALID)
def test_semantic_errors_parse_but_do_not_validate(expr):
    rc, msg = H.material_check(expr)
    assert rc == 2 and msg


@pytest.mark.parametrize("expr,msg", SYNTAX_ERRORS)
def test_syntax_errors(expr, msg):
    assert H.material_check(expr) == (1, msg)


def test_validation_does_not_descend_below_bump_and_disperse():
    """node.go:169-199: BumpMap/NormalMap/Disperse validate only their own fields."""
    assert H.material_check('bumpMap(diffuse(reflectance: {1.0, 1.0, 1.0}), "b.png")')[0] == 0
    assert H.material_check('mix(bumpMap(diffuse(reflectance: {1.0, 1.0, 1.0}), "b.png"), diffuse(), 0.5)')[0] == 0
    assert H.material_check('mix(diffuse(reflectance: {1.0, 1.0, 1.0}), diffuse(), 0.5)')[0] == 2


def test_ior_table():
    assert H.material_ior("Glass") == pytest.approx(1.51714) and H.material_ior("gLaSs") == H.material_ior("GLASS")
    assert H.material_ior("air") == pytest.approx(1.0002926)
    assert H.material_ior("Mercury (liq)") == pytest.approx(1.62)
    assert H.material_ior("unobtainium") is None
    # every name of the reference's table (asset/material/ior.go:12-257; data dumped by scripts/gen_ior_table.py)
    import json
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ior_table.json")))
    assert len(table) == 245
    for name, ior in table.items():
        assert H.material_ior(name) == np.float32(ior), name
        assert H.material_ior(name.lower()) == np.float32(ior), name
    assert H.material_ior("Zirconia, Cubic") == np.float32(2.170)


# ---- reader: wavefront_test.go -------------------------------------------------------------------
def test_select_face_coordinate():  # wavefront_test.go:84-109
    for tok, n in (("2", 1), ("-2", 1)):
        with pytest.raises(ValueError, match="index out of bounds"):
            H.select_face_index(tok, n)
    assert H.select_face_index("1", 10) == 0   # indices are 1-based
    assert H.select_face_index("-1", 10) == 9  # negative = from the end
    assert H.select_face_index("2", 10, rel_offset=5) == 6  # positive indices are relative to the including file
    with pytest.raises(ValueError, match='strconv.ParseInt: parsing "x": invalid syntax'):
        H.select_face_index("x", 10)


TRI = """
o testObj
v 0 0 0
v 1 0 0
v 0 1 0
vn 1 0 0
vt 0 0
vn 0 1 0
vt 0 1
vn 0 1 0
vt 1 0
vn 0 0 1
# Comment
f 1/1/1 2/2/2 -1/-1/-1
"""


def test_default_mesh_instance_generation():  # wavefront_test.go:111-157
    p = H.parse_obj(TRI)
    assert (p["meshes"], p["instances"], p["mesh0_primitives"]) == (1, 1, 1)
    assert np.array_equal(p["transforms"][0], np.eye(4, dtype=np.float32))
    assert np.allclose(p["center"][0], [0.5, 0.5, 0], atol=1e-3)
    assert np.allclose(p["bbox"][0], [[0, 0, 0], [1, 1, 0]], atol=1e-3)
    assert p["materials"] == [("", "1", "diffuse(reflectance: {0.700000, 0.700000, 0.700000})")]  # the default material, Kd 0.7


def test_mesh_instancing():  # wavefront_test.go:159-215
    p = H.parse_obj(TRI + """# Mesh instances
instance testObj 	1 0 1	0 0 0 	1 1 1
instance testObj 	0 0 0	0 90 0 	1 1 1
instance testObj 	0 1 0	90 0 0	10 10 10
""")
    assert p["instances"] == 3
    specs = [(0, (0, 0, 0), (1, 0, 1)), (0, (-1, 0, -1), (0, 0, 0)), (1, (1, 0, 0), (0, 0, -1)), (1, (0, 0, -1), (-1, 0, 0)),
             (2, (0, 1, 0), (0, 0, 20))]
    for inst, src, want in specs:
        got = p["transforms"][inst] @ np.array([*src, 1.0], np.float32)
        assert np.allclose(got[:3], want, atol=1e-3), (inst, src, got)
    assert np.allclose(p["bbox"][0], [[1, 0, 1], [2, 1, 1]], atol=1e-3)
    # the reference moves the mesh box by the translation only, whatever the rotation / scale (wavefront.go:514-519)
    assert np.allclose(p["bbox"][2], [[0, 1, 0], [1, 2, 0]], atol=1e-3)


def test_parse_single_faced_object():  # wavefront_test.go:217-312
    sc = H.read_scene(content=TRI)
    assert sc.vertices.shape == (3, 4)
    assert np.array_equal(sc.vertices[:, :3], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    assert np.array_equal(sc.normals[:, :3], [[1, 0, 0], [0, 1, 0], [0, 0, 1]])
    assert np.array_equal(sc.uvs, [[0, 0], [0, 1], [1, 0]])
    assert len(sc.material_nodes) == 1 and sc.material_nodes["type"][0] == T.BXDF_DIFFUSE
    assert np.allclose(sc.material_nodes["k"][0], [0.7, 0.7, 0.7, 0])


def test_parse_single_quad_faced_object():  # wavefront_test.go:314-414
    sc = H.read_scene(content="o testObj\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\n# Comment\nf 1 2 3 4\n")
    assert sc.vertices.shape == (6, 4)
    assert np.array_equal(sc.vertices[:, :3], [[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 0, 0], [1, 1, 0], [0, 1, 0]])
    assert np.array_equal(sc.normals[:, :3], np.tile([0, 0, 1], (6, 1)))  # generated face normal
    # 2 triangles < minPrimitivesPerLeaf: top leaf + one mesh leaf
    assert len(sc.bvh_nodes) == 2 and sc.bvh_nodes["rdata"][1] == 2


@pytest.mark.parametrize("payload,msg", [
    ("Kd 1.0 1.0 1.0", '[embedded: 1] error: got "Kd" without a "newmtl"'),                                           # :415-424
    ("\n\tnewmtl foo\n\tKd 1.0", '[embedded: 3] error: unsupported syntax for "Kd"; expected 3 arguments; got 1'),    # :426-437
    ("\n\tnewmtl foo\n\tNi", '[embedded: 3] error: unsupported syntax for "Ni"; expected 1 argument; got 0'),         # :439-450
    ("newmtl a\nnewmtl a", '[embedded: 2] error: material "a" already defined'),
    ("newmtl a\ninclude b", '[embedded: 2] error: could not include unknown material "b"'),
    ("newmtl a\nKd 1 x 1", '[embedded: 2] error: strconv.ParseFloat: parsing "x": invalid syntax'),
])
def test_material_loader_errors(payload, msg):
    with pytest.raises(RuntimeError) as e:
        H.parse_mtl(payload)
    assert str(e.value) == msg


def test_material_loader_and_expression_generation():  # wavefront_test.go:452-500 + wavefront.go:58-124
    mats = dict(H.parse_mtl("""
	# comment
	newmtl foo
	Kd 1.0 1.0 1.0
	Ks 0.1 0.2 0.3
	Ke 0.4    0.5 0.6
	Ni 2.5
	Nr 0
	newmtl mirror
	Ks 0.9 0.9 0.9
	newmtl lamp
	Ke 5 5 5
	KeScaler 3
	newmtl wall
	Kd 0.1234567 0.5 0.25
	map_bump b.png
	map_normal n.png
	newmtl tex
	map_Kd wood.jpg
	newmtl copy
	include lamp
	newmtl layered
	mat_expr mix("wall",   "mirror", 0.25)
	newmtl empty
"""))
    assert mats["foo"] == "dielectric(specularity: {0.100000, 0.200000, 0.300000}, intIOR: 2.5)"  # specular + Ni wins over Kd/Ke
    assert mats["mirror"] == "conductor(specularity: {0.900000, 0.900000, 0.900000})"
    assert mats["lamp"] == "emissive(radiance: {5.000000, 5.000000, 5.000000}, scale: 3)"
    # %f keeps six decimals (Vec3.String, types/vector.go:59-61); a normal map wins over a bump map
    assert mats["wall"] == 'normalMap(diffuse(reflectance: {0.123457, 0.500000, 0.250000}), "n.png")'
    assert mats["tex"] == 'diffuse(reflectance: "wood.jpg")'
    assert mats["copy"] == mats["lamp"]
    assert mats["layered"] == 'mix("wall", "mirror", 0.25)'  # tokens re-joined with single spaces
    assert mats["empty"] == "diffuse()"
    for e in mats.values():
        assert H.material_check(e)[0] == 0


def test_obj_errors_carry_file_line_and_include_stack(tmp_path):
    with pytest.raises(RuntimeError) as e:
        H.read_scene(content="v 0 0 0\nusemtl nope\n")
    assert str(e.value) == '[embedded: 2] error: undefined material with name "nope"'
    with pytest.raises(RuntimeError) as e:
        H.read_scene(content="v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3 4 5 6\n")
    assert 'unsupported syntax for "f"; expected 3 arguments for triangular face or 4 arguments for a quad face; got 6' in str(e.value)
    with pytest.raises(RuntimeError) as e:
        H.read_scene(content="v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    assert str(e.value) == "[embedded: 4] error: could not parse vertex coord for face argument 2: index out of bounds"
    with pytest.raises(RuntimeError) as e:
        H.read_scene(content="o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\ninstance b 0 0 0 0 0 0 1 1 1\n")
    assert str(e.value) == '[embedded: 6] error: unknown mesh with name "b"'
    # an error inside a called file reports where it was referenced from
    (tmp_path / "main.obj").write_text("call part.obj\n")
    (tmp_path / "part.obj").write_text("v 0 0\n")
    with pytest.raises(RuntimeError) as e:
        H.read_scene(str(tmp_path / "main.obj"))
    main, part = str(tmp_path / "main.obj"), str(tmp_path / "part.obj")
    assert str(e.value) == f'[{part}: 1] error: unsupported syntax for "v"; expected 3 arguments; got 2\nreferenced from {main}:1 [call]'
    with pytest.raises(RuntimeError, match="unsupported file format"):
        H.read_scene(str(tmp_path / "scene.ply"))


def test_call_offsets_indices_per_file(tmp_path):
    """Positive face indices are relative to the file they appear in (wavefront.go:311-317)."""
    (tmp_path / "main.obj").write_text("o a\nv 5 5 5\nv 6 5 5\nv 5 6 5\nf 1 2 3\ncall part.obj\n")
    (tmp_path / "part.obj").write_text("o b\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    sc = H.read_scene(str(tmp_path / "main.obj"))
    assert len(sc.mesh_instances) == 2
    assert sorted(map(tuple, sc.vertices[:, :3].tolist())) == sorted([(5, 5, 5), (6, 5, 5), (5, 6, 5), (0, 0, 0), (1, 0, 0), (0, 1, 0)])


# ---- textures (asset/texure/texture.go:57-147) -----------------------------------------------------
def test_texture_decoders(tmp_path):
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (6, 4, 4), dtype=np.uint8)
    grey = rng.integers(0, 256, (9, 3), dtype=np.uint8)
    want_rgba = np.concatenate([rgb, np.full((5, 7, 1), 255, np.uint8)], axis=2)  # alpha 255 added to 3-channel images

    F.write_ppm(str(tmp_path / "a.pnm"), rgb)
    F.write_png(str(tmp_path / "a.png"), rgb)
    F.write_bmp(str(tmp_path / "a.bmp"), rgb)
    F.write_tga(str(tmp_path / "a.tga"), rgb)
    for name in ("a.pnm", "a.png", "a.bmp", "a.tga"):
        fmt, w, h, data = H.texture_load(str(tmp_path / name))
        assert (fmt, w, h) == (T.TEX_RGBA8, 7, 5), name
        assert np.array_equal(data.reshape(5, 7, 4), want_rgba), name

    F.write_png(str(tmp_path / "b.png"), rgba)
    F.write_tga(str(tmp_path / "b.tga"), rgba, rle=True)
    for name in ("b.png", "b.tga"):
        fmt, w, h, data = H.texture_load(str(tmp_path / name))
        assert (fmt, w, h) == (T.TEX_RGBA8, 4, 6) and np.array_equal(data.reshape(6, 4, 4), rgba), name

    F.write_pgm(str(tmp_path / "g.pnm"), grey)
    F.write_png(str(tmp_path / "g.png"), grey)
    for name in ("g.pnm", "g.png"):
        fmt, w, h, data = H.texture_load(str(tmp_path / name))
        assert (fmt, w, h) == (T.TEX_L8, 3, 9) and np.array_equal(data.reshape(9, 3), grey), name

    # wider-than-8-bit integers become floats in [0,1] (OpenImageIO's conversion): L32F / RGBA32F
    g16 = rng.integers(0, 65536, (4, 4)).astype(np.uint16)
    F.write_pgm(str(tmp_path / "g16.pnm"), g16, maxval=65535)
    F.write_png(str(tmp_path / "g16.png"), g16, bit_depth=16)
    for name in ("g16.pnm", "g16.png"):
        fmt, w, h, data = H.texture_load(str(tmp_path / name))
        assert fmt == T.TEX_L32F and np.array_equal(data.view(np.float32).reshape(4, 4), g16.astype(np.float32) / np.float32(65535)), name
    c16 = rng.integers(0, 65536, (3, 2, 3)).astype(np.uint16)
    F.write_png(str(tmp_path / "c16.png"), c16, bit_depth=16)
    fmt, w, h, data = H.texture_load(str(tmp_path / "c16.png"))
    px = data.view(np.float32).reshape(3, 2, 4)
    assert fmt == T.TEX_RGBA32F and np.array_equal(px[..., :3], c16.astype(np.float32) / np.float32(65535)) and np.all(px[..., 3] == 1.0)

    # asset/texure/texture_test.go:14-66: a 1x1 8-bit RGBA PNG -> Rgba8, 4 bytes; a 1x1 16-bit RGBA PNG -> Rgba32F, 16 bytes
    F.write_png(str(tmp_path / "one8.png"), np.zeros((1, 1, 4), np.uint8))
    F.write_png(str(tmp_path / "one16.png"), np.zeros((1, 1, 4), np.uint16), bit_depth=16)
    fmt, w, h, data = H.texture_load(str(tmp_path / "one8.png"))
    assert (fmt, w, h, data.size) == (T.TEX_RGBA8, 1, 1, 4)
    fmt, w, h, data = H.texture_load(str(tmp_path / "one16.png"))
    assert (fmt, w, h, data.size) == (T.TEX_RGBA32F, 1, 1, 16)

    rgbe = np.array([[[128, 64, 32, 129], [255, 0, 0, 128], [0, 0, 0, 0]]], np.uint8)
    F.write_hdr(str(tmp_path / "e.hdr"), rgbe)
    fmt, w, h, data = H.texture_load(str(tmp_path / "e.hdr"))
    px = data.view(np.float32).reshape(1, 3, 4)
    assert (fmt, w, h) == (T.TEX_RGBA32F, 3, 1)
    assert np.allclose(px[0, 0, :3], (np.array([128, 64, 32]) + 0.5) * 2.0 ** (129 - 136)) and np.all(px[0, 2, :3] == 0) and np.all(px[..., 3] == 1)

    F.write_png(str(tmp_path / "ga.png"), rng.integers(0, 256, (2, 2, 2), dtype=np.uint8))  # grey+alpha
    with pytest.raises(RuntimeError, match="unsupported channel count 2"):
        H.texture_load(str(tmp_path / "ga.png"))
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    with pytest.raises(RuntimeError, match="texture: jpeg: "):
        H.texture_load(str(tmp_path / "x.jpg"))
    (tmp_path / "x.tif").write_bytes(b"II*\x00\x08\x00\x00\x00 a TIFF header and nothing else")
    with pytest.raises(RuntimeError, match="no decoder in this build"):
        H.texture_load(str(tmp_path / "x.tif"))


# ---- compiler: material trees, textures, emissives, camera ---------------------------------------------
def test_cornell_obj_compiles_to_the_expected_arrays(tmp_path, monkeypatch):
    sc = H.read_scene(F.write_cornell(str(tmp_path)), aspect=1.0)
    monkeypatch.chdir(tmp_path)  # parse_obj below resolves `mtllib room.mtl` against the working directory
    assert sc.warnings == []
    N = sc.material_nodes
    # post-order flattening (compiler.go:331-438): children before their operator; used materials in
    # definition order; "steel" is unused by geometry and only reached through tall's reference
    assert list(N["type"]) == [4, 4, 4, 2, 4, 16, 4, 10001, 32, 4, 10003]
    assert np.allclose(N["k"][3], [17, 12, 4, 0]) and N["scale"][3] == 1.0                     # emissive default scale
    assert N["tex"][4] == 0 and np.allclose(N["k"][4], [0.2, 0.2, 0.2, 0])                      # textured diffuse keeps the default Kd
    assert N["int_ior"][5] == np.float32(2.5) and N["scale"][5] == np.float32(0.25)              # "steel" IOR by name, roughness
    assert (N["left_child"][7], N["right_child"][7]) == (5, 6) and N["k"][7][0] == np.float32(0.4)
    assert N["int_ior"][8] == np.float32(1.5) and np.allclose(N["t"][8], [0.95, 0.95, 0.95, 0]) and N["ext_ior"][8] == np.float32(1.0002926)
    assert (N["left_child"][10], N["tex"][10]) == (9, 1)                                          # bumpMap(white copy, bump.png)
    assert np.all(N["roughness_tex"] == -1)
    # textures: checker.pnm -> RGBA8 8x8 at 0, bump.png -> L8 16x16 after it, dword aligned
    M = sc.texture_meta
    assert list(M["format"]) == [T.TEX_RGBA8, T.TEX_L8] and list(M["width"]) == [8, 16] and list(M["data_offset"]) == [0, 256]
    assert sc.texture_data.size == 256 + 256
    assert np.array_equal(sc.texture_data[:256].reshape(8, 8, 4)[..., :3], F.checker())
    # geometry: 12 room + 12 cube + 4 prism triangles; 4 instances over 3 meshes
    assert sc.vertices.shape[0] == 3 * 28 and len(sc.mesh_instances) == 4
    assert list(sc.mesh_instances["mesh_index"]) == [0, 1, 1, 2]
    # emissives: the light quad = 2 triangles, material node 3, area 0.18 each
    E = sc.emissives
    assert len(E) == 2 and list(E["mat_node_index"]) == [3, 3] and np.allclose(E["area"], 0.18, atol=1e-6) and np.all(E["type"] == 0)
    assert sc.scene_diffuse_mat_index == -1 and sc.scene_emissive_mat_index == -1
    # every triangle's material root is one of the used materials' roots
    assert set(sc.material_index.tolist()) <= {0, 1, 2, 3, 4, 7, 8, 10}
    # instance matrices are the INVERSE transforms (compiler.go:185-192)
    p = H.parse_obj(F.cornell_obj())
    for i in range(4):
        inv = sc.mesh_instances["inv_transform"][i].reshape(4, 4).T
        assert np.allclose(inv @ p["transforms"][i], np.eye(4), atol=1e-5)


def test_scene_wide_materials_and_missing_textures(tmp_path):
    (tmp_path / "s.mtl").write_text("newmtl scene_diffuse_material\nKd 0.2 0.3 0.4\nnewmtl scene_emissive_material\nKe 1 1 1\n"
                                    "newmtl m\nmap_Kd nowhere.png\n")
    (tmp_path / "s.obj").write_text("mtllib s.mtl\nusemtl m\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    sc = H.read_scene(str(tmp_path / "s.obj"))
    assert sc.warnings == ['"m": skipping missing texture "nowhere.png"']  # compiler.go:499-502: warn, texture index stays -1
    assert list(sc.material_nodes["type"]) == [4, 2, 4] and sc.material_nodes["tex"][2] == -1
    assert (sc.scene_diffuse_mat_index, sc.scene_emissive_mat_index) == (0, 1)
    assert len(sc.emissives) == 1 and sc.emissives["type"][0] == 1 and sc.emissives["mat_node_index"][0] == 1  # environment light


def test_material_reference_errors(tmp_path):
    def run(mtl):
        (tmp_path / "s.mtl").write_text(mtl)
        (tmp_path / "s.obj").write_text("mtllib s.mtl\nusemtl a\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        return H.read_scene(str(tmp_path / "s.obj"))

    with pytest.raises(RuntimeError, match='material "a" references undefined material "zzz"'):
        run('newmtl a\nmat_expr mix("zzz", diffuse(), 0.5)\n')
    with pytest.raises(RuntimeError, match='detected circular dependency loop while processing "a"; a -> b => a'):
        run('newmtl a\nmat_expr mix("b", diffuse(), 0.5)\nnewmtl b\nmat_expr bumpMap("a", "x.png")\n')
    # reference quirk kept: the reference list is never popped, so naming one material twice in a tree trips the same check
    with pytest.raises(RuntimeError, match="circular dependency"):
        run('newmtl a\nmat_expr mix("b", "b", 0.5)\nnewmtl b\nKd 0.5 0.5 0.5\n')
    with pytest.raises(RuntimeError, match='material "a": energy conservation violation'):
        run("newmtl a\nKd 1 1 1\n")
    sc = run('newmtl a\nmat_expr mix("b", diffuse(), 0.5)\nnewmtl b\nKd 0.5 0.5 0.5\n')
    assert list(sc.material_nodes["type"]) == [4, 4, 10001]


def test_camera_matches_the_float64_derivation():
    """scene.Camera in float32 (camera.hpp) vs polaris_amd.scenes.camera_frustum in float64."""
    obj = TRI + "camera_fov 0.9\ncamera_eye 1 2 5\ncamera_look 0 0.5 -1\ncamera_up 0 1 0\n"
    for aspect, inv in ((1.0, False), (16 / 9, False), (4 / 3, True)):
        sc = H.read_scene(content=obj, aspect=aspect, invert_y=inv)
        want = scenes.camera_frustum([1, 2, 5], [0, 0.5, -1], [0, 1, 0], np.float32(0.9), aspect, invert_y=inv)
        assert np.array_equal(sc.eye, np.array([1, 2, 5], np.float32))
        assert np.allclose(sc.frustum, want, atol=2e-5)
    assert sc.camera["fov"] == pytest.approx(0.9)
    dflt = H.read_scene(content=TRI)  # NewScene defaults, raw_scene.go:150-160
    assert dflt.camera["fov"] == 45.0 and np.array_equal(dflt.camera["look"], [0, 0, -1])


def test_oracle_traces_the_obj_scene(tmp_path, oracle):
    from oracle import pybind as ob

    sc = H.read_scene(F.write_cornell(str(tmp_path)), aspect=1.0)
    req = ob.make_request(24, 24, spp=4, bounces=4, rr=3)
    acc, st, _ = oracle.trace(sc, req, scenes.make_seeds(4, 4, base=3))
    assert np.isfinite(acc).all() and acc[..., :3].mean() > 0.05
    assert st.primary_rays == 24 * 24 * 4 and st.shaded_hits > 0 and st.occlusion_rays > 0


def test_the_references_own_scene_fixture_compiles_to_the_golden_arrays():
    """tracer/opencl/fixtures/cube.obj + cube.mtl -- the one scene file the reference ships (a 12-triangle cube `instance`d twice with scale
    arguments of 0, which types.Scale4 reads as 1: types/matrix.go:42-53) -- read WHERE IT LIES through wavefront_reader.cpp ->
    scene_compiler.cpp must give exactly the arrays tests/golden/reference_cube_32.npz was traced from by the compiled reference
    (tests/tools/make_golden.py; the golden stores arrays, never the file's text).  Build container only: the GPU box has no
    /root/reference, there the golden itself carries the fixture (tests/test_oracle_golden.py, test_hip_reproduces_reference_golden)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import make_golden

    if not os.path.exists(make_golden.REFERENCE_CUBE):
        pytest.skip("the reference tree is not on this machine")
    from conftest import load_golden

    d, want, _ = load_golden(os.path.join(ROOT, "tests", "golden", "reference_cube_32.npz"))
    sc = make_golden.build_scene("reference-cube", 1.0)
    assert sc.num_triangles == 12 and len(sc.mesh_instances) == 2 and len(sc.emissives) == 1      # 12 triangles x 2 instances + the harness's environment light
    for name in ("bvh_nodes", "mesh_instances", "material_nodes", "emissives", "vertices", "normals", "uvs", "material_index"):
        assert np.asarray(getattr(sc, name)).tobytes() == np.asarray(getattr(want, name)).tobytes(), name
    # the two instances: the cube where it is, and one unit to the left (the inverse transform carries +1 in x)
    inv = sc.mesh_instances["inv_transform"].reshape(2, 4, 4)
    assert np.array_equal(inv[0], np.eye(4, dtype=np.float32)) and inv[1][3, 0] == 1.0 and np.array_equal(inv[1][:3, :3], np.eye(3, dtype=np.float32))


def test_jpeg_textures_decode_like_libjpeg(tmp_path):
    """asset/texure/texture.go:25-150 hands every image to OpenImageIO, whose JPEG reader is libjpeg.  polaris_amd/host/jpeg.cpp restates
    libjpeg's default pipeline (islow IDCT, fancy upsampling, fixed-point YCbCr -> RGB); the texels must equal what Pillow -- libjpeg-turbo,
    bit-compatible with libjpeg -- decodes from the same file, byte for byte: greyscale and colour, 4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, sizes that are
    not multiples of the MCU, qualities 1 .. 100, Huffman-optimised, restart intervals, progressive (spectral selection + successive
    approximation), RGB-coded files; CMYK and truncated files are errors, not crashes."""
    Image = pytest.importorskip("PIL.Image")  # (the comparison needs libjpeg itself: Pillow carries it)

    rng = np.random.default_rng(3)

    def picture(w, h, mode):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 3) % 256], -1).astype(np.int32) + rng.integers(-40, 40, (h, w, 3))
        a[h // 3: h // 2, w // 4: w // 2] = (255, 0, 255)
        a = np.clip(a, 0, 255).astype(np.uint8)
        return Image.fromarray(a if mode == "RGB" else np.ascontiguousarray(a[..., 0]), mode)

    path = str(tmp_path / "t.jpg")
    cases = []
    for (w, h) in [(1, 1), (2, 3), (7, 5), (8, 8), (17, 33), (64, 48), (129, 67)]:
        for mode in ("L", "RGB"):
            for sub in ((0, 1, 2) if mode == "RGB" else (None,)):
                for q, prog in ((35, False), (75, True), (95, False), (90, True)):
                    cases.append((w, h, mode, dict(quality=q, progressive=prog, **({} if sub is None else {"subsampling": sub}))))
    big = (131, 97)
    for kw in (dict(quality=1), dict(quality=100, subsampling=0), dict(quality=60, subsampling="4:1:1"), dict(quality=85, restart_marker_blocks=3),
               dict(quality=85, restart_marker_rows=1, subsampling=2), dict(quality=70, progressive=True, restart_marker_blocks=5), dict(quality=90, keep_rgb=True, subsampling=0),
               dict(quality=50, progressive=True, subsampling="4:1:1"), dict(quality=80, optimize=True, subsampling=2), dict(quality=30, qtables="web_low")):
        cases.append((*big, "RGB", kw))
    cases.append((77, 50, "L", dict(quality=80, restart_marker_blocks=2)))
    for w, h, mode, kw in cases:
        picture(w, h, mode).save(path, "JPEG", **kw)
        want = np.asarray(Image.open(path))
        fmt, tw, th, data = H.texture_load(path)
        assert (tw, th) == (w, h) and fmt == (T.TEX_RGBA8 if mode == "RGB" else T.TEX_L8), (w, h, mode, kw)
        got = data.reshape(th, tw, 4) if mode == "RGB" else data.reshape(th, tw)
        if mode == "RGB":
            assert (got[..., 3] == 255).all()
            got = got[..., :3]
        assert np.array_equal(got, want), (w, h, mode, kw, int(np.abs(got.astype(int) - want.astype(int)).max()))
    Image.fromarray(rng.integers(0, 256, (8, 8, 4), dtype=np.uint8), "CMYK").save(path, "JPEG")
    with pytest.raises(RuntimeError, match="CMYK"):
        H.texture_load(path)
    picture(64, 48, "RGB").save(path, "JPEG", quality=80)
    whole = open(path, "rb").read()
    for cut in (3, 20, 100):
        open(path, "wb").write(whole[:cut])
        with pytest.raises(RuntimeError, match="jpeg"):
            H.texture_load(path)


def test_a_jpeg_texture_reaches_the_scene_arrays(tmp_path):
    """`map_Kd wood.jpg` goes reader -> texture.cpp (JPEG by its signature) -> bakeTexture like any other image: the compiled scene equals the
    one that names a PNG of the pixels libjpeg decodes from the same file (asset/texure/texture.go:25-150: 8-bit RGB -> Rgba8, grey -> Luminance8)."""
    Image = pytest.importorskip("PIL.Image")

    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:24, 0:40]
    rgb = np.clip(np.stack([x * 6, y * 10, (x + y) * 4], -1) + rng.integers(-20, 20, (24, 40, 3)), 0, 255).astype(np.uint8)
    for mode, fmt in (("RGB", T.TEX_RGBA8), ("L", T.TEX_L8)):
        d = tmp_path / mode
        d.mkdir()
        pic = Image.fromarray(rgb if mode == "RGB" else np.ascontiguousarray(rgb[..., 1]), mode)
        pic.save(str(d / "wood.jpg"), "JPEG", quality=80, subsampling=2 if mode == "RGB" else -1)
        F.write_png(str(d / "wood.png"), np.asarray(Image.open(str(d / "wood.jpg"))))
        scenes_read = []
        for ext in ("jpg", "png"):
            (d / f"s_{ext}.mtl").write_text(f"newmtl wood\nmap_Kd wood.{ext}\nnewmtl lamp\nKe 5 5 5\n")
            (d / f"s_{ext}.obj").write_text(f"mtllib s_{ext}.mtl\ncamera_fov 0.7\ncamera_eye 0 1 3\ncamera_look 0 0 0\ncamera_up 0 1 0\n"
                                            "v -1 0 1\nv 1 0 1\nv 1 0 -1\nv -1 0 -1\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nusemtl wood\nf 1/1 2/2 3/3 4/4\n"
                                            "v -1 2 -1\nv 1 2 -1\nv 1 2 1\nv -1 2 1\nusemtl lamp\nf 5 6 7 8\n")
            sc = H.read_scene(str(d / f"s_{ext}.obj"), aspect=1.0)
            assert sc.warnings == [], sc.warnings
            scenes_read.append(sc)
        a, b = scenes_read
        assert list(a.texture_meta["format"]) == [fmt] and (int(a.texture_meta["width"][0]), int(a.texture_meta["height"][0])) == (40, 24)
        for name in ("texture_meta", "texture_data", "material_nodes", "vertices", "uvs", "material_index", "bvh_nodes", "emissives"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), (mode, name)
