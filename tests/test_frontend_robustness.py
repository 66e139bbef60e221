"""The scene front-end parses files it did not write: every decoder and parser is run, in an
AddressSanitizer + UBSan build (CPU only), over valid, truncated and corrupted inputs.  Nothing
may crash, read out of bounds or hang; malformed input is an error value."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import obj_fixtures as F  # noqa: E402

HOST = os.path.join(ROOT, "polaris_amd", "host")
BUILD = os.path.join(ROOT, "tests", "_build")
BIN = os.path.join(BUILD, "frontend_fuzz")
SRCS = [os.path.join(ROOT, "tests", "tools", "frontend_fuzz.cpp")] + [os.path.join(HOST, f) for f in
                                                                    ("texture.cpp", "jpeg.cpp", "material_expr.cpp", "wavefront_reader.cpp", "scene_compiler.cpp")]


@pytest.fixture(scope="module")
def fuzz_bin():
    os.makedirs(BUILD, exist_ok=True)
    deps = SRCS + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")]
    if not os.path.exists(BIN) or any(os.path.getmtime(d) > os.path.getmtime(BIN) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off",
                               "-fopenmp", "-I" + os.path.join(ROOT, "include"), "-I" + HOST, *SRCS, "-lz", "-o", BIN])
    return BIN


def run(binary, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([binary, *files], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-4000:]
    ok, rejected = (int(tok.split("=")[1]) for tok in p.stdout.split())
    return ok, rejected


def mutations(data: bytes, rng, n_trunc=24, n_flip=40):
    out = []
    cuts = sorted(set([0, 1, 2, 3, 7, 8, 15, 16, 17, 18, 53, 54] + [int(x) for x in rng.integers(0, max(len(data), 1), n_trunc)]))
    for c in cuts:
        if c < len(data):
            out.append(data[:c])
    for _ in range(n_flip):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            i = int(rng.integers(0, len(b)))
            b[i] = int(rng.integers(0, 256))
        out.append(bytes(b))
    for _ in range(8):  # header fields blown up: sizes, counts, offsets
        b = bytearray(data)
        i = int(rng.integers(0, min(len(b) - 4, 64)))
        b[i:i + 4] = b"\xff\xff\xff\x7f"
        out.append(bytes(b))
    return out


def test_texture_decoders_survive_malformed_files(fuzz_bin, tmp_path):
    rng = np.random.default_rng(17)
    rgb = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (5, 6, 4), dtype=np.uint8)
    grey16 = rng.integers(0, 65536, (4, 4)).astype(np.uint16)
    seeds = {}
    F.write_ppm(str(tmp_path / "s.pnm"), rgb); seeds["pnm"] = (tmp_path / "s.pnm").read_bytes()
    F.write_pgm(str(tmp_path / "s16.pnm"), grey16, maxval=65535); seeds["pnm16"] = (tmp_path / "s16.pnm").read_bytes()
    F.write_png(str(tmp_path / "s.png"), rgba); seeds["png"] = (tmp_path / "s.png").read_bytes()
    F.write_png(str(tmp_path / "s16.png"), grey16, bit_depth=16); seeds["png16"] = (tmp_path / "s16.png").read_bytes()
    F.write_bmp(str(tmp_path / "s.bmp"), rgb); seeds["bmp"] = (tmp_path / "s.bmp").read_bytes()
    F.write_tga(str(tmp_path / "s.tga"), rgba, rle=True); seeds["tga"] = (tmp_path / "s.tga").read_bytes()
    F.write_tga(str(tmp_path / "s2.tga"), rgb); seeds["tga2"] = (tmp_path / "s2.tga").read_bytes()
    F.write_hdr(str(tmp_path / "s.hdr"), rng.integers(0, 256, (3, 9, 4), dtype=np.uint8)); seeds["hdr"] = (tmp_path / "s.hdr").read_bytes()
    # JPEG seeds (written by Pillow): baseline 4:2:0 with restart markers, progressive 4:2:2, greyscale -- every entropy decoder, both
    # fancy upsamplers, the restart logic and the progressive refinement scans see truncated and corrupted streams
    from PIL import Image  # (part of this image; the JPEG seeds need an encoder)

    pic = Image.fromarray(rng.integers(0, 256, (37, 45, 3), dtype=np.uint8), "RGB")
    pic.save(str(tmp_path / "s.jpg"), "JPEG", quality=70, subsampling=2, restart_marker_blocks=2); seeds["jpg"] = (tmp_path / "s.jpg").read_bytes()
    pic.save(str(tmp_path / "sp.jpg"), "JPEG", quality=60, subsampling=1, progressive=True); seeds["jpgp"] = (tmp_path / "sp.jpg").read_bytes()
    pic.convert("L").save(str(tmp_path / "sg.jpg"), "JPEG", quality=85); seeds["jpgg"] = (tmp_path / "sg.jpg").read_bytes()
    ext = {"pnm": "pnm", "pnm16": "pnm", "png": "png", "png16": "png", "bmp": "bmp", "tga": "tga", "tga2": "tga", "hdr": "hdr", "jpg": "jpg", "jpgp": "jpg", "jpgg": "jpg"}
    files = [str(tmp_path / n) for n in ("s.pnm", "s16.pnm", "s.png", "s16.png", "s.bmp", "s.tga", "s2.tga", "s.hdr", "s.jpg", "sp.jpg", "sg.jpg")]
    for kind, data in seeds.items():
        for k, m in enumerate(mutations(data, rng, n_flip=120 if kind.startswith("jpg") else 40)):
            p = tmp_path / f"m_{kind}_{k}.{ext[kind]}"
            p.write_bytes(m)
            files.append(str(p))
    (tmp_path / "noise.png").write_bytes(rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()); files.append(str(tmp_path / "noise.png"))
    (tmp_path / "huge.pnm").write_bytes(b"P6\n100000 100000\n255\n" + b"\0" * 64); files.append(str(tmp_path / "huge.pnm"))
    (tmp_path / "zero.pnm").write_bytes(b"P5\n0 0\n255\n"); files.append(str(tmp_path / "zero.pnm"))
    ok, rejected = run(fuzz_bin, files)
    assert ok >= 11 and rejected > 100  # the eleven valid files decode; most mutations are rejected (some still decode: flipped texels; a JPEG with a
    # damaged entropy-coded segment decodes to garbage samples, as it does in libjpeg)


def test_obj_reader_and_expression_parser_survive_malformed_text(fuzz_bin, tmp_path):
    rng = np.random.default_rng(23)
    base = tmp_path / "base"
    obj = F.write_cornell(str(base))
    files = [obj]
    text = open(obj).read()
    mtl = open(os.path.join(str(base), "room.mtl")).read()
    junk = ["f 1 2", "f 1/2/3/4 2 3", "f 0 0 0", "f -99 1 2", "f 1//1 2/2 3", "v 1 2", "v nan inf -inf", "vt", "instance", "instance room 1 2 3 4 5 6 7 8",
            "instance room 0 0 0 0 0 0 0 0 0", "usemtl", "usemtl nope", "mtllib", "mtllib missing.mtl", "call room.obj", "call missing.obj",
            "camera_fov x", "camera_eye 1", "g", "o", "f 1 2 3 4 5", "v 1e999 0 0", "f 2147483648 1 2", "f 1 2 99999999999999999999"]
    for k in range(60):  # line-level mutations of the valid scene
        lines = text.split("\n")
        for _ in range(int(rng.integers(1, 4))):
            lines.insert(int(rng.integers(0, len(lines))), junk[int(rng.integers(0, len(junk)))])
        if k % 3 == 0:
            del lines[int(rng.integers(0, len(lines)))]
        d = tmp_path / f"o{k}"
        F.write_cornell(str(d))
        (d / "room.obj").write_text("\n".join(lines))
        files.append(str(d / "room.obj"))
    mjunk = ["newmtl", "newmtl white", "Kd 1", "Kd a b c", "Ni", "map_Kd", "include nope", "mat_expr", "mat_expr mix(", 'mat_expr mix("tall", "tall", 0.5)',
             'mat_expr bumpMap("bumpy", "bump.png")', "KeScaler x", 'mat_expr diffuse(reflectance: "checker.pnm", reflectance: {0.1,0.1,0.1})', "map_bump room.obj",
             "map_Kd ../base/bump.png", 'mat_expr mixMap(diffuse(), conductor(), "room.mtl.png")']
    for k in range(40):  # material-library mutations
        lines = mtl.split("\n")
        for _ in range(int(rng.integers(1, 3))):
            lines.insert(int(rng.integers(0, len(lines))), mjunk[int(rng.integers(0, len(mjunk)))])
        d = tmp_path / f"m{k}"
        F.write_cornell(str(d))
        (d / "room.mtl").write_text("\n".join(lines))
        files.append(str(d / "room.obj"))
    # a recursive `call` must not recurse for ever
    d = tmp_path / "loop"
    d.mkdir()
    (d / "a.obj").write_text("v 0 0 0\ncall a.obj\n")
    files.append(str(d / "a.obj"))
    exprs = ["", "(", "mix(mix(mix(mix(", "diffuse(" * 2000, 'diffuse(reflectance: {1,2', '"' * 999, "diffuse(reflectance: {1e39, 0, 0})", "\xff\xfe diffuse()",
             'mix(diffuse(), conductor(), 0.5' + ")" * 50, "roughConductor(roughness: 1e-50, intIOR: 0, extIOR: 0)", "disperse(" * 300 + "diffuse()"]
    for k, e in enumerate(exprs):
        p = tmp_path / f"e{k}.expr"
        p.write_bytes(e.encode("utf-8", "surrogateescape") if isinstance(e, str) else e)
        files.append(str(p))
    ok, rejected = run(fuzz_bin, files)
    assert ok >= 1 and rejected >= 30
