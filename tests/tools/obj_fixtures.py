"""Wavefront OBJ/MTL fixtures for the scene front-end tests: a small Cornell-style room written
the way `polaris render` scenes are authored (quads, `usemtl`, `mat_expr`, textures, `instance`,
`camera_*`), plus tiny image writers (PNG/PPM/PGM/BMP/TGA/HDR) used to feed the texture decoder."""
import os
import struct
import zlib

import numpy as np


def write_ppm(path, rgb):
    h, w, _ = rgb.shape
    with open(path, "wb") as f:
        f.write(b"P6\n# written by tests\n%d %d\n255\n" % (w, h))
        f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())


def write_pgm(path, grey, maxval=255):
    h, w = grey.shape
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n%d\n" % (w, h, maxval))
        f.write(np.ascontiguousarray(grey, ">u2" if maxval > 255 else np.uint8).tobytes())


def write_png(path, img, bit_depth=8):
    """img: (h,w) grey, (h,w,3) rgb or (h,w,4) rgba; filter type cycles 0..4 over rows to exercise the decoder."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 3: 2, 4: 6, 2: 4}[ch]
    raw = img.astype(">u2" if bit_depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).reshape(h, -1).astype(np.int32)
    bpp = ch * bit_depth // 8
    out = bytearray()
    prev = np.zeros(raw.shape[1], np.int32)
    for y in range(h):
        cur = raw[y]
        ft = y % 5
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        b = prev
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) // 2
        else:
            p = a + b - c
            pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        out.append(ft)
        out += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    comp = zlib.compress(bytes(out), 6)
    half = len(comp) // 2
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", comp[:half]))  # split on purpose: IDAT chunks concatenate
        f.write(chunk(b"IDAT", comp[half:]))
        f.write(chunk(b"IEND", b""))


def write_bmp(path, rgb):
    h, w, _ = rgb.shape
    stride = (w * 3 + 3) // 4 * 4
    rows = bytearray()
    for y in range(h - 1, -1, -1):  # bottom-up
        row = np.ascontiguousarray(rgb[y, :, ::-1], np.uint8).tobytes()
        rows += row + b"\0" * (stride - len(row))
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(rows), 2835, 2835, 0, 0))
        f.write(rows)


def write_tga(path, rgba, rle=False):
    h, w, ch = rgba.shape
    px = np.ascontiguousarray(rgba[::-1, :, [2, 1, 0] + ([3] if ch == 4 else [])], np.uint8)  # bottom-up BGR(A)
    with open(path, "wb") as f:
        f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, 8 * ch, 8 if ch == 4 else 0))
        if not rle:
            f.write(px.tobytes())
        else:
            flat = px.reshape(-1, ch)
            i = 0
            while i < len(flat):
                run = 1
                while i + run < len(flat) and run < 128 and np.array_equal(flat[i + run], flat[i]):
                    run += 1
                if run > 1:
                    f.write(bytes([128 | (run - 1)]) + flat[i].tobytes())
                    i += run
                else:
                    lit = 1
                    while i + lit < len(flat) and lit < 128 and not np.array_equal(flat[i + lit], flat[i + lit - 1]):
                        lit += 1
                    f.write(bytes([lit - 1]) + flat[i:i + lit].tobytes())
                    i += lit


def write_hdr(path, rgbe):
    """rgbe: (h, w, 4) uint8, written flat (no RLE)."""
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(np.ascontiguousarray(rgbe, np.uint8).tobytes())


def checker(n=8, a=(230, 230, 230), b=(40, 60, 200)):
    yy, xx = np.mgrid[0:n, 0:n]
    m = ((xx // 2 + yy // 2) % 2).astype(bool)
    return np.where(m[..., None], np.array(a, np.uint8), np.array(b, np.uint8)).astype(np.uint8)


CORNELL_MTL = """# materials of the test room
newmtl white
Kd 0.725 0.71 0.68
newmtl red
Kd 0.63 0.065 0.05
newmtl green
Kd 0.14 0.45 0.091
newmtl light
Ke 17 12 4
newmtl steel
mat_expr roughConductor(specularity: {0.8, 0.8, 0.85}, intIOR: "steel", roughness: 0.25)
newmtl floor
map_Kd checker.pnm
newmtl tall
mat_expr mix("steel", diffuse(reflectance: {0.6, 0.6, 0.2}), 0.4)
newmtl glass
Ks 1 1 1
Tf 0.95 0.95 0.95
Ni 1.5
newmtl bumpy
include white
map_bump bump.png
newmtl unused_material
Kd 0.1 0.2 0.3
"""


def _quad(a, b, c, d):
    return [a, b, c, d]


def cornell_obj(with_instances=True):
    """A unit-ish room [-1,1]x[0,2]x[-1,1], a light quad, a box mesh instanced twice."""
    L = []
    L += ["mtllib room.mtl", "camera_fov 0.69", "camera_eye 0 1 3.4", "camera_look 0 1 0", "camera_up 0 1 0", ""]
    v = []
    f = []

    def add_quad(p, mat, uv=False):
        base = len(v)
        v.extend(p)
        if uv:
            f.append((mat, "f " + " ".join(f"{base + i + 1}/{i + 1}" for i in range(4))))
        else:
            f.append((mat, "f " + " ".join(str(base + i + 1) for i in range(4))))

    add_quad([(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], "floor", uv=True)
    add_quad([(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)], "white")
    add_quad([(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)], "bumpy", uv=True)
    add_quad([(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)], "red")
    add_quad([(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)], "green")
    add_quad([(-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3)], "light")
    L.append("o room")
    L += [f"v {x} {y} {z}" for x, y, z in v]
    L += ["vt 0 0", "vt 4 0", "vt 4 4", "vt 0 4"]
    cur = None
    for mat, line in f:
        if mat != cur:
            L.append(f"usemtl {mat}")
            cur = mat
        L.append(line)
    # a unit cube mesh around the origin, negative (relative) indices
    L.append("o cube")
    cube = [(-.5, -.5, -.5), (.5, -.5, -.5), (.5, .5, -.5), (-.5, .5, -.5), (-.5, -.5, .5), (.5, -.5, .5), (.5, .5, .5), (-.5, .5, .5)]
    L += [f"v {x} {y} {z}" for x, y, z in cube]
    faces = [(1, 4, 3, 2), (5, 6, 7, 8), (1, 2, 6, 5), (2, 3, 7, 6), (3, 4, 8, 7), (4, 1, 5, 8)]
    L.append("usemtl tall")
    for q in faces:
        L.append("f " + " ".join(str(i - 9) for i in q))
    L.append("o prism")
    L += ["v -0.2 0 -0.2", "v 0.2 0 -0.2", "v 0 0 0.2", "v 0 0.5 0"]
    L.append("usemtl glass")
    L += ["f -4 -2 -3", "f -4 -3 -1", "f -3 -2 -1", "f -2 -4 -1"]
    if with_instances:
        L += ["instance room 0 0 0 0 0 0 1 1 1",
              "instance cube -0.45 0.6 -0.3 0 20 0 0.55 1.2 0.55",
              "instance cube 0.45 0.3 0.35 0 -15 0 0.6 0.6 0.6",
              "instance prism 0.1 0 0.55 0 0 0 1 1 1"]
    return "\n".join(L) + "\n"


def write_cornell(dirpath, with_instances=True):
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "room.obj"), "w") as f:
        f.write(cornell_obj(with_instances))
    with open(os.path.join(dirpath, "room.mtl"), "w") as f:
        f.write(CORNELL_MTL)
    write_ppm(os.path.join(dirpath, "checker.pnm"), checker())
    yy, xx = np.mgrid[0:16, 0:16]
    bump = (127 + 120 * np.sin(xx * 0.8) * np.cos(yy * 0.8)).astype(np.uint8)
    write_png(os.path.join(dirpath, "bump.png"), bump)
    return os.path.join(dirpath, "room.obj")
