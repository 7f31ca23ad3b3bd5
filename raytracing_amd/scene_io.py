"""Reader/writer of the flattened scene container (.mrts) the host driver and the tests share.

The reference describes a scene as OBJ + MTL + per-texture ``.bin`` caches
(scenedata.cpp:77-271, surface.cpp:39-59).  A ``.mrts`` file holds what those loaders
produce -- the ``Mesh`` arrays of scenedata.h:143-167 and the ``Surface8`` contents of
surface.h:41-55 -- so that a machine without the asset files can still replay the default
scene.  Layout (little endian)::

    char[4] "MRTS", u32 version = 1, u32 n_textures, u32 n_meshes
    per texture: i32 w, i32 h, u8 idx[w*h], u32 pal[32*256]
    per mesh:    i32 verts, i32 tris, i32 texture, f32 refl, f32 refr, f32 rindex, f32 bounds[6],
                 f32 pos[verts*3], f32 uv[verts*2], f32 norm[verts*3], f32 N[tris*3], i32 tri[tris*3]
"""
import struct

import numpy as np

MAGIC = b"MRTS"


def save_scene(path, scene):
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<III", 1, len(scene["textures"]), len(scene["meshes"])))
        for t in scene["textures"]:
            f.write(struct.pack("<ii", t["w"], t["h"]))
            f.write(np.ascontiguousarray(t["idx"], np.uint8).tobytes())
            f.write(np.ascontiguousarray(t["pal"], "<u4").tobytes())
        for m in scene["meshes"]:
            pos = np.ascontiguousarray(m["pos"], "<f4").reshape(-1, 3)
            N = np.ascontiguousarray(m["N"], "<f4").reshape(-1, 3)
            f.write(struct.pack("<iiifff", len(pos), len(N), int(m["texture"]), m["refl"], m["refr"], m["rindex"]))
            f.write(np.ascontiguousarray(m["bounds"], "<f4").tobytes())
            f.write(pos.tobytes())
            f.write(np.ascontiguousarray(m["uv"], "<f4").tobytes())
            f.write(np.ascontiguousarray(m["norm"], "<f4").tobytes())
            f.write(N.tobytes())
            f.write(np.ascontiguousarray(m["tri"], "<i4").tobytes())


def load_scene(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != MAGIC:
        raise ValueError("%s: not a .mrts scene" % path)
    version, ntex, nmesh = struct.unpack_from("<III", buf, 4)
    if version != 1:
        raise ValueError("%s: unsupported .mrts version %d" % (path, version))
    off = 16
    scene = {"textures": [], "meshes": []}

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(buf, dtype, count, off).copy()
        off += a.nbytes
        return a

    for _ in range(ntex):
        w, h = struct.unpack_from("<ii", buf, off)
        off += 8
        scene["textures"].append({"w": w, "h": h, "idx": take(np.uint8, w * h), "pal": take("<u4", 32 * 256)})
    for _ in range(nmesh):
        verts, tris, tex, refl, refr, rindex = struct.unpack_from("<iiifff", buf, off)
        off += 24
        scene["meshes"].append({
            "texture": tex, "refl": refl, "refr": refr, "rindex": rindex,
            "bounds": take("<f4", 6),
            "pos": take("<f4", verts * 3).reshape(-1, 3), "uv": take("<f4", verts * 2).reshape(-1, 2),
            "norm": take("<f4", verts * 3).reshape(-1, 3), "N": take("<f4", tris * 3).reshape(-1, 3),
            "tri": take("<i4", tris * 3).reshape(-1, 3)})
    return scene


def soup_scene(n_tris, seed=12345, texture=None):
    """The synthetic triangle soup of the large-scene configuration.

    LCG ``s = s*1664525 + 1013904223`` (32 bit), ``rnd = (s >> 8) * 2**-24``; per triangle a
    centre ``(rnd*40-20, rnd*4-2, -rnd*40-3)``, three vertices ``centre + (rnd-0.5)*sz`` per
    component with ``sz = 40 / cbrt(n)``, uv ``rnd*0.999``, face normal
    ``normalize(cross(p1-p0, p2-p0))``; material (0, 0, 1.0).  All arithmetic in float32.
    """
    f = np.float32
    n_rnd = n_tris * (3 + 9 + 6)
    # vectorised LCG: s_k = a^k s_0 + c (a^k - 1)/(a - 1) mod 2^32, by repeated doubling
    a, c = np.uint64(1664525), np.uint64(1013904223)
    mask = np.uint64(0xFFFFFFFF)
    s = np.empty(n_rnd, np.uint64)
    mult = np.ones(1, np.uint64)
    add = np.zeros(1, np.uint64)
    # generate sequentially in blocks (simple and exact)
    state = np.uint64(seed)
    A = [a]
    C = [c]
    while (1 << len(A)) < max(n_rnd, 2):
        A.append((A[-1] * A[-1]) & mask)
        C.append((C[-1] * (A[-2] + np.uint64(1))) & mask)
    # s[k] for k = 1..n_rnd via binary decomposition applied to all k at once
    k = np.arange(1, n_rnd + 1, dtype=np.uint64)
    cur = np.full(n_rnd, state, np.uint64)
    for bit in range(len(A)):
        sel = ((k >> np.uint64(bit)) & np.uint64(1)).astype(bool)
        cur[sel] = (cur[sel] * A[bit] + C[bit]) & mask
    del mult, add, s
    r = ((cur >> np.uint64(8)).astype(np.float32) * f(1.0 / 16777216.0)).reshape(n_tris, 18)
    centre = np.stack([r[:, 0] * f(40) - f(20), r[:, 1] * f(4) - f(2), -(r[:, 2] * f(40)) - f(3)], 1).astype(f)
    sz = f(f(40.0) / f(np.cbrt(np.float64(n_tris))))
    pos = (centre[:, None, :] + (r[:, 3:12].reshape(n_tris, 3, 3) - f(0.5)) * sz).astype(f)
    uv = (r[:, 12:18].reshape(n_tris, 3, 2) * f(0.999)).astype(f)
    e1 = pos[:, 1] - pos[:, 0]
    e2 = pos[:, 2] - pos[:, 0]
    cr = np.stack([e1[:, 1] * e2[:, 2] - e2[:, 1] * e1[:, 2], e1[:, 2] * e2[:, 0] - e2[:, 2] * e1[:, 0],
                   e1[:, 0] * e2[:, 1] - e2[:, 0] * e1[:, 1]], 1).astype(f)
    sq = (cr[:, 0] * cr[:, 0] + cr[:, 1] * cr[:, 1] + cr[:, 2] * cr[:, 2]).astype(f)
    N = (cr * (f(1.0) / np.sqrt(sq))[:, None]).astype(f)
    flat = pos.reshape(-1, 3)
    if texture is None:
        idx = (np.arange(64 * 64, dtype=np.uint32) * 7 % 251).astype(np.uint8)
        pal = (np.arange(32 * 256, dtype=np.uint32) * 2654435761 & 0xFFFFFF).astype(np.uint32)
        texture = {"w": 64, "h": 64, "idx": idx, "pal": pal}
    mesh = {"pos": flat, "uv": uv.reshape(-1, 2), "norm": np.repeat(N, 3, 0), "N": N,
            "tri": np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3), "texture": 0,
            "bounds": np.concatenate([flat.min(0), flat.max(0)]).astype(f), "refl": 0.0, "refr": 0.0, "rindex": 1.0}
    return {"textures": [texture], "meshes": [mesh]}


def large_scene(maze_scene, n_tris=1000000):
    """The large-scene and rasterizer configurations' input: the triangle soup above carrying the default scene's
    512 x 512 grass texture (the second texture of the maze: assets/textures/grass.bin)."""
    grass = next(t for t in maze_scene["textures"] if int(t["w"]) == 512 and int(t["h"]) == 512)
    return soup_scene(n_tris, texture={"w": int(grass["w"]), "h": int(grass["h"]), "idx": grass["idx"], "pal": grass["pal"]})
