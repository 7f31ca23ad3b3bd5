"""Differential fuzzing of the scene loaders: the C++ host layer's Scene::Add (OBJ / MTL / .bin texture
caches, raytracing_amd/host/mrt_host.hpp) against the oracle's restatement of the reference loader
(scenedata.cpp:77-271, surface.cpp:39-59) on generated scenes, and robustness of the host loader on
damaged files.  No GPU needed: the driver only parses and dumps the scene."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from conftest import ROOT, bits
from raytracing_amd import scene_io

DRIVER = os.path.join(ROOT, "raytracing_amd", "host", "mrt_game")
pytestmark = pytest.mark.skipif(not os.path.exists(DRIVER), reason="driver not built")


def _write_bin(path, rng, w, h):
    idx = rng.integers(0, 256, w * h, dtype=np.uint8)
    pal = rng.integers(0, 1 << 24, 32 * 256, dtype=np.uint32)
    with open(path, "wb") as f:
        f.write(np.int32([w, h]).tobytes())
        f.write(idx.tobytes())
        f.write(pal.tobytes())


def _num(rng, x):
    style = rng.integers(0, 4)
    if style == 0:
        return "%.6f" % x
    if style == 1:
        return "%.3e" % x
    if style == 2:
        return "%g" % x
    return repr(float(np.float32(x)))


def make_scene(tmp, seed):
    """A random OBJ + MTL + textures inside the subset the reference parses: triangles only, every
    face after a usemtl, one face format (v/t/n or v//n) per usemtl section."""
    rng = np.random.default_rng(seed)
    d = tmp / ("s%d" % seed)
    (d / "textures").mkdir(parents=True)
    n_tex = int(rng.integers(1, 4))
    for i in range(n_tex):
        _write_bin(str(d / "textures" / ("t%d.bin" % i)), rng, int(rng.choice([4, 16, 64])), int(rng.choice([4, 8, 32])))
    n_mat = int(rng.integers(1, 5))
    eol = "\r\n" if rng.integers(0, 2) else "\n"
    with open(d / "scene.mtl", "w", newline="") as f:
        f.write("# materials" + eol)
        for i in range(n_mat):
            f.write("newmtl mat%d%s" % (i, eol))
            f.write("Kd 0.5 0.5 0.5" + eol)
            f.write("map_Kd t%d.png%s" % (int(rng.integers(0, n_tex)), eol))
            if rng.integers(0, 2):
                f.write(eol)
    nv, nn, nt = int(rng.integers(3, 60)), int(rng.integers(1, 20)), int(rng.integers(1, 30))
    lines = ["# generated", "mtllib scene.mtl"]
    if rng.integers(0, 2):
        lines.append("o thing")
    for _ in range(nv):
        p = rng.uniform(-10, 10, 3)
        lines.append("v " + " ".join(_num(rng, x) for x in p))
    for _ in range(nt):
        t = rng.uniform(0, 1, 2)
        lines.append("vt " + " ".join(_num(rng, x) for x in t))
    for _ in range(nn):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        lines.append("vn " + " ".join(_num(rng, x) for x in n))
    for s in range(int(rng.integers(1, 5))):
        if rng.integers(0, 2):
            lines.append("g group%d" % s)
        lines.append("usemtl mat%d" % int(rng.integers(0, n_mat)))
        if rng.integers(0, 3) == 0:
            lines.append("s off")
        with_uv = bool(rng.integers(0, 2))
        for _ in range(int(rng.integers(1, 40))):
            c = []
            vs = rng.choice(nv, 3, replace=False) + 1
            for v in vs:
                n = int(rng.integers(1, nn + 1))
                c.append("%d/%d/%d" % (v, int(rng.integers(1, nt + 1)), n) if with_uv else "%d//%d" % (v, n))
            lines.append("f " + " ".join(c))
        if rng.integers(0, 4) == 0:
            lines.append("")
    with open(d / "scene.obj", "w", newline="") as f:
        f.write(eol.join(lines) + eol)
    return d


def _host_scene(d, tmp, scale=None):
    out = str(tmp / (d.name + ".mrts"))
    cmd = [DRIVER, "--scene", str(d / "scene.obj"), "--dump-scene", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return scene_io.load_scene(out)


@pytest.mark.parametrize("seed", range(24))
def test_host_loader_equals_oracle_loader(tmp_path, seed):
    d = make_scene(tmp_path, seed)
    o = oracle_lib.Oracle()
    o.load_obj(str(d / "scene.obj"))
    ref = o.export_scene()
    got = _host_scene(d, tmp_path)
    assert len(got["textures"]) == len(ref["textures"]) and len(got["meshes"]) == len(ref["meshes"])
    for a, b in zip(got["textures"], ref["textures"]):
        assert (a["w"], a["h"]) == (b["w"], b["h"])
        assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["pal"], b["pal"])
    for a, b in zip(got["meshes"], ref["meshes"]):
        assert a["texture"] == b["texture"]
        assert np.array_equal(a["tri"], b["tri"])
        for k in ("pos", "uv", "N"):
            assert np.array_equal(bits(np.asarray(a[k], np.float32)), bits(np.asarray(b[k], np.float32))), k


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode in (0, 1), "driver died with %d: %s" % (r.returncode, r.stderr[-300:])
    return r


@pytest.mark.parametrize("seed", range(12))
def test_host_loader_survives_damaged_files(tmp_path, seed):
    """Truncated, bit-flipped and index-damaged inputs end in a clean load or an error message,
    never a crash."""
    rng = np.random.default_rng(1000 + seed)
    d = make_scene(tmp_path, seed)
    out = str(tmp_path / "o.mrts")
    obj = (d / "scene.obj").read_bytes()
    kind = seed % 4
    if kind == 0:                                  # cut the file anywhere
        (d / "scene.obj").write_bytes(obj[:int(rng.integers(0, len(obj)))])
    elif kind == 1:                                # flip bytes
        b = bytearray(obj)
        for _ in range(20):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        (d / "scene.obj").write_bytes(bytes(b))
    elif kind == 2:                                # faces with indices far out of range, zero and negative
        (d / "scene.obj").write_bytes(obj + b"usemtl mat0\nf 99999/1/1 2/99999/1 3/1/99999\nf 0/0/0 -1/-1/-1 -7//-7\nf 1 2 3\nf\n")
    else:                                          # truncated texture cache
        t = next((d / "textures").iterdir())
        t.write_bytes(t.read_bytes()[:int(rng.integers(0, 200))])
    _run([DRIVER, "--scene", str(d / "scene.obj"), "--dump-scene", out])
    long_line = tmp_path / "long.obj"
    long_line.write_bytes(b"mtllib " + b"x" * 5000 + b"\nusemtl " + b"y" * 5000 + b"\nv " + b"1" * 5000 + b"\nf " + b"1/1/1 " * 2000 + b"\n")
    _run([DRIVER, "--scene", str(long_line), "--dump-scene", out])
