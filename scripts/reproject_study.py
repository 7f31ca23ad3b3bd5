#!/usr/bin/env python3
"""How well do the previous frame's tile costs predict this frame's while the camera walks?  Renders the maze from a
row of camera positions (recursive shader, nothing split: MRT_SPLIT_BUDGET=0 on the diag build), reads the per-tile
clocks, and scores three predictors of frame k+1 from frame k: the tile's own cost, the largest cost within a radius,
and the cost carried along with the scene point the tile's primary rays hit (one or several sample pixels per tile).
A tuning aid.   reproject_study.py [w h]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MRT_SPLIT_BUDGET", "0")


def view(cam16, w, h):
    m = np.asarray(cam16, np.float64).reshape(4, 4).T          # columns c[0..3] -> matrix acting on column vectors
    asp = h / w
    def mul(v):
        return (m @ np.array(v, np.float64))[:3]
    p0, p1, p2 = mul([-1, asp, -1.8, 1]), mul([1, asp, -1.8, 1]), mul([-1, -asp, -1.8, 1])
    return p0, p1 - p0, p2 - p0, m[:3, 3].copy()


def main():
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_diag.so")
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    L = binding.lib()
    L.mrt_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 600)
    tx, ty = w // 8, h // 8
    n = tx * ty
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    rt = Raytracer(w, h, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream()
    frames = []
    for k in range(8):
        cam = Camera()
        cam.set_position((0.16 * k, 0.0, 0.0))
        for _ in range(2):
            rt.render_device(cam, fb.data_ptr(), w, mode=1, stream=st.cuda_stream)
        torch.cuda.synchronize()
        cost = np.zeros(n, np.uint32)
        L.mrt_debug_tile_costs(rt.h, cost.ctypes.data_as(C.c_void_p), n)
        assert not (cost >> 31).any(), "a tile was split"
        p0, d10, d20, O = view(np.asarray(cam.transform, np.float32).reshape(16), w, h)
        # primary hits of every pixel
        ys, xs = np.mgrid[0:h, 0:w]
        plane = p0 + (xs[..., None] / w) * d10 + (ys[..., None] / h) * d20
        D = plane - O
        D /= np.linalg.norm(D, axis=-1, keepdims=True)
        Of = np.broadcast_to(O.astype(np.float32), (h * w, 3))
        t, tri, _, _ = rt.trace_rays(Of, D.reshape(-1, 3).astype(np.float32), np.full(h * w, 1e8, np.float32))
        P = O + D.reshape(-1, 3) * t[:, None].astype(np.float64)
        frames.append(dict(cost=cost.astype(np.float64).reshape(ty, tx), P=P.reshape(h, w, 3), hit=(tri >= 0).reshape(h, w), view=(p0, d10, d20, O)))

    def project(P, v):
        p0, d10, d20, O = v
        nrm = np.cross(d10, d20)
        q = P - O
        qn, pn = q @ nrm, (p0 - O) @ nrm
        ok = qn * pn > 0
        s = np.where(ok, pn / np.where(ok, qn, 1.0), 0.0)
        r = q * s[..., None] - (p0 - O)
        return (r @ d10) / (d10 @ d10) * w, (r @ d20) / (d20 @ d20) * h, ok

    def carried(a, b, pixels):
        out = np.zeros((ty, tx))
        for (py, px) in pixels:
            P = a["P"][py::8, px::8]
            x, y, ok = project(P, b["view"])
            ok &= a["hit"][py::8, px::8]
            for dy in (-2.5 - (py - 3), 3.5 - (py - 3)) if len(pixels) == 1 else (0.0,):
                for dx in (-2.5 - (px - 3), 3.5 - (px - 3)) if len(pixels) == 1 else (0.0,):
                    jx, jy = np.floor((x + dx) / 8).astype(int), np.floor((y + dy) / 8).astype(int)
                    good = ok & (jx >= 0) & (jx < tx) & (jy >= 0) & (jy < ty)
                    np.maximum.at(out, (jy[good], jx[good]), a["cost"][good])
        return out

    def spread(c, r):
        out = c.copy()
        for d in range(1, r + 1):
            out[:, d:] = np.maximum(out[:, d:], c[:, :-d]); out[:, :-d] = np.maximum(out[:, :-d], c[:, d:])
        c2 = out.copy()
        for d in range(1, r + 1):
            out[d:, :] = np.maximum(out[d:, :], c2[:-d, :]); out[:-d, :] = np.maximum(out[:-d, :], c2[d:, :])
        return out

    def score(pred, actual):
        top = actual.max()
        exp = actual >= 0.5 * top
        found = (pred >= 0.35 * top) & exp
        false = (pred >= 0.35 * top) & (actual < 0.2 * top)
        return exp.sum(), found.sum(), false.sum()

    radius = max(1, tx * 40 // 1000)
    one = [(3, 3)]
    four = [(1, 1), (1, 6), (6, 1), (6, 6)]
    sixteen = [(y, x) for y in (0, 2, 5, 7) for x in (0, 2, 5, 7)]
    print("%dx%d, tiles at least half as expensive as the frame's most expensive one: found by the predictor | flagged although under a fifth" % (w, h))
    for k in range(len(frames) - 1):
        a, b = frames[k], frames[k + 1]
        row = ["%d->%d" % (k, k + 1)]
        for name, pred in (("own", a["cost"]), ("spread r=%d x0.75" % radius, np.maximum(a["cost"], 0.75 * spread(a["cost"], radius))),
                           ("carried 1", carried(a, b, one)), ("carried 4", carried(a, b, four)), ("carried 16", carried(a, b, sixteen)),
                           ("carried 16 + own", np.maximum(a["cost"], carried(a, b, sixteen)))):
            e, f, x = score(pred, b["cost"])
            # the frame's tail if the tiles predicted within 6/8 octave of the predicted maximum run as sixteenths (0.62 of their time)
            split = pred >= pred.max() / 2 ** 0.75
            tail = np.where(split, 0.62 * b["cost"], b["cost"]).max()
            row.append("%s: %d/%d | %d, %d split, tail %.0f" % (name, f, e, x, split.sum(), tail))
        print("   ".join(row) + "   (all split: %.0f, none: %.0f)" % (0.62 * b["cost"].max(), b["cost"].max()))
    rt.close()


if __name__ == "__main__":
    main()
