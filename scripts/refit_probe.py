#!/usr/bin/env python3
"""mrt_scene_refit on a large scene: wall time of the call (vertex upload over PCIe included) for moved vertices, next to
a full mrt_scene_upload of the same moved scene, and the frame both give.  Run under `rocprofv3 --kernel-trace --stats`
for the device time of the refit kernels.  usage: refit_probe.py [triangles] [repeats]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def moved(scene, amount, seed):
    f = np.float32
    rng = np.random.RandomState(seed)
    out = {"textures": scene["textures"], "meshes": []}
    for m in scene["meshes"]:
        pos = (np.asarray(m["pos"], f) + rng.uniform(-amount, amount, np.asarray(m["pos"]).shape).astype(f)).astype(f)
        tri = np.asarray(m["tri"], np.int32).reshape(-1, 3)
        e1, e2 = pos[tri[:, 1]] - pos[tri[:, 0]], pos[tri[:, 2]] - pos[tri[:, 0]]
        cr = np.cross(e1, e2).astype(f)
        N = (cr / np.maximum(np.linalg.norm(cr, axis=1, keepdims=True), 1e-30)).astype(f)
        out["meshes"].append(dict(m, pos=pos, N=N, bounds=np.concatenate([pos.min(0), pos.max(0)]).astype(f)))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from raytracing_amd import Raytracer, scene_io
    W, H = 1920, 1080
    scene = scene_io.soup_scene(n)
    rt = Raytracer(W, H, device=0)
    t0 = time.perf_counter()
    rt.build_bvh(scene)
    print("mrt_scene_upload (%d triangles, reference tree): %.1f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
    cam = np.eye(4, dtype=np.float32).reshape(16)
    rt.render(cam, want_ids=False)
    times = []
    for k in range(reps):
        m = moved(scene, 0.05, k)
        t0 = time.perf_counter()
        rt.refit(m)
        times.append((time.perf_counter() - t0) * 1e3)
    print("mrt_scene_refit, vertices moved by up to 0.05: " + " ".join("%.1f" % t for t in times) + " ms (wall, pageable host arrays)", flush=True)
    fb_refit, ids_refit = rt.render(cam)
    t0 = time.perf_counter()
    rt.build_bvh(m)
    print("mrt_scene_upload of the moved scene: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    fb_full, ids_full = rt.render(cam)
    print("frame after the refit against a fresh upload of the moved scene: %d of %d pixels differ, %d hit ids differ (other tree: exact ties may name the other triangle)"
          % (int((fb_refit != fb_full).sum()), fb_full.size, int((ids_refit != ids_full).sum())))
    rt.close()


if __name__ == "__main__":
    main()
