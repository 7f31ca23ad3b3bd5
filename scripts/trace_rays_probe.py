#!/usr/bin/env python3
"""Runs mrt_trace_rays (the per-lane nearest-hit kernel on its own, scene in global memory) on the primary rays of a
1080p default-camera frame; meant to be run under rocprofv3 --kernel-trace to read the kernel time.  A tuning aid:
the kernel takes 0.36 ms where the LDS-resident frame kernel spends 0.26 ms on the same rays."""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracing_amd import Raytracer, scene_io
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
W, H = 1920, 1080
rt = Raytracer(W, H, device=0)
rt.build_bvh(scene)
# primary rays of the default camera, tile order (8x8 tiles, lane = pixel within tile) like the frame kernels
ys, xs = np.mgrid[0:H, 0:W]
aspect = H / W
px = -1 + 2 * xs / W
py = aspect - 2 * aspect * ys / H
D = np.stack([px, py, np.full_like(px, -1.8)], -1).astype(np.float32)
D /= np.linalg.norm(D, axis=-1, keepdims=True)
D = D.reshape(H // 8, 8, W // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 3).astype(np.float32)
O = np.zeros_like(D)
for _ in range(3):
    t, tri, u, v = rt.trace_rays(O, D, np.full(len(D), 1e8, np.float32))
print("hits", int((tri >= 0).sum()))
