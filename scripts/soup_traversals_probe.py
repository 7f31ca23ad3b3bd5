#!/usr/bin/env python3
"""The two traversals of the reference on the same rays of the 1,000,000-triangle soup: BVH::TraverseRayPacket
(mrt_trace_packets: one wave per 64-ray packet, nodes and triangles read once per wave) and BVH::Traverse
(mrt_trace_rays: every lane walks the tree for itself), both with the scene in global memory.  The rays are the primary
rays of a 3840x2160 frame from a camera INSIDE the soup's bounds (from outside, BVH::Traverse turns every ray away at
the root: raytracer.cpp:947), in 8x8 tiles.  Meant to be run under rocprofv3 --kernel-trace --stats: the kernels'
average durations are the result.  A tuning aid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_lib
    from raytracing_amd import Raytracer, scene_io
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    scene = scene_io.large_scene(maze)
    W, H = 3840, 2160
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    M = oracle_lib.lookat((0, 0.5, -23), (3, 0, -30)).astype(np.float32).reshape(4, 4)
    ys, xs = np.mgrid[0:H, 0:W]
    aspect = H / W
    d = np.stack([-1 + 2 * xs / W, aspect - 2 * aspect * ys / H, np.full(xs.shape, -1.8)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    D = (d @ M[:3, :3]).astype(np.float32)          # rows of M are the camera's axes (column-major matrix read row-wise)
    D = D.reshape(H // 8, 8, W // 8, 8, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 3).astype(np.float32)
    O = np.broadcast_to(M[3, :3], D.shape).astype(np.float32).copy()
    with np.errstate(divide="ignore"):
        rD = (np.float32(1) / D).astype(np.float32)
    t0 = np.full(D.shape[:2], 1e8, np.float32)
    for _ in range(3):
        tp, trip, _, _ = rt.trace_packets(O, D, rD, t0)
    for _ in range(3):
        tr, trir, _, _ = rt.trace_rays(O.reshape(-1, 3), D.reshape(-1, 3), t0.reshape(-1))
    trir = trir.reshape(trip.shape)
    print("rays %d; packet traversal hits %d, single-ray traversal hits %d, same triangle on %d rays" %
          (trip.size, int((trip >= 0).sum()), int((trir >= 0).sum()), int((trip == trir).sum())))
    rt.close()


if __name__ == "__main__":
    main()
