#!/usr/bin/env python3
"""Per-tile shader clocks of the persistent recursive kernel (release kernels with the diagnostic accessor: make -C raytracing_amd/csrc diag):
distribution, and the bound a perfect schedule could reach.  A tuning aid."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1       # 0: packet path (needs MRT_RESIDENT_PACKET=1), 1: recursive
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_diag.so")
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    L = binding.lib()
    L.mrt_debug_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    rt = Raytracer(w, h, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    cam = Camera()
    frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3     # 1: a frame in frame order, nothing split; from 2 on the tile order is active
    for _ in range(frames):
        rt.render_device(cam, fb.data_ptr(), w, mode=mode, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    n = (w // 8) * (h // 8)
    cost = np.zeros(n, np.uint32)
    got = L.mrt_debug_tile_costs(rt.h, cost.ctypes.data_as(C.c_void_p), n)
    c = cost[:got].astype(np.float64)
    waves = 4096
    print("tiles %d  sum %.3g clocks  mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" %
          (got, c.sum(), c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max()))
    print("ten most expensive:", " ".join("%d:%.0f" % (i, c[i]) for i in np.argsort(-c)[:10]))
    print("perfect balance over %d waves: %.0f clocks per wave; heaviest tile: %.0f clocks" % (waves, c.sum() / waves, c.max()))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.save(os.path.join(out, "tile_costs_mode%d_%dx%d.npy" % (mode, w, h)), cost[:got])
    grid = c.reshape(h // 8, w // 8)
    rows = grid.mean(axis=1)
    print("mean cost per tile row (every 9th):", " ".join("%.0f" % r for r in rows[::9]))


if __name__ == "__main__":
    main()
