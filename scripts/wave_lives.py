#!/usr/bin/env python3
"""When the persistent recursive kernel's waves start and stop (diagnostic build: make -C raytracing_amd/csrc lives):
how long the frame's waves are alive, how much of the frame they spend gone, and which workgroups and tile queues
leave early or late.  A tuning aid."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_lives.so")
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    L = binding.lib()
    L.mrt_debug_wave_lives.argtypes = [C.c_void_p, C.c_void_p]
    w, h = 1920, 1080
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    rt = Raytracer(w, h, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream()
    cam = Camera()
    for _ in range(6):
        rt.render_device(cam, fb.data_ptr(), w, mode=1, stream=st.cuda_stream)
    torch.cuda.synchronize()
    out = np.zeros(3 * 4096, np.uint64)
    L.mrt_debug_wave_lives(rt.h, out.ctypes.data_as(C.c_void_p))
    rec = out.reshape(4096, 3).astype(np.float64)
    # the clock counters of different XCDs are not aligned, so only differences taken inside one wave (or one
    # workgroup) mean anything: a wave's life = its exit - its entry
    enter, leave, tiles = rec[:, 0].reshape(256, 16), rec[:, 1].reshape(256, 16), rec[:, 2].reshape(256, 16)
    life = leave - enter
    print("wave life in clocks: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f;  mean / max = %.3f" % (
        life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max(), life.mean() / life.max()))
    print("tiles per wave: min %d median %d max %d;  entry spread inside a workgroup: at most %.0f clocks" % (
        tiles.min(), np.median(tiles), tiles.max(), (enter.max(1) - enter.min(1)).max()))
    wg = life.max(1)
    print("longest life per workgroup: min %.0f median %.0f max %.0f" % (wg.min(), np.median(wg), wg.max()))
    q = np.arange(256) % 32
    per_queue = np.array([wg[q == k].max() for k in range(32)])
    print("longest life per tile queue: min %.0f median %.0f max %.0f" % (per_queue.min(), np.median(per_queue), per_queue.max()))
    np.save(os.path.join(ROOT, "gpurun_out", "wave_lives.npy"), rec)


if __name__ == "__main__":
    main()
