#!/usr/bin/env python3
"""What the waves of the rasterizer's tile kernel do on the 1,000,000-triangle 4K frame (or the maze), from the
diagnostic build's counters (make -C raytracing_amd/csrc prof).  A tuning aid.   raster_counts.py [maze]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_prof.so")
    import torch
    from raytracing_amd import Raytracer, scene_io
    L = binding.lib()
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    scene = maze if "maze" in sys.argv else scene_io.large_scene(maze)
    W, H = 3840, 2160
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    cam = np.eye(4, dtype=np.float32).reshape(16)
    out = np.zeros(16, np.uint64)
    rt.raster_device(cam, fb.data_ptr(), W, stream=torch.cuda.current_stream().cuda_stream)
    rt.raster_wait()
    L.mrt_debug_raster_counts(out.ctypes.data_as(C.c_void_p))
    rt.raster_device(cam, fb.data_ptr(), W, stream=torch.cuda.current_stream().cuda_stream)
    rt.raster_wait()
    L.mrt_debug_raster_counts(out.ctypes.data_as(C.c_void_p))
    waves, entries, cand, drawn, blocks, won = (int(x) for x in out[:6])
    print("tile kernel, one frame: %d waves; per wave: %.1f list entries looked through, %.1f candidates, %.1f drawn, %.1f blocks tested, %.1f blocks with a pixel won"
          % (waves, entries / waves, cand / waves, drawn / waves, blocks / waves, won / waves))
    clk = [float(out[k]) / waves for k in (8, 12, 9, 10, 11)]
    print("shader clocks per wave: %.0f in all = %.0f before the first batch + %.0f first looks + %.0f drawing + %.0f resolve" % tuple(clk))
    span = int(out[6]) - (~int(out[7]) & 0xffffffffffffffff)
    print("first wave's start to last wave's end: %d clocks; the waves' lives add up to %.1f times that (waves alive at a time, on average; 8192 fit)" % (span, float(out[8]) / span))
    print("longest list %d entries, longest-lived wave %d clocks, %d waves with lists over 628 entries" % (int(out[13]), int(out[14]), int(out[15])))
    rt.close()


if __name__ == "__main__":
    main()
