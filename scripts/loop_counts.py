#!/usr/bin/env python3
"""Loop iteration counts of the frame kernels from the diagnostic build (make -C raytracing_amd/csrc prof):
for every instrumented site, how often a wave executed it and how many lanes were active.  A tuning aid."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SITES = {0: "nearest: inner-node step", 1: "nearest: leaf triangle", 2: "nearest: call past root gate",
         12: "nearest: tri past back-face", 13: "nearest: tri exact tail",
         3: "shadow: work item", 4: "shadow: inner node", 5: "shadow: leaf triangle",
         6: "shared tri past back-face", 7: "shared tri exact tail",
         8: "packet: work item", 9: "packet: inner visit", 10: "packet: leaf triangle", 11: "shader round",
         15: "shadow: inner node exact fallback", 14: "longest wave life | latest scene-copy end", 18: "sum: scene copy clocks | clocks in tiles", 19: "sum: wave life clocks | waves",
         20: "nearest: group box test", 21: "shared leaf: group box test",
         16: "clocks/16 in nearest-hit traversal", 17: "clocks/16 in direct light + shadow"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp-grid", type=int, default=1)
    args = ap.parse_args()
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_prof.so")
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    L = binding.lib()
    L.mrt_debug_loop_counts.argtypes = [C.c_void_p, C.c_void_p]
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    rt = Raytracer(args.w, args.h, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((args.h, args.w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    cam = Camera()
    for mode, name in ((0, "packet"), (1, "recursive")):
        rt.render_device(cam, fb.data_ptr(), args.w, mode=mode, spp_grid=args.spp_grid, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        c = rt.ray_counts()
        out = np.zeros(44, np.uint64)
        L.mrt_debug_loop_counts(rt.h, out.ctypes.data_as(C.c_void_p))
        print("%s: %s" % (name, c))
        for s in sorted(SITES):
            w, l = int(out[2 * s]), int(out[2 * s + 1])
            if l:
                print("  %-32s wave %10d  lane %11d  lanes/wave %5.1f" % (SITES[s], w, l, l / max(w, 1)))


if __name__ == "__main__":
    main()
