#!/usr/bin/env python3
"""Times device-resident rasterizer frames of the maze (the reference's own scene) at a few sizes and poses: kernel
time from HIP events around mrt_raster_render_device.  A tuning aid.    raster_probe.py [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    st = torch.cuda.current_stream()
    for W, H in ((800, 600), (1920, 1080), (3840, 2160)):
        rt = Raytracer(W, H, device=0)
        rt.build_bvh(maze)
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        for name, pose in (("default", None), ("outside", ((6, 3, 12), (0, 0, 0)))):
            cam = Camera()
            if pose:
                cam.set_position(pose[0])
                cam.look_at(pose[1])
            for _ in range(3):
                rt.raster_device(cam, fb.data_ptr(), W, stream=st.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(frames):
                rt.raster_device(cam, fb.data_ptr(), W, stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            print("maze %dx%d %-8s %.4f ms per frame" % (W, H, name, e0.elapsed_time(e1) / frames))
        rt.close()


if __name__ == "__main__":
    main()
