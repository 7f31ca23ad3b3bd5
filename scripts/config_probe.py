#!/usr/bin/env python3
"""Renders a few frames of ONE benchmark configuration and prints the kernel time from HIP events: the program the
per-configuration rocprofv3 passes of profiles/run_profile_r02.sh run.

    config_probe.py c2|c2p|c3|c4|c5 [frames] [move]       move: the eye moves a little every frame
      c2  maze 1920x1080 recursive (uploaded with mrt_scene_upload_lbvh_exact, like bench.py's headline)
      c2r the same through mrt_scene_upload      c2p maze 1920x1080 packet       c4  maze 7680x4320, 4x4 samples, packet
      c3  1,000,000-triangle soup 3840x2160 packet                      c5  the same soup through the rasterizer
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    move = len(sys.argv) > 3 and sys.argv[3] == "move"
    import torch
    from raytracing_amd import Raytracer, scene_io
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    W, H, mode, grid, scene = {"c2": (1920, 1080, 1, 1, maze), "c2r": (1920, 1080, 1, 1, maze), "c2p": (1920, 1080, 0, 1, maze), "c4": (7680, 4320, 0, 4, maze),
                               "c3": (3840, 2160, 0, 1, None), "c5": (3840, 2160, -1, 1, None)}[cfg]
    if scene is None:
        scene = scene_io.large_scene(maze)
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene, lbvh="exact" if cfg in ("c2", "c2p", "c4") else False)       # (as bench.py uploads them)
    cam = np.eye(4, dtype=np.float32).reshape(16)
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream()

    def frame():
        if move:
            cam[12] += np.float32(1e-3)
        if mode < 0:
            rt.raster_device(cam, fb.data_ptr(), W, stream=st.cuda_stream)
        else:
            rt.render_device(cam, fb.data_ptr(), W, mode=mode, spp_grid=grid, stream=st.cuda_stream)
    for _ in range(2):
        frame()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(frames):
        frame()
    e1.record(st)
    torch.cuda.synchronize()
    print("%s: %.4f ms per frame over %d frames" % (cfg, e0.elapsed_time(e1) / frames, frames))


if __name__ == "__main__":
    main()
