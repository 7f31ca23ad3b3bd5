#!/usr/bin/env python3
"""Times the large configurations in isolation, device-resident frames, HIP events, and checks every frame against the
oracle's recorded checksum (tests/golden/full_size.json): c3 = 1,000,000-triangle soup at 3840x2160 packet path,
c4 = maze at 7680x4320 with 16 samples per pixel packet path, c2p = maze 1080p packet path.  A tuning aid.
usage: big_probe.py [c3] [c4] [c2p] [--iters N]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fbsum(fb):
    px = np.ascontiguousarray(fb, np.uint32).reshape(-1).astype(np.uint64)
    return "%016x" % int((px * np.arange(1, px.size + 1, dtype=np.uint64)).sum(dtype=np.uint64))


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["c3"]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    import torch
    from raytracing_amd import Raytracer, scene_io
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "full_size.json")))
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    cam = np.eye(4, dtype=np.float32).reshape(16)
    stream = torch.cuda.current_stream()
    for name in which:
        if name == "c3":
            scene, W, H, g, key = scene_io.large_scene(maze), 3840, 2160, 1, "c3_soup1m_4k_packet"
        elif name == "c4":
            scene, W, H, g, key = maze, 7680, 4320, 4, "c4_maze_8k_16spp_packet"
        else:
            scene, W, H, g, key = maze, 1920, 1080, 1, "c2_maze_1080p_packet"
        rt = Raytracer(W, H, device=0)
        rt.build_bvh(scene)
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        for _ in range(2):
            rt.render_device(cam, fb.data_ptr(), W, mode=0, spp_grid=g, stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            rt.render_device(cam, fb.data_ptr(), W, mode=0, spp_grid=g, stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        c = rt.ray_counts()
        rays = c["primary"] + c["shadow"] + c["secondary"]
        ok = fbsum(fb.cpu().numpy().view(np.uint32)) == golden[key]["fbsum"]
        print("%-4s %dx%d g=%d  %8.3f ms/frame  %11d rays  %8.1f Mrays/s  frame %s" % (name, W, H, g, ms, rays, rays / ms / 1e3,
                                                                                     "equals the oracle's" if ok else "DIFFERS FROM THE ORACLE'S"), flush=True)
        rt.close()
        del fb


if __name__ == "__main__":
    main()
