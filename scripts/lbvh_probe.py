#!/usr/bin/env python3
"""Upload and frame times with the GPU-built linear BVH against the reference tree: default maze at 1080p
and the 1,000,000-triangle soup at 4K.  A tuning aid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import Raytracer, scene_io
    cam = np.eye(4, dtype=np.float32).reshape(16)
    stream = torch.cuda.current_stream()
    for name, scene, W, H in (("maze", scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts")), 1920, 1080),
                              ("soup 1M", scene_io.soup_scene(1000000), 3840, 2160)):
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        frames = {}
        for lbvh in (False, True):
            rt = Raytracer(W, H, device=0)
            rt.build_bvh(scene, lbvh=lbvh)
            t0 = time.time()
            rt.build_bvh(scene, lbvh=lbvh)
            up = time.time() - t0
            line = "%-8s %-9s upload %8.1f ms" % (name, "lbvh" if lbvh else "reference", up * 1e3)
            for mode, mname in ((0, "packet"), (1, "recursive")):
                for _ in range(3):
                    rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=stream.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(10):
                    rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                line += "  %s %.3f ms" % (mname, e0.elapsed_time(e1) / 10)
                frames[(lbvh, mode)] = fb.cpu().numpy().copy()
            print(line, flush=True)
            rt.close()
        for mode in (0, 1):
            print("         mode %d: %.5f %% of pixels differ between the trees" % (mode, 100.0 * (frames[(False, mode)] != frames[(True, mode)]).mean()))


if __name__ == "__main__":
    main()
