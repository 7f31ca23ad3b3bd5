#!/usr/bin/env python3
"""The 1,000,000-triangle soup at 3840x2160 through the packet path and the recursive shader: frame time on the reference's
tree, on the GPU-built tree alone (not certified: pixels may differ) and uploaded with mrt_scene_upload_lbvh_exact, with
the upload's wall time.  A tuning aid."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import Raytracer, scene_io
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    soup = scene_io.large_scene(maze)
    W, H = 3840, 2160
    cam = np.eye(4, dtype=np.float32).reshape(16)
    st = torch.cuda.current_stream()
    frames = {}
    for name, kind in (("reference's tree", False), ("GPU-built tree", True), ("both trees", "exact")):
        rt = Raytracer(W, H, device=0)
        t0 = time.time()
        rt.build_bvh(soup, lbvh=kind)
        up = time.time() - t0
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        ids = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        for mode in (0,):
            for _ in range(2):
                rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=st.cuda_stream, ids_ptr=ids.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=st.cuda_stream, ids_ptr=ids.data_ptr())
            e1.record(st)
            torch.cuda.synchronize()
            frames[(name, mode)] = (fb.cpu().numpy().copy(), ids.cpu().numpy().copy(), dict(rt.ray_counts()))
            print("%-18s mode %d: %.3f ms per frame (upload %.2f s)  second tree: %s" % (name, mode, e0.elapsed_time(e1) / 5, up, rt.second_tree_use()), flush=True)
        rt.close()
    for mode in (0,):
        a = frames[("reference's tree", mode)]
        for other in ("GPU-built tree", "both trees"):
            b = frames[(other, mode)]
            print("mode %d %s against the reference's: %d pixels, %d hit ids differ; ray counts equal %s" % (mode, other, int((a[0] != b[0]).sum()), int((a[1] != b[1]).sum()), a[2] == b[2]))


if __name__ == "__main__":
    main()
