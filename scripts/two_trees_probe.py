#!/usr/bin/env python3
"""Frame time of the recursive shader and of the packet path on the reference's tree, on the GPU-built tree, and on both (mrt_scene_upload_lbvh_exact:
rays walk the second tree, tiles it cannot vouch for are rendered again through the first), with the number of such
tiles.  A tuning aid.   two_trees_probe.py [w h]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    st = torch.cuda.current_stream()
    frames = {}
    for name, kind in (("reference's tree", False), ("GPU-built tree", True), ("both trees", "exact")):
        rt = Raytracer(w, h, device=0)
        rt.build_bvh(scene, lbvh=kind)
        fb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        ids = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        for pose, mode in (("P0", 1), ("P1", 1), ("P0", 0), ("P1", 0)):
            cam = Camera()
            if pose == "P1":
                cam.set_position((6, 3, 12)); cam.look_at((0, 0, 0))
            for _ in range(5):
                rt.render_device(cam, fb.data_ptr(), w, mode=mode, stream=st.cuda_stream, ids_ptr=ids.data_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(20):
                rt.render_device(cam, fb.data_ptr(), w, mode=mode, stream=st.cuda_stream, ids_ptr=ids.data_ptr())
            e1.record(st)
            torch.cuda.synchronize()
            use = rt.second_tree_use()
            frames[(name, pose, mode)] = (fb.cpu().numpy().copy(), ids.cpu().numpy().copy(), dict(rt.ray_counts()))
            print("%dx%d %s %-9s %-18s %.3f ms per frame   second tree walked: %s, rays (packet path: packets) traced again: %d" %
                  (w, h, pose, "recursive" if mode else "packet", name, e0.elapsed_time(e1) / 20, use[0], use[1]))
        rt.close()
    for pose, mode in (("P0", 1), ("P1", 1), ("P0", 0), ("P1", 0)):
        a, b = frames[("reference's tree", pose, mode)], frames[("both trees", pose, mode)]
        print(pose, "recursive" if mode else "packet", "both trees against the reference's: pixels equal", np.array_equal(a[0], b[0]), " hit ids equal", np.array_equal(a[1], b[1]), " ray counts equal", a[2] == b[2])


if __name__ == "__main__":
    main()
