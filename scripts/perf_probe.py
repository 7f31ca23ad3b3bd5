#!/usr/bin/env python3
"""Times the frame kernels in isolation (device-resident output, HIP events): packet / recursive,
with and without lights, for a resolution and pose.  A tuning aid, not part of the benchmark."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pose", default="P0")
    ap.add_argument("--only", default="", help="packet|recursive")
    ap.add_argument("--lights", default="", help="0|1: only this light count")
    ap.add_argument("--one-tree", action="store_true", help="upload with mrt_scene_upload (the reference's tree alone) instead of mrt_scene_upload_lbvh_exact")
    ap.add_argument("--walk", type=float, default=0.0,
                    help="move the camera this many units sideways every frame (the frame loop walks 10 units/s, game.cpp:57-62)")
    ap.add_argument("--opaque", action="store_true", help="set refl = refr = 0 on every mesh (primary and shadow rays only)")
    args = ap.parse_args()
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    if args.opaque:
        for m in scene["meshes"]:
            m["refl"], m["refr"] = 0.0, 0.0
    cam = Camera()
    if args.pose == "P1":
        cam.set_position((6, 3, 12))
        cam.look_at((0, 0, 0))
    rt = Raytracer(args.w, args.h, device=0)
    rt.build_bvh(scene, lbvh=False if args.one_tree else "exact")
    fb = torch.zeros((args.h, args.w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    one_light = np.float32([[0, 5, 0, 1, 1, 1, 1]])
    for lights_name, lights in (("1 light", one_light), ("no light", np.zeros((0, 7), np.float32))):
        if args.lights and (args.lights == "1") != (len(lights) == 1):
            continue
        rt.set_lights(lights)
        for mode, name in ((0, "packet"), (1, "recursive")):
            if args.only and args.only != name:
                continue
            for _ in range(3):
                rt.render_device(cam, fb.data_ptr(), args.w, mode=mode, stream=stream.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(args.iters):
                if args.walk:
                    x, y, z = cam.get_position()
                    step = args.walk if (i // 25) % 2 == 0 else -args.walk      # there and back
                    cam.set_position((x + step, y, z))
                rt.render_device(cam, fb.data_ptr(), args.w, mode=mode, stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            c = rt.ray_counts()
            rays = c["primary"] + c["shadow"] + c["secondary"]
            print("%-9s %-8s resident=%s  %8.3f ms  %9d rays  %8.1f Mrays/s" % (name, lights_name, os.environ.get("MRT_RESIDENT", "auto"), ms, rays, rays / ms / 1e3))


if __name__ == "__main__":
    main()
