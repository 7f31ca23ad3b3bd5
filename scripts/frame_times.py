#!/usr/bin/env python3
"""Time of every single frame of a run of recursive frames of the default scene (HIP events around each launch): shows
what the tile order and the split of expensive tiles do from one frame to the next.  usage: frame_times.py W H [frames] [--schedule]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    W, H = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    cam = Camera()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    listed = []
    ev[0].record(stream)
    for i in range(n):
        rt.render_device(cam, fb.data_ptr(), W, mode=1, stream=stream.cuda_stream)
        ev[i + 1].record(stream)
        if "--schedule" in sys.argv:     # (waits for every frame)
            tiles, handed = rt.tile_schedule()
            listed.append((handed - tiles) // 15)
            ev[i + 1].record(stream)
    torch.cuda.synchronize()
    print("%dx%d:" % (W, H), " ".join("%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(n)), "ms")
    if listed:
        print("tiles handed out in sixteenths:", " ".join(str(e) for e in listed))


if __name__ == "__main__":
    main()
