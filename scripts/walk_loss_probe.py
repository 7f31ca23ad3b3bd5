#!/usr/bin/env python3
"""How much of a walking camera's frame time is the tile schedule lagging one frame behind (the persistent recursive
kernel orders and splits its tiles by the previous frame's costs)?  Walks the camera sideways there and back like
perf_probe.py --walk, and renders the same positions again with the camera held still at each until the schedule has
settled.  A tuning aid.   walk_loss_probe.py [w h]"""
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
    rt = Raytracer(w, h, device=0)
    rt.build_bvh(scene, lbvh="exact" if os.environ.get("MRT_PROBE_TREE", "exact") == "exact" else False)     # (as bench.py uploads it)
    fb = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream()
    xs = [0.16 * k for k in list(range(1, 26)) + list(range(24, -1, -1))]

    def timed(x, frames):
        cam = Camera()
        cam.set_position((x, 0.0, 0.0))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(frames):
            if k == frames - 1:
                e0.record(st)
            rt.render_device(cam, fb.data_ptr(), w, mode=1, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    timed(0.0, 4)
    walking = [timed(x, 1) for x in xs for _ in (0,)]
    walking = [timed(x, 1) for x in xs]             # second lap: the first one started from a cold schedule
    settled = [timed(x, 5) for x in xs]
    frame_order = []
    print("%dx%d: walking %.3f ms per frame, the same positions with the schedule settled %.3f ms (walking costs %.1f %%)" %
          (w, h, np.mean(walking), np.mean(settled), 100 * (np.mean(walking) / np.mean(settled) - 1)))
    rt.close()


if __name__ == "__main__":
    main()
