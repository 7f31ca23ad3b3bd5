#!/usr/bin/env python3
"""How evenly the work of one frame splits over ranks: times every rank's share of tile rows on one
GPU, for contiguous bands and for round-robin blocks of tile rows.  A tuning aid."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    stream = torch.cuda.current_stream()
    for name, W, H, mode, grid in (("1080p recursive", 1920, 1080, 1, 1), ("8K 4x4 packet", 7680, 4320, 0, 4)):
        rt = Raytracer(W, H, device=0)
        rt.build_bvh(scene)
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        cam = Camera()
        rows = H // 8

        def t_rows(ranges, iters=3):
            best = 1e9
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for a, b in ranges:
                    rt.render_device(cam, fb.data_ptr(), W, mode=mode, spp_grid=grid, rows=(a * 8, b * 8), stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            return best
        full = t_rows([(0, rows)])
        print("%s: full frame %.3f ms" % (name, full))
        for world in (2, 4, 8):
            base, rem = divmod(rows, world)
            cont = []
            for r in range(world):
                s = r * base + min(r, rem)
                cont.append(t_rows([(s, s + base + (1 if r < rem else 0))]))
            line = "  N=%d contiguous: max %.3f  sum %.3f  efficiency %.2f" % (world, max(cont), sum(cont), full / (world * max(cont)))
            for blk in (1, 2, 4):
                inter = []
                for r in range(world):
                    rg = [(s, min(rows, s + blk)) for s in range(r * blk, rows, world * blk)]
                    inter.append(t_rows(rg, iters=2))
                line += " | blocks of %d: max %.3f eff %.2f" % (blk, max(inter), full / (world * max(inter)))
            print(line)
        del rt


if __name__ == "__main__":
    main()
