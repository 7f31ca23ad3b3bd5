#!/usr/bin/env python3
"""Replays the maze poses of scripts/raster_fuzz.py for one seed and prints every pose whose colour fraction comes
within reach of the bar (coverage, depth winner, colours beyond 1/255), for the library MRT_LIB names.  A debugging aid:
run it for two libraries to see whether a frame near the bar is a property of the method or of a change.
usage: raster_pose_check.py [seed] [poses] [print above this colour fraction]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 54
    n_poses = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    above = float(sys.argv[3]) if len(sys.argv) > 3 else 0.006
    import oracle_lib
    from conftest import colour_mismatch_fraction
    from raytracing_amd import Raytracer, scene_io
    rng = np.random.default_rng(seed)
    maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    ora = oracle_lib.Oracle()
    ora.add_scene(maze)
    ora.build_bvh()
    ora.set_threads(8)
    sizes = [(320, 200), (256, 256), (640, 360), (1000, 504)]
    rts = {s: Raytracer(s[0], s[1], device=0) for s in sizes}
    for rt in rts.values():
        rt.build_bvh(maze)
    for i in range(n_poses):
        W, H = sizes[i % len(sizes)]
        pos = rng.uniform([-19, -0.9, -21], [19, 6, 21]).astype(np.float32)
        target = rng.uniform([-18, -1, -20], [18, 1, 20]).astype(np.float32)
        if i % 7 == 0:
            pos = np.round(pos)
        cam = oracle_lib.lookat(tuple(pos), tuple(target))
        fb0 = np.full((H, W), 0x00123456, np.uint32)
        fb, z, ids = rts[(W, H)].raster(cam, fb=fb0.copy())
        ref = ora.raster(cam, W, H)
        ref_fb = fb0.copy()
        cov_ref = ref["ids"] >= 0
        ref_fb[cov_ref] = ref["fb"][cov_ref]
        n = W * H
        c, w, col = ((ids >= 0) != cov_ref).sum() / n, (ids != ref["ids"]).sum() / n, colour_mismatch_fraction(fb, ref_fb)
        if col > above:
            print("pose %d %dx%d pos %s target %s: coverage %.6f winner %.6f colour %.6f" % (i, W, H, pos.tolist(), target.tolist(), c, w, col), flush=True)
    print("done")


if __name__ == "__main__":
    main()
