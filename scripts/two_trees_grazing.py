#!/usr/bin/env python3
"""Grazing cameras (tests/grazing_cases.py: the eye in the plane of a wall, looking along it) through a scene with two
trees (mrt_scene_upload_lbvh_exact) against the same scene with the reference's tree alone: hit ids, pixels and ray
counts of the recursive shader must be identical.  Walls are taken from ALL leaves here (min_count 1): the second
tree's leaves are small everywhere.   two_trees_grazing.py [walls] [size]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import grazing_cases
    import oracle_lib
    from raytracing_amd import MODE_RECURSIVE, Raytracer, binding, scene_io
    n_walls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    bvh = binding.bvh_build_host(scene)
    grazing_cases.long_leaf_triangles.__defaults__ = (1,)
    one = Raytracer(size, size, device=0); one.build_bvh(scene)
    two = Raytracer(size, size, device=0); two.build_bvh(scene, lbvh="exact")
    frames = bad = again = 0
    for tri, normal in grazing_cases.walls(scene, bvh, n_walls):
        for eye, target, label in grazing_cases.cameras(tri, normal):
            cam = oracle_lib.lookat(tuple(eye), tuple(target))
            if not np.isfinite(cam).all():
                continue
            a = one.render(cam, mode=MODE_RECURSIVE); ca = one.ray_counts()
            b = two.render(cam, mode=MODE_RECURSIVE); cb = two.ray_counts()
            walked, n = two.second_tree_use()
            assert walked
            again += n
            frames += 1
            if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and ca == cb):
                bad += 1
                if bad <= 10:
                    print("DIFFERS: eye %s target %s (%s): %d ids, %d pixels, counts %s / %s" % (eye, target, label, int((a[1] != b[1]).sum()), int((a[0] != b[0]).sum()), ca, cb))
    print("two_trees_grazing: %d frames of %dx%d, %d differ; %d rays traced again through the reference's tree" % (frames, size, size, bad, again))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
