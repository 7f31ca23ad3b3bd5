#!/usr/bin/env python3
"""Axis-aligned cameras through a scene with two trees (mrt_scene_upload_lbvh_exact) against the same scene with the
reference's tree alone: the centre column and row of such a frame are rays with a direction component of exactly zero
(their reciprocal is infinite; the second tree's box tests take a clamped one: kernels.hip, make_ray_second_tree), and
in the maze's axis-aligned walls their reflections keep it.  Eyes at random places and on a half-unit grid (in the
planes of walls and of box faces).  Hit ids, pixels and ray counts of both shaders must be identical; with --oracle the
one-tree frames are also held to the CPU oracle (hit ids and packet pixels identical, recursive pixels within 1/255 on
99.9 %).   two_trees_axis.py [--oracle] [cameras] [seed]"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    with_oracle = "--oracle" in sys.argv
    if with_oracle:
        sys.argv.remove("--oracle")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from raytracing_amd import Raytracer, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    W, H = 256, 192
    one, two = Raytracer(W, H, device=0), Raytracer(W, H, device=0)
    one.build_bvh(scene)
    two.build_bvh(scene, lbvh="exact")
    ora = None
    if with_oracle:
        import oracle_lib
        ora = oracle_lib.Oracle()
        ora.add_scene(scene)
        ora.build_bvh()
        ora.set_threads(16)
    # the 24 rotations that map axes to axes
    rots = []
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1.0, -1.0), repeat=3):
            R = np.zeros((3, 3), np.float32)
            for i in range(3):
                R[i, perm[i]] = signs[i]
            if np.linalg.det(R) > 0:
                rots.append(R)
    rng = np.random.default_rng(seed)
    again = walked = 0
    for i in range(n):
        if i and i % 500 == 0:
            print("... %d cameras" % i, flush=True)
        pos = rng.uniform([-19, -0.9, -21], [19, 6, 21]).astype(np.float32)
        if i % 3 == 0:
            pos = np.round(pos * 2) / 2
        cam = np.eye(4, dtype=np.float32)
        cam[:3, :3] = rots[int(rng.integers(len(rots)))]
        cam[3, :3] = pos            # (column-major 4x4 handed over as 16 floats: the eye is elements 12..14)
        cam = cam.reshape(16)
        light = np.float32([[rng.uniform(-15, 15), rng.uniform(0.5, 8), rng.uniform(-15, 15), 1, 1, 1, 1]])
        if i % 4 == 0:
            light[0, :3] = np.round(light[0, :3])
        one.set_lights(light); two.set_lights(light)
        if ora:
            ora.set_lights(light)
        for mode in (0, 1):
            a = one.render(cam, mode=mode); ca = one.ray_counts()
            b = two.render(cam, mode=mode); cb = two.ray_counts()
            if ora:
                ref = ora.render(cam, W, H, mode=mode)
                pa = a[0].view(np.uint8).reshape(-1, 4).astype(np.int16)
                pr = ref["fb"].view(np.uint8).reshape(-1, 4).astype(np.int16)
                frac = float((np.abs(pa - pr).max(axis=1) > 1).mean())
                if not np.array_equal(a[1], ref["ids"]) or (mode == 0 and not np.array_equal(a[0], ref["fb"])) or frac > 0.001:
                    print("camera %d (eye %s) mode %d against the oracle: %d hit ids differ, %.5f of pixels beyond 1/255" % (i, pos, mode, int((a[1] != ref["ids"]).sum()), frac))
                    return 1
            use = two.second_tree_use()
            walked += use[0]; again += use[1]
            if not (np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]) and ca == cb):
                print("camera %d (eye %s) mode %d: %d hit ids, %d pixels differ, ray counts %s against %s" % (
                    i, pos, mode, int((a[1] != b[1]).sum()), int((a[0] != b[0]).sum()), ca, cb))
                return 1
    print("two_trees_axis: %d axis-aligned cameras x 2 shaders identical on both uploads%s; %d frames walked the second tree, %d rays (packets) traced again" % (n, " and equal to the oracle's at its bar" if ora else "", walked, again))
    return 0


if __name__ == "__main__":
    sys.exit(main())
