#!/usr/bin/env python3
"""Does the reference ever report a triangle for a ray that passes BESIDE that triangle's own bounding box?  (For a ray
in a triangle's plane the ray/triangle test is rounding noise and may say "hit" anywhere in the plane; a tree whose boxes
hug single triangles would never offer such a triangle to the ray.)  Renders the grazing cameras of
tests/grazing_cases.py on the reference's tree and checks every primary hit in float64 against the hit triangle's box
grown by `pad`.  A tuning aid.   noise_hit_probe.py [walls] [size]"""
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
    T = grazing_cases.scene_triangles(scene).astype(np.float64)
    lo, hi = T.min(1), T.max(1)
    big = np.abs(np.concatenate([lo, hi])).max()
    rt = Raytracer(size, size, device=0); rt.build_bvh(scene)
    ys, xs = np.mgrid[0:size, 0:size]
    frames = hits = 0
    outside = {2.0 ** -16: 0, 2.0 ** -12: 0, 2.0 ** -8: 0}
    for tri, normal in grazing_cases.walls(scene, bvh, n_walls):
        for eye, target, label in grazing_cases.cameras(tri, normal):
            cam = oracle_lib.lookat(tuple(eye), tuple(target))
            if not np.isfinite(cam).all():
                continue
            fb, ids = rt.render(cam, mode=MODE_RECURSIVE)
            m = np.asarray(cam, np.float64).reshape(4, 4).T
            f = lambda v: (m @ np.array(v, np.float64))[:3]
            p0, p1, p2 = f([-1, 1, -1.8, 1]), f([1, 1, -1.8, 1]), f([-1, -1, -1.8, 1])
            O = m[:3, 3]
            D = (p0 + (xs[..., None] / size) * (p1 - p0) + (ys[..., None] / size) * (p2 - p0)) - O
            hit = ids >= 0
            i = ids[hit]
            d = D[hit]
            frames += 1
            hits += int(hit.sum())
            with np.errstate(all="ignore"):
                for pad_rel in outside:
                    pad = big * pad_rel
                    a, b = (lo[i] - pad - O) / d, (hi[i] + pad - O) / d
                    tn, tf = np.minimum(a, b).max(1), np.maximum(a, b).min(1)
                    outside[pad_rel] += int((~((tf >= tn) & (tf > 0))).sum())
    print("noise_hit_probe: %d frames, %d primary hits; hits on a triangle whose box the ray misses: %s" %
          (frames, hits, ", ".join("box grown by 2^%d of the scene: %d" % (int(np.log2(k)), v) for k, v in outside.items())))


if __name__ == "__main__":
    main()
