#!/usr/bin/env python3
"""Looks for cameras at which a library's primary hit IDs differ from the oracle's, among cameras built to graze:
the eye a few float32 steps beside the plane of a wall triangle of a long leaf, the camera rolled so that a whole
pixel row (512 rays) lies in a plane that meets the wall's plane at 1e-7 ... 1e-4 rad.  Every camera found is printed
as one JSON line {"cam": [16 hex words], "W": .., "H": .., "mode": .., "rays": differing rays}.
usage: MRT_LIB=... grazing_search.py [walls] [seed] [out.jsonl] [cameras per wall]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 512, 8


def rolled_camera(eye, along, in_plane_up, normal, angle):
    """Camera-to-world matrix (column-major float32[16]) looking along `along` turned by `angle` towards `normal`, its
    x axis along in_plane_up: the pixel row through the frame's centre then spans the plane of (forward, in_plane_up)."""
    f = np.cos(angle) * along + np.sin(angle) * normal
    x = in_plane_up
    z = -f
    y = np.cross(z, x)
    m = np.zeros((4, 4), np.float64)
    m[0, :3], m[1, :3], m[2, :3], m[3, :3] = x, y, z, eye
    m[3, 3] = 1.0
    return m.reshape(16).astype(np.float32)


def step(x, k):
    """x moved by k float32 steps"""
    x = np.float32(x)
    for _ in range(abs(k)):
        x = np.nextafter(x, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
    return x


def main():
    n_walls = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    per_wall = int(sys.argv[4]) if len(sys.argv) > 4 else 24
    import grazing_cases
    import oracle_lib
    from raytracing_amd import MODE_PACKET, MODE_RECURSIVE, Raytracer, binding, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    bvh = binding.bvh_build_host(scene)
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene)
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    ora.set_threads(8)
    rng = np.random.RandomState(seed)
    found, frames = [], 0
    for tri, normal in grazing_cases.walls(scene, bvh, n_walls):
        edges = sorted([tri[1] - tri[0], tri[2] - tri[1], tri[0] - tri[2]], key=lambda e: -np.linalg.norm(e))
        along = edges[0] / np.linalg.norm(edges[0])
        up = np.cross(normal, along)
        centre = tri.mean(0)
        axis = int(np.argmax(np.abs(normal)))
        for _ in range(per_wall):
            base = centre - rng.uniform(0.3, 9.0) * along * rng.choice([-1, 1]) + rng.uniform(-0.6, 0.6) * up
            k = int(rng.choice([0, 1, -1, 2, -2, 3, -4, 6, -8, 12, 20, -30]))
            eye = base.astype(np.float32)
            eye[axis] = step(np.float32(tri[0][axis]), k)
            h = float(eye[axis]) - float(np.float32(tri[0][axis]))
            # aim the centre ray at the plane some units ahead (when the eye is off the plane), or pick an angle
            reach = rng.uniform(0.5, 25.0)
            angle = -np.sign(normal[axis]) * h / reach if (k != 0 and rng.rand() < 0.7) else rng.choice([-1, 1]) * 10 ** rng.uniform(-7, -4)
            direction = along * rng.choice([-1, 1])
            cam = rolled_camera(eye.astype(np.float64), direction, up, normal, angle)
            for mode in (MODE_RECURSIVE, MODE_PACKET):
                fb, ids = rt.render(cam, mode=mode)
                want = ora.render(cam, W, H, mode=mode)
                frames += 1
                d = int((ids != want["ids"]).sum())
                if d:
                    found.append({"cam": ["%08x" % w for w in cam.view(np.uint32)], "W": W, "H": H, "mode": int(mode), "rays": d,
                                  "steps_off_plane": k, "angle": float(angle)})
                    print(json.dumps(found[-1]), flush=True)
        print("wall done: %d frames so far, %d found" % (frames, len(found)), flush=True)
    rt.close()
    print("grazing search: %d frames, %d with other hit ids (%d rays)" % (frames, len(found), sum(f["rays"] for f in found)))
    if out_path:
        with open(out_path, "w") as f:
            for rec in found:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
