"""Cameras whose primary rays graze the walls of long BVH leaves: the eye in (or a hair beside) the plane of a wall
triangle that sits in a leaf of six or more triangles, looking along the plane, turned out of it by up to 1e-4 rad.

For such rays every numerator of the ray/triangle test (raytracer.cpp:287-356) is rounding noise, and the reference
tests them against every triangle of every leaf they enter (raytracer.cpp:1445-1464, 1328-1346).  A renderer that
skips part of a leaf behind a bounding box must not do so for these rays; comparing hit IDs with the oracle on these
cameras is what shows it.  Test infrastructure (used by tests/test_gpu_grazing.py and scripts/grazing_probe.py).
"""
import numpy as np

ANGLES = (0.0, 1e-7, -1e-7, 1e-6, -1e-6, 4e-6, -4e-6, 1e-5, -1e-5, 1e-4, -1e-4)     # radians out of the wall's plane
OFFSETS = (0.0, 1e-6, -1e-5, 1e-4)                                                  # eye's distance from the plane


def scene_triangles(scene):
    """(n, 3, 3) float32 corner positions in the reference's flattened triangle order (mesh by mesh)."""
    out = []
    for m in scene["meshes"]:
        pos = np.asarray(m["pos"], np.float32).reshape(-1, 3)
        out.append(pos[np.asarray(m["tri"], np.int32).reshape(-1, 3)])
    return np.concatenate(out)


def long_leaf_triangles(bvh, min_count=6):
    """Global triangle ids that sit in leaves of at least min_count triangles (bvh: binding.bvh_build_host)."""
    fc, idx = bvh["node_first_cnt"], bvh["tri_idx"]
    ids, todo = [], [0]
    while todo:
        n = todo.pop()
        first, count = int(fc[n, 0]), int(fc[n, 1])
        if count == 0:
            todo += [first, first + 1]
        elif count >= min_count:
            ids += [int(i) for i in idx[first:first + count]]
    return ids


def walls(scene, bvh, n_walls):
    """The n_walls largest triangles of long leaves that lie in different planes: (corners float64 (3,3), unit normal)."""
    T = scene_triangles(scene).astype(np.float64)
    cand = long_leaf_triangles(bvh)
    n = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0])
    area = np.linalg.norm(n, axis=1)
    cand = sorted((i for i in cand if area[i] > 0), key=lambda i: -area[i])
    picked, planes = [], []
    for i in cand:
        nn = n[i] / area[i]
        d = float(nn @ T[i, 0])
        if any(abs(abs(nn @ p[0]) - 1) < 1e-6 and abs(d - np.sign(nn @ p[0]) * p[1]) < 1e-4 for p in planes):
            continue
        planes.append((nn, d))
        picked.append((T[i], nn))
        if len(picked) == n_walls:
            break
    return picked


def cameras(tri, normal, angles=ANGLES, offsets=OFFSETS):
    """Yields (eye float32 (3,), target float32 (3,), label) for one wall triangle."""
    f = np.float32
    edges = [tri[1] - tri[0], tri[2] - tri[1], tri[0] - tri[2]]
    edges.sort(key=lambda e: -np.linalg.norm(e))
    along = next((e for e in edges if abs(e[1]) / np.linalg.norm(e) < 0.9), edges[0])      # not along the camera's up vector
    along = along / np.linalg.norm(along)
    side = np.cross(normal, along)
    centre = tri.mean(0)
    axis = int(np.argmax(np.abs(normal)))
    axis_aligned = abs(abs(normal[axis]) - 1.0) < 1e-12
    for back, beside in ((3.0, 0.0), (0.6, 0.35), (8.0, -0.2)):
        base = centre - back * along + beside * side
        for off in offsets:
            eye = base + off * normal
            eye32 = eye.astype(f)
            if axis_aligned and off == 0.0:
                eye32[axis] = f(tri[0][axis])          # exactly the wall's coordinate
            for a in angles:
                direction = np.cos(a) * along + np.sin(a) * normal
                target = (eye32.astype(np.float64) + 4.0 * direction).astype(f)
                yield eye32, target, "back %.1f beside %.2f off %g angle %g" % (back, beside, off, a)
