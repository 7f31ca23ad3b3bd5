#!/usr/bin/env python3
"""Frame times of the synchronous entry points into an ordinary (pageable) caller frame: ray tracer, both shaders,
and rasterizer, default scene at 1080p.  A tuning aid."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from raytracing_amd import Raytracer, Camera, scene_io, MODE_RECURSIVE, MODE_PACKET
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rt = Raytracer(1920, 1080, device=0)
rt.build_bvh(scene_io.load_scene(os.path.join(ROOT, "tests/golden/maze.mrts")))
cam = Camera()
fb = np.zeros((1080, 1920), np.uint32)
for mode, name in ((MODE_RECURSIVE, "recursive"), (MODE_PACKET, "packet")):
    for _ in range(5): rt.render(cam, mode=mode, want_ids=False, fb=fb)
    t0 = time.perf_counter()
    for _ in range(50): rt.render(cam, mode=mode, want_ids=False, fb=fb)
    print(name, "pageable frame: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for _ in range(20): rt.raster(cam, fb=fb, want_z=False, want_ids=False)
print("raster pageable frame: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
