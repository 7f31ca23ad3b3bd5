"""Regenerates tests/golden/maze.mrts from the reference's default scene assets.

Runs only where /root/reference exists.  The OBJ/MTL/.bin files are parsed by the oracle's
restatement of the reference loaders (scenedata.cpp:77-271, surface.cpp:39-59) and the resulting
Mesh/Surface8 contents are stored in the .mrts container (raytracing_amd/scene_io.py), with the
per-mesh material scalars BVH::BVH assigns (raytracer.cpp:908-929).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib  # noqa: E402
from raytracing_amd import scene_io  # noqa: E402

REF = "/root/reference/Project"


def main():
    if not os.path.isdir(REF):
        print("reference tree absent; nothing to do")
        return
    cwd = os.getcwd()
    os.chdir(REF)                      # the reference resolves "assets/..." relative to Project/
    try:
        o = oracle_lib.Oracle()
        o.load_obj("assets/maze.obj")
        scene = o.export_scene()
    finally:
        os.chdir(cwd)
    for i, m in enumerate(scene["meshes"]):
        m["refl"], m["refr"], m["rindex"] = (0.0, 0.3, 1.5) if i == 0 else (0.0, 0.0, 1.0)
    out = os.path.join(HERE, "maze.mrts")
    scene_io.save_scene(out, scene)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
