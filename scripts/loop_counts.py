#!/usr/bin/env python3
"""Loop iteration counts of the frame kernels from the diagnostic build (make -C raytracing_amd/csrc prof):
for every instrumented site, how often a wave executed it and how many lanes were active.  A tuning aid."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SITES = {0: "nearest: inner-node step", 1: "nearest: leaf triangle", 2: "nearest: call past root gate",
         12: "nearest: tri past back-face", 13: "nearest: tri exact tail",
         3: "shadow: work item", 4: "shadow: inner node", 5: "shadow: leaf triangle",
         6: "shared tri past back-face", 7: "shared tri exact tail",
         8: "packet: work item", 9: "packet: inner visit", 10: "packet: leaf triangle", 11: "shader round",
         15: "shadow: inner node exact fallback", 14: "longest wave life | latest scene-copy end", 18: "sum: scene copy clocks | clocks in tiles", 19: "sum: wave life clocks | waves",
         20: "nearest: group box test", 21: "shared leaf: group box test",
         16: "clocks/16 in nearest-hit / packet traversal", 17: "clocks/16 in direct light + shadow"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--spp-grid", type=int, default=1)
    ap.add_argument("--scene", default="maze", help="maze | maze+spheres (the two spheres of raytracer.cpp:404-412, lifted to y = 1.5) | "
                    "soup:N (the large-scene generator) | mixed:N (that soup in three meshes: opaque, mirror, mirror + glass)")
    ap.add_argument("--pose", default="P0", help="P0 (identity) | P1 | soup (5,1,4) -> (0,0,-20) | inside (0,0.5,-23) -> (3,0,-30)")
    ap.add_argument("--gpu-tree", action="store_true", help="the tree mrt_scene_upload_lbvh builds on the GPU instead of the reference's")
    ap.add_argument("--two-trees", action="store_true", help="the scene uploaded with mrt_scene_upload_lbvh_exact (both shaders walk the second tree)")
    args = ap.parse_args()
    from raytracing_amd import binding
    binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_prof.so")
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    L = binding.lib()
    L.mrt_debug_loop_counts.argtypes = [C.c_void_p, C.c_void_p]
    objects = None
    if args.scene.startswith("maze"):
        scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
        if args.scene == "maze+spheres":
            objects = np.float32([[0, 1, 1.5, 0, 0.5, 1, 0, 1, 0, 0.5, 0.0, 1.0, 0], [0, -1, 1.5, 0, 0.5, 1, 1, 1, 1, 0.0, 0.5, 1.5, 0]])
    else:
        kind, n = args.scene.split(":")
        n = int(n)
        scene = scene_io.soup_scene(n, seed=n if kind == "mixed" else 12345)
        if kind == "mixed":
            m = scene["meshes"][0]
            third, parts = (n // 3) * 3, []
            for j, (refl, refr, rindex) in enumerate(((0.0, 0.0, 1.0), (0.4, 0.0, 1.0), (0.2, 0.3, 1.5))):
                lo, hi = j * third // 3, ((j + 1) * third // 3 if j < 2 else n)
                v0, v1 = lo * 3, hi * 3
                parts.append({"pos": m["pos"][v0:v1], "uv": m["uv"][v0:v1], "norm": m["norm"][v0:v1], "N": m["N"][lo:hi],
                              "tri": (m["tri"][lo:hi] - v0).astype(np.int32), "texture": 0,
                              "bounds": np.concatenate([m["pos"][v0:v1].min(0), m["pos"][v0:v1].max(0)]),
                              "refl": refl, "refr": refr, "rindex": rindex})
            scene["meshes"] = parts
    rt = Raytracer(args.w, args.h, device=0)
    rt.build_bvh(scene, lbvh="exact" if args.two_trees else args.gpu_tree)
    if objects is not None:
        rt.set_objects(objects)
    fb = torch.zeros((args.h, args.w), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    cam = Camera()
    if args.pose == "P1":
        cam.set_position((6, 3, 12))
        cam.look_at((0, 0, 0))
    elif args.pose == "soup":
        cam.set_position((5, 1, 4))
        cam.look_at((0, 0, -20))
    elif args.pose == "inside":         # inside the soup's bounds: the recursive shader's root gate (raytracer.cpp:947) lets rays in
        cam.set_position((0, 0.5, -23))
        cam.look_at((3, 0, -30))
    for mode, name in ((0, "packet"), (1, "recursive")):
        rt.render_device(cam, fb.data_ptr(), args.w, mode=mode, spp_grid=args.spp_grid, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        c = rt.ray_counts()
        out = np.zeros(44, np.uint64)
        L.mrt_debug_loop_counts(rt.h, out.ctypes.data_as(C.c_void_p))
        rounds = int(out[2 * 11])
        print("%s: %s" % (name, c))
        if rounds:
            print("  rays traced per shading round: %.1f of 64 lanes (%d primary + secondary rays, %d rounds)"
                  % ((c["primary"] + c["secondary"]) / rounds, c["primary"] + c["secondary"], rounds))
        for s in sorted(SITES):
            w, l = int(out[2 * s]), int(out[2 * s + 1])
            if l:
                print("  %-32s wave %10d  lane %11d  lanes/wave %5.1f" % (SITES[s], w, l, l / max(w, 1)))


if __name__ == "__main__":
    main()
