#!/usr/bin/env python3
"""Random-pose parity campaign: GPU frames (both shaders, through the C ABI) against the CPU oracle for many random
cameras in and around the default scene, random light positions, random sample grids.  Prints a summary line;
exit status 1 on the first mismatch beyond the parity bar.  usage: parity_fuzz.py [--gpu-tree | --two-trees] [poses] [seed]
--two-trees: scenes are uploaded with mrt_scene_upload_lbvh_exact (the recursive shader walks the GPU-built tree); the bar is the same."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def soups(n_scenes, seed):
    """Random triangle soups with random reflecting / refracting materials and up to three lights."""
    import oracle_lib
    from raytracing_amd import Raytracer, scene_io
    rng = np.random.default_rng(seed)
    W, H = 192, 128
    rt = Raytracer(W, H, device=0)
    rt1 = Raytracer(W, H, device=0)          # two trees: the same scenes with the reference's tree alone, held bit for bit
    worst = 0.0
    walked = [0, 0]
    any_diff = [0, 0]          # recursive pixels that differ from the oracle's at all | compared
    for i in range(n_scenes):
        if i and i % 100 == 0:
            print("... %d soups" % i, flush=True)
        n = int(rng.integers(1, 6000))
        s = scene_io.soup_scene(n, seed=int(rng.integers(1, 1 << 30)))
        m = s["meshes"][0]
        cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 3)]))
        parts = []
        for lo, hi in zip(cuts, cuts[1:]):
            if hi == lo:
                continue
            v0, v1 = lo * 3, hi * 3
            refl = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
            refr = float(rng.choice([0.0, 0.0, 0.4, 0.8]))
            parts.append({"pos": m["pos"][v0:v1], "uv": m["uv"][v0:v1], "norm": m["norm"][v0:v1], "N": m["N"][lo:hi],
                          "tri": (m["tri"][lo:hi] - v0).astype(np.int32), "texture": 0,
                          "bounds": np.concatenate([m["pos"][v0:v1].min(0), m["pos"][v0:v1].max(0)]),
                          "refl": refl, "refr": refr, "rindex": float(rng.choice([1.0, 1.33, 1.5, 0.8]))})
        s["meshes"] = parts
        ora = oracle_lib.Oracle()
        ora.add_scene(s)
        ora.build_bvh()
        ora.set_threads(16)
        rt.build_bvh(s, lbvh="exact" if TWO_TREES else False)
        if TWO_TREES:
            rt1.build_bvh(s)
        nl = int(rng.integers(0, 4))
        L = np.zeros((nl, 7), np.float32)
        L[:, :3] = rng.uniform([-20, -3, -45], [20, 8, 0], (nl, 3))
        L[:, 3:] = rng.uniform(0, 1, (nl, 4))
        ora.set_lights(L)
        rt.set_lights(L)
        rt1.set_lights(L)
        for _ in range(2):
            pos = rng.uniform([-20, -3, -40], [20, 3, 2]).astype(np.float32)
            cam = oracle_lib.lookat(tuple(pos), tuple(rng.uniform([-20, -2, -43], [20, 2, -3]).astype(np.float32)))
            for mode in (0, 1):
                fb, ids = rt.render(cam, mode=mode)
                if TWO_TREES:
                    walked[0] += rt.second_tree_use()[0]; walked[1] += 1
                    c2 = rt.ray_counts()
                    fb1, ids1 = rt1.render(cam, mode=mode)
                    if not (np.array_equal(fb, fb1) and np.array_equal(ids, ids1) and c2 == rt1.ray_counts()):
                        print("soup %d (n=%d, seed %d) mode %d: two trees against one: %d pixels, %d hit ids differ" % (i, n, seed, mode, int((fb != fb1).sum()), int((ids != ids1).sum())))
                        return 1
                ref = ora.render(cam, W, H, mode=mode)
                a = fb.view(np.uint8).reshape(-1, 4).astype(np.int16)
                b = ref["fb"].view(np.uint8).reshape(-1, 4).astype(np.int16)
                frac = float((np.abs(a - b).max(axis=1) > 1).mean())
                worst = max(worst, frac)
                if mode == 1:
                    any_diff[0] += int((fb != ref["fb"]).sum()); any_diff[1] += fb.size
                if not np.array_equal(ids, ref["ids"]) or (mode == 0 and not np.array_equal(fb, ref["fb"])) or frac > 0.001:
                    print("soup %d (n=%d, seed %d) mode %d: ids differ on %d rays, %.5f of pixels beyond 1/255" % (
                        i, n, seed, mode, int((ids != ref["ids"]).sum()), frac))
                    return 1
    if TWO_TREES:
        print("two trees: %d of %d frames walked a second tree (soups that fit the LDS have one)" % (walked[0], walked[1]))
    print("parity_fuzz: %d random soups x 2 poses x 2 shaders: hit ids identical, packet pixels identical, worst recursive "
          "fraction beyond 1/255: %.6f; recursive pixels that differ from the oracle's at all: %d of %d" % (n_scenes, worst, any_diff[0], any_diff[1]))
    return 0


TWO_TREES = False


def main():
    gpu_tree = "--gpu-tree" in sys.argv      # the clustered tree built on the GPU instead of the reference's: differences are counted, not fatal
    if gpu_tree:
        sys.argv.remove("--gpu-tree")
    global TWO_TREES
    if "--two-trees" in sys.argv:            # mrt_scene_upload_lbvh_exact: the recursive shader walks the GPU-built tree; everything must still be identical
        sys.argv.remove("--two-trees")
        TWO_TREES = True
    if len(sys.argv) > 1 and sys.argv[1] == "--soups":
        return soups(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import oracle_lib
    from raytracing_amd import Raytracer, scene_io
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    ora.set_threads(16)
    rng = np.random.default_rng(seed)
    sizes = [(256, 192), (320, 200), (128, 128), (512, 64)]
    rts = {s: Raytracer(s[0], s[1], device=0) for s in sizes}
    for rt in rts.values():
        rt.build_bvh(scene, lbvh="exact" if TWO_TREES else gpu_tree)
    # two trees: also held, bit for bit, to the same library on the reference's tree alone (the oracle's bar lets a
    # recursive pixel differ by 1/255)
    ones = {s: Raytracer(s[0], s[1], device=0) for s in sizes} if TWO_TREES else {}
    for rt in ones.values():
        rt.build_bvh(scene)
    worst = 0.0
    rays = id_diffs = px_diffs = frames_off = 0
    by_mode = {0: [0, 0], 1: [0, 0]}
    second = [0, 0]
    any_diff = [0, 0]          # recursive pixels that differ from the oracle's at all | compared
    for i in range(n_poses):
        if i and i % 1000 == 0:
            print("... %d poses" % i, flush=True)
        W, H = sizes[i % len(sizes)]
        rt = rts[(W, H)]
        pos = rng.uniform([-19, -0.9, -21], [19, 6, 21]).astype(np.float32)
        if i % 7 == 0:
            pos = np.round(pos)                       # integer coordinates: rays along box faces, exact ties
        target = rng.uniform([-18, -1, -20], [18, 1, 20]).astype(np.float32)
        if i % 11 == 0:
            target = pos + np.float32([0, -1, 0]) + np.float32([1e-3, 0, 0])     # nearly straight down
        cam = oracle_lib.lookat(tuple(pos), tuple(target))
        light = np.float32([[rng.uniform(-15, 15), rng.uniform(0.5, 8), rng.uniform(-15, 15), 1, 1, 1, 1]])
        ora.set_lights(light)
        rt.set_lights(light)
        g = 2 if i % 5 == 0 else 1
        for mode in (0, 1):
            fb, ids = rt.render(cam, mode=mode, spp_grid=g)
            ref = ora.render(cam, W, H, mode=mode, spp_grid=g)
            if gpu_tree:
                rays += ids.size
                d = int((ids != ref["ids"]).sum())
                a = fb.view(np.uint8).reshape(-1, 4).astype(np.int16)
                b = ref["fb"].view(np.uint8).reshape(-1, 4).astype(np.int16)
                e = int((np.abs(a - b).max(axis=1) > 1).sum()) if mode == 1 else int((fb != ref["fb"]).sum())
                id_diffs += d; px_diffs += e; frames_off += (d > 0 or e > 0)
                by_mode[mode][0] += d; by_mode[mode][1] += e
                continue
            if not np.array_equal(ids, ref["ids"]):
                print("pose %d mode %d: %d hit ids differ (pos %s target %s)" % (i, mode, int((ids != ref["ids"]).sum()), pos, target))
                return 1
            a = fb.view(np.uint8).reshape(-1, 4).astype(np.int16)
            b = ref["fb"].view(np.uint8).reshape(-1, 4).astype(np.int16)
            frac = float((np.abs(a - b).max(axis=1) > 1).mean())
            worst = max(worst, frac)
            if mode == 1:
                any_diff[0] += int((fb != ref["fb"]).sum()); any_diff[1] += fb.size
            if (mode == 0 and not np.array_equal(fb, ref["fb"])) or frac > 0.001:
                print("pose %d mode %d: pixels differ (%.5f beyond 1/255; pos %s target %s)" % (i, mode, frac, pos, target))
                return 1
            c = rt.ray_counts()
            if TWO_TREES:
                ones[(W, H)].set_lights(light)
                fb1, ids1 = ones[(W, H)].render(cam, mode=mode, spp_grid=g)
                if not (np.array_equal(fb, fb1) and np.array_equal(ids, ids1) and c == ones[(W, H)].ray_counts()):
                    print("pose %d mode %d: two trees against one: %d pixels, %d hit ids differ (pos %s target %s)" % (i, mode, int((fb != fb1).sum()), int((ids != ids1).sum()), pos, target))
                    return 1
            if TWO_TREES and mode == 1:
                walked, again = rt.second_tree_use()
                second[0] += walked; second[1] += again
    if TWO_TREES:
        print("two trees: %d of %d recursive frames walked the second tree, %d rays traced again through the first" % (second[0], n_poses, second[1]))
    if gpu_tree:
        print("parity_fuzz with the GPU-built tree: %d poses x 2 shaders, %d primary rays: %d hit ids and %d pixels (packet: at all, recursive: beyond 1/255) "
              "differ from the oracle's, in %d frames; packet shader %d ids %d pixels, recursive shader %d ids %d pixels%s" %
              (n_poses, rays, id_diffs, px_diffs, frames_off, by_mode[0][0], by_mode[0][1], by_mode[1][0], by_mode[1][1], ""))
        return 0
    print("parity_fuzz: %d poses x 2 shaders: hit ids identical, packet pixels identical, worst recursive fraction beyond 1/255: %.6f; "
          "recursive pixels that differ from the oracle's at all: %d of %d" % (n_poses, worst, any_diff[0], any_diff[1]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
