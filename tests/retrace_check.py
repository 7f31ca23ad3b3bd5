"""Run by tests/test_gpu_two_trees.py in a process of its own with MRT_LIB=raytracing_amd/libmrt_retrace.so (make -C
raytracing_amd/csrc retrace): the library whose two-tree frames send EVERY ray through the walks of the reference's tree
that normally serve the few rays the second tree cannot vouch for.  Such frames must equal the one-tree frames of the
same library bit for bit: hit ids, pixels, ray counts."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from raytracing_amd import MODE_PACKET, MODE_RECURSIVE, Raytracer, binding, scene_io
    assert binding.LIB_PATH.endswith("libmrt_retrace.so"), binding.LIB_PATH
    from conftest import POSES, pose_cam
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    frames = 0
    for W, H in ((256, 144), (640, 360)):
        one, two = Raytracer(W, H, device=0), Raytracer(W, H, device=0)
        one.build_bvh(scene)
        two.build_bvh(scene, lbvh="exact")
        for pose in list(POSES[:3]) + [("grid", (14.0, 4.0, 19.0), (-7.4432807, 0.4908863, 3.5839818))]:
            cam = pose_cam(pose)
            for mode, grid in ((MODE_RECURSIVE, 1), (MODE_PACKET, 1), (MODE_PACKET, 2)):
                want = one.render(cam, mode=mode, spp_grid=grid)
                counts = one.ray_counts()
                for k in range(3 if mode == MODE_RECURSIVE else 1):      # (later frames of a view hand expensive tiles out in sixteenths)
                    got = two.render(cam, mode=mode, spp_grid=grid)
                    walked, again = two.second_tree_use()
                    what = "%dx%d %s mode %d grid %d frame %d" % (W, H, pose[0], mode, grid, k)
                    assert walked, what + ": the second tree was not walked"
                    n_rays = counts["primary"] + counts["shadow"] + counts["secondary"]
                    assert again >= (n_rays // 2 if mode == MODE_RECURSIVE else (W * grid // 8) * (H * grid // 8)), what + ": only %d rays / packets traced again" % again
                    assert np.array_equal(got[1], want[1]), what + ": %d hit ids differ" % int((got[1] != want[1]).sum())
                    assert np.array_equal(got[0], want[0]), what + ": %d pixels differ" % int((got[0] != want[0]).sum())
                    assert two.ray_counts() == counts, what + ": ray counts"
                    frames += 1
        one.close(); two.close()
    print("retrace_check OK: %d frames" % frames)


if __name__ == "__main__":
    main()
