"""Primary rays that graze the walls of long leaves (tests/grazing_cases.py): hit IDs must equal the oracle's.

The frame kernels skip groups of four triangles of a long leaf whose padded box a ray misses; the reference skips
nothing (raytracer.cpp:1445-1464, 1328-1346).  For a ray in a triangle's plane the test's verdict is rounding noise that
no box predicts, so the kernels give up the skip for every group with a triangle whose plane passes within the boxes'
padding of the eye (kernels.hip: eye_grazes_group).  These cameras put the eye in such planes and steer the frame's
centre column (vertical walls) or centre row (floor) along them to 0 ... 1e-4 rad.
"""
import numpy as np
import pytest

import grazing_cases
import oracle_lib
from conftest import colour_mismatch_fraction
from raytracing_amd import MODE_PACKET, MODE_RECURSIVE, Raytracer, binding

pytestmark = pytest.mark.gpu

W = H = 64


def run_campaign(scene, n_walls, angles=grazing_cases.ANGLES, offsets=grazing_cases.OFFSETS, modes=(MODE_PACKET, MODE_RECURSIVE), tree=False):
    """Renders every grazing camera on the GPU and with the oracle; returns (frames, frames whose hit ids differ,
    rays whose hit ids differ, recursive frames beyond the colour bar, packet frames whose pixels differ)."""
    bvh = binding.bvh_build_host(scene)
    rt = Raytracer(W, H, device=0)
    rt.build_bvh(scene, lbvh=tree)
    ora = oracle_lib.Oracle()
    ora.add_scene(scene)
    ora.build_bvh()
    ora.set_threads(4)
    frames = bad_frames = bad_rays = bad_colour = bad_pixels = 0
    for tri, normal in grazing_cases.walls(scene, bvh, n_walls):
        for eye, target, _ in grazing_cases.cameras(tri, normal, angles, offsets):
            cam = oracle_lib.lookat(tuple(eye), tuple(target))
            if not np.isfinite(cam).all():
                continue
            for mode in modes:
                fb, ids = rt.render(cam, mode=mode)
                want = ora.render(cam, W, H, mode=mode)
                frames += 1
                d = int((ids != want["ids"]).sum())
                bad_rays += d
                bad_frames += d > 0
                if mode == MODE_PACKET:
                    bad_pixels += not np.array_equal(fb, want["fb"])
                else:
                    bad_colour += colour_mismatch_fraction(fb, want["fb"]) > 0.001
    rt.close()
    return frames, bad_frames, bad_rays, bad_colour, bad_pixels


@pytest.mark.parametrize("tree", [False, "exact"], ids=["reference's tree", "two trees"])
def test_primary_rays_grazing_long_leaf_walls(maze_scene, tree):
    """tree = "exact": the scene uploaded with mrt_scene_upload_lbvh_exact -- the recursive frames walk the second tree,
    whose boxes are small where the reference's leaves are long."""
    frames, bad_frames, bad_rays, bad_colour, bad_pixels = run_campaign(maze_scene, 8, tree=tree)
    assert frames > 1500
    assert bad_rays == 0, "hit ids differ on %d rays in %d of %d frames" % (bad_rays, bad_frames, frames)
    assert bad_pixels == 0 and bad_colour == 0
