"""mrt_scene_upload_lbvh_exact: the recursive shader (second tree in LDS) and the packet path (both trees in global
memory) walk a second, GPU-built tree, and every result must
still be the reference's -- hit IDs, pixels and ray counts are held to those of the SAME library on the reference's own
tree (bit for bit: the arithmetic per ray is the same, only the tree differs) and to the CPU oracle (the usual bar).
mrt_get_second_tree_use shows that the second tree was really walked, and how many rays had to ask the first.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import POSES, colour_mismatch_fraction, pose_cam
from raytracing_amd import MODE_BVH_DEPTH, MODE_PACKET, MODE_RECURSIVE, Raytracer, binding, scene_io

pytestmark = pytest.mark.gpu

# P1 looks at the maze from outside: a dozen of its rays pass two triangles within rounding of each other and are
# traced again; the integer eye of "grid" sends rays along box faces; "far" is beyond the distance up to which the second
# tree's boxes vouch for their rays (kernels.hip: slab_mul_pair; SceneDev::reach, 64 times the scene's largest coordinate):
# such a frame walks the reference's tree; "distant" is a long way off but within it
EXTRA = [("grid", (14.0, 4.0, 19.0), (-7.4432807, 0.4908863, 3.5839818)), ("low", (6.863297, 2.8962514, -9.219043), (-11.901246, -0.24103431, -7.8104954)),
         ("distant", (300.0, 250.0, 900.0), (0.0, 0.0, 0.0)), ("far", (900.0, 700.0, 2500.0), (0.0, 0.0, 0.0))]


def _pair(scene, W, H):
    one = Raytracer(W, H, device=0)
    one.build_bvh(scene)
    two = Raytracer(W, H, device=0)
    two.build_bvh(scene, lbvh="exact")
    return one, two


def _frame(rt, cam, mode, **kw):
    fb, ids = rt.render(cam, mode=mode, **kw)
    return fb, ids, rt.ray_counts()


def _same(a, b, what=""):
    assert np.array_equal(a[1], b[1]), "%s: %d hit ids differ" % (what, int((a[1] != b[1]).sum()))
    assert np.array_equal(a[0], b[0]), "%s: %d pixels differ" % (what, int((a[0] != b[0]).sum()))
    assert a[2] == b[2], "%s: ray counts %s against %s" % (what, a[2], b[2])


@pytest.mark.parametrize("W,H", [(1920, 1080), (800, 600), (256, 144)])
def test_frames_through_the_second_tree_equal_those_through_the_first(maze_scene, maze_oracle, W, H):
    one, two = _pair(maze_scene, W, H)
    n_tiles = (W // 8) * (H // 8)
    retraced = retraced_packets = split = 0
    for pose in list(POSES[:5]) + EXTRA:
        cam = pose_cam(pose)
        want = _frame(one, cam, MODE_RECURSIVE)
        for k in range(3):          # from the second frame of a view on its expensive tiles are handed out in sixteenths
            got = _frame(two, cam, MODE_RECURSIVE)
            walked, again = two.second_tree_use()
            assert walked == (pose[0] != "far"), "the second tree was %swalked" % ("" if walked else "not ")
            retraced += again
            tiles, handed = two.tile_schedule()
            assert tiles == n_tiles and (handed - tiles) % 15 == 0
            split += (handed - tiles) // 15
            _same(got, want, "%s frame %d" % (pose[0], k))
        if (W, H) == (256, 144):
            ref = maze_oracle.render(cam, W, H, mode=MODE_RECURSIVE)
            assert np.array_equal(got[1], ref["ids"]) and colour_mismatch_fraction(got[0], ref["fb"]) <= 0.001
        # the packet path
        _same(_frame(two, cam, MODE_PACKET), _frame(one, cam, MODE_PACKET), pose[0] + " packet path")
        walked, again = two.second_tree_use()
        assert walked == (pose[0] != "far")
        retraced_packets += again
        # a sample grid (never split)
        if (W, H) != (1920, 1080):
            _same(_frame(two, cam, MODE_RECURSIVE, spp_grid=2), _frame(one, cam, MODE_RECURSIVE, spp_grid=2), pose[0] + " grid 2")
            _same(_frame(two, cam, MODE_PACKET, spp_grid=4), _frame(one, cam, MODE_PACKET, spp_grid=4), pose[0] + " packet path, grid 4")
    if (W, H) == (1920, 1080):
        assert retraced > 0 and retraced_packets > 0, "no ray of these views had to ask the reference's tree: the poses no longer cover that path"
    if (W, H) == (800, 600):
        assert split > 0, "no tile was handed out in sixteenths"
    one.close(); two.close()


def test_what_goes_through_the_first_tree(maze_scene):
    W, H = 320, 200
    one, two = _pair(maze_scene, W, H)
    cam = pose_cam(POSES[1])
    got, want = two.render(cam, mode=MODE_BVH_DEPTH), one.render(cam, mode=MODE_BVH_DEPTH)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert two.second_tree_use() == (False, 0)
    # the traversal operators
    rng = np.random.default_rng(3)
    O = rng.uniform([-15, 0, -15], [15, 4, 15], (4096, 3)).astype(np.float32)
    D = rng.normal(size=(4096, 3)).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    a, b = one.trace_rays(O, D, np.full(4096, 1e8, np.float32)), two.trace_rays(O, D, np.full(4096, 1e8, np.float32))
    for x, y in zip(a, b):
        assert np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
    # analytic primitives are not in the second tree: such frames go through the first
    spheres = np.float32([[0, 1, 1.5, 0, 0.5, 1, 0, 1, 0, 0.5, 0.0, 1.0, 0], [0, -1, 1.5, 0, 0.5, 1, 1, 1, 1, 0.0, 0.5, 1.5, 0]])
    one.set_objects(spheres); two.set_objects(spheres)
    for mode in (MODE_RECURSIVE, MODE_PACKET):
        _same(_frame(two, cam, mode), _frame(one, cam, mode), "with spheres")
        assert two.second_tree_use() == (False, 0)
    one.set_objects(np.zeros((0, 12), np.float32)); two.set_objects(np.zeros((0, 12), np.float32))
    _same(_frame(two, cam, MODE_RECURSIVE), _frame(one, cam, MODE_RECURSIVE), "spheres removed")
    assert two.second_tree_use()[0]
    one.close(); two.close()


def test_lights_materials_and_refit(maze_scene):
    """New lights reach the second tree's frames; after mrt_scene_refit the second tree is no longer used (it is not
    refitted) and frames are those of a refitted first tree."""
    W, H = 256, 192
    one, two = _pair(maze_scene, W, H)
    cam = pose_cam(POSES[2])
    lights = np.float32([[3, 4, 2, 1, 0.9, 0.8, 0.7], [-6, 2, -3, 1, 0.2, 0.9, 0.4]])
    one.set_lights(lights); two.set_lights(lights)
    _same(_frame(two, cam, MODE_RECURSIVE), _frame(one, cam, MODE_RECURSIVE), "two lights")
    assert two.second_tree_use()[0]
    moved = {"textures": maze_scene["textures"], "meshes": []}
    for m in maze_scene["meshes"]:
        pos = np.asarray(m["pos"], np.float32) + np.float32([0.0, 0.25, 0.0])
        moved["meshes"].append(dict(m, pos=pos, bounds=np.concatenate([pos.min(0), pos.max(0)]).astype(np.float32)))
    one.refit(moved); two.refit(moved)
    for mode in (MODE_RECURSIVE, MODE_PACKET):
        _same(_frame(two, cam, mode), _frame(one, cam, mode), "after the refit")
        assert two.second_tree_use() == (False, 0)
    two.build_bvh(moved, lbvh="exact"); one.build_bvh(moved)
    for mode in (MODE_RECURSIVE, MODE_PACKET):
        _same(_frame(two, cam, mode), _frame(one, cam, mode), "uploaded again")
        assert two.second_tree_use()[0]
    one.close(); two.close()


def test_scenes_the_second_tree_does_not_suit():
    """A scene too large for the LDS keeps only the first tree; tiny and empty scenes upload and render."""
    W, H = 128, 64
    cam = pose_cam(("soup", (5, 1, 4), (0, 0, -20)))
    for n in (20000, 1, 2, 5, 0):
        scene = scene_io.soup_scene(max(n, 1))
        if n == 0:
            scene["meshes"] = []
        one, two = _pair(scene, W, H)
        for mode in (MODE_PACKET, MODE_RECURSIVE):
            _same(_frame(two, cam, mode), _frame(one, cam, mode), "%d triangles, mode %d" % (n, mode))
        if n == 20000:
            assert two.second_tree_use() == (False, 0)
        one.close(); two.close()


def test_device_frames_shares_and_the_entry_point_without_allocations(maze_scene):
    import torch
    W, H = 640, 360
    one, two = _pair(maze_scene, W, H)
    cam = pose_cam(POSES[1])
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    one.render_device(cam, want.data_ptr(), W, mode=MODE_RECURSIVE, stream=stream)
    torch.cuda.synchronize()
    counts = one.ray_counts()
    for call in (two.render_device, two.render_device_noalloc):
        for _ in range(3):
            fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
            call(cam, fb.data_ptr(), W, mode=MODE_RECURSIVE, stream=stream)
            torch.cuda.synchronize()
            assert torch.equal(fb, want) and two.second_tree_use()[0] and two.ray_counts() == counts
    # interleaved shares of the frame (the multi-GPU split), rendered one after the other into one frame: both shaders,
    # the packet path with a sample grid as bench.py --gpus N renders it
    for mode, grid in ((MODE_RECURSIVE, 1), (MODE_PACKET, 2)):
        whole = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        one.render_device(cam, whole.data_ptr(), W, mode=mode, spp_grid=grid, stream=stream)
        fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        for part in range(3):
            two.render_device_part(cam, fb.data_ptr(), W, part, 3, block_tile_rows=2, mode=mode, spp_grid=grid, stream=stream)
            assert two.second_tree_use()[0]
        torch.cuda.synchronize()
        assert torch.equal(fb, whole)
    # a band of rows
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    two.render_device(cam, fb.data_ptr(), W, mode=MODE_RECURSIVE, rows=(64, 200), stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(fb[64:200], want[64:200]) and int(fb[:64].abs().sum()) == 0
    # frame order instead of cost order
    got = two.render(cam, mode=MODE_RECURSIVE, extra_flags=binding.FLAG_FRAME_ORDER)
    assert np.array_equal(got[0].view(np.int32), want.cpu().numpy())
    one.close(); two.close()


def test_every_ray_through_the_walks_of_the_references_tree():
    """The walks a two-tree frame falls back on serve a handful of rays per frame; libmrt_retrace.so (the same sources,
    every ray counted as unsure) renders whole frames with them: tests/retrace_check.py, in a process of its own because a
    process binds one library."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "raytracing_amd", "libmrt_retrace.so")
    assert os.path.exists(lib), "libmrt_retrace.so is not built (make -C raytracing_amd/csrc retrace)"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "retrace_check.py")], env=dict(os.environ, MRT_LIB=lib),
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "retrace_check OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_and_axis_aligned_cameras_on_two_trees():
    """A short run of the campaigns behind the parity figures: random poses and soups through a two-tree upload against
    the oracle and, bit for bit, against the one-tree upload (scripts/parity_fuzz.py --two-trees); axis-aligned cameras,
    whose centre rays have a direction component of exactly zero (scripts/two_trees_axis.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for args in (["parity_fuzz.py", "--two-trees", "60", "5"], ["parity_fuzz.py", "--two-trees", "--soups", "20", "6"], ["two_trees_axis.py", "--oracle", "60", "3"]):
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", args[0])] + args[1:], capture_output=True, text=True, timeout=110)
        assert r.returncode == 0, " ".join(args) + ": " + r.stdout[-1500:] + r.stderr[-1500:]
