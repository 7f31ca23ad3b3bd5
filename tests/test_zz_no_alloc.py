"""mrt_render_device_noalloc against mrt_render_device, through the C ABI.

The entry point keeps its scratch memory (tile queues, ray counters, tile costs and order) in a
fixed array of the library instead of allocated buffers.  All contexts on a GPU share that array and
it keeps the last call's contents, so the checks run two calls in a row, the second on a smaller
frame, and then the first again: pixels, hit IDs and ray counters must equal what the allocating
entry point gives, bit for bit (both run the same kernels).
"""
import numpy as np
import pytest

from conftest import POSES, pose_cam
from raytracing_amd import MODE_PACKET, MODE_RECURSIVE, MrtError, Raytracer

pytestmark = pytest.mark.gpu


def _both(rt, cam, W, H, mode, grid, torch, rows=None):
    stream = torch.cuda.current_stream().cuda_stream
    out = []
    for fn in (rt.render_device, rt.render_device_noalloc):
        fb = torch.full((H, W), 0x55, dtype=torch.int32, device="cuda:0")
        ids = torch.full((H * grid, W * grid), -7, dtype=torch.int32, device="cuda:0")
        fn(cam, fb.data_ptr(), W, mode=mode, spp_grid=grid, rows=rows, ids_ptr=ids.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        out.append((fb.cpu().numpy().view(np.uint32), ids.cpu().numpy(), rt.ray_counts()))
    return out


def _same(pair):
    (fa, ia, ca), (fb, ib, cb) = pair
    assert np.array_equal(fa, fb), "%d pixels differ" % int((fa != fb).sum())
    assert np.array_equal(ia, ib)
    assert ca == cb and ca["primary"] > 0


def test_noalloc_entry_point_equals_allocating_one(maze_scene):
    import torch
    big = Raytracer(640, 360, device=0)
    big.build_bvh(maze_scene)
    small = Raytracer(256, 128, device=0)
    small.build_bvh(maze_scene)
    cam_a, cam_b = pose_cam(POSES[1]), pose_cam(POSES[2])
    for mode, grid in ((MODE_RECURSIVE, 1), (MODE_PACKET, 2)):
        _same(_both(big, cam_a, 640, 360, mode, grid, torch))           # first call
        _same(_both(small, cam_b, 256, 128, mode, grid, torch))         # second call, smaller frame, same scratch array
        _same(_both(big, cam_a, 640, 360, mode, grid, torch))           # and the larger one again: stale costs of another context
        _same(_both(big, cam_b, 640, 360, mode, grid, torch))
        _same(_both(big, cam_a, 640, 360, mode, grid, torch, rows=(64, 200)))   # a band of tile rows
    # repeated calls of the same shape take the tile order from the costs the previous call left in the array
    stream = torch.cuda.current_stream().cuda_stream
    want = _both(big, cam_a, 640, 360, MODE_RECURSIVE, 1, torch)[0][0]
    fb = torch.zeros((360, 640), dtype=torch.int32, device="cuda:0")
    for _ in range(4):
        big.render_device_noalloc(cam_a, fb.data_ptr(), 640, mode=MODE_RECURSIVE, stream=stream)
    torch.cuda.synchronize()
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), want)
    # the counters of a frame live in the shared array too: once another context has rendered into it they are gone, and
    # asking for them says so instead of reporting the other context's rays
    counts = big.ray_counts()
    fbs = torch.zeros((128, 256), dtype=torch.int32, device="cuda:0")
    small.render_device_noalloc(cam_b, fbs.data_ptr(), 256, mode=MODE_PACKET, stream=stream)
    torch.cuda.synchronize()
    with pytest.raises(MrtError, match="another context"):
        big.ray_counts()
    assert small.ray_counts()["primary"] == 256 * 128
    big.render_device_noalloc(cam_a, fb.data_ptr(), 640, mode=MODE_RECURSIVE, stream=stream)
    assert big.ray_counts() == counts
    big.close()
    small.close()


def test_noalloc_entry_point_allocates_nothing_and_rejects_large_launches(maze_scene):
    import torch
    rt = Raytracer(512, 256, device=0)
    rt.build_bvh(maze_scene)
    stream = torch.cuda.current_stream().cuda_stream
    fb = torch.zeros((256, 512), dtype=torch.int32, device="cuda:0")
    cam = pose_cam(POSES[0])
    rt.render_device_noalloc(cam, fb.data_ptr(), 512, mode=MODE_RECURSIVE, stream=stream)     # the very first frame of this context
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for mode, grid in ((MODE_RECURSIVE, 1), (MODE_PACKET, 1), (MODE_RECURSIVE, 2), (MODE_PACKET, 4)):
        rt.render_device_noalloc(cam, fb.data_ptr(), 512, mode=mode, spp_grid=grid, stream=stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "device memory changed during mrt_render_device_noalloc"
    rt.close()
    # 4096 x 4096 pixels = 262,144 tiles fit exactly; an 8 x 8 sample grid on 1024 x 512 pixels (524,288 tiles) does not
    edge = Raytracer(4096, 4096, device=0)
    edge.build_bvh(maze_scene)
    fbe = torch.zeros((4096, 4096), dtype=torch.int32, device="cuda:0")
    edge.render_device_noalloc(cam, fbe.data_ptr(), 4096, mode=MODE_PACKET, stream=stream)
    ref = torch.zeros((4096, 4096), dtype=torch.int32, device="cuda:0")
    edge.render_device(cam, ref.data_ptr(), 4096, mode=MODE_PACKET, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(fbe, ref)
    edge.close()
    over = Raytracer(1024, 512, device=0)
    over.build_bvh(maze_scene)
    fbo = torch.zeros((512, 1024), dtype=torch.int32, device="cuda:0")
    with pytest.raises(MrtError, match="MRT_NOALLOC_MAX_TILES"):
        over.render_device_noalloc(cam, fbo.data_ptr(), 1024, mode=MODE_PACKET, spp_grid=8, stream=stream)
    over.render_device_noalloc(cam, fbo.data_ptr(), 1024, mode=MODE_PACKET, spp_grid=4, stream=stream)    # 131,072 tiles: accepted
    torch.cuda.synchronize()
    over.close()
