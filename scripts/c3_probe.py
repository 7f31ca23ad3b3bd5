#!/usr/bin/env python3
"""Large-scene configuration: synthetic ~1M-triangle soup at 3840x2160, packet path.  Builds the scene,
renders on the GPU, checks hit IDs / pixels against the CPU oracle, and times the kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    W, H = (3840, 2160) if len(sys.argv) < 4 else (int(sys.argv[2]), int(sys.argv[3]))
    check = "--no-check" not in sys.argv
    import torch
    import oracle_lib
    from raytracing_amd import Raytracer, scene_io
    t0 = time.time()
    scene = scene_io.soup_scene(n)
    print("scene: %d triangles generated in %.1f s" % (n, time.time() - t0), flush=True)
    rt = Raytracer(W, H, device=0)
    t0 = time.time()
    rt.build_bvh(scene)
    print("GPU-side BuildBVH (host build + upload): %.1f s" % (time.time() - t0), flush=True)
    cam = oracle_lib.IDENTITY
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    for mode, name in ((0, "packet"), (1, "recursive")):
        for _ in range(2):
            rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            rt.render_device(cam, fb.data_ptr(), W, mode=mode, stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        c = rt.ray_counts()
        rays = c["primary"] + c["shadow"] + c["secondary"]
        print("%-9s %dx%d  %.3f ms/frame  %d rays  %.1f Mrays/s" % (name, W, H, ms, rays, rays / ms / 1e3), flush=True)
    # rasterizer path on the same scene (host framebuffer included: the C ABI has no device-resident raster entry)
    fbh = np.zeros((H, W), np.uint32)
    rt.raster(cam, fb=fbh, want_z=False, want_ids=False)
    t0 = time.time()
    for _ in range(3):
        rt.raster(cam, fb=fbh, want_z=False, want_ids=False)
    print("raster    %dx%d  %.2f ms/frame (host copy included), %.1f %% of pixels covered" % (W, H, (time.time() - t0) / 3 * 1e3, 100.0 * (fbh != 0).mean()), flush=True)
    from raytracing_amd.binding import HostFrame
    hf = HostFrame(H, W)
    hf.array[:] = 0
    rt.raster(cam, fb=hf.array, want_z=False, want_ids=False)
    t0 = time.time()
    for _ in range(5):
        rt.raster(cam, fb=hf.array, want_z=False, want_ids=False)
    print("raster    %dx%d  %.2f ms/frame into a page-locked caller frame, identical: %s" % (W, H, (time.time() - t0) / 5 * 1e3, np.array_equal(hf.array, fbh)), flush=True)
    hf.close()
    if check:
        o = oracle_lib.Oracle()
        o.add_scene(scene)
        t0 = time.time()
        o.build_bvh()
        print("oracle BVH build %.1f s" % (time.time() - t0), o.bvh_info, flush=True)
        o.set_threads(16)
        t0 = time.time()
        ref = o.render(cam, W, H, mode=0)
        print("oracle packet frame %.2f s" % (time.time() - t0), flush=True)
        got, ids = rt.render(cam, mode=0)
        print("hit ids equal:", np.array_equal(ids, ref["ids"]), " pixels equal:", np.array_equal(got, ref["fb"]),
              " hits:", int((ids >= 0).sum()), flush=True)
        t0 = time.time()
        rref = o.raster(cam, W, H)
        print("oracle raster frame %.2f s" % (time.time() - t0), flush=True)
        gfb, gz, gids = rt.raster(cam)
        n = W * H
        print("raster: coverage differs on %d px (%.4f %%), winner differs on %d px (%.4f %%), colour differs on %d px (%.4f %%)" % (
            int(((gids >= 0) != (rref["ids"] >= 0)).sum()), 100.0 * ((gids >= 0) != (rref["ids"] >= 0)).mean(),
            int((gids != rref["ids"]).sum()), 100.0 * (gids != rref["ids"]).mean(),
            int((gfb != rref["fb"]).sum()), 100.0 * (gfb != rref["fb"]).mean()), flush=True)


if __name__ == "__main__":
    main()
