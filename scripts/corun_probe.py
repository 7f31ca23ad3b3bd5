#!/usr/bin/env python3
"""Does a second kernel fit beside the persistent recursive kernel when that one is compiled for 96 VGPRs?
Context A (scene in LDS, persistent kernel) renders 7 of 8 interleaved shares of a 1080p frame one after the
other; context B (same scene, forced to the one-wave-per-tile kernel that reads the scene from global memory)
renders the eighth share on a second stream at the same time.  Compared with A rendering all eight alone.
Run with MRT_LIB pointing at a library built with -DMRT_RESIDENT_WAVES=5.  A tuning aid."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from raytracing_amd import Camera, Raytracer, scene_io
    W, H, N = 1920, 1080, 8
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    a = Raytracer(W, H, device=0)
    a.build_bvh(scene)
    os.environ["MRT_RESIDENT"] = "0"
    b = Raytracer(W, H, device=0)
    b.build_bvh(scene)
    del os.environ["MRT_RESIDENT"]
    cam = Camera()
    fa = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    fb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    sa, _same_queue, sb = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()    # the 2nd torch stream shares sa's hardware queue here

    def run(do_a, band_b, iters=20):
        def once():
            if do_a:
                a.render_device(cam, fa.data_ptr(), W, mode=1, stream=sa.cuda_stream)
            if band_b:
                b.render_device(cam, fb.data_ptr(), W, mode=1, rows=band_b, stream=sb.cuda_stream)
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        ta = tb = 0.0
        for _ in range(iters):
            e0 = torch.cuda.Event(enable_timing=True)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            sa.wait_event(e0)
            sb.wait_event(e0)
            once()
            ea.record(sa)
            eb.record(sb)
            torch.cuda.synchronize()
            ta += e0.elapsed_time(ea) / iters
            tb += e0.elapsed_time(eb) / iters
        return ta, tb

    ta, _ = run(True, None)
    print("A alone, whole frame:            %.3f ms" % ta)
    for band in ((480, 504), (480, 520), (480, 536), (0, 240), (0, 320)):
        _, tb1 = run(False, band)
        ta2, tb2 = run(True, band)
        print("B rows %4d-%4d (%5d tiles) alone %.3f ms;  together: A %.3f ms, B %.3f ms" %
              (band[0], band[1], (band[1] - band[0]) // 8 * (W // 8), tb1, ta2, tb2))


if __name__ == "__main__":
    main()
