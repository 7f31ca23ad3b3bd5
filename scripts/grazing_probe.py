#!/usr/bin/env python3
"""Aimed-grazing campaign (tests/grazing_cases.py) on the library MRT_LIB names: cameras whose primary rays run along
the walls of long leaves, hit IDs against the oracle.  usage: grazing_probe.py [walls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    n_walls = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    from raytracing_amd import scene_io
    import test_gpu_grazing
    scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
    frames, bad_frames, bad_rays, bad_colour, bad_pixels = test_gpu_grazing.run_campaign(scene, n_walls)
    print("grazing campaign: %d frames, hit ids differ on %d rays in %d frames, %d packet frames with other pixels, "
          "%d recursive frames beyond 1/255" % (frames, bad_rays, bad_frames, bad_pixels, bad_colour))
    return 1 if bad_rays or bad_pixels or bad_colour else 0


if __name__ == "__main__":
    sys.exit(main())
