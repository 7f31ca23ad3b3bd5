import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from raytracing_amd import binding
binding.LIB_PATH = os.path.join(ROOT, "raytracing_amd", "libmrt_diag.so")
from raytracing_amd import Raytracer, scene_io
maze = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
soup = scene_io.large_scene(maze)
rt = Raytracer(1920, 1080, device=0)
for k in range(2):
    print("--- lbvh upload", k, file=sys.stderr)
    rt.build_bvh(soup, lbvh=True)
rt.close()
