import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from raytracing_amd import Camera, Raytracer, scene_io
w, h = 800, 600
scene = scene_io.load_scene(os.path.join(ROOT, "tests", "golden", "maze.mrts"))
rt = Raytracer(w, h, device=0); rt.build_bvh(scene)
fb = torch.zeros((h, w), dtype=torch.int32, device="cuda"); st = torch.cuda.current_stream()
cam = Camera()
def frame():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st); rt.render_device(cam, fb.data_ptr(), w, mode=1, stream=st.cuda_stream); e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1), rt.tile_schedule()
for k in range(6): print("rest", "%.3f" % frame()[0], frame()[1])
for k in range(16):
    x, y, z = cam.get_position(); cam.set_position((x + 0.16, y, z))
    ms, sch = frame(); print("walk", "%.3f" % ms, sch)
for k in range(4): ms, sch = frame(); print("rest", "%.3f" % ms, sch)
