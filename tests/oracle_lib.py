"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "build", "liboracle.so")
REF_ASSETS = "/root/reference/Project/assets"

_lib = None


def build():
    """Compile the oracle if a compiler is available and the library is missing or stale."""
    src = [os.path.join(ORACLE_DIR, f) for f in ("oracle.cpp", "ora_math.h")]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in src):
        return
    subprocess.check_call(["make", "-C", ORACLE_DIR, "build/liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i32, f32, u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
        L.ora_create.restype = vp
        L.ora_destroy.argtypes = [vp]
        L.ora_load_obj.argtypes = [vp, C.c_char_p, f32]
        L.ora_add_texture.argtypes = [vp, i32, i32, vp, vp]
        L.ora_add_mesh.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, i32, vp]
        L.ora_set_material.argtypes = [vp, i32, f32, f32, f32]
        L.ora_default_materials.argtypes = [vp]
        L.ora_n_meshes.argtypes = [vp]
        L.ora_n_textures.argtypes = [vp]
        L.ora_mesh_info.argtypes = [vp, i32, vp, vp, vp, vp]
        L.ora_mesh_data.argtypes = [vp, i32, vp, vp, vp, vp, vp]
        L.ora_texture_info.argtypes = [vp, i32, vp, vp]
        L.ora_texture_data.argtypes = [vp, i32, vp, vp]
        L.ora_set_lights.argtypes = [vp, vp, i32]
        L.ora_build_bvh.argtypes = [vp]
        L.ora_bvh_info.argtypes = [vp, vp]
        L.ora_bvh_arrays.argtypes = [vp, vp, vp, vp]
        L.ora_idxhash.argtypes = [vp]
        L.ora_idxhash.restype = u64
        L.ora_nodehash.argtypes = [vp]
        L.ora_nodehash.restype = u64
        L.ora_fbhash.argtypes = [vp, C.c_int64]
        L.ora_fbhash.restype = u64
        L.ora_idsum.argtypes = [vp, i32, i32]
        L.ora_idsum.restype = u64
        L.ora_lookat.argtypes = [vp, vp, vp]
        L.ora_mat4_inverse.argtypes = [vp, vp]
        L.ora_view_plane.argtypes = [vp, i32, i32, vp]
        L.ora_set_threads.argtypes = [vp, i32]
        L.ora_set_threads_pinned.argtypes = [vp, i32]
        L.ora_render.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp]
        L.ora_trace_ray.argtypes = [vp, vp, vp, f32, i32, vp, vp]
        L.ora_trace_ray_brute.argtypes = [vp, vp, vp, f32, vp, vp]
        L.ora_trace_packets.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.ora_trace_rays.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        L.ora_gen_ray.argtypes = [vp, i32, i32, i32, i32, vp, vp]
        L.ora_raster_render.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.ora_raster_work.argtypes = [vp, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


IDENTITY = np.eye(4, dtype=np.float32).reshape(16)
COUNTER_NAMES = ("packet_box", "packet_tri", "packet_nodes", "packet_leaves", "scalar_box", "scalar_tri",
                 "scalar_nodes", "scalar_rays", "primary_rays", "shadow_rays", "secondary_rays", "packet_trifetch")


def lookat(pos, target):
    out = np.zeros(16, np.float32)
    lib().ora_lookat(_p(np.asarray(pos, np.float32)), _p(np.asarray(target, np.float32)), _p(out))
    return out


class Oracle:
    """One oracle context: a scene, its BVH and a tile worker pool."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.ora_create()

    def close(self):
        if self.h:
            self.L.ora_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- scene ----
    def load_obj(self, path, scale=1.0):
        if self.L.ora_load_obj(self.h, path.encode(), scale) != 0:
            raise IOError(path)

    def add_scene(self, scene):
        """scene: dict from scene_io.load_scene (textures, meshes)."""
        for t in scene["textures"]:
            idx = np.ascontiguousarray(t["idx"], np.uint8)
            pal = np.ascontiguousarray(t["pal"], np.uint32)
            self.L.ora_add_texture(self.h, t["w"], t["h"], _p(idx), _p(pal))
        for m in scene["meshes"]:
            arrs = {k: np.ascontiguousarray(m[k], np.float32) for k in ("pos", "uv", "norm", "N", "bounds")}
            tri = np.ascontiguousarray(m["tri"], np.int32)
            i = self.L.ora_add_mesh(self.h, len(m["pos"]), len(m["N"]), _p(arrs["pos"]), _p(arrs["uv"]), _p(arrs["norm"]),
                                    _p(arrs["N"]), _p(tri), int(m["texture"]), _p(arrs["bounds"]))
            self.L.ora_set_material(self.h, i, float(m["refl"]), float(m["refr"]), float(m["rindex"]))

    def default_materials(self):
        self.L.ora_default_materials(self.h)

    def export_scene(self):
        L, h = self.L, self.h
        scene = {"textures": [], "meshes": []}
        for i in range(L.ora_n_textures(h)):
            w, hh = C.c_int(), C.c_int()
            L.ora_texture_info(h, i, C.byref(w), C.byref(hh))
            idx = np.zeros(w.value * hh.value, np.uint8)
            pal = np.zeros(32 * 256, np.uint32)
            L.ora_texture_data(h, i, _p(idx), _p(pal))
            scene["textures"].append({"w": w.value, "h": hh.value, "idx": idx, "pal": pal})
        for i in range(L.ora_n_meshes(h)):
            v, t, tx = C.c_int(), C.c_int(), C.c_int()
            b = np.zeros(6, np.float32)
            L.ora_mesh_info(h, i, C.byref(v), C.byref(t), C.byref(tx), _p(b))
            pos = np.zeros((v.value, 3), np.float32)
            uv = np.zeros((v.value, 2), np.float32)
            norm = np.zeros((v.value, 3), np.float32)
            N = np.zeros((t.value, 3), np.float32)
            tri = np.zeros((t.value, 3), np.int32)
            L.ora_mesh_data(h, i, _p(pos), _p(uv), _p(norm), _p(N), _p(tri))
            scene["meshes"].append({"pos": pos, "uv": uv, "norm": norm, "N": N, "tri": tri, "texture": tx.value,
                                    "bounds": b, "refl": 0.0, "refr": 0.0, "rindex": 1.0})
        return scene

    def set_lights(self, lights):
        a = np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        self.L.ora_set_lights(self.h, _p(a), len(a))

    # ---- BVH ----
    def build_bvh(self):
        self.L.ora_build_bvh(self.h)
        info = np.zeros(5, np.int64)
        self.L.ora_bvh_info(self.h, _p(info))
        self.bvh_info = dict(nTris=int(info[0]), nNodes=int(info[1]), poolPtr=int(info[2]), maxDepth=int(info[3]))
        return self.bvh_info

    def bvh_arrays(self):
        n = self.bvh_info["poolPtr"] + 1
        mm = np.zeros((n, 6), np.float32)
        fc = np.zeros((n, 2), np.uint32)
        idx = np.zeros(self.bvh_info["nTris"], np.uint32)
        self.L.ora_bvh_arrays(self.h, _p(mm), _p(fc), _p(idx))
        return mm, fc, idx

    def idxhash(self):
        return int(self.L.ora_idxhash(self.h))

    def nodehash(self):
        return int(self.L.ora_nodehash(self.h))

    # ---- rendering ----
    def set_objects(self, objects):
        """Analytic primitives: rows of 12 floats {kind (0 sphere, 1 plane), pos.xyz, radius | dist, diffuse.xyzw, refl, refr, rindex} (a 13th column is ignored)."""
        a = np.asarray(objects, np.float32)
        a = a.reshape(0, 13) if a.size == 0 else a
        if a.ndim != 2 or a.shape[1] not in (12, 13):
            raise ValueError("objects must be (n, 12) or (n, 13)")
        if a.shape[1] == 12:
            a = np.concatenate([a, np.zeros((len(a), 1), np.float32)], axis=1)
        a = np.ascontiguousarray(a)
        self.L.ora_set_objects.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.L.ora_set_objects(self.h, _p(a), len(a))

    def set_threads(self, n, pinned=False):
        """Worker threads of the tile pool; pinned=True keeps each on its own CPU (distinct physical cores first) and
        returns how many distinct cores they occupy (0: pinning refused by the system)."""
        if pinned:
            return self.L.ora_set_threads_pinned(self.h, n)
        self.L.ora_set_threads(self.h, n)
        return 0

    def render(self, cam, W, H, mode=0, spp_grid=1, rows=None, want_ids=True, want_tuv=False, want_counters=False):
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        fb = np.zeros((H, W), np.uint32)
        g = max(1, spp_grid)
        ids = np.full((H * g, W * g), -2, np.int32) if want_ids else None
        tuv = np.zeros((H * g, W * g, 3), np.float32) if want_tuv else None
        ctr = np.zeros(len(COUNTER_NAMES), np.uint64) if want_counters else None
        y0, y1 = rows if rows else (0, H)
        rc = self.L.ora_render(self.h, _p(cam), W, H, mode, spp_grid, y0, y1, _p(fb), W, _p(ids), _p(tuv), _p(ctr))
        if rc != 0:
            raise RuntimeError("ora_render failed")
        out = {"fb": fb, "ids": ids, "tuv": tuv}
        if want_counters:
            out["counters"] = dict(zip(COUNTER_NAMES, (int(x) for x in ctr)))
        return out

    def render_band(self, cam, W, H, y0, y1, mode=0, spp_grid=1, want_counters=False):
        """Rows [y0, y1) of the frame only, without allocating the whole frame (full-size configurations):
        returns the band's pixels (y1-y0, W) and hit ids ((y1-y0)*g, W*g)."""
        cam = np.ascontiguousarray(cam, np.float32).reshape(16)
        g = max(1, spp_grid)
        fb = np.zeros((y1 - y0, W), np.uint32)
        ids = np.full(((y1 - y0) * g, W * g), -2, np.int32)
        # the library indexes both arrays by frame row; hand it the address frame row 0 would have
        fb0 = C.c_void_p(fb.ctypes.data - y0 * W * 4)
        ids0 = C.c_void_p(ids.ctypes.data - y0 * g * W * g * 4)
        ctr = np.zeros(len(COUNTER_NAMES), np.uint64) if want_counters else None
        rc = self.L.ora_render(self.h, _p(cam), W, H, mode, spp_grid, y0, y1, fb0, W, ids0, None, _p(ctr))
        if rc != 0:
            raise RuntimeError("ora_render failed")
        if want_counters:
            return fb, ids, dict(zip(COUNTER_NAMES, (int(x) for x in ctr)))
        return fb, ids

    def trace_ray(self, O, D, tmax=1e8, kind=0):
        t = C.c_float()
        uv = np.zeros(2, np.float32)
        i = self.L.ora_trace_ray(self.h, _p(np.asarray(O, np.float32)), _p(np.asarray(D, np.float32)), tmax, kind,
                                 C.byref(t), _p(uv))
        return i, t.value, uv

    def trace_ray_brute(self, O, D, tmax=1e8):
        t = C.c_float()
        uv = np.zeros(2, np.float32)
        i = self.L.ora_trace_ray_brute(self.h, _p(np.asarray(O, np.float32)), _p(np.asarray(D, np.float32)), tmax,
                                       C.byref(t), _p(uv))
        return i, t.value, uv

    def trace_packets(self, O, D, rD, t):
        O = np.ascontiguousarray(O, np.float32)
        D = np.ascontiguousarray(D, np.float32)
        rD = np.ascontiguousarray(rD, np.float32)
        t = np.array(t, np.float32, copy=True).reshape(-1, 64)
        n = t.shape[0]
        tri = np.zeros((n, 64), np.int32)
        u = np.zeros((n, 64), np.float32)
        v = np.zeros((n, 64), np.float32)
        self.L.ora_trace_packets(self.h, n, _p(O), _p(D), _p(rD), _p(t), _p(tri), _p(u), _p(v))
        return t, tri, u, v

    def trace_rays(self, O, D, t):
        O = np.ascontiguousarray(O, np.float32)
        D = np.ascontiguousarray(D, np.float32)
        t = np.array(t, np.float32, copy=True).reshape(-1)
        n = t.shape[0]
        tri = np.zeros(n, np.int32)
        u = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        self.L.ora_trace_rays(self.h, n, _p(O), _p(D), _p(t), _p(tri), _p(u), _p(v))
        return t, tri, u, v

    def gen_ray(self, cam, W, H, x, y):
        D = np.zeros(3, np.float32)
        rD = np.zeros(3, np.float32)
        self.L.ora_gen_ray(_p(np.ascontiguousarray(cam, np.float32)), W, H, x, y, _p(D), _p(rD))
        return D, rD

    def raster(self, cam, W, H, want_ids=True):
        fb = np.zeros((H, W), np.uint32)
        z = np.zeros((H, W), np.float32)
        ids = np.full((H, W), -1, np.int32) if want_ids else None
        self.L.ora_raster_render(self.h, _p(np.ascontiguousarray(cam, np.float32)), W, H, _p(fb), _p(z), _p(ids))
        work = np.zeros(3, np.uint64)
        self.L.ora_raster_work(self.h, _p(work))
        return {"fb": fb, "z": z, "ids": ids, "fragments": int(work[0]), "fragments_drawn": int(work[1]), "triangles_drawn": int(work[2])}


def fbhash(fb):
    fb = np.ascontiguousarray(fb, np.uint32)
    return int(lib().ora_fbhash(_p(fb), fb.size))


def fbhash_continue(h0, fb):
    """fbhash of a frame built up from its row bands, top to bottom: h = fbhash_continue(h, band)."""
    fb = np.ascontiguousarray(fb, np.uint32)
    L = lib()
    L.ora_fbhash_from.restype = C.c_uint64
    L.ora_fbhash_from.argtypes = [C.c_uint64, C.c_void_p, C.c_int64]
    return int(L.ora_fbhash_from(h0, _p(fb), fb.size))


def idsum(ids):
    ids = np.ascontiguousarray(ids, np.int32)
    return int(lib().ora_idsum(_p(ids), ids.shape[1], ids.shape[0]))
