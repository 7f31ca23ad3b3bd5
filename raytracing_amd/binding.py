"""ctypes binding of libmrt.so, the C ABI declared in include/mrt.h.

The classes mirror the objects game.cpp drives (game.cpp:7-9, 15-35, 84-93): ``Raytracer``
(Init / BuildBVH / Render), ``Camera`` (SetPosition / LookAt) and a ``Surface`` framebuffer
that is simply a numpy ``uint32`` array.  There is no CPU fallback: loading fails loudly when
the library is missing, and creating a ``Raytracer`` fails when no GPU is present.
"""
import ctypes as C
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRT_LIB") or os.path.join(_HERE, "libmrt.so")     # MRT_LIB: tuning builds only

MODE_PACKET = 0
MODE_RECURSIVE = 1
MODE_BVH_DEPTH = 2

SYMBOLS = ("mrt_create", "mrt_destroy", "mrt_last_error", "mrt_version", "mrt_scene_upload", "mrt_scene_upload_lbvh", "mrt_scene_upload_lbvh_exact", "mrt_get_second_tree_use", "mrt_scene_refit",
           "mrt_set_lights", "mrt_set_objects",
           "mrt_bvh_build_host", "mrt_render", "mrt_host_alloc", "mrt_host_free", "mrt_render_submit", "mrt_render_wait", "mrt_render_device", "mrt_render_device_noalloc", "mrt_render_device_part", "mrt_part_rows",
           "mrt_raster_render", "mrt_raster_render_device", "mrt_raster_wait", "mrt_trace_packets",
           "mrt_trace_rays", "mrt_get_ray_counts", "mrt_get_tile_schedule", "mrt_camera_look_at")


class MrtError(RuntimeError):
    pass


class _Mesh(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("uv", C.c_void_p), ("tri_normal", C.c_void_p), ("tri", C.c_void_p),
                ("verts", C.c_int32), ("tris", C.c_int32), ("texture_id", C.c_int32),
                ("refl", C.c_float), ("refr", C.c_float), ("rindex", C.c_float)]


class _Object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pos", C.c_float * 3), ("param", C.c_float), ("diffuse", C.c_float * 4),
                ("refl", C.c_float), ("refr", C.c_float), ("rindex", C.c_float)]


class _Texture(C.Structure):
    _fields_ = [("indices", C.c_void_p), ("palettes", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


_lib = None


def _preload_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  A PyTorch-ROCm wheel ships its own libamdhip64.so
    and asks for it under a name the system copy does not answer to, so if libmrt.so pulled in
    /opt/rocm's copy first, a later `import torch` would load a second runtime and find no GPU.
    Loading torch's copy first (when torch is installed) lets both share it."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(p):
        C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """Load libmrt.so (built in-tree by raytracing_amd/csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MrtError("%s is missing: build it with `make -C raytracing_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        _preload_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int
        L.mrt_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
        L.mrt_destroy.argtypes = [vp]
        L.mrt_destroy.restype = None
        L.mrt_last_error.argtypes = [vp]
        L.mrt_last_error.restype = C.c_char_p
        L.mrt_scene_upload.argtypes = [vp, C.POINTER(_Mesh), i32, C.POINTER(_Texture), i32]
        L.mrt_scene_upload_lbvh.argtypes = [vp, C.POINTER(_Mesh), i32, C.POINTER(_Texture), i32]
        L.mrt_scene_upload_lbvh_exact.argtypes = [vp, C.POINTER(_Mesh), i32, C.POINTER(_Texture), i32]
        L.mrt_scene_refit.argtypes = [vp, C.POINTER(_Mesh), i32]
        L.mrt_set_lights.argtypes = [vp, vp, i32]
        L.mrt_set_objects.argtypes = [vp, C.POINTER(_Object), i32]
        L.mrt_bvh_build_host.argtypes = [C.POINTER(_Mesh), i32, vp, vp, vp, vp]
        L.mrt_render.argtypes = [vp, vp, C.c_uint32, vp, i32, vp]
        L.mrt_render_device.argtypes = [vp, vp, C.c_uint32, i32, i32, vp, i32, vp, vp]
        L.mrt_render_device_noalloc.argtypes = [vp, vp, C.c_uint32, i32, i32, vp, i32, vp, vp]
        L.mrt_host_alloc.argtypes = [C.c_size_t]
        L.mrt_host_alloc.restype = vp
        L.mrt_host_free.argtypes = [vp]
        L.mrt_host_free.restype = None
        L.mrt_render_submit.argtypes = [vp, vp, C.c_uint32, vp, i32, C.POINTER(i32)]
        L.mrt_render_wait.argtypes = [vp, i32]
        L.mrt_render_device_part.argtypes = [vp, vp, C.c_uint32, i32, i32, i32, i32, vp, i32, vp, vp]
        L.mrt_part_rows.argtypes = [i32, i32, i32, i32]
        L.mrt_raster_render.argtypes = [vp, vp, vp, i32, vp, vp]
        L.mrt_raster_render_device.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        if hasattr(L, "mrt_raster_wait"):
            L.mrt_raster_wait.argtypes = [vp]
        L.mrt_trace_packets.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
        L.mrt_trace_rays.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
        L.mrt_get_ray_counts.argtypes = [vp, vp]
        if hasattr(L, "mrt_get_tile_schedule"):         # (absent from older tuning builds named by MRT_LIB)
            L.mrt_get_tile_schedule.argtypes = [vp, vp]
        if hasattr(L, "mrt_get_second_tree_use"):
            L.mrt_get_second_tree_use.argtypes = [vp, vp]
        L.mrt_camera_look_at.argtypes = [vp, vp, vp]
        L.mrt_camera_look_at.restype = None
        _lib = L
    return _lib


class HostFrame:
    """A (height, width) uint32 frame in page-locked memory (mrt_host_alloc): mrt_render writes it directly."""

    def __init__(self, height, width):
        self.L = lib()
        self.ptr = self.L.mrt_host_alloc(height * width * 4)
        if not self.ptr:
            raise MrtError("mrt_host_alloc failed (no GPU, or out of page-locked memory)")
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, C.POINTER(C.c_uint32)), shape=(height, width))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.mrt_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


FLAG_FRAME_ORDER = 1 << 16       # MRT_FLAG_FRAME_ORDER
FLAG_NO_RAY_COUNTS = 1 << 17     # MRT_FLAG_NO_RAY_COUNTS


def flags(mode=MODE_PACKET, spp_grid=1, extra=0):
    return (mode & 0xFF) | ((spp_grid & 0xFF) << 8) | extra


def _pack_scene(scene):
    """Build the mrt_mesh / mrt_texture arrays; returns (meshes, textures, keepalive)."""
    keep = []
    meshes = (_Mesh * len(scene["meshes"]))()
    for i, m in enumerate(scene["meshes"]):
        pos = np.ascontiguousarray(m["pos"], np.float32)
        uv = np.ascontiguousarray(m["uv"], np.float32)
        N = np.ascontiguousarray(m["N"], np.float32)
        tri = np.ascontiguousarray(m["tri"], np.int32)
        keep += [pos, uv, N, tri]
        meshes[i] = _Mesh(_p(pos), _p(uv), _p(N), _p(tri), pos.size // 3, N.size // 3, int(m["texture"]),
                          float(m["refl"]), float(m["refr"]), float(m["rindex"]))
    textures = (_Texture * max(1, len(scene["textures"])))()
    for i, t in enumerate(scene["textures"]):
        idx = np.ascontiguousarray(t["idx"], np.uint8)
        pal = np.ascontiguousarray(t["pal"], np.uint32)
        keep += [idx, pal]
        textures[i] = _Texture(_p(idx), _p(pal), int(t["w"]), int(t["h"]))
    return meshes, textures, keep


def bvh_build_host(scene):
    """BVH::BVH on the host (no GPU needed): returns dict(info, node_minmax, node_first_cnt, tri_idx)."""
    L = lib()
    meshes, _, keep = _pack_scene(scene)
    ntris = sum(len(m["N"]) for m in scene["meshes"])
    slots = 2 * ntris + 4
    mm = np.zeros((slots, 6), np.float32)
    fc = np.zeros((slots, 2), np.uint32)
    idx = np.zeros(max(1, ntris), np.uint32)
    info = np.zeros(4, np.int64)
    rc = L.mrt_bvh_build_host(meshes, len(scene["meshes"]), _p(mm), _p(fc), _p(idx), _p(info))
    if rc != 0:
        raise MrtError("mrt_bvh_build_host: %s" % L.mrt_last_error(None).decode())
    del keep
    return {"nTris": int(info[0]), "nNodes": int(info[1]), "poolPtr": int(info[2]), "maxDepth": int(info[3]),
            "node_minmax": mm, "node_first_cnt": fc, "tri_idx": idx[:ntris]}


class Camera:
    """scenedata.h:125-137: a camera-to-world matrix (column-major) with SetPosition / LookAt."""

    def __init__(self):
        self.transform = np.eye(4, dtype=np.float32).reshape(16)

    def set_position(self, pos):
        self.transform[12:15] = np.asarray(pos, np.float32)

    def get_position(self):
        return self.transform[12:15].copy()

    def look_at(self, target):
        out = np.zeros(16, np.float32)
        lib().mrt_camera_look_at(_p(self.get_position()), _p(np.asarray(target, np.float32)), _p(out))
        self.transform = out


class Raytracer:
    """The reference's Raytracer + Rasterizer pair behind the C ABI (game.cpp:17-19, 34, 91-93)."""

    def __init__(self, width, height, device=-1):
        self.L = lib()
        self.width, self.height = width, height
        h = C.c_void_p()
        rc = self.L.mrt_create(width, height, device, C.byref(h))
        if rc != 0:
            raise MrtError("mrt_create failed (%d): %s" % (rc, self.L.mrt_last_error(None).decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.mrt_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise MrtError("%s failed (%d): %s" % (what, rc, self.L.mrt_last_error(self.h).decode()))

    def build_bvh(self, scene, lbvh=False):
        """Raytracer::BuildBVH(meshList): flatten, build, upload.  lbvh=True: GPU-built linear BVH instead
        of the reference's tree (fast, same distances, tie-breaks may differ).  lbvh="exact": that tree for the rays and
        the reference's beside it for the tie-breaks of the single-ray traversal (mrt_scene_upload_lbvh_exact)."""
        meshes, textures, keep = _pack_scene(scene)
        name = "mrt_scene_upload_lbvh_exact" if lbvh == "exact" else ("mrt_scene_upload_lbvh" if lbvh else "mrt_scene_upload")
        self._check(getattr(self.L, name)(self.h, meshes, len(scene["meshes"]), textures, len(scene["textures"])), name)
        del keep

    def refit(self, scene):
        """mrt_scene_refit: the same meshes with moved vertices; the tree is kept, its boxes are recomputed."""
        meshes, _, keep = _pack_scene(scene)
        self._check(self.L.mrt_scene_refit(self.h, meshes, len(scene["meshes"])), "mrt_scene_refit")
        del keep

    def set_lights(self, lights):
        a = np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        self._check(self.L.mrt_set_lights(self.h, _p(a), len(a)), "mrt_set_lights")

    def set_objects(self, objects):
        """Raytracer::m_Objects: an (n, 12) array of {kind (0 sphere, 1 plane), pos.xyz, radius | dist, diffuse.xyzw, refl, refr,
        rindex}; a thirteenth column is tolerated and ignored (rows used to be padded)."""
        a = np.asarray(objects, np.float32)
        if a.size == 0:
            a = a.reshape(0, 12)
        if a.ndim != 2 or a.shape[1] not in (12, 13):
            raise ValueError("objects must be an (n, 12) array (or (n, 13) with an unused last column), got shape %s" % (a.shape,))
        a = np.ascontiguousarray(a)
        arr = (_Object * max(1, len(a)))()
        for i, r in enumerate(a):
            arr[i] = _Object(int(r[0]), (C.c_float * 3)(*r[1:4]), float(r[4]), (C.c_float * 4)(*r[5:9]), float(r[9]), float(r[10]), float(r[11]))
        self._check(self.L.mrt_set_objects(self.h, arr, len(a)), "mrt_set_objects")

    def render(self, cam, mode=MODE_PACKET, spp_grid=1, want_ids=True, fb=None, extra_flags=0):
        """Raytracer::Render(camera) into a host framebuffer; returns (fb, hit_ids)."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        if fb is None:
            fb = np.zeros((self.height, self.width), np.uint32)
        g = max(1, spp_grid)
        ids = np.full((self.height * g, self.width * g), -2, np.int32) if want_ids else None
        self._check(self.L.mrt_render(self.h, _p(cam), flags(mode, spp_grid, extra_flags), _p(fb), fb.strides[0] // 4, _p(ids)), "mrt_render")
        return fb, ids

    def render_device(self, cam, fb_ptr, pitch, mode=MODE_PACKET, spp_grid=1, rows=None, ids_ptr=None, stream=None, extra_flags=0):
        """Asynchronous launch into device memory (raw pointers, e.g. torch.Tensor.data_ptr())."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        y0, y1 = rows if rows else (0, self.height)
        self._check(self.L.mrt_render_device(self.h, _p(cam), flags(mode, spp_grid, extra_flags), y0, y1, C.c_void_p(fb_ptr), pitch,
                                             C.c_void_p(ids_ptr) if ids_ptr else None,
                                             C.c_void_p(stream) if stream else None), "mrt_render_device")

    def render_device_noalloc(self, cam, fb_ptr, pitch, mode=MODE_PACKET, spp_grid=1, rows=None, ids_ptr=None, stream=None, extra_flags=0):
        """mrt_render_device_noalloc: the same launch with its scratch memory in a fixed array of the library."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        y0, y1 = rows if rows else (0, self.height)
        self._check(self.L.mrt_render_device_noalloc(self.h, _p(cam), flags(mode, spp_grid, extra_flags), y0, y1, C.c_void_p(fb_ptr), pitch,
                                                     C.c_void_p(ids_ptr) if ids_ptr else None,
                                                     C.c_void_p(stream) if stream else None), "mrt_render_device_noalloc")

    def render_device_part(self, cam, fb_ptr, pitch, part, n_parts, block_tile_rows=1, compact=False, mode=MODE_PACKET,
                           spp_grid=1, ids_ptr=None, stream=None):
        """One share of the interleaved tile-row split (multi-GPU): see mrt_render_device_part."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        self._check(self.L.mrt_render_device_part(self.h, _p(cam), flags(mode, spp_grid), part, n_parts, block_tile_rows,
                                                  1 if compact else 0, C.c_void_p(fb_ptr), pitch,
                                                  C.c_void_p(ids_ptr) if ids_ptr else None,
                                                  C.c_void_p(stream) if stream else None), "mrt_render_device_part")

    def submit(self, cam, fb, mode=MODE_PACKET, spp_grid=1):
        """Queues one frame into a page-locked frame (HostFrame.array) and returns its ticket: mrt_render_submit."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        ticket = C.c_int32(-1)
        self._check(self.L.mrt_render_submit(self.h, _p(cam), flags(mode, spp_grid), _p(fb), fb.strides[0] // 4, C.byref(ticket)),
                    "mrt_render_submit")
        return int(ticket.value)

    def wait(self, ticket):
        """Blocks until the frame with this ticket has landed in its page-locked frame: mrt_render_wait."""
        self._check(self.L.mrt_render_wait(self.h, ticket), "mrt_render_wait")

    def ray_counts(self):
        out = np.zeros(4, np.uint64)
        self._check(self.L.mrt_get_ray_counts(self.h, _p(out)), "mrt_get_ray_counts")
        return {"primary": int(out[0]), "shadow": int(out[1]), "secondary": int(out[2]), "primary_hits": int(out[3])}

    def tile_schedule(self):
        """mrt_get_tile_schedule: (tiles of the last launch, entries its queues handed out)."""
        out = np.zeros(2, np.uint64)
        self._check(self.L.mrt_get_tile_schedule(self.h, _p(out)), "mrt_get_tile_schedule")
        return int(out[0]), int(out[1])

    def second_tree_use(self):
        """mrt_get_second_tree_use: (the last render walked the scene's second tree, tiles it rendered again through the first)."""
        out = np.zeros(2, np.uint64)
        self._check(self.L.mrt_get_second_tree_use(self.h, _p(out)), "mrt_get_second_tree_use")
        return bool(out[0]), int(out[1])

    def raster(self, cam, fb=None, want_z=True, want_ids=True):
        """Rasterizer::Render(camera)."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        if fb is None:
            fb = np.zeros((self.height, self.width), np.uint32)
        z = np.zeros((self.height, self.width), np.float32) if want_z else None
        ids = np.full((self.height, self.width), -1, np.int32) if want_ids else None
        self._check(self.L.mrt_raster_render(self.h, _p(cam), _p(fb), fb.strides[0] // 4, _p(z), _p(ids)), "mrt_raster_render")
        return fb, z, ids

    def raster_device(self, cam, fb_ptr, pitch, z_ptr=None, ids_ptr=None, stream=None):
        """Rasterizer::Render(camera) into device memory (raw pointers).  The frame is queued on `stream`; raster_wait()
        waits for it and tells whether its tile lists fitted their buffer."""
        cam = np.ascontiguousarray(getattr(cam, "transform", cam), np.float32).reshape(16)
        self._check(self.L.mrt_raster_render_device(self.h, _p(cam), C.c_void_p(fb_ptr), pitch, C.c_void_p(z_ptr) if z_ptr else None,
                                                    C.c_void_p(ids_ptr) if ids_ptr else None, C.c_void_p(stream) if stream else None),
                    "mrt_raster_render_device")

    def raster_wait(self):
        """Waits for the last frame queued by raster_device; raises MrtError (MRT_E_LIMIT) when a frame since the last call
        was incomplete (queue it again: the buffer has grown)."""
        self._check(self.L.mrt_raster_wait(self.h), "mrt_raster_wait")

    def trace_packets(self, O, D, rD, t):
        """BVH::TraverseRayPacket on (n_packets, 64, 3) ray arrays; returns (t, tri, u, v)."""
        O = np.ascontiguousarray(O, np.float32)
        D = np.ascontiguousarray(D, np.float32)
        rD = np.ascontiguousarray(rD, np.float32)
        t = np.array(t, np.float32, copy=True).reshape(-1, 64)
        n = t.shape[0]
        tri = np.zeros((n, 64), np.int32)
        u = np.zeros((n, 64), np.float32)
        v = np.zeros((n, 64), np.float32)
        self._check(self.L.mrt_trace_packets(self.h, n, _p(O), _p(D), _p(rD), _p(t), _p(tri), _p(u), _p(v)), "mrt_trace_packets")
        return t, tri, u, v

    def trace_rays(self, O, D, t):
        """BVH::Traverse on (n, 3) ray arrays; returns (t, tri, u, v)."""
        O = np.ascontiguousarray(O, np.float32)
        D = np.ascontiguousarray(D, np.float32)
        t = np.array(t, np.float32, copy=True).reshape(-1)
        n = t.shape[0]
        tri = np.zeros(n, np.int32)
        u = np.zeros(n, np.float32)
        v = np.zeros(n, np.float32)
        self._check(self.L.mrt_trace_rays(self.h, n, _p(O), _p(D), _p(t), _p(tri), _p(u), _p(v)), "mrt_trace_rays")
        return t, tri, u, v
