/*
 * mrt.h -- C ABI of the MI355X ray/scene intersection path ("mrt" = MI355X ray tracer).
 *
 * The reference program has no FFI layer: its boundary is the set of C++ member calls that
 * game.cpp makes on its global Raytracer / Rasterizer / Camera objects.  Each entry point
 * below names the reference call it stands in for (paths relative to the reference's
 * Project/ directory).  All pointers are plain host pointers unless a parameter says
 * "device"; no C++ or framework types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success and a negative MRT_E_* code otherwise;
 *     mrt_last_error() gives the text of the last failure on that context.
 *   - colours are 32-bit 0xAARRGGBB words (surface.h:16 Pixel); hit pixels have alpha 0 and
 *     a miss is 0xff4c4c4c, exactly as the reference writes them.
 *   - matrices are 16 floats, column-major (glm::mat4 memory layout), camera-to-world
 *     (scenedata.h:136 Camera::transform).
 *   - a context is bound to one GPU and is used from one thread at a time (the reference's
 *     Render is not re-entrant either: threads.cpp:155-157 keeps its state in statics).
 */
#ifndef MRT_H_
#define MRT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRT_VERSION 1

/* error codes */
#define MRT_OK            0
#define MRT_E_ARG        -1   /* bad argument (null pointer, size not a multiple of 8, ...) */
#define MRT_E_STATE      -2   /* call order (render before scene upload, ...) */
#define MRT_E_HIP        -3   /* a HIP runtime call failed */
#define MRT_E_NO_DEVICE  -4   /* no usable GPU: this library has no CPU fallback */
#define MRT_E_LIMIT      -5   /* scene exceeds a documented limit (BVH depth, counts) */

/* render flags (mrt_render / mrt_render_device) */
#define MRT_MODE_PACKET     0u        /* Raytracer::Render: 64-ray packets + direct light (raytracer.cpp:420-463, 524-574) */
#define MRT_MODE_RECURSIVE  1u        /* per-pixel GetColorFromSphere: reflection/refraction (raytracer.cpp:465-490, 575-664) */
#define MRT_MODE_BVH_DEPTH  2u        /* the BVH-depth view (raytracer.cpp:581-586, 1265-1326): whole frames, no sample grid */
#define MRT_MODE_MASK       0xffu
#define MRT_SPP_GRID(g)     (((uint32_t)(g) & 0xffu) << 8)   /* g in {1,2,4,8}: g*g samples per pixel */
#define MRT_FLAG_FRAME_ORDER   (1u << 16)  /* hand tiles out in frame order, like the counter of threads.cpp:165, instead of
                                              starting with the tiles that were expensive in the previous frame (same pixels) */
#define MRT_FLAG_NO_RAY_COUNTS (1u << 17)  /* do not accumulate the counters mrt_get_ray_counts reads */

typedef struct mrt_ctx mrt_ctx;

/* One mesh, as scenedata.h:143-167 Mesh exposes it to Raytracer::BuildBVH, plus the three
 * material scalars that BVH::BVH assigns per mesh (raytracer.cpp:908-929). */
typedef struct mrt_mesh {
    const float*   pos;         /* verts x 3, object space = world space */
    const float*   uv;          /* verts x 2 (v already flipped, scenedata.cpp:181) */
    const float*   tri_normal;  /* tris x 3, unit face normals (scenedata.cpp:251-256) */
    const int32_t* tri;         /* tris x 3 vertex indices */
    int32_t verts, tris;
    int32_t texture_id;         /* index into the textures array.  Every mesh that has triangles must carry a texture
                                   (Triangle::GetColor reads it unconditionally, raytracer.cpp:362-377): mrt_scene_upload
                                   fails with MRT_E_ARG on -1 or an index outside the array */
    float   refl, refr, rindex; /* Object::m_Refl / m_Refr / m_RIndex (raytracer.h:76-78) */
} mrt_mesh;

/* One paletted texture, as surface.h:41-55 Surface8 holds it after loading a .bin cache
 * (surface.cpp:46-59): width*height palette indices and 32 pre-scaled 256-entry palettes. */
typedef struct mrt_texture {
    const uint8_t*  indices;    /* width x height */
    const uint32_t* palettes;   /* 32 x 256, 0x00RRGGBB */
    int32_t width, height;
} mrt_texture;

/* ---- life cycle --------------------------------------------------------------------- */

/* Raytracer::Init + Rasterizer::Init (game.cpp:17-18; raytracer.cpp:384-413, rasterizer.cpp:21-46).
 * width/height must be multiples of 8 (game.h:8-9).  device = GPU ordinal, -1 = the current
 * one.  Creates the default light (0,5,0)/(1,1,1,1).  Fails with MRT_E_NO_DEVICE without a GPU. */
int  mrt_create(int width, int height, int device, mrt_ctx** out);
void mrt_destroy(mrt_ctx* ctx);
const char* mrt_last_error(const mrt_ctx* ctx);   /* ctx may be NULL: last error of mrt_create */
int  mrt_version(void);

/* ---- scene -------------------------------------------------------------------------- */

/* Raytracer::BuildBVH(meshList) (game.cpp:34; raytracer.cpp:670-673, 897-944): flattens the
 * meshes into the reference's triangle order, builds the identical BVH on the host, converts
 * it to the device layout and uploads it.  Mesh data is copied (as raytracer.h:158-170 does). */
int mrt_scene_upload(mrt_ctx* ctx, const mrt_mesh* meshes, int n_meshes,
                     const mrt_texture* textures, int n_textures);

/* Same call with the acceleration structure built on the GPU (Morton order, bottom-up clustering by surface area,
 * leaves chosen by the surface area heuristic) instead of by the reference's builder (raytracer.cpp:897-1203):
 * milliseconds instead of seconds for large scenes, and a better tree -- the default scene renders about 1.5 times as
 * fast on it as on the reference's.  The nearest distances found are the same; the reported triangle may be the other
 * one of two hit at exactly the same distance, because that choice follows the reference tree's leaf order.  Not for
 * bit-exact comparison with the reference, which is why mrt_scene_upload stays the default. */
int mrt_scene_upload_lbvh(mrt_ctx* ctx, const mrt_mesh* meshes, int n_meshes,
                          const mrt_texture* textures, int n_textures);

/* Raytracer::BuildBVH(meshList) (game.cpp:34; raytracer.cpp:670-673) with two trees: everything mrt_scene_upload
 * does -- the reference's own tree (raytracer.cpp:897-1203) serves every call as it does there --, and beside it the
 * GPU-built tree of mrt_scene_upload_lbvh over the same triangles.  Frames of the recursive shader and of the packet
 * path (MRT_MODE_RECURSIVE, MRT_MODE_PACKET; a scene that fits the LDS, with coordinates of ordinary size -- zero or
 * between 2^-27 and 2^30 --, no analytic primitives, the eye within 64 times the scene's largest coordinate of the
 * origin; any other frame walks the reference's tree as after mrt_scene_upload) walk the second tree,
 * which is the faster one, and report what BVH::Traverse and BVH::TraverseRayPacket (raytracer.cpp:1204-1264,
 * 1328-1464) report on the reference's tree: a ray's result is taken from the second tree only where it can be shown to
 * be the reference's (no other triangle as near to within rounding, and the ray passes the hit triangle's leaf box of
 * the reference's tree by the reference's arithmetic); a ray for which that cannot be shown -- on the packet path its
 * whole packet -- is traced again through the reference's tree within the same kernel.  Hit IDs, pixels and ray counts
 * are those of mrt_scene_upload; only the time differs.  The upload costs both builds.  After mrt_scene_refit the
 * second tree is no longer used. */
int mrt_scene_upload_lbvh_exact(mrt_ctx* ctx, const mrt_mesh* meshes, int n_meshes,
                                const mrt_texture* textures, int n_textures);

/* Moving geometry without a new tree.  The reference can only build again (Raytracer::BuildBVH, raytracer.cpp:670-673,
 * the whole of BVH::BVH); this call keeps the tree the context holds -- the reference's or the GPU-built one -- takes the
 * meshes' new vertex positions, normals and uvs (same meshes in the same order with the same triangle counts and vertex
 * indices' meaning; MRT_E_ARG otherwise, and the old geometry stays in place), recomputes every node's box bottom-up and
 * uploads.  Textures and lights stay.  Nearest hits are those of the moved geometry; as with any tree other than the
 * one the reference would build for it, the triangle named on an exact distance tie may be the other one. */
int mrt_scene_refit(mrt_ctx* ctx, const mrt_mesh* meshes, int n_meshes);

/* Raytracer::m_Lights (raytracer.h:244; game.cpp:69-70 moves light 0): n records of
 * {pos.xyz, colour.xyzw} = 7 floats, colour in the reference's x=A,y=R,z=G,w=B order. */
int mrt_set_lights(mrt_ctx* ctx, const float* pos_rgba7, int n);

/* Raytracer::m_Objects (raytracer.h:243): the analytic primitives of raytracer.h:84-123 / raytracer.cpp:146-213, Sphere
 * (pos = centre, param = radius) and Plane (pos = normal, param = m_Dist), with the reference's Intersect, IsOccluded
 * and HitNormal.  The reference fills this list (two spheres, raytracer.cpp:404-412) but never traces it; this library
 * does not create them by itself either.  When a list is set, every ray that has searched the BVH also intersects the
 * objects in list order (the nearest hit wins, a triangle on an exact tie), an object is shaded with its normal, its
 * diffuse colour (x=A,y=R,z=G,w=B like every colour here) and its own refl / refr / rindex, shadow rays are blocked by
 * any object whose IsOccluded says so, and hit_ids reports object k as -2 - k.  n = 0 removes the list. */
#define MRT_OBJECT_SPHERE 0
#define MRT_OBJECT_PLANE  1
typedef struct mrt_object {
    int32_t kind;
    float   pos[3];
    float   param;
    float   diffuse[4];
    float   refl, refr, rindex;
} mrt_object;
int mrt_set_objects(mrt_ctx* ctx, const mrt_object* objects, int n);

/* Host-only BVH build with the arrays handed back, for inspection and tests; needs no GPU
 * and no context.  BVH::BVH + BVHNode::Subdivide (raytracer.cpp:897-1203).
 *   node_minmax   : (2*tris+4) x 6 floats   {min.xyz, max.xyz}        (raytracer.h:251-255)
 *   node_first_cnt: (2*tris+4) x 2 uint32   {firstLeft, count}        (raytracer.h:331-332)
 *   tri_idx       : tris uint32             BVH::m_TriangleIdx        (raytracer.h:357)
 *   info[4]       : nTris, nNodes (allocated), final poolPtr, maxDepth
 * Slots 1 and 2 of the node arrays are unused, as in the reference (poolPtr starts at 2). */
int mrt_bvh_build_host(const mrt_mesh* meshes, int n_meshes,
                       float* node_minmax, uint32_t* node_first_cnt, uint32_t* tri_idx, int64_t* info);

/* ---- rendering ---------------------------------------------------------------------- */

/* Raytracer::Render(camera) (game.cpp:91; raytracer.cpp:420-463 with threads.cpp:158-225).
 * Synchronous.  fb: host, height rows of pitch_px pixels, every pixel of every tile is
 * written.  hit_ids: optional host array width*height (times g*g with MRT_SPP_GRID(g)),
 * the primary ray's triangle index into the flattened triangle list, -1 = miss. */
int mrt_render(mrt_ctx* ctx, const float cam_to_world[16], uint32_t flags,
               uint32_t* fb, int pitch_px, int32_t* hit_ids);

/* Optional: page-locked host memory for the caller's Surface (the reference allocates its pixel
 * buffer with a 64-byte aligned malloc, surface.cpp:114-120 / template.cpp:98).  mrt_render
 * recognises a framebuffer that lies in such memory and lets the kernel write the frame straight
 * into it (the transfer then runs under the rendering); any other host pointer goes through an internal page-locked staging buffer and one
 * extra memcpy.  Returns NULL when no GPU is present or memory is short. */
void* mrt_host_alloc(size_t bytes);
void mrt_host_free(void* p);

/* The frame loop with the next frame in flight.  The reference's loop is Tick (which renders)
 * followed by presenting the Surface (template.cpp:111-127); with these two calls the caller
 * presents frame i while frame i+1 renders.  mrt_render_submit is mrt_render without hit IDs that
 * returns once the frame is queued: fb MUST lie in page-locked memory (mrt_host_alloc; MRT_E_ARG
 * otherwise) and must not be read, written or freed until mrt_render_wait(ticket) has returned
 * for the ticket handed back.  Frames complete in submission order.  Up to MRT_MAX_IN_FLIGHT
 * frames may be queued; one more fails with MRT_E_STATE.  mrt_render_wait reports MRT_E_LIMIT if
 * that frame exhausted a traversal budget.  mrt_get_ray_counts refers to the frame submitted
 * last.  mrt_destroy waits for queued frames. */
#define MRT_MAX_IN_FLIGHT 4
int mrt_render_submit(mrt_ctx* ctx, const float cam_to_world[16], uint32_t flags,
                      uint32_t* fb, int pitch_px, int* ticket);
int mrt_render_wait(mrt_ctx* ctx, int ticket);

/* Same frame, device-resident and asynchronous: renders tile rows [y_begin, y_end) (multiples
 * of 8) into DEVICE memory fb_dev (row y at fb_dev + y*pitch_px) on `stream` (a hipStream_t,
 * NULL = the default stream) and returns without synchronising.  This is the launch the
 * tile scheduler of threads.cpp:158-169 / raytracer.cpp:436-457 is replaced by; a multi-GPU
 * caller gives each rank its own row band.  hit_ids_dev optional (device, full-frame indexing).
 * Launches of one context share its tile queues and counters: a launch on another stream than
 * the previous one first waits (on the device) for that one to finish. */
int mrt_render_device(mrt_ctx* ctx, const float cam_to_world[16], uint32_t flags,
                      int y_begin, int y_end, void* fb_dev, int pitch_px,
                      void* hit_ids_dev, void* stream);

/* mrt_render_device for callers that cannot allocate device memory: tight serving loops, harnesses that forbid
 * hipMalloc.  (NOT for recording into a HIP graph: which kernels a frame needs is decided on the host at every call --
 * the eye-relative arrays are rewritten only when the eye or the scene has changed, the tile order is made only when
 * the previous launch had the same shape -- and the call records, and may wait on, an event of the context; a recorded
 * graph would replay one call's decisions for every later frame.)  It has NO counterpart in the reference.  Same arguments,
 * same kernels, same pixels; but where mrt_render_device keeps the frame's counter block (tile queues,
 * ray counters, watchdog word) and the per-tile cost / order arrays of the tile scheduler in buffers it
 * allocates on demand, this call keeps them in a fixed array of the library's code object (about 4.2 MB
 * on every GPU the library runs on, until the process ends) and calls no hipMalloc / hipFree or other
 * allocating routine.  A launch of more than MRT_NOALLOC_MAX_TILES 8x8 tiles (traced size, i.e. times g*g
 * with MRT_SPP_GRID(g)) is refused with MRT_E_LIMIT.  ALL contexts on one GPU share the array and it keeps
 * the last call's contents: the call clears what must start at zero itself, but two such calls that
 * overlap in time on one GPU (different contexts or different processes' streams of one process) race --
 * order them on one stream or with events.  mrt_get_ray_counts and mrt_get_tile_schedule after it read the same array,
 * and fail with MRT_E_STATE when another context has rendered into it in between. */
#define MRT_NOALLOC_MAX_TILES 262144
int mrt_render_device_noalloc(mrt_ctx* ctx, const float cam_to_world[16], uint32_t flags,
                              int y_begin, int y_end, void* fb_dev, int pitch_px,
                              void* hit_ids_dev, void* stream);

/* The same launch for one share of an INTERLEAVED split of the frame.  The reference deals 8x8
 * tiles to its workers one at a time from a shared counter (threads.cpp:158-169), which balances
 * cheap sky tiles against expensive ones; the counterpart across GPUs is to deal 8-row tile rows
 * in blocks: share `part` of `n_parts` renders the tile rows R with
 * (R / block_tile_rows) % n_parts == part, in increasing order.  compact = 0: each row is written
 * at its place in the frame (fb_dev + y*pitch_px).  compact = 1: the share's rows are written
 * packed, its k-th pixel row at fb_dev + k*pitch_px (the buffer a rank sends to rank 0);
 * hit_ids_dev follows the same choice.  mrt_part_rows gives the number of pixel rows of a share. */
int mrt_render_device_part(mrt_ctx* ctx, const float cam_to_world[16], uint32_t flags,
                           int part, int n_parts, int block_tile_rows, int compact,
                           void* fb_dev, int pitch_px, void* hit_ids_dev, void* stream);

/* Pixel rows in share `part` of the split above for a frame of `height` rows (no context needed);
 * negative on bad arguments. */
int mrt_part_rows(int height, int part, int n_parts, int block_tile_rows);

/* Rasterizer::Render(camera) (game.cpp:93; rasterizer.cpp:53-184).  fb as above; zbuf
 * optional host width*height floats receiving the 1/z buffer (rasterizer.cpp:12,55);
 * tri_ids optional host width*height, depth-winner triangle index, -1 = uncovered.
 * Pixels no triangle covers are left untouched in fb (the reference clears before: game.cpp:87). */
int mrt_raster_render(mrt_ctx* ctx, const float cam_to_world[16],
                      uint32_t* fb, int pitch_px, float* zbuf, int32_t* tri_ids);

/* The same frame into DEVICE memory, on `stream`: fb_dev (rows of pitch_px pixels, covered pixels only are
 * written), zbuf_dev and tri_ids_dev (optional, width*height each; every pixel written).  The frame is queued and
 * the call returns; nothing is read back.  Frames of one context are queued on one stream at a time.  The lists of
 * triangles per screen tile live in a buffer sized at mrt_scene_upload (four entries per triangle and 65,536 more);
 * a frame that needs more is drawn without the entries that did not fit, and mrt_raster_wait says so. */
int mrt_raster_render_device(mrt_ctx* ctx, const float cam_to_world[16],
                             void* fb_dev, int pitch_px, void* zbuf_dev, void* tri_ids_dev, void* stream);

/* Waits for the last frame queued by mrt_raster_render_device (the counterpart of mrt_render_wait for the
 * rasterizer; no counterpart in the reference, whose Rasterizer::Render returns a finished frame).  MRT_OK: every
 * frame queued since the last call is complete.  MRT_E_LIMIT: one of them needed longer tile lists than the buffer
 * held and is incomplete; the buffer has grown to what that frame needed: queue the frame again. */
int mrt_raster_wait(mrt_ctx* ctx);

/* ---- operator-level seam ------------------------------------------------------------ */

/* BVH::TraverseRayPacket (raytracer.h:352; raytracer.cpp:1349-1464) on caller-made packets.
 * Arrays are host, packet-major: O/D/rD n_packets x 64 x 3 floats, t n_packets x 64 (in: ray
 * length, out: hit distance), out tri n_packets x 64 (-1 = none), u, v n_packets x 64. */
int mrt_trace_packets(mrt_ctx* ctx, int n_packets, const float* O, const float* D, const float* rD,
                      float* t, int32_t* tri, float* u, float* v);

/* BVH::Traverse (raytracer.h:350; raytracer.cpp:945-950, 1204-1264) on n single rays,
 * including the root gate exactly as the reference evaluates it.  O/D n x 3, t n (in/out). */
int mrt_trace_rays(mrt_ctx* ctx, int n_rays, const float* O, const float* D,
                   float* t, int32_t* tri, float* u, float* v);

/* ---- measurement -------------------------------------------------------------------- */

/* Ray counts of the most recent completed render on this context (device counters):
 * out[0] primary rays, out[1] shadow rays, out[2] reflection+refraction rays,
 * out[3] primary hits.  Synchronises the context's last stream. */
int mrt_get_ray_counts(mrt_ctx* ctx, uint64_t out[4]);

/* How the most recent render's tiles were handed to the GPU's waves: what stands in for the shared tile counter of
 * threads.cpp:158-169 / raytracer.cpp:436-444.  out[0] = 8x8 tiles of the launch; out[1] = entries its tile queues handed
 * out -- the same number, or more when the recursive shader's most expensive tiles of the previous frame were handed out
 * as sixteen 2x2-pixel parts each (15 more entries per such tile) so that a small frame does not wait for one wave.
 * Synchronises like mrt_get_ray_counts.  The pixels do not depend on any of this. */
int mrt_get_tile_schedule(mrt_ctx* ctx, uint64_t out[2]);

/* Scenes uploaded with mrt_scene_upload_lbvh_exact (no counterpart in the reference, which has one tree): out[0] = 1 if
 * the most recent render walked the scene's second tree, else 0; out[1] = rays of that render (packet path: packets)
 * that were traced again through the reference's tree because a result could not be shown to be the reference's from
 * the second tree alone.  Synchronises like mrt_get_ray_counts.  The pixels do not depend on any of this. */
int mrt_get_second_tree_use(mrt_ctx* ctx, uint64_t out[2]);

/* Camera::LookAt (scenedata.h:129-134): out = inverse(lookAt(pos, target, (0,1,0))) with the
 * vendored glm's operation order, so callers can form bit-identical camera matrices. */
void mrt_camera_look_at(const float pos[3], const float target[3], float out_cam_to_world[16]);

#ifdef __cplusplus
}
#endif
#endif /* MRT_H_ */
