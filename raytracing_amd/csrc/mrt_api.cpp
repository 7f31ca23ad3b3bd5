// C ABI of the ray path (include/mrt.h): context, scene upload, frame launches.
// Host code only; the kernels live in kernels.hip and raster.hip.  There is no CPU fallback:
// without a GPU mrt_create fails and nothing else can be called.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "bvh_builder.h"
#include "host_parallel.h"
#include "device_scene.h"
#include "mrt_math.h"
#include "raster.h"
#include "refit.h"

namespace mrt {
hipError_t launch_render(const SceneDev& S, const FrameDev& F, int mode, bool resident, int n_cus, int resident_threads, hipStream_t stream);
hipError_t launch_render_second_tree(const SceneDev& S, const FrameDev& F, int mode, int n_cus, int resident_threads, hipStream_t stream);
hipError_t launch_eye_relative(const SceneDev& S, const float eye[3], float group_pad, float4* eye_nodes, float4* eye_tri, float4* eye_g, hipStream_t stream);
hipError_t launch_eye_groups(const SceneDev& S, const float eye[3], float group_pad, uint4* out, hipStream_t stream);
hipError_t prepare_render_kernels();
hipError_t launch_depth_view(const SceneDev& S, const FrameDev& F, hipStream_t stream);
hipError_t launch_order_tiles(uint32_t* cost, uint32_t* state, uint32_t* packed, uint32_t* hist, uint32_t* order, int n, bool allow_split, int n_waves,
                              float budget_frac, int split_keys, size_t capacity, unsigned long long* n_entries,
                              uint32_t* spread_a, uint32_t* spread_b, int tiles_x, int radius, float share, hipStream_t stream);
int resident_waves(int n_tiles, int n_cus, int threads, bool ordered);
size_t resident_lds_bytes(int n_node_slots, int n_tris, int stack_entries, bool groups);
hipError_t launch_trace_packets(const SceneDev& S, int n_packets, const float* O, const float* D, const float* rD,
                                float* t, int32_t* tri, float* u, float* v, unsigned long long* flag, hipStream_t stream);
hipError_t launch_trace_rays(const SceneDev& S, int n_rays, int stack_entries, const float* O, const float* D,
                             float* t, int32_t* tri, float* u, float* v, unsigned long long* flag, hipStream_t stream);
}  // namespace mrt

using namespace mrt;

namespace {
thread_local std::string g_create_error;

// Tuning aids read from the environment exist only in the diagnostic build (make diag): the release
// library's behaviour depends on its arguments alone.
#ifdef MRT_DEBUG_API
int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
int env_int(const char*, int dflt) { return dflt; }
#endif

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t s) {
        hipError_t e = ensure(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    // a buffer dies with whatever owns it (mrt_destroy deletes the context with its device current)
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
};
}  // namespace

// ---------------------------------------------------------------------------------------------
// Scratch memory of mrt_render_device_noalloc: what mrt_render_device keeps in buffers it allocates
// (the frame's counter block with the tile queues, and the three per-tile arrays of the tile
// ordering) lives here in a fixed array of the code object, so that entry point never allocates.
// `scratch` is a host-side stand-in for the symbol, not an address: host code only ever uses the
// per-device address that hipGetSymbolAddress returned at mrt_create (scratch_of), and the kernels
// receive that address as an argument, exactly as they receive the allocated buffers elsewhere.
// Every GPU holds its own copy (about 3.2 MB) until the process ends; all contexts on a GPU share it.
// ---------------------------------------------------------------------------------------------
constexpr int kNoAllocMaxTiles = MRT_NOALLOC_MAX_TILES;
constexpr size_t kScratchCounterBytes = (size_t)kCounterSlots * sizeof(unsigned long long);
constexpr size_t kScratchBytes = kScratchCounterBytes + 4 * (size_t)kNoAllocMaxTiles * sizeof(uint32_t);
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};            // device address of `scratch` on each GPU, once known
unsigned long long g_scratch_shape[kMaxDevices] = {};       // launch the tile costs in a GPU's scratch belong to (0: none)
// who launched into a GPU's scratch last: context and that context's count of such launches.  The counters of a launch
// (ray counts, watchdog word) can only be read back while the stamp still names it.  All three arrays: g_scratch_mutex.
struct ScratchOwner { const void* ctx = nullptr; unsigned long long serial = 0; };
ScratchOwner g_scratch_owner[kMaxDevices];

// With `device` current: the address of this GPU's copy of the scratch array (looked up once per device).
hipError_t scratch_of(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct mrt_ctx {
    int width = 0, height = 0, device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    bool has_scene = false;
    int stack_entries = 16;
    HostBVH bvh;
    DevBuf<float4> d_nodes, d_triA, d_triB, d_attr, d_lights;
    DevBuf<float2> d_triC;
    DevBuf<uint4> d_triG;
    DevBuf<float4> d_triM;
    // mrt_scene_upload_lbvh_exact: a second, GPU-built tree over the same triangles, which frames of the recursive shader
    // and of the packet path walk instead when it fits the LDS (kernels.hip: hit_is_the_references); everything else
    // belongs to the first
    struct FastTree {
        DevBuf<float4> nodes, triA, triB, triM, leaf_box;
        DevBuf<float2> triC;
        DevBuf<float4> eye_nodes, eye_tri;      // its eye-relative copy for the packet path (device_scene.h: FrameDev::eye_nodes2)
        uint32_t eye_key[4] = {0, 0, 0, 0};
        int n_slots = 0, coords_ok = 0, stack_entries = 0;
        float reach = 0.0f;
        bool ok = false;
    } fast;
    DevBuf<uint2> d_place;      // per triangle id: its place in the tree's own leaf order and its leaf's first place (device_scene.h: place)
    bool has_groups = false;
    // nodes and triangles relative to the eye of the last packet-path frame (kernels.hip: SceneGlobalEyeT), sized at
    // scene upload; eye_key = the eye position and scene_serial they were made for
    DevBuf<float4> d_eye_nodes, d_eye_tri, d_eye_g;
    uint32_t eye_key[4] = {0, 0, 0, 0};
    // the group boxes as the primary rays of the recursive shader may use them from the eye of the last such frame
    // (kernels.hip: eye_groups_kernel); group_pad = the boxes' padding
    DevBuf<uint4> d_triG_eye;
    uint32_t eye_groups_key[4] = {0, 0, 0, 0};
    float group_pad = 0.0f;
    bool resident = false;      // scene small enough for the LDS-resident kernel
    unsigned scene_serial = 0;  // counts scene uploads
    int n_cus = 256;
    DevBuf<int4> d_tex;
    DevBuf<uint8_t> d_texels;
    DevBuf<uint32_t> d_pal;
    std::vector<float4> lights;     // 2 per light
    std::vector<float4> objects;    // 3 per analytic primitive (kernels.hip: trace_objects)
    DevBuf<float4> d_objects;
    SceneDev scene{};
    // frame outputs of the synchronous entry points
    DevBuf<uint32_t> d_fb;
    DevBuf<int32_t> d_ids;
    DevBuf<unsigned long long> d_counters;
    // per-tile clocks of the last persistent-kernel launch and the tile order derived from them;
    // cost_shape identifies the launch they belong to (0: none)
    DevBuf<uint32_t> d_tile_cost, d_tile_packed, d_tile_order, d_tile_state;
    DevBuf<uint32_t> d_tile_spread_a, d_tile_spread_b;   // while the camera moves: the largest previous cost near every tile (kernels.hip: tile_spread_kernel)
    float order_cam[16] = {};                            // the camera of the last frame whose tiles were ordered
    unsigned long long cost_shape = 0;
    // last_stream is only ever COMPARED with the stream of the next launch (it may be a caller's stream that no
    // longer exists); everything that has to wait for the last launch waits on last_done.
    hipStream_t last_stream = nullptr;
    hipEvent_t last_done = nullptr;   // recorded after every frame launch: the context's counters and queues are shared,
    bool launched = false;            // so a launch on another stream first waits for the previous one
    int last_n_lights = 0;                         // lights of the last launch (the shadow-ray counter is relative to it)
    int last_n_tiles = 0;                          // tiles of the last launch, and whether its queues handed out an order list
    bool last_ordered = false;
    bool last_two_trees = false;                   // the last launch walked the scene's second tree (mrt_scene_upload_lbvh_exact)
    unsigned long long* last_counters = nullptr;   // counter block the last launch used (the context's own or the GPU's scratch)
    unsigned char* scratch_dev = nullptr;          // this GPU's copy of the scratch array (mrt_render_device_noalloc)
    unsigned long long noalloc_serial = 0;         // this context's launches into it
    // ray arrays of the operator-level entry points, kept between calls (grown on demand)
    DevBuf<float> d_rayO, d_rayD, d_rayR, d_rayT, d_rayU, d_rayV;
    DevBuf<int32_t> d_rayTri;
    DevBuf<unsigned long long> d_rayFlag;
    // page-locked staging for the synchronous entry point: device -> pinned at PCIe rate, then a host copy
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // frames queued by mrt_render_submit: slot = ticket % MRT_MAX_IN_FLIGHT
    // The frame is rendered into the slot's device frame and copied to the caller's page-locked frame by a copy
    // engine on copy_stream, under the rendering of the next frame (a kernel that writes the frame over PCIe
    // itself, as the synchronous mrt_render does, runs about 10 % longer).
    struct Flight { hipEvent_t rendered = nullptr, done = nullptr; bool busy = false; int ticket = -1; DevBuf<uint32_t> d_fb; };
    Flight flight[MRT_MAX_IN_FLIGHT];
    hipStream_t copy_stream = nullptr;
    unsigned long long* flight_wd = nullptr;    // page-locked: the watchdog word of each slot's frame
    int next_ticket = 0;
    // rasterizer
    RasterState raster;
    // the tree's shape on the device, for mrt_scene_refit
    RefitState refit;
};

namespace {

int fail(mrt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
#define MRT_HIP(c, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail((c), MRT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

void refresh_scene_view(mrt_ctx* c) {
    SceneDev& S = c->scene;
    S.nodes = c->d_nodes.p; S.triA = c->d_triA.p; S.triB = c->d_triB.p; S.triC = c->d_triC.p; S.attr = c->d_attr.p;
    S.triG = c->has_groups ? c->d_triG.p : nullptr;
    S.triM = c->d_triM.p;
    S.place = c->d_place.p;
    S.tex = c->d_tex.p; S.texels = c->d_texels.p; S.pal = c->d_pal.p;
    S.lights = c->d_lights.p;
    S.objects = c->objects.empty() ? nullptr : c->d_objects.p;
    S.n_objects = (int)(c->objects.size() / 3);
    S.n_lights = (int)(c->lights.size() / 2);
    S.n_tris = (int)c->bvh.n_tris;
}

// Raytracer::Render's view-plane set-up (raytracer.cpp:422-434) for a traced image of w x h
void fill_view(FrameDev& F, const float cam16[16], int w, int h) {
    m4 cam;
    memcpy(&cam, cam16, 64);
    const float fov = 1.8f;
    const float aspect = (float)h / (float)w;          // game.h:5
    const f4 a = mul(cam, f4{-1, aspect, -fov, 1});
    const f4 b = mul(cam, f4{1, aspect, -fov, 1});
    const f4 d = mul(cam, f4{-1, -aspect, -fov, 1});
    const f3 p0 = {a.x, a.y, a.z}, p1 = {b.x, b.y, b.z}, p2 = {d.x, d.y, d.z};
    const f3 d10 = p1 - p0, d20 = p2 - p0;
    F.p0[0] = p0.x; F.p0[1] = p0.y; F.p0[2] = p0.z;
    F.d10[0] = d10.x; F.d10[1] = d10.y; F.d10[2] = d10.z;
    F.d20[0] = d20.x; F.d20[1] = d20.y; F.d20[2] = d20.z;
    F.cam[0] = cam.c[3].x; F.cam[1] = cam.c[3].y; F.cam[2] = cam.c[3].z;   // scenedata.h:130
    F.inv_h = 1.0f / (float)h;
    F.inv_w = 1.0f / (float)w;
    F.width = w; F.height = h;
    F.tiles_x = w / 8;
}

int parse_flags(mrt_ctx* c, uint32_t flags, int& mode, int& g) {
    mode = (int)(flags & MRT_MODE_MASK);
    g = (int)((flags >> 8) & 0xffu);
    if (flags & ~(MRT_MODE_MASK | 0xff00u | MRT_FLAG_FRAME_ORDER | MRT_FLAG_NO_RAY_COUNTS)) return fail(c, MRT_E_ARG, "unknown render flag");
    if (g == 0) g = 1;
    if (mode != (int)MRT_MODE_PACKET && mode != (int)MRT_MODE_RECURSIVE && mode != (int)MRT_MODE_BVH_DEPTH) return fail(c, MRT_E_ARG, "unknown render mode");
    if (mode == (int)MRT_MODE_BVH_DEPTH && g != 1) return fail(c, MRT_E_ARG, "the BVH-depth view has no supersampling grid");
    if (g != 1 && g != 2 && g != 4 && g != 8) return fail(c, MRT_E_ARG, "supersampling grid must be 1, 2, 4 or 8");
    return MRT_OK;
}

// tile rows (8 pixel rows each) of share `part`: those R < rows with (R / block) % n_parts == part
int part_tile_rows(int rows, int part, int n_parts, int block) {
    const int cycle = block * n_parts;
    const int full = rows / cycle, rest = rows - full * cycle;
    return full * block + std::min(block, std::max(0, rest - part * block));
}

struct RowShare { int part = 0, n_parts = 1, block = 1, compact = 0; };

// Device memory a frame launch needs beyond its inputs and outputs.  nullptr for the per-tile arrays: allocate
// (or reuse) the context's own; otherwise they hold `capacity` tiles and nothing is allocated.
struct FrameScratch {
    unsigned long long* counters = nullptr;
    uint32_t *tile_cost = nullptr, *tile_packed = nullptr, *tile_order = nullptr, *tile_state = nullptr;
    unsigned long long* shape = nullptr;      // where the identity of the launch that wrote tile_cost is kept
    int capacity = 0;
};

int render_common(mrt_ctx* c, const float cam[16], uint32_t flags, int y_begin, int y_end, uint32_t* fb_dev,
                  int pitch_px, int32_t* ids_dev, hipStream_t stream, const RowShare* share = nullptr,
                  const FrameScratch* fixed = nullptr) {
    int mode, g;
    if (int rc = parse_flags(c, flags, mode, g)) return rc;
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_upload has not been called");
    if (!cam || !fb_dev) return fail(c, MRT_E_ARG, "null camera or framebuffer");
    if (y_begin < 0 || y_end > c->height || y_begin > y_end || (y_begin % 8) || (y_end % 8))
        return fail(c, MRT_E_ARG, "row band must be multiples of 8 inside the frame");
    if (pitch_px < c->width) return fail(c, MRT_E_ARG, "pitch smaller than width");
    FrameDev F{};
    fill_view(F, cam, c->width * g, c->height * g);
    F.grid_g = g;
    F.tile_y0 = y_begin * g / 8;
    F.n_tiles = ((y_end - y_begin) * g / 8) * F.tiles_x;
    F.n_parts = 1; F.part_block = 1;
    if (share) {
        if (share->n_parts < 1 || share->part < 0 || share->part >= share->n_parts || share->block < 1)
            return fail(c, MRT_E_ARG, "bad share: need 0 <= part < n_parts and block_tile_rows >= 1");
        F.part = share->part; F.n_parts = share->n_parts; F.part_block = share->block; F.compact = share->compact ? 1 : 0;
        F.tile_y0 = 0;
        F.n_tiles = part_tile_rows(c->height / 8, share->part, share->n_parts, share->block) * g * F.tiles_x;
        if (share->n_parts == 1) F.compact = 0;      // the whole frame: packed and in place coincide
    }
    F.pitch = pitch_px;
    F.fb = fb_dev;
    F.hit_ids = ids_dev;
    F.counters = fixed ? fixed->counters : c->d_counters.p;
    F.stack_entries = c->stack_entries;
    if (fixed && F.n_tiles > fixed->capacity)
        return fail(c, MRT_E_LIMIT, "mrt_render_device_noalloc: " + std::to_string(F.n_tiles) + " tiles in this launch, the fixed scratch array holds " +
                                    std::to_string(fixed->capacity) + " (MRT_NOALLOC_MAX_TILES)");
    F.stats = (flags & MRT_FLAG_NO_RAY_COUNTS) ? 0 : 1;
    if (c->launched && stream != c->last_stream) MRT_HIP(c, hipStreamWaitEvent(stream, c->last_done, 0));
    MRT_HIP(c, hipMemsetAsync(F.counters, 0, kCounterSlots * sizeof(unsigned long long), stream));
    // The LDS-resident kernel pays off for the recursive shader, whose per-lane traversal reads a
    // different node or triangle in every lane.  The packet path reads wave-uniform data (scalar
    // loads through the scalar cache) and runs faster with one wave per tile at 8 waves per SIMD.
    if (mode == (int)MRT_MODE_BVH_DEPTH) {
        // colours are relative to the frame's deepest pixel, so the view is defined for whole frames only
        if (share && share->n_parts > 1) return fail(c, MRT_E_ARG, "the BVH-depth view renders whole frames only");
        if (y_begin != 0 || y_end != c->height) return fail(c, MRT_E_ARG, "the BVH-depth view renders whole frames only");
        MRT_HIP(c, launch_depth_view(c->scene, F, stream));
        MRT_HIP(c, hipEventRecord(c->last_done, stream));
        c->last_stream = stream;
        c->launched = true;
        c->last_counters = F.counters;
        c->last_n_lights = 0;
        c->last_n_tiles = F.n_tiles;
        c->last_ordered = false;
        c->last_two_trees = false;
        return MRT_OK;
    }
    // a scene with a second tree (mrt_scene_upload_lbvh_exact): the recursive shader walks that one in the persistent kernel,
    // the packet path in the one-wave-per-tile kernel
    // (its boxes stand back from their triangles by a margin that covers the rounding of the box tests for eyes within
    // `reach` of the origin: kernels.hip, slab_mul_pair; frames from farther away walk the reference's tree)
    const bool eye_in_reach = fabsf(cam[12]) <= c->fast.reach && fabsf(cam[13]) <= c->fast.reach && fabsf(cam[14]) <= c->fast.reach;
    const bool second_tree = c->fast.ok && c->scene.n_objects == 0 && eye_in_reach && env_int("MRT_TWO_TREES", 1) != 0;
    const bool two_trees = second_tree && mode == (int)MRT_MODE_RECURSIVE;
    bool resident = two_trees || (c->resident && mode == (int)MRT_MODE_RECURSIVE && c->scene.n_objects == 0);
    if (env_int("MRT_RESIDENT_PACKET", 0)) resident = c->resident && c->scene.n_objects == 0;
    const bool two_trees_packet = second_tree && mode == (int)MRT_MODE_PACKET && !resident;
    if (resident && F.n_tiles < (1 << 24) && !(flags & MRT_FLAG_FRAME_ORDER)) {
        // Launches of the same shape follow each other in a frame loop with a slowly moving camera:
        // start the tiles that were expensive last time first.
        const unsigned long long shape = ((unsigned long long)F.n_tiles << 24) ^ ((unsigned long long)F.tile_y0 << 12) ^
                                         ((unsigned long long)g << 8) ^ ((unsigned long long)F.part << 4) ^
                                         ((unsigned long long)F.n_parts << 16) ^ ((unsigned long long)F.part_block << 40) ^ ((unsigned long long)mode << 1) ^ 1ull;
        uint32_t *cost, *packed, *order, *state;
        unsigned long long* last_shape;
        unsigned long long key = shape;
        size_t order_capacity;
        if (fixed) {
            cost = fixed->tile_cost; packed = fixed->tile_packed; order = fixed->tile_order; state = fixed->tile_state; last_shape = fixed->shape;
            order_capacity = (size_t)fixed->capacity;
            key ^= (unsigned long long)(uintptr_t)c * 0x9e3779b97f4a7c15ull ^ ((unsigned long long)c->scene_serial << 52);   // the scratch is shared by every context on the GPU
            key |= 1ull;
        } else {
            MRT_HIP(c, c->d_tile_cost.ensure((size_t)F.n_tiles));
            MRT_HIP(c, c->d_tile_packed.ensure((size_t)F.n_tiles));
            order_capacity = (size_t)F.n_tiles + kOrderExtra;
            MRT_HIP(c, c->d_tile_order.ensure(order_capacity));
            MRT_HIP(c, c->d_tile_state.ensure((size_t)F.n_tiles));
            MRT_HIP(c, c->d_tile_spread_a.ensure((size_t)F.n_tiles));
            MRT_HIP(c, c->d_tile_spread_b.ensure((size_t)F.n_tiles));
            cost = c->d_tile_cost.p; packed = c->d_tile_packed.p; order = c->d_tile_order.p; state = c->d_tile_state.p; last_shape = &c->cost_shape;
        }
        bool same_shape;
        {
            // the scratch's slot is shared by every context (and host thread) on the GPU
            std::unique_lock<std::mutex> lock(g_scratch_mutex, std::defer_lock);
            if (fixed) lock.lock();
            same_shape = *last_shape == key;
            *last_shape = key;
        }
        if (same_shape) {
            // expensive tiles of the recursive shader may be listed as sixteenths (kernels.hip: tile_order_kernel); the packet
            // traversal needs all 64 rays of a tile in one wave, and a sample grid is resolved across the tile's lanes
            const bool allow_split = mode == (int)MRT_MODE_RECURSIVE && g == 1;
            const int threads = env_int("MRT_RESIDENT_THREADS", kResidentThreads);
            // A camera that has moved since the frame the costs come from: what was expensive there is expensive a few
            // tiles further on here, so the order also looks at the costs within 4 % of the frame's width of a tile.
            const bool moved = memcmp(c->order_cam, cam, 64) != 0;
            const int radius = moved && !fixed ? std::max(1, F.tiles_x * env_int("MRT_SPREAD_PERMILLE", 40) / 1000) : 0;
            MRT_HIP(c, launch_order_tiles(cost, state, packed, reinterpret_cast<uint32_t*>(F.counters + kOrderHistBase), order, F.n_tiles, allow_split,
                                          resident_waves(F.n_tiles, c->n_cus, threads, true), (float)env_int("MRT_SPLIT_BUDGET", 50) * 0.01f, env_int("MRT_SPLIT_KEYS", 6), order_capacity,
                                          F.counters + kOrderCountSlot, fixed ? nullptr : c->d_tile_spread_a.p, fixed ? nullptr : c->d_tile_spread_b.p, F.tiles_x, radius,
                                          (float)env_int("MRT_SPREAD_SHARE", 75) * 0.01f, stream));
            F.tile_order = order;
        }
        memcpy(c->order_cam, cam, 64);
        F.tile_cost = cost;
    }
    if (mode == (int)MRT_MODE_PACKET && !resident) {
        // Frames of one context run one after the other (last_done above), so the arrays are rewritten only when
        // the eye has moved or the scene has changed since the frame that made them.
        uint32_t key[4];
        memcpy(key, F.cam, 12);
        key[3] = c->scene_serial;
        if (memcmp(key, c->eye_key, sizeof key) != 0) {
            MRT_HIP(c, launch_eye_relative(c->scene, F.cam, c->group_pad, c->d_eye_nodes.p, c->d_eye_tri.p, c->d_eye_g.p, stream));
            memcpy(c->eye_key, key, sizeof key);
        }
        F.eye_nodes = c->d_eye_nodes.p; F.eye_tri = c->d_eye_tri.p; F.eye_g = c->d_eye_g.p;
    }
    SceneDev S = c->scene;
    SceneDev fast = c->scene;
    if (two_trees || two_trees_packet) {
        fast.nodes = c->fast.nodes.p; fast.triA = c->fast.triA.p; fast.triB = c->fast.triB.p; fast.triC = c->fast.triC.p;
        fast.triG = nullptr; fast.triM = c->fast.triM.p; fast.place = nullptr; fast.leaf_box = c->fast.leaf_box.p;
        fast.n_node_slots = c->fast.n_slots; fast.coords_ok = c->fast.coords_ok; fast.reach = c->fast.reach;
        fast.ref_nodes = S.nodes; fast.ref_triA = S.triA; fast.ref_triB = S.triB; fast.ref_triC = S.triC; fast.ref_triM = S.triM; fast.ref_triG = S.triG;
        fast.ref_place = S.place; fast.ref_coords_ok = S.coords_ok;
    }
    if (two_trees_packet) {
        uint32_t key[4];
        memcpy(key, F.cam, 12);
        key[3] = c->scene_serial;
        if (memcmp(key, c->fast.eye_key, sizeof key) != 0) {
            MRT_HIP(c, launch_eye_relative(fast, F.cam, c->group_pad, c->fast.eye_nodes.p, c->fast.eye_tri.p, c->d_eye_g.p, stream));
            memcpy(c->fast.eye_key, key, sizeof key);
        }
        F.eye_nodes2 = c->fast.eye_nodes.p; F.eye_tri2 = c->fast.eye_tri.p;
    }
    if (two_trees || two_trees_packet) {
        FrameDev Ff = F;
        if (two_trees) Ff.stack_entries = c->fast.stack_entries;        // (enough for either tree: mrt_scene_upload_lbvh_exact)
        MRT_HIP(c, launch_render_second_tree(fast, Ff, mode, c->n_cus, env_int("MRT_RESIDENT_THREADS", kResidentThreads), stream));
    } else {
    if (mode == (int)MRT_MODE_RECURSIVE && S.triG) {
        // the recursive shader's primary rays read the scene's own group boxes: this frame's copy of them stands
        // back where the eye grazes a triangle (rewritten when the eye or the scene has changed, like the arrays above)
        uint32_t key[4];
        memcpy(key, F.cam, 12);
        key[3] = c->scene_serial;
        if (memcmp(key, c->eye_groups_key, sizeof key) != 0) {
            MRT_HIP(c, launch_eye_groups(c->scene, F.cam, c->group_pad, c->d_triG_eye.p, stream));
            memcpy(c->eye_groups_key, key, sizeof key);
        }
        S.triG = c->d_triG_eye.p;
    }
    MRT_HIP(c, launch_render(S, F, mode, resident, c->n_cus, env_int("MRT_RESIDENT_THREADS", kResidentThreads), stream));
    }
    MRT_HIP(c, hipEventRecord(c->last_done, stream));
    c->last_stream = stream;
    c->launched = true;
    c->last_counters = F.counters;
    c->last_n_lights = c->scene.n_lights;
    c->last_n_tiles = F.n_tiles;
    c->last_ordered = F.tile_order != nullptr;
    c->last_two_trees = two_trees || two_trees_packet;
    return MRT_OK;
}

}  // namespace

extern "C" {

int mrt_version(void) { return MRT_VERSION; }

const char* mrt_last_error(const mrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int mrt_create(int width, int height, int device, mrt_ctx** out) {
    if (!out) return fail(nullptr, MRT_E_ARG, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (width % 8) || (height % 8))
        return fail(nullptr, MRT_E_ARG, "width and height must be positive multiples of 8");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, MRT_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) return fail(nullptr, MRT_E_ARG, "device ordinal out of range");
    MRT_HIP(nullptr, hipSetDevice(device));
    mrt_ctx* c = new mrt_ctx();
    c->width = width; c->height = height; c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->last_done, hipEventDisableTiming);
    if (e == hipSuccess) e = c->d_counters.ensure(kCounterSlots);
    if (e == hipSuccess) e = prepare_render_kernels();          // per-device kernel attributes, so that no frame launch has to set them
    if (e == hipSuccess) e = scratch_of(device, &c->scratch_dev);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) c->n_cus = prop.multiProcessorCount;
    if (e != hipSuccess) {
        std::string msg = std::string("context set-up: ") + hipGetErrorString(e);
        if (c->last_done) (void)hipEventDestroy(c->last_done);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        c->d_counters.release();
        delete c;
        return fail(nullptr, MRT_E_HIP, msg);
    }
    // the one light Raytracer::Init creates (raytracer.cpp:396-399)
    c->lights = {make_float4(0, 5, 0, 0), make_float4(1, 1, 1, 1)};
    *out = c;
    return MRT_OK;
}

void mrt_destroy(mrt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->launched) { (void)hipEventSynchronize(c->last_done); }     // the last frame may be on a caller's stream
    {
        // a context created later may get this one's address: nothing in the shared scratch may pass for its own
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (c->device >= 0 && c->device < kMaxDevices) {
            g_scratch_shape[c->device] = 0;
            if (g_scratch_owner[c->device].ctx == c) g_scratch_owner[c->device] = ScratchOwner{};
        }
    }
    c->d_rayO.release(); c->d_rayD.release(); c->d_rayR.release(); c->d_rayT.release(); c->d_rayU.release(); c->d_rayV.release();
    c->d_rayTri.release(); c->d_rayFlag.release();
    c->d_nodes.release(); c->d_triA.release(); c->d_triB.release(); c->d_triC.release(); c->d_attr.release(); c->d_lights.release();
    c->d_triG.release(); c->d_triM.release(); c->d_place.release(); c->d_eye_nodes.release(); c->d_eye_tri.release(); c->d_eye_g.release(); c->d_triG_eye.release();
    c->d_tex.release(); c->d_texels.release(); c->d_pal.release(); c->d_objects.release();
    c->d_fb.release(); c->d_ids.release(); c->d_counters.release();
    c->d_tile_cost.release(); c->d_tile_packed.release(); c->d_tile_order.release(); c->d_tile_state.release();
    c->d_tile_spread_a.release(); c->d_tile_spread_b.release();
    c->fast.nodes.release(); c->fast.triA.release(); c->fast.triB.release(); c->fast.triC.release(); c->fast.triM.release(); c->fast.leaf_box.release();
    c->fast.eye_nodes.release(); c->fast.eye_tri.release();
    raster_release(c->raster);
    refit_release(c->refit);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& f : c->flight) {
        if (f.done) (void)hipEventDestroy(f.done);
        if (f.rendered) (void)hipEventDestroy(f.rendered);
        f.d_fb.release();
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->flight_wd) (void)hipHostFree(c->flight_wd);
    if (c->last_done) (void)hipEventDestroy(c->last_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int mrt_set_lights(mrt_ctx* c, const float* pos_rgba7, int n) {
    if (!c) return MRT_E_ARG;
    if (n < 0 || (n > 0 && !pos_rgba7)) return fail(c, MRT_E_ARG, "bad light array");
    MRT_HIP(c, hipSetDevice(c->device));
    c->lights.clear();
    for (int i = 0; i < n; ++i) {
        const float* p = pos_rgba7 + 7 * i;
        c->lights.push_back(make_float4(p[0], p[1], p[2], 0));
        c->lights.push_back(make_float4(p[3], p[4], p[5], p[6]));
    }
    MRT_HIP(c, c->d_lights.upload(c->lights, c->stream));
    MRT_HIP(c, hipStreamSynchronize(c->stream));
    refresh_scene_view(c);
    return MRT_OK;
}

enum class TreeSource { Reference, Lbvh, LbvhExact };
static int scene_upload(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes, const mrt_texture* textures, int n_textures, TreeSource source);

int mrt_set_objects(mrt_ctx* c, const mrt_object* objects, int n) {
    if (!c) return MRT_E_ARG;
    if (n < 0 || (n > 0 && !objects)) return fail(c, MRT_E_ARG, "bad object array");
    for (int i = 0; i < n; ++i)
        if (objects[i].kind != MRT_OBJECT_SPHERE && objects[i].kind != MRT_OBJECT_PLANE) return fail(c, MRT_E_ARG, "unknown object kind");
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));       // a frame in flight may still read the old list
    c->objects.clear();
    for (int i = 0; i < n; ++i) {
        const mrt_object& o = objects[i];
        c->objects.push_back(make_float4(o.pos[0], o.pos[1], o.pos[2], o.param));
        c->objects.push_back(make_float4(o.diffuse[0], o.diffuse[1], o.diffuse[2], o.diffuse[3]));
        c->objects.push_back(make_float4(o.refl, o.refr, o.rindex, bits2f((uint32_t)o.kind)));
    }
    MRT_HIP(c, c->d_objects.upload(c->objects, c->stream));
    MRT_HIP(c, hipStreamSynchronize(c->stream));
    refresh_scene_view(c);
    c->cost_shape = 0;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        g_scratch_shape[c->device] = 0;        // (tile costs in the shared scratch may be this context's: they go with the old list)
    }
    return MRT_OK;
}

int mrt_scene_upload(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes, const mrt_texture* textures, int n_textures) {
    return scene_upload(c, meshes, n_meshes, textures, n_textures, TreeSource::Reference);
}

int mrt_scene_upload_lbvh(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes, const mrt_texture* textures, int n_textures) {
    return scene_upload(c, meshes, n_meshes, textures, n_textures, TreeSource::Lbvh);
}

int mrt_scene_upload_lbvh_exact(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes, const mrt_texture* textures, int n_textures) {
    return scene_upload(c, meshes, n_meshes, textures, n_textures, TreeSource::LbvhExact);
}

static int scene_refit(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes);
int mrt_scene_refit(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes) {
    return scene_refit(c, meshes, n_meshes);
}

// mrt_scene_refit: the tree stays, everything that depends on vertex positions is recomputed on the device (refit.hip)
// from the rasterizer's vertex arrays, which are refreshed first.  The triangles' vertex indices are those of the upload.
static int scene_refit(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes) {
    if (!c) return MRT_E_ARG;
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_refit needs a scene: call mrt_scene_upload first");
    if (n_meshes < 0 || (n_meshes > 0 && !meshes)) return fail(c, MRT_E_ARG, "meshes is null");
    // the old geometry stays in place if the new one is refused: everything is checked before anything is written
    size_t tris = 0;
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        if (M.tris < 0 || M.verts < 0) return fail(c, MRT_E_ARG, "negative vertex/triangle count");
        if (M.tris > 0 && (!M.pos || !M.uv || !M.tri_normal || !M.tri)) return fail(c, MRT_E_ARG, "mesh array is null");
        tris += (size_t)M.tris;
    }
    if (tris != c->bvh.tris.size())
        return fail(c, MRT_E_ARG, "mrt_scene_refit: " + std::to_string(tris) + " triangles, the uploaded scene has " +
                                  std::to_string(c->bvh.tris.size()) + " (a refit keeps the tree: same meshes, same triangle counts)");
    if ((size_t)n_meshes != c->raster.meshes.size()) return fail(c, MRT_E_ARG, "mrt_scene_refit: another number of meshes than the uploaded scene has");
    for (int m = 0; m < n_meshes; ++m)
        if (meshes[m].verts != c->raster.meshes[m].verts || meshes[m].tris != c->raster.meshes[m].tris)
            return fail(c, MRT_E_ARG, "mrt_scene_refit: mesh " + std::to_string(m) + " has another number of vertices or triangles than at upload "
                                      "(a refit keeps the tree: same meshes, same triangle counts)");
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));      // a frame in flight may still read the old arrays
    std::string err;
    c->has_scene = false;
    if (int rc = raster_refit(c->raster, meshes, n_meshes, c->stream, err)) return fail(c, rc, err);
    float largest = 0.0f;
    bool coords_ok = true;
    MRT_HIP(c, refit_run(c->refit, c->raster.d_pos, c->raster.d_uv, c->raster.d_N, c->raster.d_tri, c->d_nodes.p, c->d_triA.p, c->d_triB.p, c->d_triC.p,
                         c->d_triM.p, c->has_groups ? c->d_triG.p : nullptr, c->d_attr.p, c->stream, &largest, &coords_ok));
    if (tris == 0) MRT_HIP(c, hipStreamSynchronize(c->stream));
    if (tris > 0) {
        c->group_pad = largest * 0x1p-12f;
        c->scene.coords_ok = coords_ok ? 1 : 0;
    }
    c->fast.ok = false;         // (the second tree of mrt_scene_upload_lbvh_exact is not refitted: frames go through the first from here on)
    c->has_scene = true;
    c->cost_shape = 0;          // tile costs of the old geometry say little about the new one
    ++c->scene_serial;          // (the eye-relative copies of the old arrays are stale)
    return MRT_OK;
}

// The order of the triangles INSIDE a long leaf, as the device arrays hold them.  The reference tests a leaf's
// triangles one after the other and keeps a hit only if it is strictly nearer (raytracer.cpp:1328-1346, 1445-1464): the
// outcome is the nearest triangle, and of several at exactly the same distance the one that comes first in the leaf.
// With that rule made explicit (kernels.hip: Tie -- on equal distance the triangle with the lower place in the
// reference's order wins, if both belong to the leaf), the order in which a leaf is tested is free, and the traversals
// skip groups of four consecutive triangles whose box a ray misses: so the triangles of a leaf of kGroupMinLeaf or more
// are laid out by recursive bisection of their centroids (longest axis, halves ending on group boundaries), which makes
// neighbours in the arrays neighbours in space and the groups' boxes small.  In the reference's own order the leaf is
// sorted along one axis only and a group's box spans the leaf in the other two.
static void arrange_long_leaves(const HostBVH& B, std::vector<uint32_t>& order) {
    std::vector<uint32_t> leaves;
    for (uint32_t i = 0; i <= B.pool_ptr && B.n_tris > 0; ++i)
        if (i != 1 && i != 2 && B.count[i] >= (uint32_t)kGroupMinLeaf) leaves.push_back(i);
    parallel_for(leaves.size(), [&](size_t lb, size_t le) {
        std::vector<uint32_t> todo;
        for (size_t li = lb; li < le; ++li) {
            const uint32_t first = B.first[leaves[li]], end = first + B.count[leaves[li]];
            todo.assign({first, end});
            while (!todo.empty()) {
                const uint32_t hi = todo.back(); todo.pop_back();
                const uint32_t lo = todo.back(); todo.pop_back();
                if (hi - lo <= 4u && (lo >> 2) == ((hi - 1u) >> 2)) continue;          // inside one group
                f3 cl = {INFINITY, INFINITY, INFINITY}, ch = {-INFINITY, -INFINITY, -INFINITY};
                for (uint32_t k = lo; k < hi; ++k) {
                    const f3 c = B.tris[order[k]].centroid;
                    cl = {std::min(cl.x, c.x), std::min(cl.y, c.y), std::min(cl.z, c.z)};
                    ch = {std::max(ch.x, c.x), std::max(ch.y, c.y), std::max(ch.z, c.z)};
                }
                const float ex = ch.x - cl.x, ey = ch.y - cl.y, ez = ch.z - cl.z;
                const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
                auto key = [&](uint32_t id) { const f3 c = B.tris[id].centroid; return axis == 0 ? c.x : axis == 1 ? c.y : c.z; };
                std::sort(order.begin() + lo, order.begin() + hi, [&](uint32_t a, uint32_t b) { const float ka = key(a), kb = key(b); return ka < kb || (ka == kb && a < b); });
                // the halves end on a boundary of groups of four (of sixteen while the run is long): places are global, as the groups are
                const uint32_t align = hi - lo > 32u ? 16u : 4u;
                uint32_t mid = ((lo + hi) / 2u + align / 2u) / align * align;
                if (mid <= lo || mid >= hi) mid = ((lo + hi) / 2u + 2u) / 4u * 4u;
                if (mid <= lo || mid >= hi) mid = (lo + hi) / 2u;
                if (mid <= lo || mid >= hi) continue;
                todo.push_back(lo); todo.push_back(mid);
                todo.push_back(mid); todo.push_back(hi);
            }
        }
    }, 16);
}

static int scene_upload(mrt_ctx* c, const mrt_mesh* meshes, int n_meshes, const mrt_texture* textures, int n_textures, TreeSource source) {
    if (!c) return MRT_E_ARG;
    const bool lbvh = source == TreeSource::Lbvh;
    if (n_textures < 0 || (n_textures > 0 && !textures)) return fail(c, MRT_E_ARG, "bad texture array");
    for (int i = 0; i < n_textures; ++i)
        if (!textures[i].indices || !textures[i].palettes || textures[i].width <= 0 || textures[i].height <= 0)
            return fail(c, MRT_E_ARG, "bad texture");
    MRT_HIP(c, hipSetDevice(c->device));
    std::string err;
    const bool timing = env_int("MRT_TIMING", 0) != 0;             // stage times of the upload on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto stage = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "mrt_scene_upload: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    c->has_scene = false;
    if (!flatten_meshes(meshes, n_meshes, n_textures, true, c->bvh.tris, err)) return fail(c, MRT_E_ARG, err);
    stage("flatten");
    if (lbvh && !c->bvh.tris.empty()) {
        if (!build_lbvh_device(c->bvh, c->stream, err, (float)env_int("MRT_LBVH_COST", 0) * 0.1f)) return fail(c, MRT_E_HIP, err);
    } else {
        build_bvh(c->bvh);
    }
    stage(lbvh ? "tree (GPU)" : "tree (host)");
    const HostBVH& B = c->bvh;
    if (B.max_depth + 2 > kStackDepth) return fail(c, MRT_E_LIMIT, "BVH deeper than the traversal stack (62 levels)");
    if (B.pool_ptr >= (1u << 26)) return fail(c, MRT_E_LIMIT, "too many BVH nodes for the packet work list encoding");
    c->stack_entries = B.max_depth + 2;

    // nodes: slots 0..pool_ptr (1 and 2 unused); leaves point into the leaf-ordered records
    const size_t slots = (size_t)B.pool_ptr + 1;
    std::vector<float4> nodes(2 * slots);
    // exact-division shortcut of the single-ray traversal: every bound must be 0 or of ordinary size
    std::atomic<bool> coords_all_ok{true};
    parallel_for(slots, [&](size_t b, size_t e) {
        bool ok = true;
        for (size_t i = b; i < e; ++i) {
            nodes[2 * i] = make_float4(B.nmin[i].x, B.nmin[i].y, B.nmin[i].z, bits2f(B.first[i]));
            nodes[2 * i + 1] = make_float4(B.nmax[i].x, B.nmax[i].y, B.nmax[i].z, bits2f(B.count[i]));
            if (i == 1 || i == 2) continue;
            const float c6[6] = {B.nmin[i].x, B.nmin[i].y, B.nmin[i].z, B.nmax[i].x, B.nmax[i].y, B.nmax[i].z};
            for (float v : c6) ok = ok && div_shortcut_coord_ok(v);
        }
        if (!ok) coords_all_ok = false;
    });
    const bool coords_ok = coords_all_ok;
    // the device arrays' triangle order: the tree's leaf order (BVH::m_TriangleIdx), long leaves rearranged inside;
    // place[id] = where triangle id stands in the tree's own order, which decides between equal distances
    std::vector<uint32_t> order(B.idx.begin(), B.idx.begin() + B.n_tris);
    std::vector<uint2> place(B.n_tris);          // {place, first place of the triangle's leaf}
    for (uint32_t i = 0; i <= B.pool_ptr && B.n_tris > 0; ++i)
        if (i != 1 && i != 2)
            for (uint32_t k = B.first[i]; k < B.first[i] + B.count[i]; ++k) place[B.idx[k]] = make_uint2(k, B.first[i]);
    arrange_long_leaves(B, order);
    stage("long leaves arranged");
    std::vector<float4> triA(B.n_tris), triB(B.n_tris), attr(4 * (size_t)B.n_tris);
    std::vector<float2> triC(B.n_tris);
    std::vector<float4> triM(3 * (size_t)B.n_tris);
    parallel_for(B.n_tris, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; ++k) {
            const uint32_t id = order[k];
            const HostTri& t = B.tris[id];
            const f3 e1 = t.p1 - t.p0, e2 = t.p2 - t.p0;      // raytracer.cpp:289-290
            triA[k] = make_float4(e1.x, e1.y, e1.z, e2.x);
            triB[k] = make_float4(e2.y, e2.z, t.p0.x, t.p0.y);
            triC[k] = make_float2(t.p0.z, bits2f(id));
            triM[3 * k] = triA[k]; triM[3 * k + 1] = triB[k]; triM[3 * k + 2] = make_float4(t.p0.z, bits2f(id), 0.0f, 0.0f);
        }
        for (size_t id = b; id < e; ++id) {
            const HostTri& t = B.tris[id];
            attr[4 * id] = make_float4(t.normal.x, t.normal.y, t.normal.z, t.refl);
            attr[4 * id + 1] = make_float4(t.uv0.x, t.uv0.y, t.uv1.x, t.uv1.y);
            attr[4 * id + 2] = make_float4(t.uv2.x, t.uv2.y, t.refr, t.rindex);
            attr[4 * id + 3] = make_float4(bits2f((uint32_t)t.tex), 0, 0, 0);
        }
    });
    // boxes around groups of four consecutive leaf-order slots (device_scene.h: triG)
    float largest = 0.0f;
    for (int a = 0; a < 3 && B.n_tris > 0; ++a) {
        const float lo = a == 0 ? B.nmin[0].x : a == 1 ? B.nmin[0].y : B.nmin[0].z, hi = a == 0 ? B.nmax[0].x : a == 1 ? B.nmax[0].y : B.nmax[0].z;
        largest = std::max(largest, std::max(fabsf(lo), fabsf(hi)));
    }
    // binary16 keeps 11 bits between 2^-14 and 65504: scenes far from that range go without groups
    c->has_groups = B.n_tris > 0 && largest >= 0x1p-6f && largest <= 0x1p14f;
    if (c->has_groups) {
        // Worth it?  Only leaves of kGroupMinLeaf triangles or more consult the boxes, and a box helps when it is
        // small beside its leaf's: compare surface areas (the chance that a ray through the leaf also meets the
        // group), weighted by triangles.  A soup of unrelated triangles has short leaves or groups as large as
        // the leaf, and goes without.
        auto area = [](f3 lo, f3 hi) {
            const double dx = std::max(0.0f, hi.x - lo.x), dy = std::max(0.0f, hi.y - lo.y), dz = std::max(0.0f, hi.z - lo.z);
            return dx * dy + dy * dz + dz * dx;
        };
        double long_tris = 0, weighted = 0;
        for (size_t i = 0; i <= B.pool_ptr; ++i) {
            if (i == 1 || i == 2 || B.count[i] < (uint32_t)kGroupMinLeaf) continue;
            const double leaf_area = area(B.nmin[i], B.nmax[i]);
            const size_t first = B.first[i], end = first + B.count[i];
            for (size_t k = first; k < end;) {
                const size_t gend = std::min(end, (k | 3) + 1);
                f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
                for (size_t j = k; j < gend; ++j) {
                    const HostTri& t = B.tris[order[j]];
                    for (const f3& p : {t.p0, t.p1, t.p2}) {
                        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
                        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
                    }
                }
                weighted += (double)(gend - k) * (leaf_area > 0 ? std::min(1.0, area(lo, hi) / leaf_area) : 1.0);
                long_tris += (double)(gend - k);
                k = gend;
            }
        }
        c->has_groups = long_tris >= 0.05 * (double)B.n_tris && weighted <= 0.6 * long_tris;
        if (timing) fprintf(stderr, "mrt_scene_upload: group boxes %s: %.0f of %u triangles in long leaves, mean box-to-leaf area %.2f\n",
                            c->has_groups ? "on" : "off", long_tris, B.n_tris, long_tris > 0 ? weighted / long_tris : 0.0);
    }
    const size_t n_groups = ((size_t)B.n_tris + 3) / 4;
    std::vector<uint4> triG(c->has_groups ? n_groups : 0);
    c->group_pad = largest * 0x1p-12f;
    if (c->has_groups) {
        const float pad = c->group_pad;
        parallel_for(n_groups, [&](size_t b, size_t e) {
            for (size_t g = b; g < e; ++g) {
                f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
                bool sliver = false;
                for (size_t k = 4 * g; k < std::min<size_t>(4 * g + 4, B.n_tris); ++k) {
                    const HostTri& t = B.tris[order[k]];
                    sliver = sliver || triangle_is_sliver(t.p0, t.p1, t.p2);
                    for (const f3& p : {t.p0, t.p1, t.p2}) {
                        lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
                        hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
                    }
                }
                if (sliver) {           // a box no ray misses
                    triG[g] = make_uint4(MRT_GROUP_EVERYWHERE_X, MRT_GROUP_EVERYWHERE_Y, MRT_GROUP_EVERYWHERE_Z, 0u);
                    continue;
                }
                const uint32_t h[6] = {half_outward(lo.x - pad, false), half_outward(lo.y - pad, false), half_outward(lo.z - pad, false),
                                       half_outward(hi.x + pad, true), half_outward(hi.y + pad, true), half_outward(hi.z + pad, true)};
                triG[g] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), 0u);
            }
        });
    }
    std::vector<int4> tex(n_textures);
    std::vector<uint8_t> texels;
    std::vector<uint32_t> pal((size_t)n_textures * 32 * 256);
    for (int i = 0; i < n_textures; ++i) {
        const size_t n = (size_t)textures[i].width * textures[i].height;
        tex[i] = make_int4((int)texels.size(), textures[i].width, textures[i].height, 0);
        texels.insert(texels.end(), textures[i].indices, textures[i].indices + n);
        memcpy(pal.data() + (size_t)i * 32 * 256, textures[i].palettes, 32 * 256 * 4);
        if (texels.size() > 0x7fffffffu) return fail(c, MRT_E_LIMIT, "textures exceed 2 GiB of texels");
    }
    stage("pack arrays");
    MRT_HIP(c, c->d_nodes.upload(nodes, c->stream));
    MRT_HIP(c, c->d_triA.upload(triA, c->stream));
    MRT_HIP(c, c->d_triB.upload(triB, c->stream));
    MRT_HIP(c, c->d_triC.upload(triC, c->stream));
    MRT_HIP(c, c->d_triM.upload(triM, c->stream));
    MRT_HIP(c, c->d_place.upload(place, c->stream));
    if (c->has_groups) MRT_HIP(c, c->d_triG.upload(triG, c->stream));
    MRT_HIP(c, c->d_attr.upload(attr, c->stream));
    MRT_HIP(c, c->d_tex.upload(tex, c->stream));
    MRT_HIP(c, c->d_texels.upload(texels, c->stream));
    MRT_HIP(c, c->d_pal.upload(pal, c->stream));
    MRT_HIP(c, c->d_lights.upload(c->lights, c->stream));
    MRT_HIP(c, c->d_eye_nodes.ensure(std::max<size_t>(1, 2 * slots)));
    MRT_HIP(c, c->d_eye_tri.ensure(std::max<size_t>(4, 4 * (size_t)B.n_tris)));
    MRT_HIP(c, c->d_eye_g.ensure(std::max<size_t>(2, 2 * n_groups)));
    MRT_HIP(c, c->d_triG_eye.ensure(std::max<size_t>(1, n_groups)));
    stage("upload ray-path arrays");
    if (int rc = raster_upload(c->raster, meshes, n_meshes, order.size() == (size_t)B.n_tris ? order.data() : nullptr, c->stream, err)) return fail(c, rc, err);
    MRT_HIP(c, hipStreamSynchronize(c->stream));
    stage("rasterizer arrays");
    MRT_HIP(c, refit_prepare(c->refit, c->bvh, order, c->stream));
    stage("tree shape for refits");
    refresh_scene_view(c);
    c->scene.n_textures = n_textures;
    c->scene.n_node_slots = (int)slots;
    c->scene.coords_ok = coords_ok ? 1 : 0;
    // LDS residency: nodes + triangles + 16 per-lane stacks must fit one CU's LDS, node indices 16 bits
    c->resident = B.n_tris > 0 && slots < 65536 &&
                  resident_lds_bytes((int)slots, (int)B.n_tris, c->stack_entries, true) + 256 <= (size_t)kLdsBudget;
    c->resident = c->resident && env_int("MRT_RESIDENT", 1) != 0;      // 0 forces the global-memory kernel
    c->fast.ok = false;
    // (a reference tree that is one leaf has nothing to gain, and its leaf is entered through the root gate's own
    // arithmetic, raytracer.cpp:947, not through a parent's box test, which is what the second tree's rays are checked by)
    if (source == TreeSource::LbvhExact && B.n_tris > 0 && B.count[0] == 0) {
        // the second tree: built on the GPU over the same triangles, kept only if the persistent kernel can hold it in LDS
        // (with ever larger leaves until its nodes, the triangles and the waves' stacks fit)
        HostBVH T;
        T.tris = B.tris;
        size_t fslots = 0;
        int fstack = 0;
        bool fits = false;
        const int first_cost = env_int("MRT_SECOND_TREE_COST", 0);
        // (the triangles alone, with the fewest nodes a tree of leaves of eight can have, must fit: otherwise no build is tried)
        const bool hopeless = resident_lds_bytes((int)(B.n_tris / 4 + 3), (int)B.n_tris, c->stack_entries, false) + 256 > (size_t)kLdsBudget;
        for (float inner_cost : {first_cost > 0 ? (float)first_cost * 0.1f : 1.5f, 2.0f, 2.5f, 3.0f, 3.5f, 4.0f, 5.0f, 6.0f, 8.0f, 12.0f, 20.0f}) {
            if (hopeless) break;
            if (!build_lbvh_device(T, c->stream, err, inner_cost)) return fail(c, MRT_E_HIP, err);
            fslots = (size_t)T.pool_ptr + 1;
            fstack = std::max(T.max_depth + 2, c->stack_entries);        // the waves' stacks serve walks of either tree
            fits = fstack <= kStackDepth && fslots < 65536 && resident_lds_bytes((int)fslots, (int)T.n_tris, fstack, false) + 256 <= (size_t)kLdsBudget;
            if (timing) fprintf(stderr, "mrt_scene_upload: second tree at inner cost %.1f: %zu node slots, %u triangles, stack %d, LDS %zu of %d bytes\n", inner_cost, fslots,
                                T.n_tris, fstack, resident_lds_bytes((int)fslots, (int)T.n_tris, fstack, false) + 256, kLdsBudget);
            if (fits) break;
        }
        stage("second tree (GPU)");
        if (fits && env_int("MRT_RESIDENT", 1) != 0) {
            // The reference's boxes decide what the reference finds, rounding included; this tree's must only never hide a
            // triangle from a ray that passes its test.  A ray through a triangle's edge that lies in a face of its box can
            // miss the box by the rounding of the slab quotients (2^-22 of the distances involved), so every box but the
            // root's (the gate every ray passes first: the same box, the same arithmetic in both trees) stands back by
            // 2^-16 of the scene's largest coordinate.
            const float pad = largest * 0x1p-16f;
            std::vector<float4> fnodes(2 * fslots), fA(T.n_tris), fB(T.n_tris), fM(3 * (size_t)T.n_tris), fbox(2 * (size_t)T.n_tris);
            std::vector<float2> fC(T.n_tris);
            bool fcoords = true;
            for (size_t i = 0; i < fslots; ++i) {
                const float p = i == 0 ? 0.0f : pad;
                const f3 lo = {T.nmin[i].x - p, T.nmin[i].y - p, T.nmin[i].z - p}, hi = {T.nmax[i].x + p, T.nmax[i].y + p, T.nmax[i].z + p};
                fnodes[2 * i] = make_float4(lo.x, lo.y, lo.z, bits2f(T.first[i]));
                fnodes[2 * i + 1] = make_float4(hi.x, hi.y, hi.z, bits2f(T.count[i]));
                if (i == 1 || i == 2) continue;
                for (float v : {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}) fcoords = fcoords && div_shortcut_coord_ok(v);
            }
            for (size_t k = 0; k < T.n_tris; ++k) {
                const uint32_t id = T.idx[k];
                const HostTri& t = T.tris[id];
                const f3 e1 = t.p1 - t.p0, e2 = t.p2 - t.p0;      // raytracer.cpp:289-290
                fA[k] = make_float4(e1.x, e1.y, e1.z, e2.x);
                fB[k] = make_float4(e2.y, e2.z, t.p0.x, t.p0.y);
                fC[k] = make_float2(t.p0.z, bits2f(id));
                fM[3 * k] = fA[k]; fM[3 * k + 1] = fB[k]; fM[3 * k + 2] = make_float4(t.p0.z, bits2f(id), 0.0f, 0.0f);
            }
            for (uint32_t i = 0; i <= B.pool_ptr; ++i) {
                if (i == 1 || i == 2) continue;
                for (uint32_t k = B.first[i]; k < B.first[i] + B.count[i]; ++k) {       // (count 0: an inner node)
                    fbox[2 * (size_t)B.idx[k]] = make_float4(B.nmin[i].x, B.nmin[i].y, B.nmin[i].z, 0.0f);
                    fbox[2 * (size_t)B.idx[k] + 1] = make_float4(B.nmax[i].x, B.nmax[i].y, B.nmax[i].z, 0.0f);
                }
            }
            MRT_HIP(c, c->fast.nodes.upload(fnodes, c->stream));
            MRT_HIP(c, c->fast.triA.upload(fA, c->stream));
            MRT_HIP(c, c->fast.triB.upload(fB, c->stream));
            MRT_HIP(c, c->fast.triC.upload(fC, c->stream));
            MRT_HIP(c, c->fast.triM.upload(fM, c->stream));
            MRT_HIP(c, c->fast.eye_nodes.ensure(2 * fslots));
            MRT_HIP(c, c->fast.eye_tri.ensure(4 * (size_t)T.n_tris));
            memset(c->fast.eye_key, 0, sizeof c->fast.eye_key);
            MRT_HIP(c, c->fast.leaf_box.upload(fbox, c->stream));
            MRT_HIP(c, hipStreamSynchronize(c->stream));
            c->fast.n_slots = (int)fslots;
            c->fast.coords_ok = fcoords ? 1 : 0;
            c->fast.reach = 64.0f * largest;
            c->fast.stack_entries = fstack;
            c->fast.ok = fcoords;       // (coordinates beyond 2^30 or below 2^-27: kernels.hip, make_ray_second_tree; such a scene keeps one tree)
            stage("second tree packed");
        }
    }
    c->has_scene = true;
    c->cost_shape = 0;          // tile costs of the previous scene say nothing about this one
    ++c->scene_serial;
    return MRT_OK;
}

int mrt_bvh_build_host(const mrt_mesh* meshes, int n_meshes, float* node_minmax, uint32_t* node_first_cnt,
                       uint32_t* tri_idx, int64_t* info) {
    if (!node_minmax || !node_first_cnt || !tri_idx || !info) return fail(nullptr, MRT_E_ARG, "null output array");
    HostBVH B;
    std::string err;
    if (!flatten_meshes(meshes, n_meshes, 0, false, B.tris, err)) return fail(nullptr, MRT_E_ARG, err);
    build_bvh(B);
    const size_t slots = (size_t)B.n_nodes_alloc + 3;
    for (size_t i = 0; i < slots; ++i) {
        node_minmax[6 * i] = B.nmin[i].x; node_minmax[6 * i + 1] = B.nmin[i].y; node_minmax[6 * i + 2] = B.nmin[i].z;
        node_minmax[6 * i + 3] = B.nmax[i].x; node_minmax[6 * i + 4] = B.nmax[i].y; node_minmax[6 * i + 5] = B.nmax[i].z;
        node_first_cnt[2 * i] = B.first[i];
        node_first_cnt[2 * i + 1] = B.count[i];
    }
    memcpy(tri_idx, B.idx.data(), sizeof(uint32_t) * B.n_tris);
    info[0] = B.n_tris; info[1] = B.n_nodes_alloc; info[2] = B.pool_ptr; info[3] = B.max_depth;
    return MRT_OK;
}

int mrt_render_device(mrt_ctx* c, const float cam[16], uint32_t flags, int y_begin, int y_end, void* fb_dev,
                      int pitch_px, void* hit_ids_dev, void* stream) {
    if (!c) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    return render_common(c, cam, flags, y_begin, y_end, (uint32_t*)fb_dev, pitch_px, (int32_t*)hit_ids_dev, (hipStream_t)stream);
}

int mrt_render_device_noalloc(mrt_ctx* c, const float cam[16], uint32_t flags, int y_begin, int y_end, void* fb_dev,
                              int pitch_px, void* hit_ids_dev, void* stream) {
    if (!c) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    if (!c->scratch_dev) return fail(c, MRT_E_STATE, "no scratch array on this device");
    FrameScratch fs;
    fs.counters = reinterpret_cast<unsigned long long*>(c->scratch_dev);
    fs.tile_cost = reinterpret_cast<uint32_t*>(c->scratch_dev + kScratchCounterBytes);
    fs.tile_packed = fs.tile_cost + kNoAllocMaxTiles;
    fs.tile_order = fs.tile_packed + kNoAllocMaxTiles;
    fs.tile_state = fs.tile_order + kNoAllocMaxTiles;
    fs.capacity = kNoAllocMaxTiles;
    fs.shape = &g_scratch_shape[c->device];
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        g_scratch_owner[c->device] = ScratchOwner{c, ++c->noalloc_serial};
    }
    return render_common(c, cam, flags, y_begin, y_end, (uint32_t*)fb_dev, pitch_px, (int32_t*)hit_ids_dev, (hipStream_t)stream, nullptr, &fs);
}

void* mrt_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void mrt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int mrt_render_device_part(mrt_ctx* c, const float cam[16], uint32_t flags, int part, int n_parts, int block_tile_rows,
                           int compact, void* fb_dev, int pitch_px, void* hit_ids_dev, void* stream) {
    if (!c) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    RowShare sh;
    sh.part = part; sh.n_parts = n_parts; sh.block = block_tile_rows; sh.compact = compact;
    return render_common(c, cam, flags, 0, c->height, (uint32_t*)fb_dev, pitch_px, (int32_t*)hit_ids_dev, (hipStream_t)stream, &sh);
}

int mrt_part_rows(int height, int part, int n_parts, int block_tile_rows) {
    if (height <= 0 || (height % 8) || n_parts < 1 || part < 0 || part >= n_parts || block_tile_rows < 1) return MRT_E_ARG;
    return 8 * part_tile_rows(height / 8, part, n_parts, block_tile_rows);
}

int mrt_render(mrt_ctx* c, const float cam[16], uint32_t flags, uint32_t* fb, int pitch_px, int32_t* hit_ids) {
    if (!c) return MRT_E_ARG;
    if (!fb) return fail(c, MRT_E_ARG, "null framebuffer");
    if (pitch_px < c->width) return fail(c, MRT_E_ARG, "pitch smaller than width");
    int mode, g;
    if (int rc = parse_flags(c, flags, mode, g)) return rc;
    MRT_HIP(c, hipSetDevice(c->device));
    const size_t npx = (size_t)c->width * c->height;
    MRT_HIP(c, c->d_fb.ensure(npx));
    if (hit_ids) MRT_HIP(c, c->d_ids.ensure(npx * g * g));
    // a framebuffer in page-locked memory (mrt_host_alloc, or registered by the caller) is written by the
    // kernel itself through its device address, so the transfer over PCIe runs under the rendering
    bool direct = false;
    uint32_t* fb_target = c->d_fb.p;
    int fb_pitch = c->width;
    {
        hipPointerAttribute_t at;
        void* dp = nullptr;
        if (hipPointerGetAttributes(&at, fb) == hipSuccess && at.type == hipMemoryTypeHost &&
            hipHostGetDevicePointer(&dp, fb, 0) == hipSuccess && dp && !env_int("MRT_NO_DIRECT_HOST", 0)) {
            direct = true;
            fb_target = static_cast<uint32_t*>(dp);
            fb_pitch = pitch_px;
        }
        (void)hipGetLastError();
    }
    if (int rc = render_common(c, cam, flags, 0, c->height, fb_target, fb_pitch, hit_ids ? c->d_ids.p : nullptr, c->stream))
        return rc;
    const size_t fb_bytes = direct ? 0 : npx * 4, id_bytes = hit_ids ? npx * g * g * sizeof(int32_t) : 0, need = fb_bytes + id_bytes + 64;
    if (c->h_stage_bytes < need) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr; c->h_stage_bytes = 0;
        MRT_HIP(c, hipHostMalloc(&c->h_stage, need, hipHostMallocDefault));
        c->h_stage_bytes = need;
    }
    unsigned char* stage = static_cast<unsigned char*>(c->h_stage);
    unsigned long long* wd_stage = reinterpret_cast<unsigned long long*>(stage + fb_bytes + id_bytes);
    if (!direct) MRT_HIP(c, hipMemcpyAsync(stage, c->d_fb.p, fb_bytes, hipMemcpyDeviceToHost, c->stream));
    if (hit_ids) MRT_HIP(c, hipMemcpyAsync(stage + fb_bytes, c->d_ids.p, id_bytes, hipMemcpyDeviceToHost, c->stream));
    MRT_HIP(c, hipMemcpyAsync(wd_stage, c->d_counters.p + 5, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    MRT_HIP(c, hipStreamSynchronize(c->stream));
    // out of the staging buffer on a few threads: one thread copies a 1080p frame in about 0.4 ms
    if (!direct)
        parallel_for((size_t)c->height, [&](size_t y0, size_t y1) {
            if (pitch_px == c->width) memcpy(fb + y0 * c->width, stage + y0 * c->width * 4, (y1 - y0) * c->width * 4);
            else for (size_t y = y0; y < y1; ++y) memcpy(fb + y * pitch_px, stage + y * c->width * 4, (size_t)c->width * 4);
        }, std::max<size_t>(1, ((size_t)1 << 19) / (size_t)c->width));
    if (hit_ids)
        parallel_for(id_bytes / 4, [&](size_t b, size_t e) { memcpy(hit_ids + b, stage + fb_bytes + 4 * b, (e - b) * 4); }, (size_t)1 << 19);
    const unsigned long long wd = *wd_stage;
    if (wd != 0) return fail(c, MRT_E_LIMIT, "a traversal loop exhausted its iteration budget (code " + std::to_string(wd) + ")");
    return MRT_OK;
}

int mrt_render_submit(mrt_ctx* c, const float cam[16], uint32_t flags, uint32_t* fb, int pitch_px, int* ticket) {
    if (!c) return MRT_E_ARG;
    if (!fb || !ticket) return fail(c, MRT_E_ARG, "null framebuffer or ticket");
    MRT_HIP(c, hipSetDevice(c->device));
    mrt_ctx::Flight& f = c->flight[c->next_ticket % MRT_MAX_IN_FLIGHT];
    if (f.busy) return fail(c, MRT_E_STATE, "MRT_MAX_IN_FLIGHT frames are queued already: call mrt_render_wait for the oldest first");
    hipPointerAttribute_t at;
    void* dp = nullptr;
    const bool locked = hipPointerGetAttributes(&at, fb) == hipSuccess && at.type == hipMemoryTypeHost &&
                        hipHostGetDevicePointer(&dp, fb, 0) == hipSuccess && dp;
    (void)hipGetLastError();
    if (!locked) return fail(c, MRT_E_ARG, "mrt_render_submit needs a framebuffer in page-locked memory (mrt_host_alloc)");
    if (!c->flight_wd) {
        MRT_HIP(c, hipHostMalloc((void**)&c->flight_wd, MRT_MAX_IN_FLIGHT * sizeof(unsigned long long), hipHostMallocDefault));
    }
    if (pitch_px < c->width) return fail(c, MRT_E_ARG, "pitch smaller than width");
    if (!c->copy_stream) MRT_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!f.done) MRT_HIP(c, hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
    if (!f.rendered) MRT_HIP(c, hipEventCreateWithFlags(&f.rendered, hipEventDisableTiming));
    MRT_HIP(c, f.d_fb.ensure((size_t)c->width * c->height));
    if (int rc = render_common(c, cam, flags, 0, c->height, f.d_fb.p, c->width, nullptr, c->stream)) return rc;
    const int slot = c->next_ticket % MRT_MAX_IN_FLIGHT;
    // the watchdog word is cleared by the next frame's set-up, so it is read on the rendering stream
    MRT_HIP(c, hipMemcpyAsync(c->flight_wd + slot, c->d_counters.p + 5, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    MRT_HIP(c, hipEventRecord(f.rendered, c->stream));
    MRT_HIP(c, hipStreamWaitEvent(c->copy_stream, f.rendered, 0));
    MRT_HIP(c, hipMemcpy2DAsync(dp, (size_t)pitch_px * 4, f.d_fb.p, (size_t)c->width * 4, (size_t)c->width * 4, (size_t)c->height,
                                hipMemcpyDeviceToHost, c->copy_stream));
    MRT_HIP(c, hipEventRecord(f.done, c->copy_stream));
    f.busy = true;
    f.ticket = c->next_ticket;
    *ticket = c->next_ticket++;
    return MRT_OK;
}

int mrt_render_wait(mrt_ctx* c, int ticket) {
    if (!c) return MRT_E_ARG;
    if (ticket < 0) return fail(c, MRT_E_ARG, "unknown ticket");
    mrt_ctx::Flight& f = c->flight[ticket % MRT_MAX_IN_FLIGHT];
    if (!f.busy || f.ticket != ticket) return fail(c, MRT_E_ARG, "unknown ticket (never handed out, or waited for already)");
    MRT_HIP(c, hipSetDevice(c->device));
    MRT_HIP(c, hipEventSynchronize(f.done));
    f.busy = false;
    const unsigned long long wd = c->flight_wd[ticket % MRT_MAX_IN_FLIGHT];
    if (wd != 0) return fail(c, MRT_E_LIMIT, "a traversal loop exhausted its iteration budget (code " + std::to_string(wd) + ")");
    return MRT_OK;
}

// The counters of the context's last launch can be read: they are its own, or the GPU's shared scratch array and no other
// launch has gone into that since.
static int counters_still_ours(mrt_ctx* c) {
    if (!c->last_counters || c->last_counters != reinterpret_cast<unsigned long long*>(c->scratch_dev)) return MRT_OK;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    const ScratchOwner& o = g_scratch_owner[c->device];
    if (o.ctx == c && o.serial == c->noalloc_serial) return MRT_OK;
    return fail(c, MRT_E_STATE, "the counters of this context's last frame were in the GPU's shared scratch array (mrt_render_device_noalloc) and "
                                "another context has rendered into it since");
}

int mrt_get_ray_counts(mrt_ctx* c, uint64_t out[4]) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));
    if (int rc = counters_still_ours(c)) return rc;
    std::vector<unsigned long long> tmp(kCounterSlots);
    MRT_HIP(c, hipMemcpy(tmp.data(), c->last_counters ? c->last_counters : c->d_counters.p, kCounterSlots * sizeof(unsigned long long),
                         hipMemcpyDeviceToHost));
    unsigned long long tiles = 0, hits = 0, extra_shadow = 0, secondary = 0;
    for (int s = 0; s < kStatSlots; ++s) {
        const unsigned long long* g = &tmp[kStatBase + 16 * s];
        tiles += g[0] & 0xffffffffull; hits += g[0] >> 32;
        extra_shadow += g[1];                // differences, two's complement: the sum wraps to the right total
        secondary += g[2];
    }
    out[0] = 64 * tiles;
    out[1] = hits * (unsigned long long)c->last_n_lights + extra_shadow;
    out[2] = secondary;
    out[3] = hits;
    const unsigned long long wd = tmp[5];
    if (wd != 0) return fail(c, MRT_E_LIMIT, "a traversal loop exhausted its iteration budget (code " + std::to_string(wd) + ")");
    return MRT_OK;
}

int mrt_get_tile_schedule(mrt_ctx* c, uint64_t out[2]) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    out[0] = out[1] = 0;
    if (!c->launched) return MRT_OK;
    MRT_HIP(c, hipEventSynchronize(c->last_done));
    out[0] = out[1] = (uint64_t)c->last_n_tiles;
    if (c->last_ordered) {
        if (int rc = counters_still_ours(c)) return rc;
        unsigned long long v = 0;
        MRT_HIP(c, hipMemcpy(&v, c->last_counters + kOrderCountSlot, sizeof v, hipMemcpyDeviceToHost));
        out[1] = v;
    }
    return MRT_OK;
}

int mrt_get_second_tree_use(mrt_ctx* c, uint64_t out[2]) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    out[0] = out[1] = 0;
    if (!c->launched || !c->last_two_trees) return MRT_OK;
    MRT_HIP(c, hipEventSynchronize(c->last_done));
    if (int rc = counters_still_ours(c)) return rc;
    unsigned long long v = 0;
    MRT_HIP(c, hipMemcpy(&v, c->last_counters + kRedoCountSlot, sizeof v, hipMemcpyDeviceToHost));
    out[0] = 1;
    out[1] = v;
    return MRT_OK;
}

#ifdef MRT_LOOP_COUNTS
// diagnostic build only: out[2*site] = wave-level executions, out[2*site+1] = lane-level executions
extern "C" int mrt_debug_loop_counts(mrt_ctx* c, uint64_t* out) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));
    MRT_HIP(c, hipMemcpy(out, c->d_counters.p + kProfBase, 2 * kProfSites * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MRT_OK;
}
#endif
#ifdef MRT_WAVE_LIVES
// diagnostic build only: out[3*w .. 3*w+2] = clock at entry, clock at exit, tiles rendered of persistent wave w of the last launch
extern "C" int mrt_debug_wave_lives(mrt_ctx* c, uint64_t* out) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));
    MRT_HIP(c, hipMemcpy(out, c->d_counters.p + kLivesBase, 3 * kLivesWaves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MRT_OK;
}
#endif
#if defined(MRT_LOOP_COUNTS) || defined(MRT_DEBUG_API)
// diagnostic builds only: shader clocks per tile of the last persistent-kernel launch; returns the tile count
extern "C" int mrt_debug_tile_costs(mrt_ctx* c, uint32_t* out, int capacity) {
    if (!c || !out) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    if (c->launched) MRT_HIP(c, hipEventSynchronize(c->last_done));
    const int n = (int)std::min<size_t>(c->d_tile_cost.n, (size_t)std::max(capacity, 0));
    if (n > 0) MRT_HIP(c, hipMemcpy(out, c->d_tile_cost.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return n;
}
#endif

// Shared body of the two operator-level entry points: ray arrays up, one launch, results down.  rD == nullptr
// selects the single-ray operator.  The device arrays stay with the context and grow on demand.
static int trace_common(mrt_ctx* c, const char* what, size_t n, int n_packets, const float* O, const float* D, const float* rD,
                        float* t, int32_t* tri, float* u, float* v) {
    MRT_HIP(c, hipSetDevice(c->device));
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) { if (e == hipSuccess) e = x; return e == hipSuccess; };
    unsigned long long flag = 0;
    if (ok(c->d_rayO.ensure(3 * n)) && ok(c->d_rayD.ensure(3 * n)) && (!rD || ok(c->d_rayR.ensure(3 * n))) && ok(c->d_rayT.ensure(n)) &&
        ok(c->d_rayU.ensure(n)) && ok(c->d_rayV.ensure(n)) && ok(c->d_rayTri.ensure(n)) && ok(c->d_rayFlag.ensure(1)) &&
        ok(hipMemsetAsync(c->d_rayFlag.p, 0, sizeof(unsigned long long), c->stream)) &&
        ok(hipMemcpyAsync(c->d_rayO.p, O, 12 * n, hipMemcpyHostToDevice, c->stream)) &&
        ok(hipMemcpyAsync(c->d_rayD.p, D, 12 * n, hipMemcpyHostToDevice, c->stream)) &&
        (!rD || ok(hipMemcpyAsync(c->d_rayR.p, rD, 12 * n, hipMemcpyHostToDevice, c->stream))) &&
        ok(hipMemcpyAsync(c->d_rayT.p, t, 4 * n, hipMemcpyHostToDevice, c->stream)) &&
        ok(rD ? launch_trace_packets(c->scene, n_packets, c->d_rayO.p, c->d_rayD.p, c->d_rayR.p, c->d_rayT.p, c->d_rayTri.p, c->d_rayU.p,
                                     c->d_rayV.p, c->d_rayFlag.p, c->stream)
              : launch_trace_rays(c->scene, (int)n, c->stack_entries, c->d_rayO.p, c->d_rayD.p, c->d_rayT.p, c->d_rayTri.p, c->d_rayU.p,
                                  c->d_rayV.p, c->d_rayFlag.p, c->stream)) &&
        ok(hipMemcpyAsync(t, c->d_rayT.p, 4 * n, hipMemcpyDeviceToHost, c->stream)) &&
        ok(hipMemcpyAsync(tri, c->d_rayTri.p, 4 * n, hipMemcpyDeviceToHost, c->stream)) &&
        ok(hipMemcpyAsync(u, c->d_rayU.p, 4 * n, hipMemcpyDeviceToHost, c->stream)) &&
        ok(hipMemcpyAsync(v, c->d_rayV.p, 4 * n, hipMemcpyDeviceToHost, c->stream)) &&
        ok(hipMemcpyAsync(&flag, c->d_rayFlag.p, sizeof(flag), hipMemcpyDeviceToHost, c->stream)))
        ok(hipStreamSynchronize(c->stream));
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->stream);
        return fail(c, MRT_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
    if (flag != 0) return fail(c, MRT_E_LIMIT, "a traversal loop exhausted its iteration budget (code " + std::to_string(flag) + ")");
    return MRT_OK;
}

int mrt_trace_packets(mrt_ctx* c, int n_packets, const float* O, const float* D, const float* rD, float* t,
                      int32_t* tri, float* u, float* v) {
    if (!c) return MRT_E_ARG;
    if (n_packets < 0 || !O || !D || !rD || !t || !tri || !u || !v) return fail(c, MRT_E_ARG, "null ray array");
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_upload has not been called");
    if (n_packets == 0) return MRT_OK;
    return trace_common(c, "mrt_trace_packets", (size_t)n_packets * 64, n_packets, O, D, rD, t, tri, u, v);
}

int mrt_trace_rays(mrt_ctx* c, int n_rays, const float* O, const float* D, float* t, int32_t* tri, float* u, float* v) {
    if (!c) return MRT_E_ARG;
    if (n_rays < 0 || !O || !D || !t || !tri || !u || !v) return fail(c, MRT_E_ARG, "null ray array");
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_upload has not been called");
    if (n_rays == 0) return MRT_OK;
    return trace_common(c, "mrt_trace_rays", (size_t)n_rays, 0, O, D, nullptr, t, tri, u, v);
}

int mrt_raster_render(mrt_ctx* c, const float cam[16], uint32_t* fb, int pitch_px, float* zbuf, int32_t* tri_ids) {
    if (!c) return MRT_E_ARG;
    if (!cam || !fb) return fail(c, MRT_E_ARG, "null camera or framebuffer");
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_upload has not been called");
    if (pitch_px < c->width) return fail(c, MRT_E_ARG, "pitch smaller than width");
    MRT_HIP(c, hipSetDevice(c->device));
    std::string err;
    RasterTextures tx{c->d_tex.p, c->d_texels.p, c->d_pal.p};
    if (int rc = raster_render(c->raster, tx, c->width, c->height, cam, fb, pitch_px, zbuf, tri_ids, false, c->stream, err))
        return fail(c, rc, err);
    return MRT_OK;
}

int mrt_raster_render_device(mrt_ctx* c, const float cam[16], void* fb_dev, int pitch_px, void* zbuf_dev, void* tri_ids_dev, void* stream) {
    if (!c) return MRT_E_ARG;
    if (!cam || !fb_dev) return fail(c, MRT_E_ARG, "null camera or framebuffer");
    if (!c->has_scene) return fail(c, MRT_E_STATE, "mrt_scene_upload has not been called");
    if (pitch_px < c->width) return fail(c, MRT_E_ARG, "pitch smaller than width");
    MRT_HIP(c, hipSetDevice(c->device));
    std::string err;
    RasterTextures tx{c->d_tex.p, c->d_texels.p, c->d_pal.p};
    if (int rc = raster_render(c->raster, tx, c->width, c->height, cam, (uint32_t*)fb_dev, pitch_px, (float*)zbuf_dev, (int32_t*)tri_ids_dev, true,
                               (hipStream_t)stream, err))
        return fail(c, rc, err);
    return MRT_OK;
}

int mrt_raster_wait(mrt_ctx* c) {
    if (!c) return MRT_E_ARG;
    MRT_HIP(c, hipSetDevice(c->device));
    std::string err;
    if (int rc = raster_wait(c->raster, err)) return fail(c, rc, err);
    return MRT_OK;
}

}  // extern "C"
