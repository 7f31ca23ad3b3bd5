// Device-side views of the flattened scene and of one frame, passed to the kernels by value.
//
// HBM layout (all arrays 16-byte aligned, uploaded once per mrt_scene_upload):
//   nodes : 2 x float4 per node slot i, the reference's 32-byte BVHNode (raytracer.h:327-341):
//             [2i]   = {min.x, min.y, min.z, bits(firstLeft)}
//             [2i+1] = {max.x, max.y, max.z, bits(count)}
//           An inner node's two children are slots firstLeft and firstLeft+1, i.e. 64
//           contiguous bytes.  For a leaf, firstLeft is the first record in the triangle arrays.
//   triA/triB/triC : 40 bytes per LEAF-ORDER slot k (position in BVH::m_TriangleIdx), so a
//           leaf's triangles are contiguous and are read front to back like the reference does:
//             triA[k] = {e1.x, e1.y, e1.z, e2.x}       e1 = p1-p0, e2 = p2-p0: the two subtractions
//             triB[k] = {e2.y, e2.z, p0.x, p0.y}       Triangle::Intersect performs first
//             triC[k] = {p0.z, bits(triangle id)}      (raytracer.cpp:289-290)
//           The edges come first: the first 24 bytes decide whether the triangle faces away from a
//           ray, which dismisses about half of all tests before the rest is read.
//           Inside a leaf of kGroupMinLeaf or more triangles the slots are NOT in the reference's order: they are
//           arranged so that neighbours in the array are neighbours in space (mrt_api.cpp: arrange_long_leaves), and
//   place : per triangle ID {the triangle's place in the reference's order, the first place of its leaf}: settles what
//           the order of the tests used to settle -- of two triangles of ONE leaf hit at exactly the same distance the
//           one with the lower place is kept (kernels.hip: Tie).
//   triM  : the triangle arrays once more as ONE record of 3 float4 per leaf-order slot, {triA, triB, {triC, 0, 0}}:
//           the traversals in which all lanes test the same triangle (packet leaves, shadow rays) of the
//           global-memory kernels read a triangle through one address instead of three.
//   triG  : one uint4 per group of four consecutive leaf-order slots (4g .. 4g+3, whatever leaves they belong
//           to): a box around those triangles, padded by 2^-12 of the scene's largest coordinate and rounded
//           outwards to binary16: {lo.x | lo.y << 16, lo.z | hi.x << 16, hi.y | hi.z << 16, 0}.  A ray that
//           misses the box cannot hit any of the four, so the single-ray traversal skips their tests in long
//           leaves (kernels.hip: traverse_nearest).  nullptr when the scene's coordinates do not suit binary16.
//   attr  : 4 x float4 per triangle ID (shading only, touched once per hit):
//             [4t]   = {normal.xyz, refl}
//             [4t+1] = {uv0.x, uv0.y, uv1.x, uv1.y}
//             [4t+2] = {uv2.x, uv2.y, refr, rindex}
//             [4t+3] = {bits(texture id), 0, 0, 0}
//   tex   : int4 per texture {offset of its texels in `texels`, width, height, 0}
//   texels: all palette-index bytes back to back
//   pal   : 32 x 256 uint32 per texture (surface.h:53), level 15 is what the tracer reads
//   lights: 2 float4 per light {pos.xyz, 0}, {colour.xyzw}
//
// Small scenes (nodes + triangle arrays up to about 110 KB, e.g. the default maze: 25 KB +
// 85 KB) are additionally copied into every workgroup's LDS by the resident kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrt {

struct SceneDev {
    const float4* nodes;
    const float4* triA;
    const float4* triB;
    const float2* triC;
    const uint4* triG;
    const float4* triM;
    const uint2* place;      // per triangle id: {its place in the tree's own leaf order, the first place of its leaf}
    const float4* attr;
    const int4* tex;
    const uint8_t* texels;
    const uint32_t* pal;
    const float4* lights;
    const float4* objects;   // analytic primitives, 3 float4 each (kernels.hip: trace_objects); nullptr without any
    int n_objects;
    int n_lights;
    int n_tris;
    int n_textures;
    int n_node_slots;     // node slots in use (poolPtr + 1)
    int coords_ok;        // every node bound is 0 or 2^-27 <= |c| <= 2^30: exact-division shortcut allowed
    // Only when nodes / triA / triB / triC are the second, GPU-built tree of a scene uploaded with
    // mrt_scene_upload_lbvh_exact (kernels.hip: hit_is_the_references): per triangle ID the box of its leaf in the
    // REFERENCE's tree, {min.xyz, 0}, {max.xyz, 0}; and the reference's tree itself, for the rays that have to ask it.
    const float4* leaf_box;
    float reach;          // 64 times the scene's largest coordinate: ray origins within it may use the second tree's cheap box test (kernels.hip: slab_mul_pair)
    const float4* ref_nodes;
    const float4* ref_triA;
    const float4* ref_triB;
    const float2* ref_triC;
    const float4* ref_triM;
    const uint4* ref_triG;
    const uint2* ref_place;
    int ref_coords_ok;
};

struct FrameDev {
    float p0[3], d10[3], d20[3], cam[3];   // view-plane corner, p1-p0, p2-p0, eye (raytracer.cpp:422-434)
    float inv_w, inv_h;
    int width, height;       // of the traced image (output size times the supersampling grid)
    int tiles_x;             // width / 8
    int tile_y0;             // first tile row of this launch (contiguous band, n_parts <= 1)
    int n_tiles;             // tiles in this launch
    // Interleaved share of the frame (n_parts > 1): this launch renders the OUTPUT tile rows R
    // (8 output rows each) with (R / part_block) % n_parts == part, in increasing order.  With
    // `compact` set, the rows are written packed (the launch's k-th row of pixels at row k of
    // fb / hit_ids) instead of at their place in the frame.
    int part, n_parts, part_block, compact;
    int grid_g;              // supersampling grid g: 1, 2, 4 or 8
    int pitch;               // output row pitch in pixels
    int stack_entries;       // per-ray traversal stack entries this BVH needs (depth + 2)
    int stats;               // accumulate the ray counters
    int n_queues;            // tile queues of the persistent kernel (set by its launcher: at most one per workgroup)
    uint32_t* fb;            // output frame (output resolution)
    int32_t* hit_ids;        // optional, traced-image resolution
    // counters[0..15]: [5] watchdog code, [6] deepest pixel of the BVH-depth view, [7] entries of the tile order list,
    // [8] rays traced again through the reference's tree (scenes with a second tree).
    // counters[kQueueBase + 16*q]: next index of tile queue q
    // (resident kernel), one cache line per queue.
    // Ray statistics live in kStatSlots groups of 16 words (128 bytes, one cache line each) after
    // the queues: group s, word 0 = tiles (low 32 bits; 64 primary rays each) and primary hits (high 32 bits),
    // word 1 = shadow rays minus (primary hits x lights), two's complement, word 2 = secondary rays.  A tile
    // adds into group (tile % kStatSlots) so that the atomics of concurrent waves spread over
    // many lines instead of queueing on one address; the host sums the groups.
    unsigned long long* counters;
    // Temporal load balancing of the persistent kernel (both optional).  tile_cost[tile]: shader
    // clocks this launch spent on the tile.  tile_order: a permutation of [0, n_tiles), most
    // expensive tiles of the previous launch first; the queues hand tiles out in that order.  An entry is
    // tile | sub << 24: sub 0 = the whole tile, 1 ... 16 = one sixteenth of it (kernels.hip: render_tile), and the
    // list has counters[kOrderCountSlot] entries.
    uint32_t* tile_cost;
    const uint32_t* tile_order;
    // The scene relative to this frame's eye, for the packet traversal of the primary rays (kernels.hip:
    // SceneGlobalEyeT, eye_relative_kernel); set for every packet-path launch of the one-wave-per-tile kernel.
    const float4* eye_nodes;
    const float4* eye_tri;
    const float4* eye_g;
    // the same of a scene's second tree (SceneDev::leaf_box; no group boxes), for packet-path launches through it
    const float4* eye_nodes2;
    const float4* eye_tri2;
};

constexpr int kOrderCountSlot = 7;     // counters[7]: entries of the frame's tile order list (tile_order_kernel)
constexpr int kRedoCountSlot = 8;      // counters[8]: rays of a frame through a scene's second tree that were traced again through the reference's
constexpr int kOrderExtra = 16 * 2048; // entries the list may hold beyond one per tile (tiles handed out as sixteenths)
constexpr int kStatSlots = 256;
constexpr int kTileQueues = 32;
constexpr int kQueueBase = 16;
constexpr int kStatBase = kQueueBase + 16 * kTileQueues;
// 256 32-bit words after the statistics: histogram of the tile-ordering sort.  Zeroed with the rest
// at the start of every frame; the ordering kernels run between that and the frame kernel.
constexpr int kOrderHistBase = kStatBase + 16 * kStatSlots;
#ifdef MRT_LOOP_COUNTS
// Diagnostic build only (make prof): per-site loop iteration counts, wave level and lane level.
constexpr int kProfSites = 22;
constexpr int kProfBase = kOrderHistBase + 128;
constexpr int kCounterSlots = kProfBase + 2 * kProfSites;
#elif defined(MRT_WAVE_LIVES)
// Diagnostic build only (make lives): the release kernels, and every persistent wave leaves {clock at entry, clock
// at exit, tiles rendered} behind (scripts/wave_lives.py).
constexpr int kLivesBase = kOrderHistBase + 128;
constexpr int kLivesWaves = 4096;
constexpr int kCounterSlots = kLivesBase + 3 * kLivesWaves;
#else
constexpr int kCounterSlots = kOrderHistBase + 128;
#endif
constexpr int kGroupMinLeaf = 6;       // shortest leaf whose triangles are tested group by group (triG)
constexpr int kStackDepth = 64;        // per-ray traversal stack entries; BVH depth must stay below
constexpr int kResidentThreads = 1024; // workgroup size of the LDS-resident kernel (16 waves)
constexpr int kLdsBudget = 160 * 1024; // bytes of LDS per CU on gfx950

}  // namespace mrt
