// Rasterizer path (Rasterizer::Render, rasterizer.cpp:53-184) -- host-side state and entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "mrt_math.h"

namespace mrt {

struct RasterTextures {
    const int4* tex;
    const uint8_t* texels;
    const uint32_t* pal;
};

struct RasterMeshHost {
    int verts = 0, tris = 0, texture = -1;
    uint32_t vert_off = 0, tri_off = 0;    // offsets into the concatenated device arrays
    f3 bounds[2];                          // scenedata.cpp:258-263
};

// One triangle after set-up: what the coverage and resolve kernels need (device memory).  32 words on a 128-byte
// boundary: a record is one memory line, a wave fetches it with one load (lane i takes word i), and the first six words
// are all the tile kernel's first look at a list entry needs.
struct alignas(128) RasterTri {
    uint32_t box_x, box_y;    // pixel bounding box (inclusive, already clamped to the drawable area): x0 | x1 << 16, y0 | y1 << 16
    float zA, zB, zC;         // 1/z      = zA*x + zB*y + zC
    uint32_t info;            // edge count | palette level (0..31) << 8 | texture index << 16; edge count 0 = culled / clipped away
    float uA, uB, uC;         // u/z      likewise
    float vA, vB, vC;         // v/z      likewise
    // polygon edges after the two clip planes (up to 5), each oriented top to bottom:
    // x(y) = ex + dx * (y - ey) on rows iy0..iy1 (rasterizer.cpp:146-160); rows = iy0 | iy1 << 16
    struct Edge { float ex, ey, dx; uint32_t rows; } edge[5];
};
static_assert(sizeof(RasterTri) == 128, "a set-up record is one 128-byte line");
constexpr int kRasterMaxDim = 32768;       // frame width and height the packed pixel boxes and row ranges can hold
constexpr int kRasterMaxTexture = 65535;

struct RasterState {
    std::vector<RasterMeshHost> meshes;
    float* d_pos = nullptr;          // all vertices, 3 floats each
    float* d_uv = nullptr;           // 2 floats each
    float* d_N = nullptr;            // per triangle, 3 floats
    int32_t* d_tri = nullptr;        // per triangle, 3 vertex indices into d_pos/d_uv (already offset)
    int32_t* d_tri_mesh = nullptr;   // per triangle: mesh index (uploaded once)
    // The order in which set-up takes the triangles (place -> triangle id): the tree's leaf order, in which neighbours
    // are neighbours in space.  And what set-up reads, gathered in that order at upload (and again after a refit) so
    // that lane i reads 16 bytes at i of each array:
    // in_a = {p0.xyz, p1.x}, in_b = {p1.yz, p2.xy}, in_c = {p2.z, N.xyz}, in_u = {t0, t1}, in_v = {t2, mesh, triangle id}
    uint32_t* d_order = nullptr;
    uint4* d_in_a = nullptr;
    uint4* d_in_b = nullptr;
    uint4* d_in_c = nullptr;
    uint4* d_in_u = nullptr;
    uint4* d_in_v = nullptr;
    int32_t* d_mesh_tex = nullptr;   // per mesh, every frame: texture index, or -1 when the mesh is culled / untextured
    RasterTri* d_setup = nullptr;
    uint2* d_bins = nullptr;          // per place of the set-up order: tile range, depth class, drawn flag (8 bytes; what the binning passes read)
    uint32_t* d_chunk_base = nullptr; // n_tris words: list of the places whose triangles overlap many tiles
    uint32_t* d_total = nullptr;     // [0] (tile, triangle) pairs this frame, [1] length of the big-triangle list (inside d_tile_count's allocation)
    uint32_t* d_tile_count = nullptr; // per (32x32 tile, depth class): triangle count, then fill cursor
    uint32_t* d_tile_start = nullptr; // per (tile, depth class) (+1): start of its list segment in d_pairs
    uint32_t* d_scan_sums = nullptr;  // sums of chunks of d_tile_count (the scan's first pass)
    uint32_t n_tile_slots = 0;
    uint32_t* d_pairs = nullptr;     // triangle indices, grouped by tile
    size_t pair_capacity = 0;
    // Frames are queued without waiting for their list length.  The scan kernel raises d_short to the length a frame
    // needed when that exceeds the capacity it was queued with; every frame copies the word to h_short (page-locked)
    // behind itself and records frame_done.  short_seen: a frame since the last raster_wait was incomplete.
    uint32_t* d_short = nullptr;
    volatile uint32_t* h_short = nullptr;
    hipEvent_t frame_done = nullptr;
    bool frame_queued = false, short_seen = false;
    uint32_t n_verts = 0, n_tris = 0;
    std::vector<int32_t> h_tri_mesh; // per triangle: mesh index
    std::vector<int32_t> h_mesh_tex; // host copy of d_mesh_tex for the frame being queued
    uint32_t* d_fb = nullptr;
    float* d_z = nullptr;
    int32_t* d_ids = nullptr;
    size_t frame_px = 0;
    void* h_stage = nullptr;          // page-locked staging for the frame coming back
    size_t h_stage_bytes = 0;
};

// order: n triangle ids (over the meshes concatenated), the order set-up works in; null = as given
int raster_upload(RasterState& R, const mrt_mesh* meshes, int n_meshes, const uint32_t* order, hipStream_t stream, std::string& err);
// The same meshes with moved vertices (same vertex and triangle counts, checked by the caller): positions, uvs and face
// normals are copied over the ones on the device, mesh bounds are recomputed; nothing is allocated.  Asynchronous on `stream`.
int raster_refit(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err);
int raster_render(RasterState& R, const RasterTextures& tx, int W, int H, const float cam[16], uint32_t* fb, int pitch_px,
                  float* zbuf, int32_t* tri_ids, bool device_out, hipStream_t stream, std::string& err);
// Waits for the last frame queued into device memory.  MRT_E_LIMIT: a frame since the last call needed longer tile
// lists than the buffer held and is incomplete; the buffer has grown to what it needed, render it again.
int raster_wait(RasterState& R, std::string& err);
void raster_release(RasterState& R);

}  // namespace mrt
