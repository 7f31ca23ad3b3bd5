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

// One triangle after set-up: what the coverage and resolve kernels need (device memory).
struct RasterTri {
    // polygon edges after the two clip planes (up to 5), each oriented top to bottom:
    // x(y) = ex + dx * (y - ey) on rows iy0..iy1 (rasterizer.cpp:146-160)
    float ex[5], ey[5], dx[5];
    int iy0[5], iy1[5];
    int n;                    // edge count, 0 = culled / clipped away
    int x0, x1, y0, y1;       // pixel bounding box (inclusive), already clamped to the drawable area
    float zA, zB, zC;         // 1/z      = zA*x + zB*y + zC
    float uA, uB, uC;         // u/z      likewise
    float vA, vB, vC;         // v/z      likewise
    int texture, level;       // texture index and palette level (0..31)
};

struct RasterState {
    std::vector<RasterMeshHost> meshes;
    float* d_pos = nullptr;          // all vertices, 3 floats each
    float* d_uv = nullptr;           // 2 floats each
    float* d_N = nullptr;            // per triangle, 3 floats
    int32_t* d_tri = nullptr;        // per triangle, 3 vertex indices into d_pos/d_uv (already offset)
    int32_t* d_tri_mesh = nullptr;   // per triangle: mesh index (uploaded once)
    int32_t* d_mesh_tex = nullptr;   // per mesh, every frame: texture index, or -1 when the mesh is culled / untextured
    RasterTri* d_setup = nullptr;
    uint2* d_bins = nullptr;          // per triangle: tile range, near / drawn flags (8 bytes; what the binning passes read)
    uint32_t* d_chunk_base = nullptr; // n_tris words: list of the triangles that overlap many tiles
    uint32_t* d_total = nullptr;     // [0] (tile, triangle) pairs this frame, [1] length of the big-triangle list (inside d_tile_count's allocation)
    uint32_t* d_tile_count = nullptr; // per (32x32 tile, depth class): triangle count, then fill cursor
    uint32_t* d_tile_start = nullptr; // per (tile, depth class) (+1): start of its list segment in d_pairs
    uint32_t* d_scan_sums = nullptr;  // sums of chunks of d_tile_count (the scan's first pass)
    uint32_t n_tile_slots = 0;
    uint32_t* d_pairs = nullptr;     // triangle indices, grouped by tile
    size_t pair_capacity = 0;
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

int raster_upload(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err);
// The same meshes with moved vertices (same vertex and triangle counts, checked by the caller): positions, uvs and face
// normals are copied over the ones on the device, mesh bounds are recomputed; nothing is allocated.  Asynchronous on `stream`.
int raster_refit(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err);
int raster_render(RasterState& R, const RasterTextures& tx, int W, int H, const float cam[16], uint32_t* fb, int pitch_px,
                  float* zbuf, int32_t* tri_ids, bool device_out, hipStream_t stream, std::string& err);
void raster_release(RasterState& R);

}  // namespace mrt
