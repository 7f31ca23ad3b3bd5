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

struct RasterMesh {
    int verts = 0, tris = 0, texture = -1, tri_base = 0;
    f3 bounds[2];
    size_t vert_off = 0, tri_off = 0;    // offsets into the concatenated device arrays
};

struct RasterState {
    std::vector<RasterMesh> meshes;
    float* d_pos = nullptr;      // all vertices, 3 floats each
    float* d_uv = nullptr;       // 2 floats each
    float* d_N = nullptr;        // per triangle, 3 floats
    int32_t* d_tri = nullptr;    // per triangle, 3 vertex indices (mesh-local)
    int32_t* d_tri_mesh = nullptr;   // per triangle, mesh index
    size_t n_verts = 0, n_tris = 0;
    unsigned long long* d_depth = nullptr;   // W*H packed {orderable 1/z, triangle index}
    uint32_t* d_fb = nullptr;
    float* d_z = nullptr;
    int32_t* d_ids = nullptr;
    size_t frame_px = 0;
    void* d_meshinfo = nullptr;
};

int raster_upload(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err);
int raster_render(RasterState& R, const RasterTextures& tx, int W, int H, const float cam[16], uint32_t* fb, int pitch_px,
                  float* zbuf, int32_t* tri_ids, hipStream_t stream, std::string& err);
void raster_release(RasterState& R);

}  // namespace mrt
