// Device-side refit of the ray path's arrays after the scene's vertices have moved (refit.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "bvh_builder.h"

namespace mrt {

// The shape of the tree a context holds, as the refit kernels need it (device memory, made at scene upload).
struct RefitState {
    uint32_t* d_idx = nullptr;        // slot of the device arrays -> triangle id
    uint32_t* d_inner = nullptr;      // slots of the inner nodes, sorted by height (1 = both children are leaves)
    std::vector<uint32_t> level_start;  // host: d_inner[level_start[h] .. level_start[h + 1]) are the nodes of height h
    uint32_t* d_leaves = nullptr;     // slots of the leaves
    uint32_t* d_result = nullptr;     // [0] bits of the root box's largest coordinate, [1] a coordinate outside the division shortcut's range
    uint32_t n_tris = 0, n_slots = 0, n_leaves = 0;
};

hipError_t refit_prepare(RefitState& R, const HostBVH& B, const std::vector<uint32_t>& order, hipStream_t stream);
hipError_t refit_run(RefitState& R, const float* pos, const float* uv, const float* N, const int32_t* tri, float4* nodes, float4* triA, float4* triB,
                     float2* triC, float4* triM, uint4* triG, float4* attr, hipStream_t stream, float* largest, bool* coords_ok);
void refit_release(RefitState& R);

}  // namespace mrt
