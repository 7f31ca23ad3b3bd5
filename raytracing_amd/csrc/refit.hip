// mrt_scene_refit on the device: new boxes, triangle records and shading attributes for the tree the context holds,
// from vertices that have moved.  The reference can only build its tree again (Raytracer::BuildBVH, raytracer.cpp:670-673,
// the whole of BVH::BVH :897-1203, seconds for a large scene); a refit keeps which triangles a leaf holds and how the
// nodes hang together, and recomputes everything that depends on positions:
//   refit_triangles  one thread per leaf-order slot: the 40-byte triangle records (triA/B/C and the merged triM) from the
//                    moved corners, with the subtractions of Triangle::Intersect (raytracer.cpp:289-290);
//   refit_attributes one thread per triangle: face normal and the three uv pairs (raytracer.h:158-170 copies them out of
//                    the mesh when the tree is built);
//   refit_leaf_bounds, refit_inner_bounds
//                    one thread per leaf: the leaf's box over its triangles' corners (compare + select from +-1e30 as
//                    raytracer.h:258-262 does); then the inner nodes from the bottom up, one launch per height (a node's
//                    height is one more than the taller of its children's, so a launch reads only boxes that earlier
//                    launches wrote: no atomics and no fences inside a launch).  The thread that completes the root
//                    leaves the scene's largest coordinate behind, and every thread reports a coordinate outside the
//                    range of the exact-division shortcut (mrt_math.h).  (One launch in which the second of two children
//                    to arrive at a node merged and went on, with arrival counters and a device-scope fence per step,
//                    took 3.4 ms for a tree of 1,000,000 triangles: the fences, one behind the other up the tree.)
//   refit_groups     one thread per group of four leaf-order slots: the padded binary16 box (device_scene.h: triG).
// Vertex data is read from the rasterizer's device arrays (all meshes' vertices back to back, global vertex indices per
// triangle), which mrt_scene_refit has refreshed before.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "bvh_builder.h"
#include "device_scene.h"
#include "mrt_math.h"
#include "refit.h"

namespace mrt {

namespace {

__device__ __forceinline__ f3 vertex(const float* pos, int32_t v) { return {pos[3 * (size_t)v], pos[3 * (size_t)v + 1], pos[3 * (size_t)v + 2]}; }

__global__ __launch_bounds__(256) void refit_triangles(uint32_t n_tris, const uint32_t* idx, const int32_t* tri, const float* pos,
                                                       float4* triA, float4* triB, float2* triC, float4* triM) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tris) return;
    const uint32_t id = idx[k];
    const f3 p0 = vertex(pos, tri[3 * (size_t)id]), p1 = vertex(pos, tri[3 * (size_t)id + 1]), p2 = vertex(pos, tri[3 * (size_t)id + 2]);
    const f3 e1 = p1 - p0, e2 = p2 - p0;
    const float4 A = make_float4(e1.x, e1.y, e1.z, e2.x), B = make_float4(e2.y, e2.z, p0.x, p0.y);
    triA[k] = A; triB[k] = B;
    triC[k] = make_float2(p0.z, __uint_as_float(id));
    triM[3 * (size_t)k] = A; triM[3 * (size_t)k + 1] = B; triM[3 * (size_t)k + 2] = make_float4(p0.z, __uint_as_float(id), 0.0f, 0.0f);
}

// attr[4t] = {normal, refl}, [4t+1] = {uv0, uv1}, [4t+2] = {uv2, refr, rindex}: the material scalars stay
__global__ __launch_bounds__(256) void refit_attributes(uint32_t n_tris, const int32_t* tri, const float* uv, const float* N, float4* attr) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    const int32_t a = tri[3 * (size_t)t], b = tri[3 * (size_t)t + 1], c = tri[3 * (size_t)t + 2];
    const float4 old0 = attr[4 * (size_t)t], old2 = attr[4 * (size_t)t + 2];
    attr[4 * (size_t)t] = make_float4(N[3 * (size_t)t], N[3 * (size_t)t + 1], N[3 * (size_t)t + 2], old0.w);
    attr[4 * (size_t)t + 1] = make_float4(uv[2 * (size_t)a], uv[2 * (size_t)a + 1], uv[2 * (size_t)b], uv[2 * (size_t)b + 1]);
    attr[4 * (size_t)t + 2] = make_float4(uv[2 * (size_t)c], uv[2 * (size_t)c + 1], old2.z, old2.w);
}

__device__ __forceinline__ bool box_coords_ok(f3 lo, f3 hi) {
    return div_shortcut_coord_ok(lo.x) && div_shortcut_coord_ok(lo.y) && div_shortcut_coord_ok(lo.z) &&
           div_shortcut_coord_ok(hi.x) && div_shortcut_coord_ok(hi.y) && div_shortcut_coord_ok(hi.z);
}

// result[0] = bits of the largest absolute coordinate of the root box, result[1] = 1 when some box has a coordinate the
// exact-division shortcut does not cover (zero on entry).
__device__ __forceinline__ void store_box(float4* nodes, uint32_t node, f3 lo, f3 hi, bool is_root, uint32_t* result) {
    nodes[2 * (size_t)node] = make_float4(lo.x, lo.y, lo.z, nodes[2 * (size_t)node].w);
    nodes[2 * (size_t)node + 1] = make_float4(hi.x, hi.y, hi.z, nodes[2 * (size_t)node + 1].w);
    if (!box_coords_ok(lo, hi)) atomicOr(&result[1], 1u);
    if (is_root) {
        const float largest = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
        result[0] = __float_as_uint(largest);
    }
}
__global__ __launch_bounds__(256) void refit_leaf_bounds(uint32_t n_leaves, const uint32_t* leaves, const uint32_t* idx, const int32_t* tri, const float* pos,
                                                         float4* nodes, uint32_t* result) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    const uint32_t node = leaves[i];
    const uint32_t first = __float_as_uint(nodes[2 * (size_t)node].w), count = __float_as_uint(nodes[2 * (size_t)node + 1].w);
    f3 lo = {1e30f, 1e30f, 1e30f}, hi = {-1e30f, -1e30f, -1e30f};
    for (uint32_t k = first; k < first + count; ++k) {
        const uint32_t id = idx[k];
        for (int c = 0; c < 3; ++c) {
            const f3 p = vertex(pos, tri[3 * (size_t)id + c]);
            lo = {gmin(lo.x, p.x), gmin(lo.y, p.y), gmin(lo.z, p.z)};
            hi = {gmax(hi.x, p.x), gmax(hi.y, p.y), gmax(hi.z, p.z)};
        }
    }
    store_box(nodes, node, lo, hi, node == 0u, result);       // (a tree of one leaf: the leaf is the root)
}
// inner[0 .. n): the inner nodes of one height
__global__ __launch_bounds__(256) void refit_inner_bounds(uint32_t n, const uint32_t* inner, float4* nodes, uint32_t* result) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = inner[i];
    const uint32_t l = __float_as_uint(nodes[2 * (size_t)node].w), r = l + 1;
    const float4 l0 = nodes[2 * (size_t)l], l1 = nodes[2 * (size_t)l + 1], r0 = nodes[2 * (size_t)r], r1 = nodes[2 * (size_t)r + 1];
    const f3 lo = {gmin(l0.x, r0.x), gmin(l0.y, r0.y), gmin(l0.z, r0.z)};
    const f3 hi = {gmax(l1.x, r1.x), gmax(l1.y, r1.y), gmax(l1.z, r1.z)};
    store_box(nodes, node, lo, hi, node == 0u, result);
}

// groups: the scene has group boxes.  result[0] = largest coordinate of the new root box (refit_bounds): the padding is
// 2^-12 of it, and a scene that has left the range binary16 covers gets the box no ray misses everywhere.
__global__ __launch_bounds__(256) void refit_groups(uint32_t n_tris, const uint32_t* idx, const int32_t* tri, const float* pos,
                                                    const uint32_t* result, uint4* triG) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (n_tris + 3u) / 4u) return;
    const float largest = __uint_as_float(result[0]);
    const float pad = largest * 0x1p-12f;
    f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
    bool open = !(largest >= 0x1p-6f && largest <= 0x1p14f);
    for (uint32_t k = 4u * g; k < 4u * g + 4u && k < n_tris; ++k) {
        const uint32_t id = idx[k];
        const f3 p0 = vertex(pos, tri[3 * (size_t)id]), p1 = vertex(pos, tri[3 * (size_t)id + 1]), p2 = vertex(pos, tri[3 * (size_t)id + 2]);
        open = open || triangle_is_sliver(p0, p1, p2);
        for (const f3 p : {p0, p1, p2}) {
            lo = {fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)};
            hi = {fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z)};
        }
    }
    if (open) { triG[g] = make_uint4(MRT_GROUP_EVERYWHERE_X, MRT_GROUP_EVERYWHERE_Y, MRT_GROUP_EVERYWHERE_Z, 0u); return; }
    const uint32_t h[6] = {half_outward(lo.x - pad, false), half_outward(lo.y - pad, false), half_outward(lo.z - pad, false),
                           half_outward(hi.x + pad, true), half_outward(hi.y + pad, true), half_outward(hi.z + pad, true)};
    triG[g] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), 0u);
}

}  // namespace

void refit_release(RefitState& R) {
    for (void* p : {(void*)R.d_idx, (void*)R.d_inner, (void*)R.d_leaves, (void*)R.d_result})
        if (p) (void)hipFree(p);
    R = RefitState();
}

#define REFIT_HIP(call)                                                     \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return e_;                                    \
    } while (0)

// At scene upload: what a later refit needs of the tree's shape -- the triangle ids in the order of the device arrays
// (`order`: the leaf order with long leaves rearranged inside), the list of leaves, and the inner nodes sorted by height
// (1 = both children are leaves ... the root's).  (Slots 1 and 2 of the node pool are unused and belong to no tree.)
hipError_t refit_prepare(RefitState& R, const HostBVH& B, const std::vector<uint32_t>& order, hipStream_t stream) {
    refit_release(R);
    R.n_tris = B.n_tris;
    R.n_slots = B.n_tris ? B.pool_ptr + 1 : 0;
    if (R.n_tris == 0) return hipSuccess;
    // nodes in an order in which every node comes before its children (the root first); heights from its reverse
    std::vector<uint32_t> leaves, top_down, todo{0u};
    leaves.reserve(R.n_slots / 2 + 1);
    top_down.reserve(R.n_slots / 2 + 1);
    while (!todo.empty()) {
        const uint32_t n = todo.back();
        todo.pop_back();
        if (B.count[n] != 0) { leaves.push_back(n); continue; }
        top_down.push_back(n);
        const uint32_t l = B.first[n];
        todo.push_back(l); todo.push_back(l + 1);
    }
    std::vector<uint32_t> height(R.n_slots, 0u);
    uint32_t tallest = 0;
    for (size_t k = top_down.size(); k-- > 0;) {
        const uint32_t n = top_down[k], l = B.first[n];
        height[n] = 1u + std::max(height[l], height[l + 1]);
        tallest = std::max(tallest, height[n]);
    }
    R.level_start.assign(tallest + 2, 0u);                  // level_start[h] .. level_start[h + 1]: the nodes of height h in `inner`
    for (uint32_t n : top_down) ++R.level_start[height[n] + 1];
    for (uint32_t h = 1; h < tallest + 2; ++h) R.level_start[h] += R.level_start[h - 1];
    std::vector<uint32_t> inner(top_down.size()), at(R.level_start.begin(), R.level_start.end() - 1);
    for (uint32_t n : top_down) inner[at[height[n]]++] = n;
    R.n_leaves = (uint32_t)leaves.size();
    REFIT_HIP(hipMalloc((void**)&R.d_idx, (size_t)R.n_tris * 4));
    REFIT_HIP(hipMalloc((void**)&R.d_inner, std::max<size_t>(1, inner.size()) * 4));
    REFIT_HIP(hipMalloc((void**)&R.d_leaves, (size_t)R.n_leaves * 4));
    REFIT_HIP(hipMalloc((void**)&R.d_result, 2 * sizeof(uint32_t)));
    REFIT_HIP(hipMemcpyAsync(R.d_idx, order.data(), (size_t)R.n_tris * 4, hipMemcpyHostToDevice, stream));
    if (!inner.empty()) REFIT_HIP(hipMemcpyAsync(R.d_inner, inner.data(), inner.size() * 4, hipMemcpyHostToDevice, stream));
    REFIT_HIP(hipMemcpyAsync(R.d_leaves, leaves.data(), (size_t)R.n_leaves * 4, hipMemcpyHostToDevice, stream));
    return hipStreamSynchronize(stream);           // the staging vectors go out of scope
}

// The kernels above, on `stream`; pos / uv / N / tri are the rasterizer's device arrays with the moved vertices in them.
// Returns after the stream has drained: largest = largest absolute coordinate of the new root box, coords_ok = every
// box coordinate suits the exact-division shortcut.
hipError_t refit_run(RefitState& R, const float* pos, const float* uv, const float* N, const int32_t* tri, float4* nodes, float4* triA, float4* triB,
                     float2* triC, float4* triM, uint4* triG, float4* attr, hipStream_t stream, float* largest, bool* coords_ok) {
    *largest = 0.0f; *coords_ok = true;
    if (R.n_tris == 0) return hipSuccess;
    const dim3 block(256);
    REFIT_HIP(hipMemsetAsync(R.d_result, 0, 2 * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(refit_triangles, dim3((R.n_tris + 255u) / 256u), block, 0, stream, R.n_tris, (const uint32_t*)R.d_idx, tri, pos, triA, triB, triC, triM);
    hipLaunchKernelGGL(refit_attributes, dim3((R.n_tris + 255u) / 256u), block, 0, stream, R.n_tris, tri, uv, N, attr);
    hipLaunchKernelGGL(refit_leaf_bounds, dim3((R.n_leaves + 255u) / 256u), block, 0, stream, R.n_leaves, (const uint32_t*)R.d_leaves,
                       (const uint32_t*)R.d_idx, tri, pos, nodes, R.d_result);
    for (size_t h = 1; h + 1 < R.level_start.size(); ++h) {
        const uint32_t n = R.level_start[h + 1] - R.level_start[h];
        if (n == 0) continue;
        hipLaunchKernelGGL(refit_inner_bounds, dim3((n + 255u) / 256u), block, 0, stream, n, (const uint32_t*)R.d_inner + R.level_start[h], nodes, R.d_result);
    }
    if (triG)
        hipLaunchKernelGGL(refit_groups, dim3(((R.n_tris + 3u) / 4u + 255u) / 256u), block, 0, stream, R.n_tris, (const uint32_t*)R.d_idx, tri, pos,
                           (const uint32_t*)R.d_result, triG);
    REFIT_HIP(hipGetLastError());
    uint32_t out[2] = {0, 0};
    REFIT_HIP(hipMemcpyAsync(out, R.d_result, sizeof out, hipMemcpyDeviceToHost, stream));
    REFIT_HIP(hipStreamSynchronize(stream));
    *largest = bits2f(out[0]);
    *coords_ok = out[1] == 0;
    return hipSuccess;
}

}  // namespace mrt
