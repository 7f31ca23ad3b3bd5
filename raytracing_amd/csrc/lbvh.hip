// GPU construction of a linear BVH (Morton order + binary radix tree) over the flattened triangles,
// as an ALTERNATIVE to the reference's own builder (raytracer.cpp:897-1203, restated on the host in
// bvh_builder.cpp).  The reference tree defines the bit-exact hit IDs -- which triangle wins when two
// are hit at exactly the same distance depends on leaf order -- so this tree is only used when the
// caller asks for it (mrt_scene_upload_lbvh): it is built in milliseconds instead of seconds, gives
// the same nearest distances, and may name the other triangle of an exact tie.
//
// Output has the layout every traversal kernel expects (device_scene.h): node slots with the two
// children of an inner node in adjacent slots, leaves pointing at runs of the triangle order.
//   1. 30-bit Morton code of every centroid inside the centroid bounds; radix sort (hipCUB).
//   2. One thread per internal node of the binary radix tree (Karras 2012): range and split from the
//      longest-common-prefix function, ties between equal codes broken by position.
//   3. Bounds bottom-up: one thread per triangle starts at its leaf; at every internal node the
//      second thread to arrive merges the two child boxes and moves on.
//   4. One thread per internal node writes its two children into slots 3+2i and 4+2i (slots 1 and 2
//      stay unused as in the reference layout); a subtree of at most kLeafMax triangles becomes a leaf.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>

#include <string>
#include <vector>

#include <mutex>

#include "bvh_builder.h"
#include "host_parallel.h"

namespace mrt {

namespace {

constexpr int kLeafMax = 4;

struct Tri9 { float v[9]; };

__device__ __forceinline__ uint32_t spread10(uint32_t x) {      // 10 bits -> every third bit
    x &= 0x3ffu;
    x = (x | (x << 16)) & 0x030000ffu;
    x = (x | (x << 8)) & 0x0300f00fu;
    x = (x | (x << 4)) & 0x030c30c3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void lbvh_codes(const Tri9* tris, uint32_t n, float3 lo, float3 inv_extent, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* v = tris[i].v;
    const float c[3] = {(v[0] + v[3] + v[6]) * (1.0f / 3.0f), (v[1] + v[4] + v[7]) * (1.0f / 3.0f), (v[2] + v[5] + v[8]) * (1.0f / 3.0f)};
    const float l[3] = {lo.x, lo.y, lo.z}, s[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        float f = (c[a] - l[a]) * s[a] * 1024.0f;
        f = f < 0.0f ? 0.0f : (f > 1023.0f ? 1023.0f : f);      // NaN -> 0 through the first comparison's else branch order
        q[a] = (uint32_t)f;
    }
    keys[i] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    vals[i] = i;
}

// length of the common prefix of the keys at sorted positions i and j (-1 outside the array);
// equal keys are told apart by their positions
__device__ __forceinline__ int prefix(const uint32_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a != b ? __clz(a ^ b) : 32 + __clz((uint32_t)i ^ (uint32_t)j);
}

// Internal node i of n-1; child references: index with bit 31 set = leaf (sorted position).
__global__ __launch_bounds__(256) void lbvh_tree(const uint32_t* keys, int n, uint32_t* left, uint32_t* right, int* range_lo, int* range_hi,
                                                 int* parent_internal, int* parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix(keys, n, i, i - d);
    int lmax = 2;
    while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = prefix(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int split = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    range_lo[i] = lo; range_hi[i] = hi;
    if (split == lo) { left[i] = 0x80000000u | (uint32_t)split; parent_leaf[split] = i; }
    else { left[i] = (uint32_t)split; parent_internal[split] = i; }
    if (split + 1 == hi) { right[i] = 0x80000000u | (uint32_t)(split + 1); parent_leaf[split + 1] = i; }
    else { right[i] = (uint32_t)(split + 1); parent_internal[split + 1] = i; }
    if (i == 0) parent_internal[0] = -1;
}

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ Box6 tri_box(const Tri9& t) {
    Box6 b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(t.v[a], fminf(t.v[3 + a], t.v[6 + a]));
        b.hi[a] = fmaxf(t.v[a], fmaxf(t.v[3 + a], t.v[6 + a]));
    }
    return b;
}

__global__ __launch_bounds__(256) void lbvh_bounds(const Tri9* tris, const uint32_t* vals, int n, const uint32_t* left, const uint32_t* right,
                                                   const int* parent_internal, const int* parent_leaf, Box6* leaf_box, Box6* node_box,
                                                   unsigned int* arrived, int* max_depth) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    leaf_box[k] = tri_box(tris[vals[k]]);
    __threadfence();
    int node = parent_leaf[k], depth = 1;
    while (node >= 0) {
        if (atomicAdd(&arrived[node], 1u) == 0u) return;           // the sibling's thread will come by later and do the merge
        __threadfence();
        const uint32_t l = left[node], r = right[node];
        const Box6 a = (l & 0x80000000u) ? leaf_box[l & 0x7fffffffu] : node_box[l];
        const Box6 b = (r & 0x80000000u) ? leaf_box[r & 0x7fffffffu] : node_box[r];
        Box6 m;
        for (int x = 0; x < 3; ++x) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
        node_box[node] = m;
        __threadfence();
        node = parent_internal[node];
        ++depth;
    }
    (void)depth;
    (void)max_depth;
}

// depth of every leaf (edges to the root), maximum over all: the traversal stacks are sized from it
__global__ __launch_bounds__(256) void lbvh_depth(int n, const int* parent_internal, const int* parent_leaf, int* max_depth) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int d = 0;
    for (int node = parent_leaf[k]; node >= 0; node = parent_internal[node]) ++d;
    atomicMax(max_depth, d);
}

__device__ __forceinline__ void write_slot(float4* nodes, uint32_t slot, const Box6& b, uint32_t first, uint32_t count) {
    nodes[2 * (size_t)slot] = make_float4(b.lo[0], b.lo[1], b.lo[2], __uint_as_float(first));
    nodes[2 * (size_t)slot + 1] = make_float4(b.hi[0], b.hi[1], b.hi[2], __uint_as_float(count));
}

__device__ __forceinline__ void emit_child(float4* nodes, uint32_t slot, uint32_t ref, const Box6* leaf_box, const Box6* node_box,
                                           const int* range_lo, const int* range_hi) {
    if (ref & 0x80000000u) {
        const uint32_t k = ref & 0x7fffffffu;
        write_slot(nodes, slot, leaf_box[k], k, 1u);
    } else {
        const int size = range_hi[ref] - range_lo[ref] + 1;
        if (size <= kLeafMax) write_slot(nodes, slot, node_box[ref], (uint32_t)range_lo[ref], (uint32_t)size);
        else write_slot(nodes, slot, node_box[ref], 3u + 2u * ref, 0u);
    }
}

__global__ __launch_bounds__(256) void lbvh_emit(int n, const uint32_t* left, const uint32_t* right, const Box6* leaf_box, const Box6* node_box,
                                                 const int* range_lo, const int* range_hi, float4* nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    emit_child(nodes, 3u + 2u * (uint32_t)i, left[i], leaf_box, node_box, range_lo, range_hi);
    emit_child(nodes, 4u + 2u * (uint32_t)i, right[i], leaf_box, node_box, range_lo, range_hi);
    if (i == 0) {
        if (n <= kLeafMax) write_slot(nodes, 0u, node_box[0], 0u, (uint32_t)n);
        else write_slot(nodes, 0u, node_box[0], 3u, 0u);
    }
}

template <typename T>
struct Dev {
    T* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

}  // namespace

#define LBVH_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

// Fills bvh.idx / nmin / nmax / first / count / pool_ptr / max_depth from bvh.tris.  Returns false with
// err set on a HIP failure.  bvh.tris must not be empty.
bool build_lbvh_device(HostBVH& B, hipStream_t stream, std::string& err) {
    const uint32_t n = (uint32_t)B.tris.size();
    B.n_tris = n;
    B.n_nodes_alloc = 2 * n + 1;
    const size_t slots = (size_t)B.n_nodes_alloc + 2;
    std::vector<Tri9> host(n);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::mutex mu;
    parallel_for(n, [&](size_t b, size_t e) {
        float l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = b; i < e; ++i) {
            const HostTri& t = B.tris[i];
            const float v[9] = {t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z};
            for (int k = 0; k < 9; ++k) host[i].v[k] = v[k];
            const float c[3] = {t.centroid.x, t.centroid.y, t.centroid.z};
            for (int a = 0; a < 3; ++a) { if (c[a] < l[a]) l[a] = c[a]; if (c[a] > h[a]) h[a] = c[a]; }
        }
        std::lock_guard<std::mutex> lk(mu);
        for (int a = 0; a < 3; ++a) { if (l[a] < lo[a]) lo[a] = l[a]; if (h[a] > hi[a]) hi[a] = h[a]; }
    });
    float inv[3];
    for (int a = 0; a < 3; ++a) {
        if (!(lo[a] <= hi[a])) { lo[a] = 0; hi[a] = 0; }            // all-NaN axis
        const float e = hi[a] - lo[a];
        inv[a] = (e > 0.0f && e < INFINITY) ? 1.0f / e : 0.0f;
    }
    Dev<Tri9> d_tris;
    Dev<uint32_t> k0, k1, v0, v1, left, right;
    Dev<int> rlo, rhi, pint, pleaf, depth;
    Dev<Box6> leaf_box, node_box;
    Dev<unsigned int> arrived;
    Dev<float4> nodes;
    Dev<unsigned char> temp;
    LBVH_HIP(d_tris.alloc(n)); LBVH_HIP(k0.alloc(n)); LBVH_HIP(k1.alloc(n)); LBVH_HIP(v0.alloc(n)); LBVH_HIP(v1.alloc(n));
    LBVH_HIP(left.alloc(n)); LBVH_HIP(right.alloc(n)); LBVH_HIP(rlo.alloc(n)); LBVH_HIP(rhi.alloc(n));
    LBVH_HIP(pint.alloc(n)); LBVH_HIP(pleaf.alloc(n)); LBVH_HIP(depth.alloc(1));
    LBVH_HIP(leaf_box.alloc(n)); LBVH_HIP(node_box.alloc(n)); LBVH_HIP(arrived.alloc(n)); LBVH_HIP(nodes.alloc(2 * slots));
    LBVH_HIP(hipMemcpyAsync(d_tris.p, host.data(), (size_t)n * sizeof(Tri9), hipMemcpyHostToDevice, stream));
    LBVH_HIP(hipMemsetAsync(arrived.p, 0, (size_t)n * sizeof(unsigned int), stream));
    LBVH_HIP(hipMemsetAsync(depth.p, 0, sizeof(int), stream));
    LBVH_HIP(hipMemsetAsync(nodes.p, 0, 2 * slots * sizeof(float4), stream));
    LBVH_HIP(hipMemsetAsync(pleaf.p, 0xff, (size_t)n * sizeof(int), stream));      // -1: a lone leaf has no parent
    LBVH_HIP(hipMemsetAsync(pint.p, 0xff, (size_t)n * sizeof(int), stream));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(lbvh_codes, grid, block, 0, stream, d_tris.p, n, make_float3(lo[0], lo[1], lo[2]), make_float3(inv[0], inv[1], inv[2]), k0.p, v0.p);
    size_t temp_bytes = 0;
    LBVH_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 30, stream));
    LBVH_HIP(temp.alloc(temp_bytes));
    LBVH_HIP(hipcub::DeviceRadixSort::SortPairs(temp.p, temp_bytes, k0.p, k1.p, v0.p, v1.p, (int)n, 0, 30, stream));
    if (n > 1) {
        hipLaunchKernelGGL(lbvh_tree, grid, block, 0, stream, k1.p, (int)n, left.p, right.p, rlo.p, rhi.p, pint.p, pleaf.p);
        hipLaunchKernelGGL(lbvh_bounds, grid, block, 0, stream, d_tris.p, v1.p, (int)n, left.p, right.p, pint.p, pleaf.p, leaf_box.p, node_box.p,
                           arrived.p, depth.p);
        hipLaunchKernelGGL(lbvh_depth, grid, block, 0, stream, (int)n, pint.p, pleaf.p, depth.p);
        hipLaunchKernelGGL(lbvh_emit, grid, block, 0, stream, (int)n, left.p, right.p, leaf_box.p, node_box.p, rlo.p, rhi.p, nodes.p);
    }
    LBVH_HIP(hipGetLastError());
    B.idx.resize(n);
    std::vector<float4> h_nodes(2 * slots);
    int h_depth = 0;
    LBVH_HIP(hipMemcpyAsync(B.idx.data(), v1.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipMemcpyAsync(h_nodes.data(), nodes.p, 2 * slots * sizeof(float4), hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipMemcpyAsync(&h_depth, depth.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipStreamSynchronize(stream));
    B.nmin.assign(slots, f3{0, 0, 0});
    B.nmax.assign(slots, f3{0, 0, 0});
    B.first.assign(slots, 0);
    B.count.assign(slots, 0);
    if (n == 1) {
        const float* v = host[0].v;
        B.nmin[0] = {fminf(v[0], fminf(v[3], v[6])), fminf(v[1], fminf(v[4], v[7])), fminf(v[2], fminf(v[5], v[8]))};
        B.nmax[0] = {fmaxf(v[0], fmaxf(v[3], v[6])), fmaxf(v[1], fmaxf(v[4], v[7])), fmaxf(v[2], fmaxf(v[5], v[8]))};
        B.first[0] = 0; B.count[0] = 1;
        B.pool_ptr = 2;
        B.max_depth = 0;
        return true;
    }
    parallel_for(slots, [&](size_t sb, size_t se) {
        for (size_t i = sb; i < se; ++i) {
            const float4 a = h_nodes[2 * i], b = h_nodes[2 * i + 1];
            B.nmin[i] = {a.x, a.y, a.z};
            B.nmax[i] = {b.x, b.y, b.z};
            memcpy(&B.first[i], &a.w, 4);
            memcpy(&B.count[i], &b.w, 4);
        }
    });
    B.pool_ptr = (uint32_t)slots - 1;          // highest slot in use
    B.max_depth = h_depth;
    return true;
}

}  // namespace mrt
