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
//   2. The tree's shape by clustering along the Morton order (ploc_* below): merges by smallest surface area, every
//      cluster looking kRadius places to either side.  (lbvh_tree, the binary radix tree of Karras 2012 that splits the
//      order at its highest differing bit, is kept for comparison: its trees rendered the default scene at half the
//      speed of the reference's tree, the clustered ones render it one and a half times as fast.)
//   3. Bounds bottom-up: one thread per triangle starts at its leaf; at every internal node the
//      second thread to arrive merges the two child boxes and moves on.  The same pass prices every subtree by the
//      surface area heuristic -- as a leaf (area x triangles) and as an inner node (area x kInnerCost + the children's
//      prices) -- and marks it as a leaf where that is cheaper and it holds at most kLeafMax triangles.
//   4. An internal node is kept when neither it nor any of its ancestors is such a leaf; the kept nodes are numbered
//      by a prefix sum, and one thread per kept node writes its two children into slots 3+2q and 4+2q (q = its number;
//      slots 1 and 2 stay unused as in the reference layout), so the node array holds no slot that is not reached.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>

#include <string>
#include <vector>

#include <algorithm>
#include <mutex>

#include "bvh_builder.h"
#include "host_parallel.h"

namespace mrt {

namespace {

constexpr int kLeafMax = 8;            // most triangles a subtree may hold to be made a leaf
constexpr float kInnerCost = 1.5f;     // price of visiting an inner node (two box tests) in units of one triangle test

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

__device__ __forceinline__ float half_area(const Box6& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// price[node]: cheapest way to render the subtree (surface area heuristic); as_leaf[node]: that way is one leaf
__global__ __launch_bounds__(256) void lbvh_bounds(const Tri9* tris, const uint32_t* vals, int n, const uint32_t* left, const uint32_t* right,
                                                   const int* parent_internal, const int* parent_leaf, const int* range_lo, const int* range_hi,
                                                   Box6* leaf_box, Box6* node_box, unsigned int* arrived, float* price, uint32_t* as_leaf,
                                                   int leaf_max, float inner_cost) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    leaf_box[k] = tri_box(tris[vals[k]]);
    __threadfence();
    int node = parent_leaf[k];
    while (node >= 0) {
        if (atomicAdd(&arrived[node], 1u) == 0u) return;           // the sibling's thread will come by later and do the merge
        __threadfence();
        const uint32_t l = left[node], r = right[node];
        const Box6 a = (l & 0x80000000u) ? leaf_box[l & 0x7fffffffu] : node_box[l];
        const Box6 b = (r & 0x80000000u) ? leaf_box[r & 0x7fffffffu] : node_box[r];
        Box6 m;
        for (int x = 0; x < 3; ++x) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
        node_box[node] = m;
        const float pl = (l & 0x80000000u) ? half_area(a) : price[l], pr = (r & 0x80000000u) ? half_area(b) : price[r];
        const int size = range_hi[node] - range_lo[node] + 1;
        const float area = half_area(m), leaf_price = area * (float)size, inner_price = area * inner_cost + pl + pr;
        const bool leaf = size <= leaf_max && !(inner_price < leaf_price);
        price[node] = leaf ? leaf_price : inner_price;
        as_leaf[node] = leaf ? 1u : 0u;
        __threadfence();
        node = parent_internal[node];
    }
}

// kept[i] = 1 when internal node i appears in the tree that is written out: neither it nor an ancestor is a leaf.
// A kept node with a child that is a leaf reports that leaf's depth (kept nodes on the way to the root).
__global__ __launch_bounds__(256) void lbvh_keep(int n, const uint32_t* left, const uint32_t* right, const int* parent_internal, const uint32_t* as_leaf,
                                                 uint32_t* kept, int* max_depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    bool keep = true;
    int depth = 0;
    for (int node = i; node >= 0; node = parent_internal[node]) { keep = keep && !as_leaf[node]; ++depth; }
    kept[i] = keep ? 1u : 0u;
    const uint32_t l = left[i], r = right[i];
    const bool leaf_child = (l & 0x80000000u) || as_leaf[l] || (r & 0x80000000u) || as_leaf[r];
    if (keep && leaf_child) atomicMax(max_depth, depth);
}

__device__ __forceinline__ void write_slot(float4* nodes, uint32_t slot, const Box6& b, uint32_t first, uint32_t count) {
    nodes[2 * (size_t)slot] = make_float4(b.lo[0], b.lo[1], b.lo[2], __uint_as_float(first));
    nodes[2 * (size_t)slot + 1] = make_float4(b.hi[0], b.hi[1], b.hi[2], __uint_as_float(count));
}

// number[i]: how many kept nodes come before node i (exclusive prefix sum of kept)
__device__ __forceinline__ void emit_child(float4* nodes, uint32_t slot, uint32_t ref, const Box6* leaf_box, const Box6* node_box,
                                           const int* range_lo, const int* range_hi, const uint32_t* as_leaf, const uint32_t* number) {
    if (ref & 0x80000000u) {
        const uint32_t k = ref & 0x7fffffffu;
        write_slot(nodes, slot, leaf_box[k], k, 1u);
    } else {
        const int size = range_hi[ref] - range_lo[ref] + 1;
        if (as_leaf[ref]) write_slot(nodes, slot, node_box[ref], (uint32_t)range_lo[ref], (uint32_t)size);
        else write_slot(nodes, slot, node_box[ref], 3u + 2u * number[ref], 0u);
    }
}

__global__ __launch_bounds__(256) void lbvh_emit(int n, const uint32_t* left, const uint32_t* right, const Box6* leaf_box, const Box6* node_box,
                                                 const int* range_lo, const int* range_hi, const uint32_t* as_leaf, const uint32_t* kept,
                                                 const uint32_t* number, float4* nodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    if (i == 0) {
        if (as_leaf[0]) write_slot(nodes, 0u, node_box[0], 0u, (uint32_t)n);
        else write_slot(nodes, 0u, node_box[0], 3u, 0u);           // (the root is the first kept node: number 0)
    }
    if (!kept[i]) return;
    emit_child(nodes, 3u + 2u * number[i], left[i], leaf_box, node_box, range_lo, range_hi, as_leaf, number);
    emit_child(nodes, 4u + 2u * number[i], right[i], leaf_box, node_box, range_lo, range_hi, as_leaf, number);
}

// ---------------------------------------------------------------------------------------------------------------
// The tree's shape by parallel locally-ordered clustering (Meister and Bittner 2017) instead of by splitting the Morton
// order at its highest differing bit: clusters, at first the triangles in Morton order, are merged bottom-up; in every
// round each cluster looks kRadius places to either side for the neighbour with which it would have the smallest box
// (surface area), the pairs that chose each other become a new node in the place of the earlier of the two, and the
// array is compacted.  The globally cheapest pair always chooses each other (ties go to the lower positions), so every
// round merges, n - 1 merges in all.  Splitting by Morton bits gives bloated boxes wherever large and small triangles
// lie side by side (long walls between small steps); merging by area keeps the small ones together and puts the
// large ones high up.  Ids: 0 ... n-1 the triangles by Morton position, n ... 2n-2 the nodes in the order they are made.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRadius = 100;

__global__ __launch_bounds__(256) void ploc_init(const Tri9* tris, const uint32_t* vals, uint32_t n, Box6* box, uint32_t* cid, uint32_t* size) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    box[p] = tri_box(tris[vals[p]]);
    cid[p] = p;
    size[p] = 1u;
}

__global__ __launch_bounds__(256) void ploc_nearest(uint32_t c, int radius, const uint32_t* cid, const Box6* box, uint32_t* nn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const Box6 a = box[cid[i]];
    const uint32_t lo = i > (uint32_t)radius ? i - (uint32_t)radius : 0u, hi = i + (uint32_t)radius < c ? i + (uint32_t)radius : c - 1u;
    float best = INFINITY;
    uint32_t best_j = i == lo ? hi : lo;       // (c >= 2: some neighbour exists)
    uint32_t best_key_lo = 0xffffffffu;
    for (uint32_t j = lo; j <= hi; ++j) {
        if (j == i) continue;
        const Box6 b = box[cid[j]];
        Box6 m;
        for (int x = 0; x < 3; ++x) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
        const float area = half_area(m);
        // pairs are ordered by (area, lower position, higher position): the same order seen from both sides
        const uint32_t key_lo = i < j ? i : j;
        if (area < best || (area == best && (key_lo < best_key_lo || (key_lo == best_key_lo && j < best_j)))) { best = area; best_j = j; best_key_lo = key_lo; }
    }
    nn[i] = best_j;
}

// flags[i]: low word 1 = position i survives the round (alone, or as the place of a new node), high word 1 = a node is made there
__global__ __launch_bounds__(256) void ploc_mark(uint32_t c, const uint32_t* nn, unsigned long long* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > c) return;
    if (i == c) { flags[c] = 0ull; return; }       // (one more element: its prefix sum holds the totals)
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    flags[i] = (mutual && j < i) ? 0ull : (mutual ? (1ull << 32) | 1ull : 1ull);
}

__global__ __launch_bounds__(256) void ploc_merge(uint32_t c, uint32_t next_id, const uint32_t* cid, const uint32_t* nn, const unsigned long long* flags,
                                                  const unsigned long long* sums, Box6* box, uint32_t* child_l, uint32_t* child_r, uint32_t* size,
                                                  uint32_t* parent, uint32_t n, uint32_t* cid_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const unsigned long long f = flags[i];
    if (!(f & 1ull)) return;
    const uint32_t at = (uint32_t)(sums[i] & 0xffffffffull);
    if (f >> 32) {
        const uint32_t id = next_id + (uint32_t)(sums[i] >> 32), a = cid[i], b = cid[nn[i]];
        const Box6 ba = box[a], bb = box[b];
        Box6 m;
        for (int x = 0; x < 3; ++x) { m.lo[x] = fminf(ba.lo[x], bb.lo[x]); m.hi[x] = fmaxf(ba.hi[x], bb.hi[x]); }
        box[id] = m;
        child_l[id - n] = a; child_r[id - n] = b;
        size[id] = size[a] + size[b];
        parent[a] = id; parent[b] = id;
        cid_out[at] = id;
    } else {
        cid_out[at] = cid[i];
    }
}

// Where the subtree of `id` starts in an order of the triangles in which every subtree is one run: the sizes of the
// left siblings of the nodes on the way to the root, added up.
__device__ __forceinline__ uint32_t ploc_start(uint32_t id, uint32_t root, uint32_t n, const uint32_t* parent, const uint32_t* child_l, const uint32_t* size) {
    uint32_t at = 0;
    while (id != root) {
        const uint32_t up = parent[id], l = child_l[up - n];
        if (l != id) at += size[l];
        id = up;
    }
    return at;
}

// The clustered tree in the arrays the rest of the build reads (those lbvh_tree makes for the Morton tree): node id ->
// internal node number 2n-2 - id (the root, made last, is number 0); a triangle's reference is its place in the new order.
__global__ __launch_bounds__(256) void ploc_order(uint32_t n, const uint32_t* vals, const uint32_t* parent, const uint32_t* child_l, const uint32_t* size,
                                                  uint32_t* vals_out, uint32_t* place, int* parent_leaf) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t root = 2u * n - 2u, at = ploc_start(p, root, n, parent, child_l, size);
    vals_out[at] = vals[p];
    place[p] = at;
    parent_leaf[at] = (int)(root - parent[p]);
}
__global__ __launch_bounds__(256) void ploc_links(uint32_t n, const uint32_t* parent, const uint32_t* child_l, const uint32_t* child_r, const uint32_t* size,
                                                  const uint32_t* place, uint32_t* left, uint32_t* right, int* range_lo, int* range_hi, int* parent_internal) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n - 1u) return;
    const uint32_t root = 2u * n - 2u, id = n + k, q = root - id;
    const uint32_t at = ploc_start(id, root, n, parent, child_l, size);
    range_lo[q] = (int)at; range_hi[q] = (int)(at + size[id] - 1u);
    const uint32_t l = child_l[k], r = child_r[k];
    left[q] = l < n ? (0x80000000u | place[l]) : root - l;
    right[q] = r < n ? (0x80000000u | place[r]) : root - r;
    parent_internal[q] = id == root ? -1 : (int)(root - parent[id]);
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
// err set on a HIP failure.  bvh.tris must not be empty.  inner_cost: the price of an inner node in triangle tests
// (0: kInnerCost); the higher, the larger the leaves and the fewer the nodes.
bool build_lbvh_device(HostBVH& B, hipStream_t stream, std::string& err, float inner_cost_arg) {
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
    Dev<float> price;
    Dev<uint32_t> as_leaf, kept, number;
    Dev<float4> nodes;
    Dev<unsigned char> temp;
    LBVH_HIP(d_tris.alloc(n)); LBVH_HIP(k0.alloc(n)); LBVH_HIP(k1.alloc(n)); LBVH_HIP(v0.alloc(n)); LBVH_HIP(v1.alloc(n));
    LBVH_HIP(left.alloc(n)); LBVH_HIP(right.alloc(n)); LBVH_HIP(rlo.alloc(n)); LBVH_HIP(rhi.alloc(n));
    LBVH_HIP(pint.alloc(n)); LBVH_HIP(pleaf.alloc(n)); LBVH_HIP(depth.alloc(1));
    LBVH_HIP(leaf_box.alloc(n)); LBVH_HIP(node_box.alloc(n)); LBVH_HIP(arrived.alloc(n)); LBVH_HIP(nodes.alloc(2 * slots));
    LBVH_HIP(price.alloc(n)); LBVH_HIP(as_leaf.alloc(n)); LBVH_HIP(kept.alloc(n + 1)); LBVH_HIP(number.alloc(n + 1));
    LBVH_HIP(hipMemsetAsync(kept.p, 0, ((size_t)n + 1) * sizeof(uint32_t), stream));
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
    const int leaf_max = kLeafMax;
    const float inner_cost = inner_cost_arg > 0.0f ? inner_cost_arg : kInnerCost;
    uint32_t n_kept = 0;
    constexpr bool morton_tree = false;         // true: the tree by Morton bits (lbvh_tree) instead of by clustering, for comparison
    uint32_t* order = v1.p;             // triangle ids in the tree's leaf order
    if (n > 1 && !morton_tree) {
        const int radius = kRadius;
        Dev<Box6> box;
        Dev<uint32_t> cid_a, cid_b, nn, child_l, child_r, size, parent, place;
        Dev<unsigned long long> flags, sums;
        Dev<unsigned char> scan_temp;
        LBVH_HIP(box.alloc(2 * (size_t)n)); LBVH_HIP(cid_a.alloc(n)); LBVH_HIP(cid_b.alloc(n)); LBVH_HIP(nn.alloc(n)); LBVH_HIP(child_l.alloc(n));
        LBVH_HIP(child_r.alloc(n)); LBVH_HIP(size.alloc(2 * (size_t)n)); LBVH_HIP(parent.alloc(2 * (size_t)n)); LBVH_HIP(place.alloc(n));
        LBVH_HIP(flags.alloc((size_t)n + 1)); LBVH_HIP(sums.alloc((size_t)n + 1));
        size_t scan_bytes = 0;
        LBVH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flags.p, sums.p, (int)n + 1, stream));
        LBVH_HIP(scan_temp.alloc(scan_bytes));
        hipLaunchKernelGGL(ploc_init, grid, block, 0, stream, d_tris.p, v1.p, n, box.p, cid_a.p, size.p);
        uint32_t c = n, next_id = n;
        uint32_t *cid = cid_a.p, *cid_next = cid_b.p;
        while (c > 1) {
            const dim3 g((c + 256) / 256);
            hipLaunchKernelGGL(ploc_nearest, g, block, 0, stream, c, radius, (const uint32_t*)cid, (const Box6*)box.p, nn.p);
            hipLaunchKernelGGL(ploc_mark, g, block, 0, stream, c, (const uint32_t*)nn.p, flags.p);
            LBVH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, flags.p, sums.p, (int)c + 1, stream));
            hipLaunchKernelGGL(ploc_merge, g, block, 0, stream, c, next_id, (const uint32_t*)cid, (const uint32_t*)nn.p, (const unsigned long long*)flags.p,
                               (const unsigned long long*)sums.p, box.p, child_l.p, child_r.p, size.p, parent.p, n, cid_next);
            unsigned long long totals = 0;
            LBVH_HIP(hipMemcpyAsync(&totals, sums.p + c, sizeof totals, hipMemcpyDeviceToHost, stream));
            LBVH_HIP(hipStreamSynchronize(stream));
            const uint32_t survivors = (uint32_t)(totals & 0xffffffffull), made = (uint32_t)(totals >> 32);
            if (made == 0 || survivors + made != c) { err = "clustering made no progress"; return false; }
            c = survivors; next_id += made;
            std::swap(cid, cid_next);
        }
        if (next_id != 2 * n - 1) { err = "clustering ended with the wrong number of nodes"; return false; }
        hipLaunchKernelGGL(ploc_order, grid, block, 0, stream, n, (const uint32_t*)v1.p, (const uint32_t*)parent.p, (const uint32_t*)child_l.p,
                           (const uint32_t*)size.p, v0.p, place.p, pleaf.p);
        hipLaunchKernelGGL(ploc_links, grid, block, 0, stream, n, (const uint32_t*)parent.p, (const uint32_t*)child_l.p, (const uint32_t*)child_r.p,
                           (const uint32_t*)size.p, (const uint32_t*)place.p, left.p, right.p, rlo.p, rhi.p, pint.p);
        LBVH_HIP(hipGetLastError());
        LBVH_HIP(hipStreamSynchronize(stream));            // (the clustering's arrays go out of scope)
        order = v0.p;
    }
    if (n > 1) {
        if (morton_tree) hipLaunchKernelGGL(lbvh_tree, grid, block, 0, stream, k1.p, (int)n, left.p, right.p, rlo.p, rhi.p, pint.p, pleaf.p);
        hipLaunchKernelGGL(lbvh_bounds, grid, block, 0, stream, d_tris.p, order, (int)n, left.p, right.p, pint.p, pleaf.p, rlo.p, rhi.p, leaf_box.p,
                           node_box.p, arrived.p, price.p, as_leaf.p, leaf_max, inner_cost);
        hipLaunchKernelGGL(lbvh_keep, grid, block, 0, stream, (int)n, left.p, right.p, pint.p, as_leaf.p, kept.p, depth.p);
        size_t scan_bytes = 0;
        LBVH_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, kept.p, number.p, (int)n + 1, stream));
        Dev<unsigned char> scan_temp;
        LBVH_HIP(scan_temp.alloc(scan_bytes));
        LBVH_HIP(hipcub::DeviceScan::ExclusiveSum(scan_temp.p, scan_bytes, kept.p, number.p, (int)n + 1, stream));
        hipLaunchKernelGGL(lbvh_emit, grid, block, 0, stream, (int)n, left.p, right.p, leaf_box.p, node_box.p, rlo.p, rhi.p, as_leaf.p, kept.p,
                           number.p, nodes.p);
        LBVH_HIP(hipGetLastError());
        LBVH_HIP(hipMemcpyAsync(&n_kept, number.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));      // the sum's last element: all kept nodes
        LBVH_HIP(hipStreamSynchronize(stream));        // (scan_temp goes out of scope)
    }
    LBVH_HIP(hipGetLastError());
    const size_t used_slots = 3 + 2 * (size_t)n_kept;         // slots 0 .. 2 + 2 * kept
    B.idx.resize(n);
    std::vector<float4> h_nodes(2 * slots);
    int h_depth = 0;
    LBVH_HIP(hipMemcpyAsync(B.idx.data(), order, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    LBVH_HIP(hipMemcpyAsync(h_nodes.data(), nodes.p, 2 * used_slots * sizeof(float4), hipMemcpyDeviceToHost, stream));
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
    B.pool_ptr = (uint32_t)used_slots - 1;     // highest slot in use
    B.max_depth = h_depth;
    return true;
}

}  // namespace mrt
