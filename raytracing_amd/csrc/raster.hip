// Rasterizer path: Rasterizer::Render / RenderMesh (rasterizer.cpp:53-184) as a tiled pipeline.
//
//   raster_gather     at upload and after a refit: the corners, texture coordinates and face normal of every triangle
//                     gathered into arrays that set-up reads 16 bytes per lane, neighbouring lanes neighbouring words.
//   raster_setup      one thread per triangle: vertex transform, back-face cull, Sutherland-Hodgman
//                     clip against the near and left planes, projection, palette level, per-edge
//                     constants and the screen-space plane equations of 1/z, u/z and v/z
//                     (rasterizer.cpp:100-141); it also counts the triangle into the 32x32-pixel
//                     screen tiles its pixel bounding box overlaps.
//   raster_count_big, raster_tile_sums, raster_tile_scan, raster_fill
//                     exclusive scan of the tile counts, then the tiles' triangle lists are filled (a
//                     triangle overlapping more than 16 tiles is counted and filled by a whole wave,
//                     lanes striding over its tiles).  The add that counts a triangle into a tile also tells its
//                     place in the tile's list, so filling is a plain store per entry.
//   raster_tile       one workgroup per tile, depth cells of the tile in registers.  A wave owns a 32x8 strip
//                     of the tile and walks the tile's whole list; a lane owns one pixel in each of the
//                     strip's four 8x8 blocks.  A pixel is covered exactly when the reference's scan
//                     conversion would plot it (rows y0 < y <= y1 of the spanning edges, columns
//                     xl < x <= xr of the row's extent, rasterizer.cpp:143-182), and the depth test keeps
//                     the minimum of {orderable 1/z, triangle index}: the reference draws triangles in
//                     order and skips a pixel when the new 1/z is >= the stored one, i.e. the smallest
//                     1/z wins and, among equals, the first triangle drawn.  Every lane then resolves
//                     its own pixels: texel fetch and palette lookup of the winner
//                     (rasterizer.cpp:178-180), optional 1/z and triangle-id planes.
//
// The reference interpolates along edges and spans by repeated addition; here the same affine
// functions are evaluated directly, so results differ only where a pixel centre lies within
// rounding distance of an edge, a depth tie or a texel border.
#include "raster.h"
#include "host_parallel.h"

#include <algorithm>
#include <mutex>

namespace mrt {

constexpr int kInlineMeshes = 16;
constexpr uint32_t kTotalsGap = 64;       // words between the tile counts and the frame's totals
struct RasterFrame {
    float M[16];             // inverse camera matrix (world -> view), column-major
    float frustum[2][4];     // clip planes 0 (near) and 1 (left): rasterizer.cpp:37-38
    int W, H;
    // per mesh: texture index, or -1 when the mesh is culled this frame or untextured.  A scene of few meshes carries
    // the table in the kernel's arguments (n_inline > 0) and needs no copy of it per frame.
    int n_inline;
    int mesh_tex_inline[kInlineMeshes];
};

constexpr uint32_t kUncovered = 0xff000000u;

__device__ __forceinline__ unsigned int orderable(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(unsigned int k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// vec3(M * vec4(p, 1)) with glm's operation order (detail/type_mat4x4.inl:676-687)
__device__ __forceinline__ f3 xform(const float* M, f3 p) {
    const f4 c0 = {M[0], M[1], M[2], M[3]}, c1 = {M[4], M[5], M[6], M[7]}, c2 = {M[8], M[9], M[10], M[11]}, c3 = {M[12], M[13], M[14], M[15]};
    const f4 a0 = c0 * p.x + c1 * p.y;
    const f4 a1 = c2 * p.z + c3 * 1.0f;
    const f4 r = a0 + a1;
    return {r.x, r.y, r.z};
}

constexpr int kTile = 32, kTileShift = 5;
constexpr int kSmallTiles = 16;          // a triangle overlapping more tiles than this is binned by a whole wave
constexpr int kBigWaves = 1024;

struct TileRange { int tx0, ty0, tw, th; };
__device__ __forceinline__ int box_lo(uint32_t packed) { return (int)(packed & 0xffffu); }
__device__ __forceinline__ int box_hi(uint32_t packed) { return (int)(packed >> 16); }
__device__ __forceinline__ TileRange tile_range(const RasterTri& T) {
    TileRange r;
    r.tx0 = box_lo(T.box_x) >> kTileShift; r.ty0 = box_lo(T.box_y) >> kTileShift;
    r.tw = (box_hi(T.box_x) >> kTileShift) - r.tx0 + 1; r.th = (box_hi(T.box_y) >> kTileShift) - r.ty0 + 1;
    return r;
}
// Depth class of a triangle for the order of the tile lists: view-space z at the middle of its bounding box,
// kDepthClasses equal steps from the nearest to the farthest corner of the visible meshes (class 0 nearest).
// Only the order in which a tile meets its triangles depends on it, never a pixel.
constexpr int kDepthClasses = 8;
__device__ __forceinline__ int depth_class(const RasterTri& T, float z_near, float z_scale) {
    const float cx = 0.5f * (float)(box_lo(T.box_x) + box_hi(T.box_x)), cy = 0.5f * (float)(box_lo(T.box_y) + box_hi(T.box_y));
    const float z = 1.0f / (T.zA * cx + T.zB * cy + T.zC);              // negative in front of the camera
    const float c = (z_near - z) * z_scale;
    return c >= 0.0f ? min((int)c, kDepthClasses - 1) : 0;              // a NaN goes to class 0
}
// What binning needs of a triangle, 8 bytes instead of the 128-byte set-up record:
// lo = first tile column | first tile row << 12 | depth class << 24 | drawn << 27, hi = tile columns | tile rows << 12
__device__ __forceinline__ uint2 pack_bin(const TileRange& r, int cls) {
    return make_uint2((uint32_t)r.tx0 | ((uint32_t)r.ty0 << 12) | ((uint32_t)cls << 24) | (1u << 27), (uint32_t)r.tw | ((uint32_t)r.th << 12));
}
__device__ __forceinline__ bool bin_drawn(uint2 b) { return (b.x >> 27) & 1u; }
__device__ __forceinline__ uint32_t bin_class(uint2 b) { return (b.x >> 24) & 7u; }
__device__ __forceinline__ TileRange bin_range(uint2 b) {
    return TileRange{(int)(b.x & 0xfffu), (int)((b.x >> 12) & 0xfffu), (int)(b.y & 0xfffu), (int)((b.y >> 12) & 0xfffu)};
}

// A convex polygon in view space with its texture coordinates.  A triangle gains at most one corner per clip
// plane and two planes are applied (the near plane and one side plane: rasterizer.cpp:119 stops at p < 2).
struct ClipPoly {
    static constexpr int kMax = 8;
    f3 p[kMax];
    f2 t[kMax];
    int n;
    __device__ __forceinline__ void add(f3 pos, f2 uv) { if (n < kMax) { p[n] = pos; t[n] = uv; ++n; } }
};
// The part of `in` on the inner side of the plane dot(normal, x) = offset (rasterizer.cpp:117-133).  The polygon is
// walked edge by edge; an edge that crosses the plane contributes the crossing point A + s (B - A) with
// s = da / (da - db), an edge that ends inside contributes its end corner -- in that order, which is the
// reference's, so the corner order of the result (and with it the vertex triple the attribute planes are fitted
// to further down) is the reference's too.  "Inside" is da >= 0 and "outside" da < 0 as two separate tests: a
// NaN distance is neither, and such an edge contributes nothing.
__device__ __forceinline__ ClipPoly clip_against(const ClipPoly& in, f3 normal, float offset) {
    ClipPoly out;
    out.n = 0;
    for (int j = 0; j < in.n; ++j) {
        const int k = j + 1 == in.n ? 0 : j + 1;
        const float da = dot(normal, in.p[j]) - offset, db = dot(normal, in.p[k]) - offset;
        const bool a_in = da >= 0, a_out = da < 0, b_in = db >= 0, b_out = db < 0;
        if ((a_out && b_in) || (a_in && b_out)) {
            const float s = da / (da - db);
            out.add(in.p[j] + s * (in.p[k] - in.p[j]), in.t[j] + s * (in.t[k] - in.t[j]));
        }
        if (b_in && (a_in || a_out)) out.add(in.p[k], in.t[k]);
    }
    return out;
}

// Second half of the set-up of one polygon (3 to NMAX corners, in the clip's output order): palette level, screen
// coordinates, attribute planes, pixel box and the edge table, written into the record T (which lives in LDS: its
// edge table is filled at a running index; the caller has set T.info = 0, "not drawn", and it stays so on every early return).  NMAX = 3 is the triangle no clip plane touched: every index is then a
// compile-time constant and the corner arrays stay in registers; the five-corner form indexes them at run time and
// the compiler keeps them in private memory, which only the lanes with a clipped triangle then touch.
template <int NMAX>
__device__ __forceinline__ void setup_polygon(const f3* p, const f2* t, int nin, f3 Nt, int texture, int W, int H, RasterTri& T) {
    const uint32_t level = (uint32_t)(int)(gmax(0.0f, Nt.z) * 31);    // rasterizer.cpp:136
    float sx[NMAX], sy[NMAX], iz[NMAX], uz[NMAX], vz[NMAX];
    float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
    for (int v = 0; v < NMAX; ++v) {
        if (v >= nin) break;
        const f3 q = p[v];
        sx[v] = ((q.x * W) / -q.z) + W / 2;                         // :140-141
        sy[v] = ((q.y * W) / q.z) + H / 2;
        iz[v] = 1.0f / q.z;                                        // :151
        uz[v] = t[v].x * iz[v];                                    // :152-153
        vz[v] = t[v].y * iz[v];
        xmin = fminf(xmin, sx[v]); xmax = fmaxf(xmax, sx[v]);
        ymin = fminf(ymin, sy[v]); ymax = fmaxf(ymax, sy[v]);
    }
    // affine attribute planes through the polygon's widest vertex triple
    int b = 1, c = 2;
    if (NMAX > 3) {
        float best = 0.0f;
        for (int j = 1; j < nin; ++j)
            for (int k = j + 1; k < nin; ++k) {
                const float ar = fabsf((sx[j] - sx[0]) * (sy[k] - sy[0]) - (sx[k] - sx[0]) * (sy[j] - sy[0]));
                if (ar > best) { best = ar; b = j; c = k; }
            }
    }
    const double x1 = (double)sx[b] - sx[0], y1 = (double)sy[b] - sy[0], x2 = (double)sx[c] - sx[0], y2 = (double)sy[c] - sy[0];
    const double det = x1 * y2 - x2 * y1;
    if (!(det != 0.0 && xmax >= 0.0f && ymax >= 1.0f && xmin < (float)W && ymin < (float)H)) return;
    auto plane = [&](const float* a, float& A, float& B, float& C) {
        const double d1 = (double)a[b] - a[0], d2 = (double)a[c] - a[0];
        const double pa = (d1 * y2 - d2 * y1) / det, pb = (x1 * d2 - x2 * d1) / det;
        A = (float)pa; B = (float)pb; C = (float)((double)a[0] - pa * sx[0] - pb * sy[0]);
    };
    float A, B, C;
    plane(iz, A, B, C); T.zA = A; T.zB = B; T.zC = C;
    plane(uz, A, B, C); T.uA = A; T.uB = B; T.uC = C;
    plane(vz, A, B, C); T.vA = A; T.vB = B; T.vC = C;
    // rows max(1,(int)y0+1) .. min(H-2,(int)y1), columns (int)x0+1 .. min(W-2,(int)x1): :149, :170
    const int by0 = max(1, (int)fmaxf(ymin, -1.0f) + 1), by1 = min(H - 2, (int)fminf(ymax, (float)H));
    const int bx0 = max(0, (int)fmaxf(xmin, -1.0f) + 1), bx1 = min(W - 2, (int)fminf(xmax, (float)W));
    if (!(by0 <= by1 && bx0 <= bx1)) return;
    T.box_x = (uint32_t)bx0 | ((uint32_t)bx1 << 16); T.box_y = (uint32_t)by0 | ((uint32_t)by1 << 16);
    int ne = 0;
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {                               // rasterizer.cpp:143-151
        if (j >= nin) break;
        const int j2 = (j + 1 == nin) ? 0 : j + 1;
        float xa = sx[j], ya = sy[j], xb = sx[j2], yb = sy[j2];       // the edge, oriented top to bottom
        if (ya > yb) { float h = xa; xa = xb; xb = h; h = ya; ya = yb; yb = h; }
        if ((ya == yb) || (ya >= H) || (yb < 1)) continue;
        const float rydiff = 1.0f / (yb - ya);
        const int r0 = max(1, (int)ya + 1), r1 = min(H - 2, (int)yb);
        if (r0 > r1) continue;
        T.edge[ne] = RasterTri::Edge{xa, ya, (xb - xa) * rydiff, (uint32_t)r0 | ((uint32_t)r1 << 16)};
        ++ne;
    }
    T.info = (uint32_t)ne | (level << 8) | ((uint32_t)texture << 16);
}
__device__ __forceinline__ bool record_drawn(uint32_t info) { return (info & 0xffu) != 0u; }

// One thread per place i of the set-up order: the corners, texture coordinates, face normal and mesh of triangle
// order[i] out of the indexed vertex arrays into the five arrays raster_setup reads (RasterState::d_in_a ...).
__global__ __launch_bounds__(256) void raster_gather(const float* pos, const float* uv, const float* N, const int32_t* tri, const int32_t* tri_mesh,
                                                      const uint32_t* order, uint32_t n_tris, uint4* in_a, uint4* in_b, uint4* in_c, uint4* in_u, uint4* in_v) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const uint32_t t = order[i];
    const int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
    auto w = [](float f) { return __float_as_uint(f); };
    in_a[i] = make_uint4(w(pos[3 * v0]), w(pos[3 * v0 + 1]), w(pos[3 * v0 + 2]), w(pos[3 * v1]));
    in_b[i] = make_uint4(w(pos[3 * v1 + 1]), w(pos[3 * v1 + 2]), w(pos[3 * v2]), w(pos[3 * v2 + 1]));
    in_c[i] = make_uint4(w(pos[3 * v2 + 2]), w(N[3 * t]), w(N[3 * t + 1]), w(N[3 * t + 2]));
    in_u[i] = make_uint4(w(uv[2 * v0]), w(uv[2 * v0 + 1]), w(uv[2 * v1]), w(uv[2 * v1 + 1]));
    in_v[i] = make_uint4(w(uv[2 * v2]), w(uv[2 * v2 + 1]), (uint32_t)tri_mesh[t], t);
}

constexpr uint32_t kSetupSlots = 224;      // records a workgroup of raster_setup stages at a time
constexpr uint32_t kRecWordsU = sizeof(RasterTri) / 4, kInfoWord = offsetof(RasterTri, info) / 4;
__global__ __launch_bounds__(256) void raster_setup(RasterFrame F, const uint4* __restrict__ in_a, const uint4* __restrict__ in_b, const uint4* __restrict__ in_c,
                                                     const uint4* __restrict__ in_u, const uint4* __restrict__ in_v,
                                                     const int32_t* __restrict__ mesh_tex, RasterTri* out, uint32_t n_tris, uint2* bins, int tiles_x,
                                                     uint32_t* tile_count, uint32_t* big, uint32_t* big_count, float z_near, float z_scale) {
    // Thread i sets up the triangle at place i of the set-up order (the tree's leaf order: neighbours in the order are
    // neighbours in space, so the counters a wave adds to lie in few memory lines); its record goes to out[triangle id],
    // its bin record to bins[i].
    // The 128-byte records are built in LDS and written from there 16 bytes per lane, eight lanes a record, instead
    // of each thread scattering 32 words.  Only triangles that face the eye get a staging slot (the others are in no
    // tile's list, nobody reads their record and it is not written: only their bin record says so), and a workgroup
    // stages kSetupSlots records at a time: with about half of a scene's triangles facing away one round is the rule.
    // A workgroup with more front-facing triangles than slots takes another round.
    // (The kernel's 112 vector registers allow four waves per SIMD; held to 96 for five, it spills and the frame takes 15 us longer.)
    __shared__ __align__(16) uint32_t stage[kSetupSlots * kRecWordsU];
    __shared__ uint32_t slot_tri[kSetupSlots];         // id of the triangle a slot holds
    // Triangles a clip plane cuts take the long way (a polygon whose corner count is only known at run time, kept
    // in private memory).  They are few and scattered over the waves, so they are set aside and worked off
    // afterwards by the first threads of the workgroup, side by side in as few waves as they fill.
    __shared__ uint32_t cut_list[256];                 // owner thread | slot << 16
    __shared__ uint32_t cut_count;
    __shared__ uint32_t wave_front[4];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = i < n_tris;
    const int W = F.W, H = F.H;
    struct Corners { f3 p0, p1, p2, Nt; };
    auto fl = [](uint32_t u) { return __uint_as_float(u); };
    auto corners = [&](uint32_t k) {
        const uint4 a = in_a[k], b = in_b[k], cc = in_c[k];
        Corners c;
        m4 M;
        memcpy(&M, F.M, 64);
        c.Nt = mul3(M, f3{fl(cc.y), fl(cc.z), fl(cc.w)});                        // rasterizer.cpp:112
        c.p0 = xform(F.M, f3{fl(a.x), fl(a.y), fl(a.z)});                        // rasterizer.cpp:100
        c.p1 = xform(F.M, f3{fl(a.w), fl(b.x), fl(b.y)});
        c.p2 = xform(F.M, f3{fl(b.z), fl(b.w), fl(cc.x)});
        return c;
    };
    auto texture_of = [&](uint32_t mesh) { return F.n_inline > 0 ? F.mesh_tex_inline[mesh] : mesh_tex[mesh]; };
    const Corners c = corners(in_range ? i : 0u);
    bool front = in_range && !(dot(c.p0, c.Nt) > 0);                           // back-face cull, :113
    bool untouched = true;
    for (int plane = 0; plane < 2; ++plane) {      // every corner on the inner side of both planes (a NaN distance is not)
        const f3 n = {F.frustum[plane][0], F.frustum[plane][1], F.frustum[plane][2]};
        const float o = F.frustum[plane][3];
        untouched = untouched && (dot(n, c.p0) - o >= 0) && (dot(n, c.p1) - o >= 0) && (dot(n, c.p2) - o >= 0);
    }
    // texture coordinates, mesh and id: of the triangles that face the eye only
    uint4 tu = make_uint4(0u, 0u, 0u, 0u), tv = make_uint4(0u, 0u, 0u, 0u);
    int texture = -1;                                        // -1 when the mesh is culled or has no texture
    if (front) {
        tu = in_u[i]; tv = in_v[i];
        texture = texture_of(tv.z);
        front = texture >= 0;
    }
    // rank of this thread's triangle among the workgroup's front-facing ones
    const unsigned long long fmask = __ballot(front);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wave_front[wave] = (uint32_t)__popcll(fmask);
    if (threadIdx.x == 0) cut_count = 0;
    __syncthreads();
    uint32_t rank = (uint32_t)__popcll(fmask & ((1ull << lane) - 1ull)), n_front = 0;
    for (uint32_t w = 0; w < 4u; ++w) { const uint32_t x = wave_front[w]; n_front += x; if (w < wave) rank += x; }
    uint2 bin = make_uint2(0u, 0u);
    bool is_big = false;
    int cls = 0;
    TileRange r = {0, 0, 0, 0};
    for (uint32_t r0 = 0; r0 < n_front; r0 += kSetupSlots) {
        const bool mine = front && rank >= r0 && rank < r0 + kSetupSlots;
        const uint32_t slot = rank - r0;
        if (mine) {
            RasterTri& T = reinterpret_cast<RasterTri*>(stage)[slot];
            T.info = 0;
            slot_tri[slot] = tv.w;
            if (untouched) {
                // what the two clips make of such a triangle: each hands the corners on rotated by one (clip_against)
                const f3 q[3] = {c.p2, c.p0, c.p1};
                const f2 tq[3] = {f2{fl(tv.x), fl(tv.y)}, f2{fl(tu.x), fl(tu.y)}, f2{fl(tu.z), fl(tu.w)}};
                setup_polygon<3>(q, tq, 3, c.Nt, texture, W, H, T);
            } else {
                cut_list[atomicAdd(&cut_count, 1u)] = threadIdx.x | (slot << 16);
            }
        }
        __syncthreads();
        if (threadIdx.x < cut_count) {
            const uint32_t e = cut_list[threadIdx.x], owner = e & 0xffffu, k = blockIdx.x * blockDim.x + owner;
            const Corners cc = corners(k);
            const uint4 ku = in_u[k], kv = in_v[k];
            ClipPoly poly;
            poly.n = 0;
            poly.add(cc.p0, f2{fl(ku.x), fl(ku.y)}); poly.add(cc.p1, f2{fl(ku.z), fl(ku.w)}); poly.add(cc.p2, f2{fl(kv.x), fl(kv.y)});
            for (int plane = 0; plane < 2; ++plane)
                poly = clip_against(poly, f3{F.frustum[plane][0], F.frustum[plane][1], F.frustum[plane][2]}, F.frustum[plane][3]);
            if (poly.n >= 3 && poly.n <= 5)
                setup_polygon<5>(poly.p, poly.t, poly.n, cc.Nt, texture_of(kv.z), W, H, reinterpret_cast<RasterTri*>(stage)[e >> 16]);
        }
        __syncthreads();
        {
            const uint32_t n_here = min(kSetupSlots, n_front - r0);
            for (uint32_t d = threadIdx.x; d < n_here * 8u; d += 256u) {
                const uint32_t s = d >> 3, part = d & 7u;
                if (record_drawn(stage[s * kRecWordsU + kInfoWord]))
                    reinterpret_cast<uint4*>(out + slot_tri[s])[part] = reinterpret_cast<const uint4*>(stage)[s * 8u + part];
            }
        }
        if (mine) {
            // binning, first pass: the triangle's tile range for the fill pass
            const RasterTri& T = reinterpret_cast<const RasterTri*>(stage)[slot];
            if (record_drawn(T.info)) {
                r = tile_range(T);
                cls = depth_class(T, z_near, z_scale);
                bin = pack_bin(r, cls);
                is_big = r.tw * r.th > kSmallTiles;
            }
        }
        if (r0 + kSetupSlots < n_front) {      // another round: the slots are used again
            __syncthreads();
            if (threadIdx.x == 0) cut_count = 0;
            __syncthreads();
        }
    }
    if (!in_range) return;
    bins[i] = bin;
    if (!bin_drawn(bin)) return;
    // and the tiles' triangle counts (adds that hand nothing back: the wave does not wait for them)
    if (is_big) { big[atomicAdd(big_count, 1u)] = i; return; }
    for (int j = 0; j < r.th; ++j)
        for (int k = 0; k < r.tw; ++k) atomicAdd(&tile_count[((r.ty0 + j) * tiles_x + r.tx0 + k) * kDepthClasses + cls], 1u);
}

// Texel fetch and palette lookup of the depth winners of a lane's NB pixels (rasterizer.cpp:178-180), and the optional
// 1/z and triangle-id planes.  A pixel's colour is four loads deep (the winner's record, its texture's descriptor, the
// texel, the palette entry); the NB pixels go through each of the four steps side by side, so a lane waits for four
// round trips to memory and not for 4 x NB.
// at[b]: index into the W-pitched z and id planes; at_fb[b]: index into fb (the caller's pitch when fb is the caller's
// own frame, in which case pixels nothing covers are left as they are: keep_uncovered); inside[b]: the pixel exists.
template <int NB>
__device__ __forceinline__ void resolve_pixels(const RasterTri* __restrict__ setup, const unsigned long long (&key)[NB], const RasterTextures& tx,
                                               const int (&x)[NB], int y, const bool (&inside)[NB], const size_t (&at)[NB], const size_t (&at_fb)[NB],
                                               uint32_t* fb, bool keep_uncovered, float* zbuf, int32_t* ids) {
    bool hit[NB];
    uint4 ra[NB], rb[NB];           // record words 4..7 {zC, info, uA, uB} and 8..11 {uC, vA, vB, vC}
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t t = (uint32_t)key[b];
        hit[b] = inside[b] && t != 0xffffffffu;
        const uint4* rec = reinterpret_cast<const uint4*>(setup + (hit[b] ? t : 0u));
        ra[b] = hit[b] ? rec[1] : make_uint4(0u, 0u, 0u, 0u);
        rb[b] = hit[b] ? rec[2] : make_uint4(0u, 0u, 0u, 0u);
    }
    int4 td[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) td[b] = hit[b] ? tx.tex[ra[b].y >> 16] : make_int4(0, 1, 1, 0);
    uint32_t texel[NB];
    float z0[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        z0[b] = from_orderable((unsigned int)(key[b] >> 32));
        const float xf = (float)x[b], yf = (float)y;
        const float u0 = __uint_as_float(ra[b].z) * xf + __uint_as_float(ra[b].w) * yf + __uint_as_float(rb[b].x);
        const float v0 = __uint_as_float(rb[b].y) * xf + __uint_as_float(rb[b].z) * yf + __uint_as_float(rb[b].w);
        const float tw = (float)td[b].y, th = (float)td[b].z;
        const int umask = td[b].y - 1, vmask = td[b].z - 1;
        const float z = 1.0f / z0[b];
        const int u = (int)(u0 * z * tw) & umask, v = (int)(v0 * z * th) & vmask;
        texel[b] = hit[b] ? tx.texels[(size_t)td[b].x + (size_t)(u + v * (umask + 1))] : 0u;
    }
    uint32_t colour[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint32_t texture = ra[b].y >> 16, level = (ra[b].y >> 8) & 0xffu;
        colour[b] = hit[b] ? tx.pal[(size_t)texture * (32 * 256) + (size_t)level * 256 + texel[b]] : kUncovered;   // palette entries are 0x00RRGGBB, so this value marks "nothing drawn here"
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (!inside[b]) continue;
        if (hit[b] || !keep_uncovered) fb[at_fb[b]] = colour[b];
        if (zbuf) zbuf[at[b]] = hit[b] ? z0[b] : 0.0f;
        if (ids) ids[at[b]] = hit[b] ? (int32_t)(uint32_t)key[b] : -1;
    }
}

// ---------------------------------------------------------------------------------------
// Tiled path: triangles are binned to 32x32-pixel screen tiles, and one workgroup per tile keeps
// the tile's depth cells on chip (in the registers of the lanes that own them), so the depth test
// needs no device-scope atomic in memory, and resolves its own pixels.  The result is the minimum over
// {1/z, triangle index} per pixel, whatever the order in which a tile meets its triangles.
// ---------------------------------------------------------------------------------------

// Binning in two passes.  Counting: raster_setup itself counts the triangles of every tile (adds that hand nothing
// back) and lists the big triangles, raster_count_big counts those.  Filling, after the scan has turned counts into list
// starts and cleared them: raster_fill writes the lists (the count is now the running fill position per segment).
// A tile's list is kDepthClasses consecutive segments, nearest class first (counts, starts and fill positions are
// kept per (tile, class)).  The tile kernel walks the list front to back, so near triangles lower the blocks'
// depth bounds before the far ones are met, most of which are then skipped.  The order changes no pixel.
// `capacity`: entries the list buffer holds.  A frame that needs more is found out afterwards and rendered again
// after the buffer has grown, so entries past the end are only kept from being written.
// (Taking the fill positions from the counting adds, so that filling needs no second add per entry, was tried: adds
// that hand their result back cost the waves of raster_setup, few per CU, twice the time they saved raster_fill.)
__device__ __forceinline__ uint32_t bin_slot(const TileRange& r, int k, int tiles_x, uint32_t cls) {
    return (uint32_t)((r.ty0 + k / r.tw) * tiles_x + r.tx0 + k % r.tw) * kDepthClasses + cls;
}
// triangles that overlap many tiles, one wave each, the lanes striding over the triangle's tiles: the counting pass
__global__ __launch_bounds__(256) void raster_count_big(const uint2* bins, int tiles_x, uint32_t* tile_count, const uint32_t* big, const uint32_t* big_count) {
    const uint32_t n_big = *big_count;
    const int lane = threadIdx.x & 63;
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n_big; i += kBigWaves) {
        const uint2 b = bins[big[i]];
        const TileRange r = bin_range(b);
        const uint32_t cls = bin_class(b);
        for (int k = lane; k < r.tw * r.th; k += 64) atomicAdd(&tile_count[bin_slot(r, k, tiles_x, cls)], 1u);
    }
}
// The second pass, one launch: the first kBigWaves / 4 workgroups take the big triangles, one wave each (the longest
// pieces of work start first), the other workgroups the small ones, one thread per triangle.  The adds hand back the
// entry's position and are served on the memory side: a thread asks for four at a time (one that waited for each
// before it asked for the next kept its whole wave waiting through as many round trips as the wave's largest
// triangle has tiles), and the small triangles are dealt to the threads with a stride, because neighbours in the
// set-up order are neighbours on the screen and a wave of neighbours sent most of its adds to the same few words,
// which are then served one after the other (72 us instead of 60 for the 1,000,000-triangle frame).
// (Tried: the scan in one launch, every workgroup adding up the counts before its chunk itself, with the fill pass
// counting the segments' counts down instead of up from cleared ones: the frame took 13 us longer.)
constexpr uint32_t kFillStride = 4099;       // (a prime: eight = thread / 8 * stride mod n visits every eight once when n is no multiple of it)
__global__ __launch_bounds__(256) void raster_fill(const uint2* bins, const uint32_t* order, uint32_t n_tris, int tiles_x, uint32_t* tile_count, const uint32_t* tile_start,
                                                    uint32_t* pairs, const uint32_t* big, const uint32_t* big_count, uint32_t capacity) {
    constexpr uint32_t kBigBlocks = kBigWaves / 4;
    if (blockIdx.x < kBigBlocks) {
        const uint32_t n_big = *big_count;
        const int lane = threadIdx.x & 63;
        for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n_big; i += kBigWaves) {
            const uint32_t place = big[i], t = order[place];
            const uint2 b = bins[place];
            const TileRange r = bin_range(b);
            const uint32_t cls = bin_class(b);
            const int n = r.tw * r.th;
            for (int k0 = lane; k0 < n; k0 += 256) {
                uint32_t slot[4], at[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    slot[u] = bin_slot(r, min(k0 + 64 * u, n - 1), tiles_x, cls);
                    at[u] = k0 + 64 * u < n ? atomicAdd(&tile_count[slot[u]], 1u) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k0 + 64 * u < n) {
                        const uint32_t p = tile_start[slot[u]] + at[u];
                        if (p < capacity) pairs[p] = t;
                    }
            }
        }
        return;
    }
    // (eight consecutive places stay together: their bin records are one 64-byte read)
    const uint32_t thread = (blockIdx.x - kBigBlocks) * blockDim.x + threadIdx.x;
    const uint32_t n_eights = (n_tris + 7u) / 8u, eight = thread >> 3;
    if (eight >= n_eights) return;
    const uint32_t place = (n_eights % kFillStride != 0u ? (uint32_t)(((unsigned long long)eight * kFillStride) % n_eights) : eight) * 8u + (thread & 7u);
    if (place >= n_tris) return;
    const uint2 b = bins[place];
    if (!bin_drawn(b)) return;
    const TileRange r = bin_range(b);
    const int n = r.tw * r.th;
    if (n > kSmallTiles) return;
    const uint32_t t = order[place];
    const uint32_t cls = bin_class(b);
    for (int k0 = 0; k0 < n; k0 += 4) {
        uint32_t slot[4], at[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            slot[u] = bin_slot(r, min(k0 + u, n - 1), tiles_x, cls);
            at[u] = k0 + u < n ? atomicAdd(&tile_count[slot[u]], 1u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < n) {
                const uint32_t p = tile_start[slot[u]] + at[u];
                if (p < capacity) pairs[p] = t;
            }
    }
}
// Exclusive scan of the list-segment counts in two small launches: raster_tile_sums adds up chunks of kScanChunk
// consecutive counts (one workgroup each), raster_tile_scan lets every workgroup add the sums of the chunks before
// its own and scan its chunk.  The counts are cleared on the way: they become the fill cursors of the second
// binning pass.  tile_start[n] and total[0] receive the grand total.  (One workgroup scanning everything took
// 28 us for the 65,280 segments of a 4K frame.)
constexpr uint32_t kScanChunk = 2048;        // 256 threads x 8 counts
__global__ __launch_bounds__(256) void raster_tile_sums(const uint32_t* tile_count, uint32_t n, uint32_t* chunk_sum) {
    __shared__ uint32_t wave_sum[4];
    const uint32_t first = blockIdx.x * kScanChunk + 8u * threadIdx.x;
    uint32_t mine = 0;
    if (first + 8u <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(tile_count + first), b = *reinterpret_cast<const uint4*>(tile_count + first + 4);
        mine = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
    } else {
        for (uint32_t k = first; k < n; ++k) mine += tile_count[k];
    }
    for (int o = 32; o > 0; o >>= 1) mine += (uint32_t)__shfl_xor((int)mine, o, 64);
    if ((threadIdx.x & 63u) == 0u) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) chunk_sum[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}
__global__ __launch_bounds__(256) void raster_tile_scan(uint32_t* tile_count, uint32_t* tile_start, uint32_t n, const uint32_t* chunk_sum,
                                                         uint32_t* total, uint32_t capacity, uint32_t* too_short) {
    __shared__ uint32_t wave_sum[4];
    __shared__ uint32_t before_chunk;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    {   // the chunks before this one
        uint32_t s = 0;
        for (uint32_t k = t; k < blockIdx.x; k += 256u) s += chunk_sum[k];
        for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o, 64);
        if (lane == 0u) wave_sum[wave] = s;
        __syncthreads();
        if (t == 0) before_chunk = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        __syncthreads();
    }
    const uint32_t first = blockIdx.x * kScanChunk + 8u * t;
    uint32_t v[8], mine = 0;
    const bool whole = first + 8u <= n;
    if (whole) {
        const uint4 a = *reinterpret_cast<const uint4*>(tile_count + first), b = *reinterpret_cast<const uint4*>(tile_count + first + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        *reinterpret_cast<uint4*>(tile_count + first) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(tile_count + first + 4) = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = first + k < n ? tile_count[first + k] : 0u;
            if (first + k < n) tile_count[first + k] = 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) mine += v[k];
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < 4u; ++w) { const uint32_t x = wave_sum[w]; all += x; if (w < wave) before += x; }
    uint32_t run = before_chunk + before + incl - mine;
    if (whole) {
        uint4 a, b;
        a.x = run; run += v[0]; a.y = run; run += v[1]; a.z = run; run += v[2]; a.w = run; run += v[3];
        b.x = run; run += v[4]; b.y = run; run += v[5]; b.z = run; run += v[6]; b.w = run; run += v[7];
        *reinterpret_cast<uint4*>(tile_start + first) = a;
        *reinterpret_cast<uint4*>(tile_start + first + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (first + k < n) tile_start[first + k] = run; run += v[k]; }
    }
    if (blockIdx.x == gridDim.x - 1 && t == 0) {
        tile_start[n] = before_chunk + all; total[0] = before_chunk + all;
        // a frame whose lists do not fit the buffer it was queued with leaves the length it needed (the largest of such frames)
        if (before_chunk + all > capacity) atomicMax(too_short, before_chunk + all);
    }
}

#ifdef MRT_LOOP_COUNTS
// Diagnostic build only: what the tile kernel's waves did, summed over the launches since the last read
// (scripts/raster_counts.py): [0] waves, [1] list entries looked through, [2] candidates after the first look,
// [3] candidates still alive when their turn came (drawn), [4] blocks tested, [5] blocks in which a pixel was won.
// clocks (shader cycles, per wave): [8] whole wave, [9] first look at the batches (up to the ballot), [10] drawing, [11] resolve,
// [12] before the first batch (list bounds, first entries and heads); [13] longest list, [14] longest-lived wave's clocks, [15] waves with lists over 628 entries
__device__ unsigned long long g_raster_counts[16];
extern "C" int mrt_debug_raster_counts(unsigned long long* out) {
    unsigned long long zero[16] = {};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_raster_counts), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_raster_counts), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#define RASTER_COUNT(slot, n) (counts[slot] += (unsigned int)(n))
#else
#define RASTER_COUNT(slot, n) ((void)0)
#endif

// word offsets of a RasterTri's fields (the record as 32 consecutive words)
constexpr int kRecWords = (int)(sizeof(RasterTri) / 4);
constexpr int kRecBox = (int)(offsetof(RasterTri, box_x) / 4), kRecZ = (int)(offsetof(RasterTri, zA) / 4), kRecInfo = (int)(offsetof(RasterTri, info) / 4);
constexpr int kRecEdge = (int)(offsetof(RasterTri, edge) / 4);
static_assert(kRecWords == 32 && kRecBox == 0 && kRecZ == 2, "the tile kernel reads a record's first words as {box_x, box_y, zA, zB} and {zC}");

// Maximum of an unsigned value over the 64 lanes of a wave, returned wave-uniform: four DPP steps inside the rows of
// 16 lanes (swap neighbours, swap pairs, mirror the half row, mirror the row), then the four rows are combined from
// scalar reads.  All 64 lanes must be active.
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));     // row_half_mirror
    v = max(v, (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));     // row_mirror
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

// The tile kernel: one workgroup per 32x32 tile, the tile's depth cells in registers.  A wave owns one 32x8 strip of
// the tile (wave w: rows 8w .. 8w+7), a lane one pixel in each of the strip's four 8x8 blocks (the same row, columns
// 8 apart), and every wave walks the tile's WHOLE list: a cell has one owner, so the depth test is a compare and a
// select on the owner's registers (no atomics, no LDS), and the result is the minimum over {1/z, triangle index} per
// pixel.  A lane's four pixels share their row, so the row's extent in a triangle (which columns the reference's scan
// conversion plots, rasterizer.cpp:143-170) is worked out once per triangle.
// Per 8x8 block the wave keeps an upper bound of the depth keys its cells hold (wave-uniform, in scalar registers).
// Cells only ever decrease, so the bound stays valid until it is refreshed; a triangle whose smallest 1/z over the block lies above it cannot win any of the block's pixels (equal 1/z is not
// skipped: the lower triangle index wins ties) and the block is not tested at all.  1/z is evaluated as
// (zA*x + zB*y) + zC with every operation rounded, which is monotone in x and in y, so its minimum over a block is
// taken at the corner the signs of zA and zB pick, in exactly the arithmetic of the per-pixel test.  The lists are sorted nearest class first, so
// the bounds drop early and most far triangles are skipped.
// The list is taken 32 entries at a time, one per lane, to find the triangles that reach into a block of the strip
// whose bound they are not above; those are then drawn one after the other in list order (each block's bound is
// looked at again then, before the triangle's record is asked for).  The record of the triangle being drawn is the
// same for all lanes: it is read by scalar loads.
// (Tried and measured on the 1,000,000-triangle 4K frame, none faster: the record as one vector load with lane i taking
// word i and the fields picked out by v_readlane, with two records on their way ahead; the same with two records per
// load; the records of all of a batch's candidates requested at once and parked in LDS; 8 or 16 waves per tile with
// narrower strips (0.38 and 0.46 ms instead of 0.36).  What did shorten the frame was fewer vector instructions per
// drawn triangle: the scalar loads, and one corner instead of four per block bound.)
// (An earlier form kept the cells in LDS, gave each wave every fourth triangle of the list and tested depth with a
// 64-bit LDS atomicMin: its waves spent more than half their time waiting for those atomics.)
template <int NB>      // 8x8 blocks per wave: 4 (a 32x8 strip, 4 waves per tile), 2 (16x8, 8 waves) or 1 (8x8, 16 waves)
__global__ __launch_bounds__(1024 / NB) void raster_tile(const RasterTri* __restrict__ setup, const uint32_t* __restrict__ tile_start,
                                                         const uint32_t* __restrict__ pairs, int tiles_x,
                                                         RasterTextures tx, uint32_t* fb, int fb_pitch, int keep_uncovered, float* zbuf, int32_t* ids,
                                                         int W, int H, uint32_t capacity) {
    // list entries looked through at a time, one per lane of the lower half of the wave: with 64 more of them are dismissed
    // when their turn comes, because the bounds they were looked through against are older by then
    constexpr uint32_t kBatch = 32;
    const int tile = blockIdx.x, px0 = (tile % tiles_x) << kTileShift, py0 = (tile / tiles_x) << kTileShift;
    const int lane = threadIdx.x & 63, wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int kPerRow = 4 / NB;                                                  // waves side by side in a row of blocks
    const int sy0 = py0 + ((wave / kPerRow) << 3);                                   // the wave's first row
    const int sx0 = px0 + (wave % kPerRow) * (NB << 3);                              // and its first column
    const int lx = lane & 7, y = sy0 + (lane >> 3);                                  // this lane's row
    const float yf = (float)y, xf0 = (float)(sx0 + lx);          // (sums of small integers: exact in float)
    const unsigned long long empty = ((unsigned long long)orderable(0.0f) << 32) | 0xffffffffull;   // the cleared z-buffer, no triangle
    unsigned long long k0 = empty, k1 = empty, k2 = empty, k3 = empty;              // this lane's cells at columns lx, lx + 8, lx + 16, lx + 24
    unsigned int m0 = orderable(0.0f), m1 = m0, m2 = m0, m3 = m0;                   // wave-uniform bounds of the four blocks
    const uint32_t begin = tile_start[tile * kDepthClasses], end = min(tile_start[(tile + 1) * kDepthClasses], capacity);
    const float ya = (float)sy0, yb = (float)(sy0 + 7);
    // The list entries and what the 64-wide pass needs of their records (pixel box, 1/z plane) are read ahead: the
    // entries of the batch after the next and the records of the next batch are requested before the current batch
    // is looked through and drawn.
    struct Head { int x0, x1, y0, y1; float zA, zB, zC; };
    auto entry = [&](uint32_t base) -> uint32_t { return (uint32_t)lane < kBatch && base + lane < end ? pairs[base + lane] : 0xffffffffu; };
    auto head = [&](uint32_t t) -> Head {
        Head h = {1, 0, 1, 0, 0.0f, 0.0f, 0.0f};
        if (t != 0xffffffffu) {         // two loads: the record's first 16 bytes and the word behind them
            const uint4 a = *reinterpret_cast<const uint4*>(setup + t);
            const float zc = setup[t].zC;
            h = Head{box_lo(a.x), box_hi(a.x), box_lo(a.y), box_hi(a.y), __uint_as_float(a.z), __uint_as_float(a.w), zc};
        }
        return h;
    };
#ifdef MRT_LOOP_COUNTS
    unsigned int counts[16] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const unsigned long long clk_enter = __builtin_readcyclecounter();
    unsigned long long clk_mark = clk_enter;
#define RASTER_CLOCK(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); counts[slot] += (unsigned int)(now_ - clk_mark); clk_mark = now_; } while (0)
#else
#define RASTER_CLOCK(slot) ((void)0)
#endif
    uint32_t t_cur = entry(begin), t_next = entry(begin + kBatch);
    Head h_cur = head(t_cur);
    RASTER_CLOCK(12);
    for (uint32_t base = begin; base < end; base += kBatch) {
        const uint32_t t_after = entry(base + 2 * kBatch);
        const Head h_next = head(t_next);
        const uint32_t t_lane = t_cur;
        // zk0..zk3: per block of the strip, the orderable form of the entry's smallest 1/z over the block (taken at a
        // corner, in the arithmetic of the per-pixel test); 0 when that is a NaN (a NaN decides nothing: never above a
        // bound), all ones when the entry does not reach the block (above every bound).  The entry can still win a
        // pixel of block b while zk_b <= the block's bound: asked now of all entries at once, and again of each
        // candidate when its turn comes, against the bounds as they are then.
        unsigned int zk0 = ~0u, zk1 = ~0u, zk2 = ~0u, zk3 = ~0u;
        {
            const Head& R = h_cur;
            if (t_lane != 0xffffffffu && R.y0 <= sy0 + 7 && R.y1 >= sy0 && R.x0 <= sx0 + 8 * NB - 1 && R.x1 >= sx0) {
                const int bx0 = max(R.x0, sx0) >> 3, bx1 = min(R.x1, sx0 + 8 * NB - 1) >> 3;
                // (zA*x + zB*y) + zC with every operation rounded grows with x when zA > 0 and falls when zA < 0, and likewise
                // in y: its smallest value over the block is taken at the corner the two signs pick
                const float zy = R.zB * (R.zB < 0.0f ? yb : ya);
                const int x_off = R.zA < 0.0f ? 7 : 0;
                auto key = [&](int bi) -> unsigned int {
                    const int bx = (sx0 >> 3) + bi;
                    if (bx < bx0 || bx > bx1) return ~0u;
                    const float zmin = R.zA * (float)((bx << 3) + x_off) + zy + R.zC;
                    return zmin == zmin ? orderable(zmin) : 0u;
                };
                zk0 = key(0);
                if constexpr (NB > 1) zk1 = key(1);
                if constexpr (NB > 2) { zk2 = key(2); zk3 = key(3); }
            }
        }
        const bool cand = zk0 <= m0 || zk1 <= m1 || zk2 <= m2 || zk3 <= m3;
        unsigned long long todo = __ballot(cand);
        RASTER_COUNT(1, __popcll(__ballot(t_lane != 0xffffffffu)));
        RASTER_COUNT(2, __popcll(todo));
        RASTER_CLOCK(9);
        // draws triangle t (the entry of lane j) into this wave's cells.  t is the same in every lane, so the record's
        // fields are read by scalar loads: they arrive in scalar registers and cost the vector unit nothing.
        auto draw = [&](const int j) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)t_lane, j);
            const uint32_t* __restrict__ rec = reinterpret_cast<const uint32_t*>(setup + t);
            auto wi = [&](int w) { return (int)rec[w]; };
            auto wf = [&](int w) { return __uint_as_float(rec[w]); };
            // which of the wave's blocks can the triangle still win a pixel in, now?
            const unsigned int blocks = ((unsigned int)__builtin_amdgcn_readlane((int)zk0, j) <= m0 ? 0x1u : 0u) |
                                        ((unsigned int)__builtin_amdgcn_readlane((int)zk1, j) <= m1 ? 0x10u : 0u) |
                                        ((unsigned int)__builtin_amdgcn_readlane((int)zk2, j) <= m2 ? 0x100u : 0u) |
                                        ((unsigned int)__builtin_amdgcn_readlane((int)zk3, j) <= m3 ? 0x1000u : 0u);
            if (blocks == 0) return;        // the bounds have dropped below the triangle since the batch was looked through
            RASTER_COUNT(3, 1);
            RASTER_COUNT(4, __popc(blocks));
            const uint32_t bxw = (uint32_t)wi(kRecBox), byw = (uint32_t)wi(kRecBox + 1);
            const int tx0 = box_lo(bxw), tx1 = box_hi(bxw), ty0 = box_lo(byw), ty1 = box_hi(byw);
            const float zA = wf(kRecZ), zB = wf(kRecZ + 1), zC = wf(kRecZ + 2);
            const int n_edges = wi(kRecInfo) & 0xff;
            // This lane's row: columns ix0 .. ix1 are plotted (none when ix0 > ix1): every edge contributes to the rows it
            // spans, the row's extent is [min x, max x] over those edges, columns (int)xl+1 .. min(W-2, (int)xr).  Without branches
            // on what differs between lanes: an edge that does not span the row leaves the extent alone.
            float xl = (float)(W - 1), xr = 0.0f;
            bool any = false;
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                if (e >= n_edges) break;
                const uint32_t rows = (uint32_t)wi(kRecEdge + 4 * e + 3);
                const bool spans = y >= box_lo(rows) && y <= box_hi(rows);
                const float xe = wf(kRecEdge + 4 * e) + wf(kRecEdge + 4 * e + 2) * (yf - wf(kRecEdge + 4 * e + 1));
                xl = (spans && xe < xl) ? xe : xl;
                xr = (spans && xe > xr) ? xe : xr;
                any = any || spans;
            }
            const bool row = any && y >= ty0 && y <= ty1;
            const int ix0 = row ? max((int)xl + 1, tx0) : 1, ix1 = row ? min(min(W - 2, (int)xr), tx1) : 0;
            const float zrow = zB * yf;
            auto block = [&](int bi, unsigned long long& k, unsigned int& bm) {
                if (!((blocks >> (4 * bi)) & 1u)) return;
                const int x = sx0 + (bi << 3) + lx;
                const float z0 = zA * (xf0 + (float)(bi << 3)) + zrow + zC;
                // the z-buffer is cleared to 0 and the test is "z0 >= zbuf: skip" (:55, :177); a negative z0 has its sign bit set
                const unsigned long long key = ((unsigned long long)~__float_as_uint(z0) << 32) | t;
                const bool won = x >= ix0 && x <= ix1 && z0 < 0.0f && key < k;
                k = won ? key : k;
                if (__builtin_amdgcn_ballot_w64(won) != 0ull) { bm = wave_max_u32((unsigned int)(k >> 32)); RASTER_COUNT(5, 1); }
            };
            block(0, k0, m0);
            if constexpr (NB > 1) block(1, k1, m1);
            if constexpr (NB > 2) { block(2, k2, m2); block(3, k3, m3); }
        };
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            draw(j);
        }
        t_cur = t_next; t_next = t_after; h_cur = h_next;
        RASTER_CLOCK(10);
    }
    {
        unsigned long long keys[NB];
        int xs[NB];
        bool inside[NB];
        size_t at[NB], at_fb[NB];
        keys[0] = k0;
        if constexpr (NB > 1) keys[1] = k1;
        if constexpr (NB > 2) { keys[2] = k2; keys[3] = k3; }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            xs[b] = sx0 + (b << 3) + lx;
            inside[b] = xs[b] < W && y < H;
            at[b] = (size_t)y * W + xs[b];
            at_fb[b] = (size_t)y * fb_pitch + xs[b];
        }
        resolve_pixels<NB>(setup, keys, tx, xs, y, inside, at, at_fb, fb, keep_uncovered != 0, zbuf, ids);
    }
#ifdef MRT_LOOP_COUNTS
    __builtin_amdgcn_s_waitcnt(0);
    RASTER_CLOCK(11);
    counts[8] = (unsigned int)(__builtin_readcyclecounter() - clk_enter);
    for (int k = 0; k < 13; ++k)
        if (lane == k && k != 6 && k != 7) atomicAdd(&g_raster_counts[k], (unsigned long long)counts[k]);
    if (lane == 6) atomicMax(&g_raster_counts[6], (unsigned long long)__builtin_readcyclecounter());     // the last wave's end
    if (lane == 7) atomicMax(&g_raster_counts[7], ~clk_enter);                                            // the first wave's start, complemented
    if (lane == 13) atomicMax(&g_raster_counts[13], (unsigned long long)(end - begin));       // the longest list
    if (lane == 14) atomicMax(&g_raster_counts[14], (unsigned long long)counts[8]);           // the longest-lived wave
    if (lane == 15 && end - begin > 628u) atomicAdd(&g_raster_counts[15], 1ull);              // waves with more than four times the mean list
#endif
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
#define RASTER_HIP(call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return MRT_E_HIP; } \
    } while (0)

void raster_release(RasterState& R) {
    for (void* p : {(void*)R.d_pos, (void*)R.d_uv, (void*)R.d_N, (void*)R.d_tri, (void*)R.d_tri_mesh, (void*)R.d_mesh_tex, (void*)R.d_setup, (void*)R.d_bins,
                    (void*)R.d_in_a, (void*)R.d_in_b, (void*)R.d_in_c, (void*)R.d_in_u, (void*)R.d_in_v, (void*)R.d_order, (void*)R.d_short,
                    (void*)R.d_pairs, (void*)R.d_chunk_base, (void*)R.d_tile_count, (void*)R.d_tile_start, (void*)R.d_scan_sums,
                    (void*)R.d_fb, (void*)R.d_z, (void*)R.d_ids})
        if (p) (void)hipFree(p);
    if (R.h_stage) (void)hipHostFree(R.h_stage);
    if (R.h_short) (void)hipHostFree((void*)R.h_short);
    if (R.frame_done) (void)hipEventDestroy(R.frame_done);
    R = RasterState();
}

static int gather(RasterState& R, hipStream_t stream, std::string& err) {
    if (R.n_tris == 0) return MRT_OK;
    hipLaunchKernelGGL(raster_gather, dim3((R.n_tris + 255) / 256), dim3(256), 0, stream, R.d_pos, R.d_uv, R.d_N, R.d_tri, R.d_tri_mesh, R.d_order, R.n_tris,
                       R.d_in_a, R.d_in_b, R.d_in_c, R.d_in_u, R.d_in_v);
    RASTER_HIP(hipGetLastError());
    return MRT_OK;
}

int raster_upload(RasterState& R, const mrt_mesh* meshes, int n_meshes, const uint32_t* order, hipStream_t stream, std::string& err) {
    raster_release(R);
    std::vector<float> pos, uv, N;
    std::vector<int32_t> tri;
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        RasterMeshHost h;
        if (M.texture_id > kRasterMaxTexture) { err = "raster_upload: texture index above " + std::to_string(kRasterMaxTexture); return MRT_E_LIMIT; }
        h.verts = M.verts; h.tris = M.tris; h.texture = M.texture_id;
        h.vert_off = (uint32_t)(pos.size() / 3); h.tri_off = (uint32_t)(tri.size() / 3);
        h.bounds[0] = {1e30f, 1e30f, 1e30f}; h.bounds[1] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < M.verts; ++i) {                                  // scenedata.cpp:258-263
            const float* p = M.pos + 3 * i;
            h.bounds[0] = {gmin(h.bounds[0].x, p[0]), gmin(h.bounds[0].y, p[1]), gmin(h.bounds[0].z, p[2])};
            h.bounds[1] = {gmax(h.bounds[1].x, p[0]), gmax(h.bounds[1].y, p[1]), gmax(h.bounds[1].z, p[2])};
        }
        pos.insert(pos.end(), M.pos, M.pos + 3 * (size_t)M.verts);
        uv.insert(uv.end(), M.uv, M.uv + 2 * (size_t)M.verts);
        N.insert(N.end(), M.tri_normal, M.tri_normal + 3 * (size_t)M.tris);
        for (int i = 0; i < 3 * M.tris; ++i) tri.push_back(M.tri[i] + (int32_t)h.vert_off);
        for (int i = 0; i < M.tris; ++i) R.h_tri_mesh.push_back(m);
        R.meshes.push_back(h);
    }
    R.n_verts = (uint32_t)(pos.size() / 3);
    R.n_tris = (uint32_t)(tri.size() / 3);
    if (R.n_tris == 0) return MRT_OK;
    RASTER_HIP(hipMalloc((void**)&R.d_pos, pos.size() * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_uv, uv.size() * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_N, N.size() * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_tri, tri.size() * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_tri_mesh, (size_t)R.n_tris * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_mesh_tex, R.meshes.size() * 4));
    RASTER_HIP(hipMemcpyAsync(R.d_tri_mesh, R.h_tri_mesh.data(), (size_t)R.n_tris * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMalloc((void**)&R.d_setup, (size_t)R.n_tris * sizeof(RasterTri)));
    RASTER_HIP(hipMalloc((void**)&R.d_bins, (size_t)R.n_tris * sizeof(uint2)));
    RASTER_HIP(hipMalloc((void**)&R.d_chunk_base, (size_t)R.n_tris * 4));
    RASTER_HIP(hipMemcpyAsync(R.d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_uv, uv.data(), uv.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_N, N.data(), N.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_tri, tri.data(), tri.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMalloc((void**)&R.d_in_a, (size_t)R.n_tris * sizeof(uint4)));
    RASTER_HIP(hipMalloc((void**)&R.d_in_b, (size_t)R.n_tris * sizeof(uint4)));
    RASTER_HIP(hipMalloc((void**)&R.d_in_c, (size_t)R.n_tris * sizeof(uint4)));
    RASTER_HIP(hipMalloc((void**)&R.d_in_u, (size_t)R.n_tris * sizeof(uint4)));
    RASTER_HIP(hipMalloc((void**)&R.d_in_v, (size_t)R.n_tris * sizeof(uint4)));
    RASTER_HIP(hipMalloc((void**)&R.d_order, (size_t)R.n_tris * 4));
    std::vector<uint32_t> as_given;
    if (!order) {
        as_given.resize(R.n_tris);
        for (uint32_t i = 0; i < R.n_tris; ++i) as_given[i] = i;
        order = as_given.data();
    }
    RASTER_HIP(hipMemcpyAsync(R.d_order, order, (size_t)R.n_tris * 4, hipMemcpyHostToDevice, stream));
    if (int rc = gather(R, stream, err)) return rc;
    // Room for the tile lists: four entries per triangle and then some.  A frame that needs more is found out
    // afterwards (raster_wait, or the end of a frame that comes back to the host) and the buffer grows then.
    R.pair_capacity = 4 * (size_t)R.n_tris + 65536;
    RASTER_HIP(hipMalloc((void**)&R.d_pairs, R.pair_capacity * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_short, 256));
    RASTER_HIP(hipMemsetAsync(R.d_short, 0, 256, stream));
    RASTER_HIP(hipHostMalloc((void**)&R.h_short, 64, hipHostMallocDefault));
    *R.h_short = 0;
    RASTER_HIP(hipEventCreateWithFlags(&R.frame_done, hipEventDisableTiming));
    RASTER_HIP(hipStreamSynchronize(stream));      // the staging vectors go out of scope
    return MRT_OK;
}

int raster_refit(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err) {
    if ((size_t)n_meshes != R.meshes.size()) { err = "raster_refit: another number of meshes"; return MRT_E_ARG; }
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        RasterMeshHost& h = R.meshes[m];
        if (M.verts != h.verts || M.tris != h.tris) { err = "raster_refit: a mesh has changed its size"; return MRT_E_ARG; }
        if (M.verts == 0) continue;
        // scenedata.cpp:258-263, a few threads for a large mesh
        std::mutex mu;
        f3 lo = {1e30f, 1e30f, 1e30f}, hi = {-1e30f, -1e30f, -1e30f};
        parallel_for((size_t)M.verts, [&](size_t b, size_t e) {
            f3 l = {1e30f, 1e30f, 1e30f}, u = {-1e30f, -1e30f, -1e30f};
            for (size_t i = b; i < e; ++i) {
                const float* p = M.pos + 3 * i;
                l = {gmin(l.x, p[0]), gmin(l.y, p[1]), gmin(l.z, p[2])};
                u = {gmax(u.x, p[0]), gmax(u.y, p[1]), gmax(u.z, p[2])};
            }
            std::lock_guard<std::mutex> lock(mu);
            lo = {gmin(lo.x, l.x), gmin(lo.y, l.y), gmin(lo.z, l.z)};
            hi = {gmax(hi.x, u.x), gmax(hi.y, u.y), gmax(hi.z, u.z)};
        }, (size_t)1 << 16);
        h.bounds[0] = lo; h.bounds[1] = hi;
        if (R.n_tris == 0) continue;
        RASTER_HIP(hipMemcpyAsync(R.d_pos + 3 * (size_t)h.vert_off, M.pos, 12 * (size_t)M.verts, hipMemcpyHostToDevice, stream));
        RASTER_HIP(hipMemcpyAsync(R.d_uv + 2 * (size_t)h.vert_off, M.uv, 8 * (size_t)M.verts, hipMemcpyHostToDevice, stream));
        if (M.tris > 0) RASTER_HIP(hipMemcpyAsync(R.d_N + 3 * (size_t)h.tri_off, M.tri_normal, 12 * (size_t)M.tris, hipMemcpyHostToDevice, stream));
    }
    return gather(R, stream, err);
}

// The five planes Rasterizer::Init sets up for a W x H target (rasterizer.cpp:31-41): the near plane z = -0.2 and
// the four sides of the pyramid from the eye through the corners of the drawable pixel area
// [0.5, W - 1.5] x [0.5, H - 1.5], left, top, right, bottom.  A side through the eye and the directions
// a = (ax, ay, 1), b = (bx, by, 1) has the normal cross(a, b - a) = (-(by - ay), bx - ax, ax (by - ay) - (bx - ax) ay):
// with a zero third component of b - a the general cross product reduces to this without a change in rounding.
static void frustum_planes(int W, int H, f4 out[5]) {
    const float lo_x = 0.5f, hi_x = W - 1.5f, lo_y = 0.5f, hi_y = H - 1.5f;
    auto dir = [&](float px, float py) { return f2{((px - W / 2) * -1.0f) / W, ((py - H / 2) * -1.0f) / W}; };
    const f2 corner[4] = {dir(lo_x, lo_y), dir(hi_x, lo_y), dir(hi_x, hi_y), dir(lo_x, hi_y)};
    auto side = [](f2 a, f2 b) {
        const float dx = b.x - a.x, dy = b.y - a.y;
        const f3 n = normalize(f3{a.y * 0.0f - dy * 1.0f, 1.0f * dx - 0.0f * a.x, a.x * dy - dx * a.y});
        return f4{n.x, n.y, n.z, 0};
    };
    out[0] = {0, 0, -1, 0.2f};
    for (int k = 0; k < 4; ++k) out[1 + k] = side(corner[k], corner[(k + 3) % 4]);     // from corner k back to the one before it
}

// The tile lists' buffer after a frame has been found short of room: `needed` entries and a quarter more.  Nothing
// may be in flight on the rasterizer's arrays (the callers have waited).
static int grow_pairs(RasterState& R, uint32_t needed, std::string& err) {
    if (R.d_pairs) (void)hipFree(R.d_pairs);
    R.d_pairs = nullptr; R.pair_capacity = 0;
    const size_t cap = (size_t)needed + needed / 4 + 1024;
    RASTER_HIP(hipMalloc((void**)&R.d_pairs, cap * 4));
    R.pair_capacity = cap;
    RASTER_HIP(hipMemset(R.d_short, 0, 4));
    *R.h_short = 0;
    return MRT_OK;
}

int raster_wait(RasterState& R, std::string& err) {
    if (R.frame_queued) RASTER_HIP(hipEventSynchronize(R.frame_done));
    R.frame_queued = false;
    bool was_short = R.short_seen;
    R.short_seen = false;
    if (R.h_short && *R.h_short > R.pair_capacity) {
        was_short = true;
        if (int rc = grow_pairs(R, *R.h_short, err)) return rc;
    }
    if (was_short) {
        err = "a rasterized frame needed longer tile lists than the buffer held and is incomplete; the buffer has grown: render the frame again";
        return MRT_E_LIMIT;
    }
    return MRT_OK;
}

// device_out: fb, zbuf and tri_ids are DEVICE memory and are written by the kernels directly (covered pixels only for
// fb, like a page-locked host frame); the frame is queued and the call returns (raster_wait).
int raster_render(RasterState& R, const RasterTextures& tx, int W, int H, const float cam16[16], uint32_t* fb, int pitch_px,
                  float* zbuf, int32_t* tri_ids, bool device_out, hipStream_t stream, std::string& err) {
    const size_t npx = (size_t)W * H;
    if (W > kRasterMaxDim || H > kRasterMaxDim) { err = "rasterizer: frames up to " + std::to_string(kRasterMaxDim) + " pixels a side"; return MRT_E_LIMIT; }
    // an earlier frame is known to have been short of list room: the buffer grows before this frame is queued
    if (R.h_short && *R.h_short > R.pair_capacity) {
        R.short_seen = true;
        if (R.frame_queued) RASTER_HIP(hipEventSynchronize(R.frame_done));
        RASTER_HIP(hipStreamSynchronize(stream));
        if (int rc = grow_pairs(R, *R.h_short, err)) return rc;
    }
    if (!device_out && R.frame_px != npx) {
        for (void* p : {(void*)R.d_fb, (void*)R.d_z, (void*)R.d_ids}) if (p) (void)hipFree(p);
        R.d_fb = nullptr; R.d_z = nullptr; R.d_ids = nullptr;
        RASTER_HIP(hipMalloc((void**)&R.d_fb, npx * 4));
        RASTER_HIP(hipMalloc((void**)&R.d_z, npx * 4));
        RASTER_HIP(hipMalloc((void**)&R.d_ids, npx * 4));
        R.frame_px = npx;
    }
    m4 cam;
    memcpy(&cam, cam16, 64);
    const m4 inv = inverse(cam);                                              // rasterizer.cpp:57
    // rasterizer.cpp:65-72: scene root -> OBJ root -> group -> mesh, four identity local transforms
    const m4 I = identity();
    const m4 M = mul(mul(mul(mul(inv, I), I), I), I);
    f4 frustum[5];
    frustum_planes(W, H, frustum);
    // per-mesh frustum cull (rasterizer.cpp:92-98) and the "texture required" rule (:102)
    // (a scene of many meshes copies the table below to the device per frame: the frame queued before this one must have
    // read its copy before the host rewrites it)
    if (R.meshes.size() > (size_t)kInlineMeshes && R.frame_queued) RASTER_HIP(hipEventSynchronize(R.frame_done));
    R.h_mesh_tex.assign(R.meshes.size(), -1);
    float z_hi = -1e30f, z_lo = 1e30f;          // view-space z range of the visible meshes' bounding boxes
    for (size_t m = 0; m < R.meshes.size(); ++m) {
        const RasterMeshHost& h = R.meshes[m];
        bool visible = h.tris > 0 && h.texture >= 0;
        f3 c[8];
        if (visible) {
            for (int i = 0; i < 8; i++) {
                const f4 v = mul(M, f4{h.bounds[i & 1].x, h.bounds[(i >> 1) & 1].y, h.bounds[i >> 2].z, 1});
                c[i] = {v.x, v.y, v.z};
            }
            for (int p = 0; p < 5 && visible; p++) {
                int i;
                for (i = 0; i < 8; i++)
                    if ((dot(f3{frustum[p].x, frustum[p].y, frustum[p].z}, c[i]) - frustum[p].w) > 0) break;
                if (i == 8) visible = false;
            }
        }
        if (visible) {
            R.h_mesh_tex[m] = h.texture;
            for (int i = 0; i < 8; i++) { z_hi = std::max(z_hi, c[i].z); z_lo = std::min(z_lo, c[i].z); }
        }
    }
    // depth classes for the order of the tile lists: equal steps of view-space z (negative in front of the camera)
    // from the nearest to the farthest corner of the visible meshes; only a heuristic for the order
    const float z_near = std::min(z_hi, -0.2f);                 // nothing nearer than the near plane is drawn
    const float z_scale = z_lo < z_near ? (float)kDepthClasses / (z_near - z_lo) : 0.0f;
    RasterFrame F;
    memcpy(F.M, &M, 64);
    for (int p = 0; p < 2; ++p) { F.frustum[p][0] = frustum[p].x; F.frustum[p][1] = frustum[p].y; F.frustum[p][2] = frustum[p].z; F.frustum[p][3] = frustum[p].w; }
    F.W = W; F.H = H;
    F.n_inline = R.h_mesh_tex.size() <= (size_t)kInlineMeshes ? (int)R.h_mesh_tex.size() : 0;
    for (int m = 0; m < kInlineMeshes; ++m) F.mesh_tex_inline[m] = m < F.n_inline ? R.h_mesh_tex[(size_t)m] : -1;
    const int tiles_x = (W + kTile - 1) / kTile, tiles_y = (H + kTile - 1) / kTile;
    const uint32_t n_tiles = (uint32_t)(tiles_x * tiles_y);
    const uint32_t n_slots = n_tiles * (uint32_t)kDepthClasses;       // one list segment per (tile, depth class)
    if (R.n_tile_slots < n_slots + 1) {
        for (void* p : {(void*)R.d_tile_count, (void*)R.d_tile_start}) if (p) (void)hipFree(p);
        R.d_tile_count = nullptr; R.d_tile_start = nullptr; R.d_total = nullptr; R.n_tile_slots = 0;
        // the frame's two totals live behind the counts, a cache line away from them (the list atomics keep taking
        // the counts' lines), so that one fill clears both
        RASTER_HIP(hipMalloc((void**)&R.d_tile_count, (size_t)(n_slots + 1 + kTotalsGap + 4) * 4));
        RASTER_HIP(hipMalloc((void**)&R.d_tile_start, (size_t)(n_slots + 1) * 4));
        if (R.d_scan_sums) (void)hipFree(R.d_scan_sums);
        R.d_scan_sums = nullptr;
        RASTER_HIP(hipMalloc((void**)&R.d_scan_sums, (size_t)((n_slots + kScanChunk - 1) / kScanChunk + 1) * 4));
        R.n_tile_slots = n_slots + 1;
    }
    R.d_total = R.d_tile_count + n_slots + 1 + kTotalsGap;
    // A frame in page-locked memory (mrt_host_alloc) is written by the kernel itself, covered pixels only; any
    // other frame comes back through page-locked staging and its covered pixels are merged in on the host
    // (the reference only plots covered pixels, so the rest must stay as the caller had it).
    uint32_t* fb_direct = device_out ? fb : nullptr;
    if (!device_out) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, fb) == hipSuccess && at.type == hipMemoryTypeHost) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, fb, 0) == hipSuccess) fb_direct = static_cast<uint32_t*>(dp);
        }
        (void)hipGetLastError();
    }
    const size_t out_px = device_out ? 0 : npx;        // pixels that come back through the staging buffer, per plane
    const size_t need = out_px * 4 + (tri_ids ? out_px * 4 : 0) + (zbuf ? out_px * 4 : 0);
    if (R.h_stage_bytes < need) {
        if (R.h_stage) (void)hipHostFree(R.h_stage);
        R.h_stage = nullptr; R.h_stage_bytes = 0;
        RASTER_HIP(hipHostMalloc(&R.h_stage, need, hipHostMallocDefault));
        R.h_stage_bytes = need;
    }
    unsigned char* stage = static_cast<unsigned char*>(R.h_stage);
    uint32_t* h_fb = reinterpret_cast<uint32_t*>(stage);
    unsigned char* ids_stage = stage + out_px * 4;
    unsigned char* z_stage = ids_stage + (tri_ids ? out_px * 4 : 0);
    // The list buffers keep their size and the frame is queued without waiting for this frame's list length.  A frame
    // that comes back to the host is looked at when it has arrived, and rendered once more after the buffers have
    // grown; a frame into device memory is not waited for at all (raster_wait tells).
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint32_t capacity = (uint32_t)std::min<size_t>(R.pair_capacity, 0xffffffffu);
        // counts; behind them the frame's totals: [0] list entries, [1] big triangles
        RASTER_HIP(hipMemsetAsync(R.d_tile_count, 0, (size_t)(n_slots + 1 + kTotalsGap + 4) * 4, stream));
        if (R.n_tris > 0) {
            const dim3 per_tri((R.n_tris + 255) / 256), b256(256);
            if (F.n_inline == 0) RASTER_HIP(hipMemcpyAsync(R.d_mesh_tex, R.h_mesh_tex.data(), R.h_mesh_tex.size() * 4, hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(raster_setup, per_tri, b256, 0, stream, F, R.d_in_a, R.d_in_b, R.d_in_c, R.d_in_u, R.d_in_v, R.d_mesh_tex, R.d_setup, R.n_tris,
                               R.d_bins, tiles_x, R.d_tile_count, R.d_chunk_base, R.d_total + 1, z_near, z_scale);
            hipLaunchKernelGGL(raster_count_big, dim3(kBigWaves / 4), b256, 0, stream, R.d_bins, tiles_x, R.d_tile_count, R.d_chunk_base, R.d_total + 1);
            const uint32_t n_chunks = (n_slots + kScanChunk - 1) / kScanChunk;
            hipLaunchKernelGGL(raster_tile_sums, dim3(n_chunks), b256, 0, stream, R.d_tile_count, n_slots, R.d_scan_sums);
            hipLaunchKernelGGL(raster_tile_scan, dim3(n_chunks), b256, 0, stream, R.d_tile_count, R.d_tile_start, n_slots, R.d_scan_sums, R.d_total, capacity, R.d_short);
            hipLaunchKernelGGL(raster_fill, dim3(per_tri.x + kBigWaves / 4), b256, 0, stream, R.d_bins, R.d_order, R.n_tris, tiles_x, R.d_tile_count, R.d_tile_start,
                               R.d_pairs, R.d_chunk_base, R.d_total + 1, capacity);
        }
#ifndef MRT_RASTER_NB
#define MRT_RASTER_NB 4
#endif
        hipLaunchKernelGGL(raster_tile<MRT_RASTER_NB>, dim3(n_tiles), dim3(1024 / MRT_RASTER_NB), 0, stream, R.d_setup, R.d_tile_start, R.d_pairs, tiles_x, tx,
                           fb_direct ? fb_direct : R.d_fb, fb_direct ? pitch_px : W, fb_direct ? 1 : 0,
                           zbuf ? (device_out ? zbuf : R.d_z) : nullptr, tri_ids ? (device_out ? tri_ids : R.d_ids) : nullptr, W, H, capacity);
        RASTER_HIP(hipGetLastError());
        if (R.n_tris > 0) RASTER_HIP(hipMemcpyAsync((void*)R.h_short, R.d_short, 4, hipMemcpyDeviceToHost, stream));
        if (device_out) {
            if (R.n_tris > 0) {
                RASTER_HIP(hipEventRecord(R.frame_done, stream));
                R.frame_queued = true;
            }
            return MRT_OK;
        }
        if (!fb_direct) RASTER_HIP(hipMemcpyAsync(h_fb, R.d_fb, npx * 4, hipMemcpyDeviceToHost, stream));
        if (tri_ids) RASTER_HIP(hipMemcpyAsync(ids_stage, R.d_ids, npx * 4, hipMemcpyDeviceToHost, stream));
        if (zbuf) RASTER_HIP(hipMemcpyAsync(z_stage, R.d_z, npx * 4, hipMemcpyDeviceToHost, stream));
        RASTER_HIP(hipStreamSynchronize(stream));
        if (R.n_tris == 0 || *R.h_short <= R.pair_capacity) break;
        if (int rc = grow_pairs(R, *R.h_short, err)) return rc;      // this frame needs longer lists than any before it: once more
    }
    if (!fb_direct)
        parallel_for((size_t)H, [&](size_t y0, size_t y1) {
            for (size_t y = y0; y < y1; ++y) {
                const uint32_t* src = h_fb + y * W;
                uint32_t* dst = fb + y * pitch_px;
                for (int x = 0; x < W; ++x) dst[x] = src[x] == kUncovered ? dst[x] : src[x];
            }
        }, 64);
    if (tri_ids) parallel_for(npx, [&](size_t b, size_t e) { memcpy(tri_ids + b, ids_stage + 4 * b, (e - b) * 4); }, (size_t)1 << 18);
    if (zbuf) parallel_for(npx, [&](size_t b, size_t e) { memcpy(zbuf + b, z_stage + 4 * b, (e - b) * 4); }, (size_t)1 << 18);
    return MRT_OK;
}

}  // namespace mrt
