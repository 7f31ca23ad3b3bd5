// Rasterizer path: Rasterizer::Render / RenderMesh (rasterizer.cpp:53-184) as three kernels.
//
//   raster_setup    one thread per triangle: vertex transform, back-face cull, Sutherland-Hodgman
//                   clip against the near and left planes, projection, palette level, and the
//                   screen-space plane equations of 1/z, u/z and v/z (rasterizer.cpp:100-141).
//   raster_emit     cuts every triangle's pixel bounding box into chunks of kChunkBlocks 8x8 blocks
//                   and appends one work item per chunk, so a screen-filling triangle is spread
//                   over hundreds of waves while a small one costs a single wave.
//   raster_cover    one wave per chunk, one lane per pixel of an 8x8 block;
//                   a pixel is covered exactly when the reference's scan conversion would plot it
//                   (rows y0 < y <= y1 of the spanning edges, columns xl < x <= xr of the row's
//                   extent, rasterizer.cpp:143-182), and the depth test is a 64-bit atomicMin on
//                   {orderable 1/z, triangle index}: the reference draws triangles in order and
//                   skips a pixel when the new 1/z is >= the stored one, i.e. the smallest 1/z
//                   wins and, among equals, the first triangle drawn.
//   raster_resolve  one thread per pixel: texel fetch and palette lookup of the winner
//                   (rasterizer.cpp:178-180), optional 1/z and triangle-id planes.
//
// The reference interpolates along edges and spans by repeated addition; here the same affine
// functions are evaluated directly, so results differ only where a pixel centre lies within
// rounding distance of an edge, a depth tie or a texel border.
#include "raster.h"

#include <algorithm>

namespace mrt {

struct RasterFrame {
    float M[16];             // inverse camera matrix (world -> view), column-major
    float frustum[2][4];     // clip planes 0 (near) and 1 (left): rasterizer.cpp:37-38
    int W, H;
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

__global__ __launch_bounds__(256) void raster_setup(RasterFrame F, const float* pos, const float* uv, const float* N, const int32_t* tri,
                                                     const int32_t* tri_tex, RasterTri* out, uint32_t n_tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    RasterTri T;
    T.n = 0;
    const int texture = tri_tex[i];
    if (texture >= 0) {
        const int W = F.W, H = F.H;
        m4 M;
        memcpy(&M, F.M, 64);
        const f3 Nt = mul3(M, f3{N[3 * i], N[3 * i + 1], N[3 * i + 2]});       // rasterizer.cpp:112
        f3 cpos[2][8];
        f2 cuv[2][8];
        for (int v = 0; v < 3; ++v) {
            const int vi = tri[3 * i + v];
            cpos[0][v] = xform(F.M, f3{pos[3 * vi], pos[3 * vi + 1], pos[3 * vi + 2]});   // rasterizer.cpp:100
            cuv[0][v] = f2{uv[2 * vi], uv[2 * vi + 1]};
        }
        if (!(dot(cpos[0][0], Nt) > 0)) {                                      // back-face cull, :113
            int nin = 3, nout = 0, from = 0, to = 1;
            for (int p = 0; p < 2; p++, from = 1 - from, to = 1 - to, nin = nout, nout = 0)   // :119-133
                for (int v = 0; v < nin; v++) {
                    const f3 A = cpos[from][v], B = cpos[from][(v + 1) % nin];
                    const f2 Auv = cuv[from][v], Buv = cuv[from][(v + 1) % nin];
                    const f3 pn = {F.frustum[p][0], F.frustum[p][1], F.frustum[p][2]};
                    const float pw = F.frustum[p][3];
                    const float t1 = dot(pn, A) - pw, t2 = dot(pn, B) - pw;
                    if ((t1 < 0) && (t2 >= 0)) {
                        const float f = t1 / (t1 - t2);
                        cuv[to][nout] = Auv + f * (Buv - Auv); cpos[to][nout++] = A + f * (B - A);
                        cuv[to][nout] = Buv; cpos[to][nout++] = B;
                    } else if ((t1 >= 0) && (t2 >= 0)) { cuv[to][nout] = Buv; cpos[to][nout++] = B; }
                    else if ((t1 >= 0) && (t2 < 0)) {
                        const float f = t1 / (t1 - t2);
                        cuv[to][nout] = Auv + f * (Buv - Auv); cpos[to][nout++] = A + f * (B - A);
                    }
                }
            if (nin >= 3 && nin <= 5) {
                T.level = (int)(gmax(0.0f, Nt.z) * 31);                        // :136
                T.texture = texture;
                float sx[5], sy[5], iz[5], uz[5], vz[5];
                float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
                for (int v = 0; v < nin; ++v) {
                    const f3 p = cpos[from][v];
                    sx[v] = ((p.x * W) / -p.z) + W / 2;                         // :140-141
                    sy[v] = ((p.y * W) / p.z) + H / 2;
                    iz[v] = 1.0f / p.z;                                        // :151
                    uz[v] = cuv[from][v].x * iz[v];                            // :152-153
                    vz[v] = cuv[from][v].y * iz[v];
                    xmin = fminf(xmin, sx[v]); xmax = fmaxf(xmax, sx[v]);
                    ymin = fminf(ymin, sy[v]); ymax = fmaxf(ymax, sy[v]);
                }
                // affine attribute planes through the polygon's widest vertex triple
                int b = 1, c = 2;
                float best = 0.0f;
                for (int j = 1; j < nin; ++j)
                    for (int k = j + 1; k < nin; ++k) {
                        const float ar = fabsf((sx[j] - sx[0]) * (sy[k] - sy[0]) - (sx[k] - sx[0]) * (sy[j] - sy[0]));
                        if (ar > best) { best = ar; b = j; c = k; }
                    }
                const double x1 = (double)sx[b] - sx[0], y1 = (double)sy[b] - sy[0], x2 = (double)sx[c] - sx[0], y2 = (double)sy[c] - sy[0];
                const double det = x1 * y2 - x2 * y1;
                if (det != 0.0 && xmax >= 0.0f && ymax >= 1.0f && xmin < (float)W && ymin < (float)H) {
                    auto plane = [&](const float* a, float& A, float& B, float& C) {
                        const double d1 = (double)a[b] - a[0], d2 = (double)a[c] - a[0];
                        const double pa = (d1 * y2 - d2 * y1) / det, pb = (x1 * d2 - x2 * d1) / det;
                        A = (float)pa; B = (float)pb; C = (float)((double)a[0] - pa * sx[0] - pb * sy[0]);
                    };
                    plane(iz, T.zA, T.zB, T.zC);
                    plane(uz, T.uA, T.uB, T.uC);
                    plane(vz, T.vA, T.vB, T.vC);
                    // rows max(1,(int)y0+1) .. min(H-2,(int)y1), columns (int)x0+1 .. min(W-2,(int)x1): :149, :170
                    T.y0 = max(1, (int)fmaxf(ymin, -1.0f) + 1);
                    T.y1 = min(H - 2, (int)fminf(ymax, (float)H));
                    T.x0 = max(0, (int)fmaxf(xmin, -1.0f) + 1);
                    T.x1 = min(W - 2, (int)fminf(xmax, (float)W));
                    if (T.y0 <= T.y1 && T.x0 <= T.x1) {
                        int ne = 0;
                        for (int j = 0; j < nin; ++j) {                      // rasterizer.cpp:143-151
                            int a = j, b2 = (j + 1 == nin) ? 0 : j + 1;
                            if (sy[a] > sy[b2]) { const int h = a; a = b2; b2 = h; }
                            const float ya = sy[a], yb = sy[b2];
                            if ((ya == yb) || (ya >= H) || (yb < 1)) continue;
                            const float rydiff = 1.0f / (yb - ya);
                            T.ex[ne] = sx[a]; T.ey[ne] = ya; T.dx[ne] = (sx[b2] - sx[a]) * rydiff;
                            T.iy0[ne] = max(1, (int)ya + 1); T.iy1[ne] = min(H - 2, (int)yb);
                            if (T.iy0[ne] <= T.iy1[ne]) ++ne;
                        }
                        T.n = ne;
                    }
                }
            }
        }
    }
    out[i] = T;
}

// Is pixel (x, y) plotted by the reference's scan conversion of this polygon?  The row extent of
// rasterizer.cpp:143-170: every edge contributes to the rows it spans; the row's extent is
// [min x, max x] over those edges; columns (int)xl+1 .. min(W-2, (int)xr).
__device__ __forceinline__ bool covers(const RasterTri& T, int x, int y, int W) {
    float xl = (float)(W - 1), xr = 0.0f;             // the outline tables' idle values, :28, :164
    bool any = false;
    for (int j = 0; j < T.n; ++j) {
        if (y < T.iy0[j] || y > T.iy1[j]) continue;
        const float xe = T.ex[j] + T.dx[j] * ((float)y - T.ey[j]);
        if (xe < xl) xl = xe;
        if (xe > xr) xr = xe;
        any = true;
    }
    if (!any) return false;
    const int ix0 = (int)xl + 1, ix1 = min(W - 2, (int)xr);
    return x >= ix0 && x <= ix1;
}

constexpr int kChunkBlocks = 16;     // 8x8 blocks per coverage work item

__device__ __forceinline__ uint32_t tri_blocks(const RasterTri& T) {
    if (T.n == 0) return 0;
    const uint32_t bw = (uint32_t)((T.x1 >> 3) - (T.x0 >> 3) + 1), bh = (uint32_t)((T.y1 >> 3) - (T.y0 >> 3) + 1);
    return bw * bh;
}

// pass 1: reserve chunk slots per triangle
__global__ __launch_bounds__(256) void raster_count(const RasterTri* setup, uint32_t n_tris, uint32_t* chunk_base, uint32_t* total) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    const uint32_t nb = tri_blocks(setup[t]);
    const uint32_t nc = (nb + kChunkBlocks - 1) / kChunkBlocks;
    chunk_base[t] = nc ? atomicAdd(total, nc) : 0u;
}
// pass 2: write the work items
__global__ __launch_bounds__(256) void raster_emit(const RasterTri* setup, uint32_t n_tris, const uint32_t* chunk_base, uint2* chunks) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    const uint32_t nb = tri_blocks(setup[t]);
    const uint32_t nc = (nb + kChunkBlocks - 1) / kChunkBlocks;
    for (uint32_t c = 0; c < nc; ++c) chunks[chunk_base[t] + c] = make_uint2(t, c * kChunkBlocks);
}

__global__ __launch_bounds__(64) void raster_cover(const RasterTri* setup, const uint2* chunks, uint32_t n_chunks,
                                                   unsigned long long* depth, int W, int H) {
    if (blockIdx.x >= n_chunks) return;
    const uint2 item = chunks[blockIdx.x];
    const uint32_t t = item.x;
    const RasterTri T = setup[t];
    const int lane = threadIdx.x;
    const int bx0 = T.x0 >> 3, by0 = T.y0 >> 3;
    const uint32_t bw = (uint32_t)((T.x1 >> 3) - bx0 + 1);
    const uint32_t nb = tri_blocks(T);
    const uint32_t end = min(item.y + (uint32_t)kChunkBlocks, nb);
    for (uint32_t b = item.y; b < end; ++b) {
        const int x = ((bx0 + (int)(b % bw)) << 3) + (lane & 7), y = ((by0 + (int)(b / bw)) << 3) + (lane >> 3);
        if (x < T.x0 || x > T.x1 || y < T.y0 || y > T.y1) continue;
        if (!covers(T, x, y, W)) continue;
        const float z0 = T.zA * (float)x + T.zB * (float)y + T.zC;
        if (!(z0 < 0.0f)) continue;                     // the z-buffer is cleared to 0 and the test is "z0 >= zbuf: skip" (:55, :177)
        const unsigned long long key = ((unsigned long long)orderable(z0) << 32) | t;
        atomicMin(&depth[(size_t)y * W + x], key);
    }
}

__global__ __launch_bounds__(256) void raster_resolve(const RasterTri* setup, const unsigned long long* depth, RasterTextures tx,
                                                       uint32_t* fb, float* zbuf, int32_t* ids, int W, int H) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const unsigned long long key = depth[p];
    const uint32_t t = (uint32_t)key;
    if (t == 0xffffffffu) {
        if (zbuf) zbuf[p] = 0.0f;
        if (ids) ids[p] = -1;
        fb[p] = kUncovered;              // palette entries are 0x00RRGGBB, so this value marks "nothing drawn here"
        return;
    }
    const RasterTri T = setup[t];
    const float z0 = from_orderable((unsigned int)(key >> 32));
    const float u0 = T.uA * (float)x + T.uB * (float)y + T.uC;
    const float v0 = T.vA * (float)x + T.vB * (float)y + T.vC;
    const int4 td = tx.tex[T.texture];
    const float tw = (float)td.y, th = (float)td.z;
    const int umask = td.y - 1, vmask = td.z - 1;
    const float z = 1.0f / z0;                                                  // rasterizer.cpp:178-180
    const int u = (int)(u0 * z * tw) & umask, v = (int)(v0 * z * th) & vmask;
    const uint32_t texel = tx.texels[(size_t)td.x + (size_t)(u + v * (umask + 1))];
    fb[p] = tx.pal[(size_t)T.texture * (32 * 256) + (size_t)T.level * 256 + texel];
    if (zbuf) zbuf[p] = z0;
    if (ids) ids[p] = (int32_t)t;
}

__global__ void raster_clear(unsigned long long* depth, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 1/z = 0 (the cleared z-buffer) with no triangle: every negative 1/z is smaller
    if (i < n) depth[i] = ((unsigned long long)orderable(0.0f) << 32) | 0xffffffffull;
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
    for (void* p : {(void*)R.d_pos, (void*)R.d_uv, (void*)R.d_N, (void*)R.d_tri, (void*)R.d_tri_tex, (void*)R.d_setup,
                    (void*)R.d_chunks, (void*)R.d_chunk_base, (void*)R.d_total,
                    (void*)R.d_depth, (void*)R.d_fb, (void*)R.d_z, (void*)R.d_ids})
        if (p) (void)hipFree(p);
    if (R.h_stage) (void)hipHostFree(R.h_stage);
    R = RasterState();
}

int raster_upload(RasterState& R, const mrt_mesh* meshes, int n_meshes, hipStream_t stream, std::string& err) {
    raster_release(R);
    std::vector<float> pos, uv, N;
    std::vector<int32_t> tri;
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        RasterMeshHost h;
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
    RASTER_HIP(hipMalloc((void**)&R.d_tri_tex, (size_t)R.n_tris * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_setup, (size_t)R.n_tris * sizeof(RasterTri)));
    RASTER_HIP(hipMalloc((void**)&R.d_chunk_base, (size_t)R.n_tris * 4));
    RASTER_HIP(hipMalloc((void**)&R.d_total, 16));
    RASTER_HIP(hipMemcpyAsync(R.d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_uv, uv.data(), uv.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_N, N.data(), N.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipMemcpyAsync(R.d_tri, tri.data(), tri.size() * 4, hipMemcpyHostToDevice, stream));
    RASTER_HIP(hipStreamSynchronize(stream));      // the staging vectors go out of scope
    return MRT_OK;
}

// Rasterizer::Init's frustum planes (rasterizer.cpp:31-41) for a W x H target
static void frustum_planes(int W, int H, f4 out[5]) {
    const float C = -1.0f, x1 = 0.5f, x2 = W - 1.5f, y1 = 0.5f, y2 = H - 1.5f;
    const f3 p0 = {0, 0, 0};
    const f3 p1 = {((x1 - W / 2) * C) / W, ((y1 - H / 2) * C) / W, 1.0f};
    const f3 p2 = {((x2 - W / 2) * C) / W, ((y1 - H / 2) * C) / W, 1.0f};
    const f3 p3 = {((x2 - W / 2) * C) / W, ((y2 - H / 2) * C) / W, 1.0f};
    const f3 p4 = {((x1 - W / 2) * C) / W, ((y2 - H / 2) * C) / W, 1.0f};
    auto plane = [](f3 n) { return f4{n.x, n.y, n.z, 0}; };
    out[0] = {0, 0, -1, 0.2f};
    out[1] = plane(normalize(cross(p1 - p0, p4 - p1)));
    out[2] = plane(normalize(cross(p2 - p0, p1 - p2)));
    out[3] = plane(normalize(cross(p3 - p0, p2 - p3)));
    out[4] = plane(normalize(cross(p4 - p0, p3 - p4)));
}

int raster_render(RasterState& R, const RasterTextures& tx, int W, int H, const float cam16[16], uint32_t* fb, int pitch_px,
                  float* zbuf, int32_t* tri_ids, hipStream_t stream, std::string& err) {
    const size_t npx = (size_t)W * H;
    if (R.frame_px != npx) {
        for (void* p : {(void*)R.d_depth, (void*)R.d_fb, (void*)R.d_z, (void*)R.d_ids}) if (p) (void)hipFree(p);
        R.d_depth = nullptr; R.d_fb = nullptr; R.d_z = nullptr; R.d_ids = nullptr;
        RASTER_HIP(hipMalloc((void**)&R.d_depth, npx * 8));
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
    std::vector<int32_t> tri_tex(R.n_tris, -1);
    for (size_t m = 0; m < R.meshes.size(); ++m) {
        const RasterMeshHost& h = R.meshes[m];
        bool visible = h.tris > 0 && h.texture >= 0;
        if (visible) {
            f3 c[8];
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
        if (visible) std::fill(tri_tex.begin() + h.tri_off, tri_tex.begin() + h.tri_off + h.tris, h.texture);
    }
    RasterFrame F;
    memcpy(F.M, &M, 64);
    for (int p = 0; p < 2; ++p) { F.frustum[p][0] = frustum[p].x; F.frustum[p][1] = frustum[p].y; F.frustum[p][2] = frustum[p].z; F.frustum[p][3] = frustum[p].w; }
    F.W = W; F.H = H;
    hipLaunchKernelGGL(raster_clear, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, R.d_depth, npx);
    if (R.n_tris > 0) {
        RASTER_HIP(hipMemcpyAsync(R.d_tri_tex, tri_tex.data(), (size_t)R.n_tris * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(raster_setup, dim3((R.n_tris + 255) / 256), dim3(256), 0, stream, F, R.d_pos, R.d_uv, R.d_N, R.d_tri,
                           R.d_tri_tex, R.d_setup, R.n_tris);
        RASTER_HIP(hipMemsetAsync(R.d_total, 0, 4, stream));
        hipLaunchKernelGGL(raster_count, dim3((R.n_tris + 255) / 256), dim3(256), 0, stream, R.d_setup, R.n_tris, R.d_chunk_base, R.d_total);
        uint32_t n_chunks = 0;
        RASTER_HIP(hipMemcpyAsync(&n_chunks, R.d_total, 4, hipMemcpyDeviceToHost, stream));
        RASTER_HIP(hipStreamSynchronize(stream));
        if (n_chunks > R.chunk_capacity) {
            if (R.d_chunks) (void)hipFree(R.d_chunks);
            R.d_chunks = nullptr; R.chunk_capacity = 0;
            const size_t cap = (size_t)n_chunks + n_chunks / 4 + 1024;
            RASTER_HIP(hipMalloc((void**)&R.d_chunks, cap * sizeof(uint2)));
            R.chunk_capacity = cap;
        }
        if (n_chunks > 0) {
            hipLaunchKernelGGL(raster_emit, dim3((R.n_tris + 255) / 256), dim3(256), 0, stream, R.d_setup, R.n_tris, R.d_chunk_base, R.d_chunks);
            hipLaunchKernelGGL(raster_cover, dim3(n_chunks), dim3(64), 0, stream, R.d_setup, R.d_chunks, n_chunks, R.d_depth, W, H);
        }
    }
    hipLaunchKernelGGL(raster_resolve, dim3((unsigned)((W + 255) / 256), (unsigned)H), dim3(256), 0, stream, R.d_setup, R.d_depth, tx,
                       R.d_fb, zbuf ? R.d_z : nullptr, tri_ids ? R.d_ids : nullptr, W, H);
    RASTER_HIP(hipGetLastError());
    // Pixels no triangle covers are left as the caller had them (the reference only plots covered
    // pixels): the frame comes back through page-locked staging and covered pixels are merged in.
    const size_t need = npx * 4 + (tri_ids ? npx * 4 : 0) + (zbuf ? npx * 4 : 0);
    if (R.h_stage_bytes < need) {
        if (R.h_stage) (void)hipHostFree(R.h_stage);
        R.h_stage = nullptr; R.h_stage_bytes = 0;
        RASTER_HIP(hipHostMalloc(&R.h_stage, need, hipHostMallocDefault));
        R.h_stage_bytes = need;
    }
    unsigned char* stage = static_cast<unsigned char*>(R.h_stage);
    uint32_t* h_fb = reinterpret_cast<uint32_t*>(stage);
    unsigned char* next = stage + npx * 4;
    RASTER_HIP(hipMemcpyAsync(h_fb, R.d_fb, npx * 4, hipMemcpyDeviceToHost, stream));
    if (tri_ids) { RASTER_HIP(hipMemcpyAsync(next, R.d_ids, npx * 4, hipMemcpyDeviceToHost, stream)); }
    unsigned char* ids_stage = next;
    if (tri_ids) next += npx * 4;
    if (zbuf) RASTER_HIP(hipMemcpyAsync(next, R.d_z, npx * 4, hipMemcpyDeviceToHost, stream));
    RASTER_HIP(hipStreamSynchronize(stream));      // tri_tex goes out of scope after this
    for (int y = 0; y < H; ++y) {
        const uint32_t* src = h_fb + (size_t)y * W;
        uint32_t* dst = fb + (size_t)y * pitch_px;
        for (int x = 0; x < W; ++x) dst[x] = src[x] == kUncovered ? dst[x] : src[x];
    }
    if (tri_ids) memcpy(tri_ids, ids_stage, npx * 4);
    if (zbuf) memcpy(zbuf, next, npx * 4);
    return MRT_OK;
}

}  // namespace mrt
