// HIP kernels of the ray path for gfx950 (CDNA4, 64-lane wavefronts).
//
// Mapping: one wavefront = one 8x8 pixel tile = one 64-ray packet of the reference
// (raytracer.h:6 PACKETSIZE, game.h:6 TILESIZE); lane k is ray k of the packet, i.e. pixel
// (tile.x + k%8, tile.y + k/8) exactly as threads.cpp:196-211 numbers them.  The packet
// traversal keeps the reference's "first active ray" semantics with wave-uniform control
// flow (ballot + find-first-set); shadow and secondary rays use an independent per-lane
// traversal with a per-lane stack in LDS.
//
// All arithmetic that decides a hit or a colour follows the reference's operation order;
// this file must be compiled with -ffp-contract=off and without fast-math.
#include <hip/hip_runtime.h>

#include "device_scene.h"
#include "mrt_math.h"

namespace mrt {

#define MRT_BACKGROUND f4{1.0f, 0.3f, 0.3f, 0.3f}   // raytracer.cpp:5
#define MRT_AMBIENT    f4{1.0f, 0.0f, 0.0f, 0.0f}   // raytracer.cpp:6
constexpr float kEpsilon = 1e-4f;                   // raytracer.cpp:7
constexpr int kMaxReflPasses = 5, kMaxRefrPasses = 5;   // raytracer.cpp:9-10

struct Hit { int tri; float u, v; };

__device__ __forceinline__ f3 xyz(float4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ uint32_t wbits(float4 v) { return __float_as_uint(v.w); }

// ---------------------------------------------------------------------------------------
// Ray/box: the predicate of raytracer.cpp:876-883 (CheckBox), :1227-1237 and :1389-1393.
// min/max are compare+select as in glm (a NaN or a +-0 must pick the same operand).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_terms(f3 tMin, f3 tMax, float& tNear, float& tFar) {
    const f3 t1 = {gmin(tMin.x, tMax.x), gmin(tMin.y, tMax.y), gmin(tMin.z, tMax.z)};
    const f3 t2 = {gmax(tMin.x, tMax.x), gmax(tMin.y, tMax.y), gmax(tMin.z, tMax.z)};
    tNear = gmax(gmax(t1.x, t1.y), t1.z);
    tFar = gmin(gmin(t2.x, t2.y), t2.z);
}
__device__ __forceinline__ bool slab_pass(float tNear, float tFar, float t) {
    return (tFar >= tNear) && (tNear < t) && (tFar > 0);
}
// multiply by a (supposed) reciprocal: packet path, and the scalar root gate which passes D
__device__ __forceinline__ bool slab_mul(f3 bmin, f3 bmax, f3 O, f3 rD, float t, float& tNear) {
    float tFar;
    slab_terms((bmin - O) * rD, (bmax - O) * rD, tNear, tFar);
    return slab_pass(tNear, tFar, t);
}
// true division by D: scalar path children (raytracer.cpp:1227, 1233)
__device__ __forceinline__ bool slab_div(f3 bmin, f3 bmax, f3 O, f3 D, float t, float& tNear) {
    float tFar;
    slab_terms((bmin - O) / D, (bmax - O) / D, tNear, tFar);
    return slab_pass(tNear, tFar, t);
}

// ---------------------------------------------------------------------------------------
// Ray/triangle: Triangle::Intersect / Intersect2 (raytracer.cpp:287-356).  e1 = p1-p0 and
// e2 = p2-p0 are the reference's first two subtractions, precomputed at upload.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_test(f3 p0, f3 e1, f3 e2, int id, f3 O, f3 D, float& t, Hit& h) {
    const f3 pvec = cross(D, e2);
    const float det = dot(e1, pvec);
    if (det < -kEpsilon) return;
    const float invDet = 1 / det;
    const f3 tvec = O - p0;
    const float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(D, qvec) * invDet;
    if (v < 0 || u + v > 1) return;
    const float T = dot(e2, qvec) * invDet;
    if (T < 0.0f) return;
    if (T < t) { t = T; h.tri = id; h.u = u; h.v = v; }
}

// ---------------------------------------------------------------------------------------
// 64-ray packet traversal: BVH::TraverseRayPacket, BVHNode::TraversePacket and
// IntersectPrimitivesPacket (raytracer.cpp:1349-1464).
//
// The reference recursion "visit(node, firstActive)" becomes a wave-uniform work list of
// (child, f) scan tasks: pop a task, find the first lane i >= f whose ray passes the child's
// box with its CURRENT t, and visit the child with firstActive = i.  Visiting an inner node
// orders its children by lane i's entry distance and pushes two scan tasks (far one first, so
// the near one is scanned next and the far one only after the near subtree has updated t).
// Visiting a leaf tests every lane >= i against every triangle of the leaf, in leaf order.
// The work list lives in one VGPR: entry j is lane j of `stk` (select-write / v_readlane).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void traverse_packet(const SceneDev& S, f3 O, f3 D, f3 rD, float& t, Hit& h, int lane) {
    if (S.n_tris == 0) return;
    uint32_t stk = 0;
    int sp = 0;
    auto push = [&](uint32_t node, int f) {
        stk = (lane == sp) ? ((node << 6) | (uint32_t)f) : stk;
        ++sp;
    };
    push(0u, 0);
    while (sp > 0) {
        --sp;
        const uint32_t e = __builtin_amdgcn_readlane(stk, sp);
        const uint32_t c = e >> 6;
        const int f = (int)(e & 63u);
        const float4 n0 = S.nodes[2 * c], n1 = S.nodes[2 * c + 1];
        float tn;
        const bool pass = slab_mul(xyz(n0), xyz(n1), O, rD, t, tn);
        const unsigned long long mask = __ballot(pass) & (~0ull << f);
        if (mask == 0) continue;
        const int i = __ffsll((long long)mask) - 1;
        const uint32_t first = wbits(n0), count = wbits(n1);
        if (count != 0) {
            if (lane >= i) {
                for (uint32_t k = first; k < first + count; ++k) {
                    const float4 a = S.ltri[3 * k], b = S.ltri[3 * k + 1], cc = S.ltri[3 * k + 2];
                    tri_test(xyz(a), xyz(b), xyz(cc), (int)wbits(a), O, D, t, h);
                }
            }
        } else {
            const uint32_t L = first, R = first + 1;
            const float4 l0 = S.nodes[2 * L], l1 = S.nodes[2 * L + 1];
            const float4 r0 = S.nodes[2 * R], r1 = S.nodes[2 * R + 1];
            float tl, tr, unused;
            slab_terms((xyz(l0) - O) * rD, (xyz(l1) - O) * rD, tl, unused);
            slab_terms((xyz(r0) - O) * rD, (xyz(r1) - O) * rD, tr, unused);
            const float tNear = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl), i));
            const float tNear2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tr), i));
            if (tNear2 < tNear) { push(L, i); push(R, i); }   // right is scanned first
            else                { push(R, i); push(L, i); }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Single-ray traversal: BVH::Traverse + BVHNode::Traverse + IntersectPrimitives
// (raytracer.cpp:945-950, 1204-1264, 1328-1346).  Returns true when a hit was recorded.
// `stack` points at this lane's column of an LDS array laid out [entry][64 lanes].
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool traverse_scalar(const SceneDev& S, f3 O, f3 D, float& t, Hit& h, uint32_t* stack) {
    h.tri = -1;
    if (S.n_tris == 0) return false;
    {
        // root gate: the reference hands CheckBox the direction where it expects the reciprocal
        const float4 r0 = S.nodes[0], r1 = S.nodes[1];
        float tn;
        if (!slab_mul(xyz(r0), xyz(r1), O, D, t, tn)) return false;
    }
    int sp = 0;
    uint32_t node = 0;
    for (;;) {
        const float4 n0 = S.nodes[2 * node], n1 = S.nodes[2 * node + 1];
        const uint32_t first = wbits(n0), count = wbits(n1);
        bool descend = false;
        if (count != 0) {
            for (uint32_t k = first; k < first + count; ++k) {
                const float4 a = S.ltri[3 * k], b = S.ltri[3 * k + 1], c = S.ltri[3 * k + 2];
                tri_test(xyz(a), xyz(b), xyz(c), (int)wbits(a), O, D, t, h);
            }
        } else {
            const uint32_t L = first, R = first + 1;
            const float4 l0 = S.nodes[2 * L], l1 = S.nodes[2 * L + 1];
            const float4 r0 = S.nodes[2 * R], r1 = S.nodes[2 * R + 1];
            float tNear, tNear2;
            const bool leftN = slab_div(xyz(l0), xyz(l1), O, D, t, tNear);
            const bool rightN = slab_div(xyz(r0), xyz(r1), O, D, t, tNear2);
            if (leftN && rightN) {
                const bool rightFirst = tNear2 < tNear;
                stack[sp * 64] = rightFirst ? L : R;
                ++sp;
                node = rightFirst ? R : L;
                descend = true;
            } else if (leftN) { node = L; descend = true; }
            else if (rightN) { node = R; descend = true; }
        }
        if (!descend) {
            if (sp == 0) break;
            --sp;
            node = stack[sp * 64];
        }
    }
    return h.tri >= 0;
}

// ---------------------------------------------------------------------------------------
// Shading
// ---------------------------------------------------------------------------------------
// raytracer.cpp:37-45 PixelToVec
__device__ __forceinline__ f4 pixel_to_vec(uint32_t c) {
    return {((float)((c & 0xff000000u) >> 24)) / 255.0f, ((float)((c & 0x00ff0000u) >> 16)) / 255.0f,
            ((float)((c & 0x0000ff00u) >> 8)) / 255.0f, ((float)(c & 0x000000ffu)) / 255.0f};
}
// raytracer.cpp:28-36 / threads.cpp:179-187: truncating float -> 8 bit per channel
__device__ __forceinline__ uint32_t vec_to_pixel(f4 c) {
    uint32_t r = 0;
    r += (uint32_t)(c.x * 255) << 24;
    r += (uint32_t)(c.y * 255) << 16;
    r += (uint32_t)(c.z * 255) << 8;
    r += (uint32_t)(c.w * 255);
    return r;
}

struct Surf {       // what shading needs from the triangle that was hit
    f3 normal;
    float refl, refr, rindex;
    f4 colorP;
};

// Triangle::GetColor (raytracer.cpp:362-377): palette level 15, truncating texel address.  The
// reference applies no wrap mask, so uv outside [0,1) is undefined there; the linear texel
// index is clamped into the texture here.
__device__ __forceinline__ Surf fetch_surface(const SceneDev& S, const Hit& h) {
    const float4 a0 = S.attr[4 * h.tri], a1 = S.attr[4 * h.tri + 1], a2 = S.attr[4 * h.tri + 2], a3 = S.attr[4 * h.tri + 3];
    Surf s;
    s.normal = xyz(a0);
    s.refl = a0.w; s.refr = a2.z; s.rindex = a2.w;
    const f2 uv0 = {a1.x, a1.y}, uv1 = {a1.z, a1.w}, uv2 = {a2.x, a2.y};
    const f2 uvPos = (uv0 + h.u * (uv1 - uv0)) + h.v * (uv2 - uv0);
    const int tex = (int)__float_as_uint(a3.x);
    const int4 td = S.tex[tex];
    const int xBuffer = (int)(uvPos.x * td.y);
    const int yBuffer = (int)(uvPos.y * td.z);
    long long li = (long long)yBuffer * td.y + xBuffer;
    const long long n = (long long)td.y * td.z;
    li = li < 0 ? 0 : (li >= n ? n - 1 : li);
    const uint32_t texel = S.texels[(size_t)td.x + (size_t)li];
    s.colorP = pixel_to_vec(S.pal[(size_t)tex * (32 * 256) + 15 * 256 + texel]);
    return s;
}

// Direct light of raytracer.cpp:551-570 / 604-623: one shadow ray per light, starting exactly
// on the surface, traced to its nearest hit and used as "anything hit = occluded".
__device__ __forceinline__ void direct_light(const SceneDev& S, const Surf& s, f3 hit, f4& finalColor,
                                             uint32_t* stack, uint32_t& n_shadow) {
    const float lightStrength = 1.0f - (s.refr + s.refl);
    if (lightStrength > 0.0f) {
        finalColor = finalColor + (MRT_AMBIENT * s.colorP) * lightStrength;
        for (int i = 0; i < S.n_lights; ++i) {
            const float4 lp = S.lights[2 * i], lc = S.lights[2 * i + 1];
            const f3 L = normalize(xyz(lp) - hit);
            float lt = INFINITY;
            Hit sh;
            ++n_shadow;
            if (!traverse_scalar(S, hit, L, lt, sh, stack)) {
                float d = dot(s.normal, L);
                if (d < 0.0f) d = 0.0f;
                finalColor = finalColor + ((d * f4{lc.x, lc.y, lc.z, lc.w}) * s.colorP) * lightStrength;
            }
        }
    }
}

// raytracer.cpp:60-78
__device__ __forceinline__ f3 reflect(f3 I, f3 N) {
    const float nl = dot(I, N);
    const f3 projN = nl * N;
    const f3 tan = I - projN;
    return tan - projN;
}

// Raytracer::GetColorFromSphere (raytracer.cpp:575-664) without recursion: an explicit frame
// stack replays "shade, then maybe reflect, then maybe refract, then clamp".  The two pass
// counters are shared by the whole tree of one pixel (the reference passes them by reference),
// so they are a budget of 5 + 5 secondary rays, which also bounds the frame stack.
struct ShadeFrame {
    f4 finalColor, colorP;
    f3 hit, normal, D;
    float t, refl, refr, rindex, curIndex;
    f3 transparency;
    int stage;      // 0: reflection not tried yet, 1: refraction not tried yet, 2: done
};

__device__ __forceinline__ f4 shade_recursive(const SceneDev& S, f3 O, f3 D, float tmax, uint32_t* stack,
                                              Hit& primary, float& primary_t, uint32_t& n_shadow, uint32_t& n_secondary) {
    ShadeFrame frames[kMaxReflPasses + kMaxRefrPasses + 1];
    int depth = 0;
    int reflPass = 0, refrPass = 0;
    float curIndex = 1.0f;
    f4 ret = MRT_BACKGROUND;
    ShadeFrame F;
    bool first = true;
    enum { CALL, CONT, RET } state = CALL;
    for (;;) {
        if (state == CALL) {
            float t = tmax;
            Hit h;
            const bool isHit = traverse_scalar(S, O, D, t, h, stack);
            if (first) { primary = h; primary_t = t; first = false; }
            if (!isHit) { ret = MRT_BACKGROUND; state = RET; continue; }
            const Surf s = fetch_surface(S, h);
            F.hit = O + D * t;
            F.normal = s.normal;
            F.finalColor = {0, 0, 0, 0};
            F.colorP = s.colorP;
            F.D = D; F.t = t;
            F.refl = s.refl; F.refr = s.refr; F.rindex = s.rindex; F.curIndex = curIndex;
            F.stage = 0;
            direct_light(S, s, F.hit, F.finalColor, stack, n_shadow);
            state = CONT;
        }
        if (state == CONT) {
            if (F.stage == 0) {
                F.stage = 1;
                if (F.refl > 0.0f && reflPass < kMaxReflPasses) {
                    reflPass++;
                    ++n_secondary;
                    frames[depth++] = F;
                    O = F.hit; D = reflect(F.D, F.normal); tmax = INFINITY; curIndex = F.curIndex;
                    state = CALL;
                    continue;
                }
            }
            if (F.stage == 1) {
                F.stage = 2;
                if (F.refr > 0.0f && refrPass < kMaxRefrPasses) {
                    refrPass++;
                    const float rindex = F.rindex;
                    const float n = F.curIndex / rindex;
                    const float cosI = -dot(F.normal, F.D);
                    const float cosT2 = 1.0f - n * n * (1.0f - cosI * cosI);
                    if (cosT2 > 0.0f) {
                        const f3 T = (n * F.D) + (n * cosI - sqrtf(cosT2)) * F.normal;
                        const f4 absorbance = (F.colorP * 0.15f) * -F.t;
                        F.transparency = {expf(absorbance.x), expf(absorbance.y), expf(absorbance.z)};
                        ++n_secondary;
                        frames[depth++] = F;
                        O = F.hit + T * kEpsilon; D = T; tmax = INFINITY; curIndex = rindex;
                        state = CALL;
                        continue;
                    }
                }
            }
            ret = clamp01(F.finalColor);
            state = RET;
        }
        // state == RET: hand `ret` to the caller frame, if any
        if (depth == 0) break;
        F = frames[--depth];
        if (F.stage == 1) F.finalColor = F.finalColor + (ret * F.colorP) * F.refl;
        else F.finalColor = F.finalColor + ret * f4{F.transparency.x, F.transparency.y, F.transparency.z, 1.0f};
        state = CONT;
    }
    return ret;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// Frame kernel: ray generation (threads.cpp:189-211), trace, shade, pixel pack + store
// (threads.cpp:215-224).  MODE 0 = packet path, 1 = recursive shader.  SD = per-lane stack
// entries.  One 64-thread workgroup (one wave) per 8x8 tile of the traced image.
// ---------------------------------------------------------------------------------------
template <int MODE, int SD>
__global__ __launch_bounds__(64) void render_tiles(SceneDev S, FrameDev F) {
    __shared__ uint32_t lds_stack[SD * 64];
    const int lane = threadIdx.x;
    const int tile = blockIdx.x;
    if (tile >= F.n_tiles) return;
    const int tx = tile % F.tiles_x, ty = F.tile_y0 + tile / F.tiles_x;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    uint32_t* stack = lds_stack + lane;

    const f3 p0 = {F.p0[0], F.p0[1], F.p0[2]}, d10 = {F.d10[0], F.d10[1], F.d10[2]}, d20 = {F.d20[0], F.d20[1], F.d20[2]};
    const f3 O = {F.cam[0], F.cam[1], F.cam[2]};
    const float u = (float)x * F.inv_w;
    const float v = (float)y * F.inv_h;
    const f3 planepos = (p0 + u * d10) + v * d20;
    const f3 D = normalize(planepos - O);

    uint32_t n_shadow = 0, n_secondary = 0;
    Hit h;
    h.tri = -1; h.u = 0; h.v = 0;
    f4 color;
    if (MODE == 0) {
        const f3 rD = 1.0f / D;
        float t = 100000000;
        traverse_packet(S, O, D, rD, t, h, lane);
        if (h.tri < 0) color = MRT_BACKGROUND;
        else {
            const f3 hit = O + D * t;
            const Surf s = fetch_surface(S, h);
            f4 finalColor = {0, 0, 0, 0};
            direct_light(S, s, hit, finalColor, stack, n_shadow);
            color = clamp01(finalColor);
        }
    } else {
        float pt;
        color = shade_recursive(S, O, D, 100000000, stack, h, pt, n_shadow, n_secondary);
    }
    const uint32_t px = vec_to_pixel(color);

    if (F.hit_ids) F.hit_ids[(size_t)y * F.width + x] = h.tri;
    const int g = F.grid_g;
    if (g == 1) {
        F.fb[(size_t)y * F.pitch + x] = px;
    } else {
        // box filter over g x g blocks of QUANTISED pixels, rounded down per channel
        uint32_t ar = ((px >> 24) << 16) | ((px >> 16) & 255u);      // A | R   (16 bits each)
        uint32_t gb = (((px >> 8) & 255u) << 16) | (px & 255u);      // G | B
        for (int o = 1; o < g; o <<= 1) {
            ar += __shfl_xor(ar, o, 64);      gb += __shfl_xor(gb, o, 64);
            ar += __shfl_xor(ar, o * 8, 64);  gb += __shfl_xor(gb, o * 8, 64);
        }
        if (((lane & 7) % g) == 0 && ((lane >> 3) % g) == 0) {
            const uint32_t n = (uint32_t)(g * g);
            const uint32_t out = (((ar >> 16) / n) << 24) | (((ar & 0xffffu) / n) << 16) | (((gb >> 16) / n) << 8) | ((gb & 0xffffu) / n);
            F.fb[(size_t)(y / g) * F.pitch + (x / g)] = out;
        }
    }
    if (F.counters) {
        const uint32_t hits = (uint32_t)__popcll(__ballot(h.tri >= 0));
        const uint32_t sh = wave_sum(n_shadow), se = wave_sum(n_secondary);
        if (lane == 0) {
            atomicAdd(&F.counters[0], 64ull);
            if (sh) atomicAdd(&F.counters[1], (unsigned long long)sh);
            if (se) atomicAdd(&F.counters[2], (unsigned long long)se);
            if (hits) atomicAdd(&F.counters[3], (unsigned long long)hits);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Operator-level kernels on caller-supplied rays
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void trace_packets_kernel(SceneDev S, int n_packets, const float* O, const float* D,
                                                           const float* rD, float* t, int32_t* tri, float* u, float* v) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_packets) return;
    const size_t r = (size_t)p * 64 + lane;
    const f3 o = {O[3 * r], O[3 * r + 1], O[3 * r + 2]};
    const f3 d = {D[3 * r], D[3 * r + 1], D[3 * r + 2]};
    const f3 rd = {rD[3 * r], rD[3 * r + 1], rD[3 * r + 2]};
    float tt = t[r];
    Hit h;
    h.tri = -1; h.u = 0; h.v = 0;
    traverse_packet(S, o, d, rd, tt, h, lane);
    t[r] = tt; tri[r] = h.tri; u[r] = h.u; v[r] = h.v;
}

template <int SD>
__global__ __launch_bounds__(64) void trace_rays_kernel(SceneDev S, int n_rays, const float* O, const float* D,
                                                        float* t, int32_t* tri, float* u, float* v) {
    __shared__ uint32_t lds_stack[SD * 64];
    const int lane = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * 64 + lane;
    if (r >= (size_t)n_rays) return;
    const f3 o = {O[3 * r], O[3 * r + 1], O[3 * r + 2]};
    const f3 d = {D[3 * r], D[3 * r + 1], D[3 * r + 2]};
    float tt = t[r];
    Hit h;
    h.u = 0; h.v = 0;
    traverse_scalar(S, o, d, tt, h, lds_stack + lane);
    t[r] = tt; tri[r] = h.tri; u[r] = h.u; v[r] = h.v;
}

// ---------------------------------------------------------------------------------------
// Host-callable launchers (used by the C ABI in mrt_api.cpp)
// ---------------------------------------------------------------------------------------
template <int MODE>
static hipError_t launch_render_mode(const SceneDev& S, const FrameDev& F, int stack_entries, hipStream_t stream) {
    const dim3 grid((unsigned)F.n_tiles), block(64);
    if (stack_entries <= 16) hipLaunchKernelGGL((render_tiles<MODE, 16>), grid, block, 0, stream, S, F);
    else if (stack_entries <= 32) hipLaunchKernelGGL((render_tiles<MODE, 32>), grid, block, 0, stream, S, F);
    else hipLaunchKernelGGL((render_tiles<MODE, kStackDepth>), grid, block, 0, stream, S, F);
    return hipGetLastError();
}

hipError_t launch_render(const SceneDev& S, const FrameDev& F, int mode, int stack_entries, hipStream_t stream) {
    if (F.n_tiles <= 0) return hipSuccess;
    return mode == 0 ? launch_render_mode<0>(S, F, stack_entries, stream) : launch_render_mode<1>(S, F, stack_entries, stream);
}

hipError_t launch_trace_packets(const SceneDev& S, int n_packets, const float* O, const float* D, const float* rD,
                                float* t, int32_t* tri, float* u, float* v, hipStream_t stream) {
    if (n_packets <= 0) return hipSuccess;
    hipLaunchKernelGGL(trace_packets_kernel, dim3((unsigned)n_packets), dim3(64), 0, stream, S, n_packets, O, D, rD, t, tri, u, v);
    return hipGetLastError();
}

hipError_t launch_trace_rays(const SceneDev& S, int n_rays, int stack_entries, const float* O, const float* D,
                             float* t, int32_t* tri, float* u, float* v, hipStream_t stream) {
    if (n_rays <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_rays + 63) / 64)), block(64);
    if (stack_entries <= 16) hipLaunchKernelGGL((trace_rays_kernel<16>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v);
    else if (stack_entries <= 32) hipLaunchKernelGGL((trace_rays_kernel<32>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v);
    else hipLaunchKernelGGL((trace_rays_kernel<kStackDepth>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v);
    return hipGetLastError();
}

}  // namespace mrt
