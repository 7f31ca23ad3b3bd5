// Vector arithmetic shared by the host code and the HIP kernels.
//
// Operation ORDER is part of the contract: results must be bit-identical to the reference,
// which takes its math from glm 0.9.5.4 (paths below are relative to the reference's
// Project/lib/glm/).  Build with -ffp-contract=off; never with fast-math.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MRT_HD __host__ __device__ __forceinline__
#else
#define MRT_HD inline
#endif

namespace mrt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct m4 { f4 c[4]; };   // column-major: c[i] is column i

// detail/func_common.inl:367-399 -- compare + select, not fminf/fmaxf (NaN and +-0 differ)
MRT_HD float gmin(float x, float y) { return x < y ? x : y; }
MRT_HD float gmax(float x, float y) { return x > y ? x : y; }
// detail/func_common.inl:403-415
MRT_HD float gclamp(float x, float lo, float hi) { return gmin(hi, gmax(lo, x)); }

// detail/type_vec3.inl:540-616, detail/type_vec4.inl:602-626 -- component-wise
MRT_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
MRT_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
MRT_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
MRT_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
MRT_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
MRT_HD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
MRT_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
MRT_HD f3 operator/(float s, f3 a) { return {s / a.x, s / a.y, s / a.z}; }
MRT_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
MRT_HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
MRT_HD f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }
MRT_HD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
MRT_HD f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
MRT_HD f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
MRT_HD f4 operator*(f4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
MRT_HD f4 operator*(float s, f4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }
MRT_HD f4 clamp01(f4 a) {
    return {gclamp(a.x, 0.0f, 1.0f), gclamp(a.y, 0.0f, 1.0f), gclamp(a.z, 0.0f, 1.0f), gclamp(a.w, 0.0f, 1.0f)};
}

// detail/func_geometric.inl:66-73 -- three products, then (x + y) + z
MRT_HD float dot(f3 a, f3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// detail/func_geometric.inl:215-228
MRT_HD f3 cross(f3 x, f3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
// detail/func_geometric.inl:256-265 + detail/func_exponential.inl:226-229: IEEE sqrt, then an
// IEEE divide, then three multiplies (not rsqrt)
MRT_HD f3 normalize(f3 v) {
    float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
    float r = 1.0f / sqrtf(sqr);
    return v * r;
}

// ---- exact division by a divisor whose correctly rounded reciprocal is known ------------
// q = a / d, correctly rounded, from r = RN(1/d): one multiply and two fused residual
// corrections.  After the first correction the quotient is within one ulp, the residual
// a - d*q is then exactly representable, and adding residual * r rounds to RN(a/d) (Markstein's
// final-step theorem) PROVIDED that
//   - d's significand is not all ones, 2^-40 <= |d| <= 2^12,
//   - a == 0 or 2^-50 <= |a| <= 2^31   (no overflow, and the residuals stay exact even when
//     they are subnormal).
// div_shortcut_dir_ok / div_shortcut_coord_ok test these ranges; coordinates that are 0 or in
// [2^-27, 2^30] differ by 0 or by at least 2^-50, which is how the range of a = bound - origin
// is guaranteed without looking at every difference.
MRT_HD float div_by_recip(float a, float d, float r) {
    const float q0 = a * r;
    const float e1 = __builtin_fmaf(-d, q0, a);
    const float q1 = __builtin_fmaf(e1, r, q0);
    const float e2 = __builtin_fmaf(-d, q1, a);
    return __builtin_fmaf(e2, r, q1);
}
MRT_HD bool div_shortcut_dir_ok(float d) {
    uint32_t u;
    memcpy(&u, &d, 4);
    const float a = __builtin_fabsf(d);
    return (a >= 0x1p-40f) && (a <= 0x1p12f) && ((u & 0x7fffffu) != 0x7fffffu);
}
MRT_HD bool div_shortcut_coord_ok(float c) {
    const float a = __builtin_fabsf(c);
    return (c == 0.0f) || ((a >= 0x1p-27f) && (a <= 0x1p30f));
}

// Iteration budget of the outer traversal loops of one tile or one operator call: far above what a
// correct traversal of a tree with this many node slots and triangles can need, computed in 64 bits
// and saturated (the product passes 2^31 near 3.7 million triangles).
MRT_HD int watch_budget(int n_node_slots, int n_tris) {
    const long long b = 64ll * (4ll * (long long)n_node_slots + (long long)n_tris + 64ll);
    return b > 0x7fffffffll ? 0x7fffffff : (int)b;
}

// ---- binary16 bounds that never shrink a box -------------------------------------------------
// half_outward(x, up): the binary16 value nearest to x on the side asked for -- not below x (up) or not
// above x (!up) -- restricted to zero, the normal numbers and the infinities, so that the result reads back
// the same whether or not the reader flushes subnormal halves.  Finite x only.
MRT_HD uint16_t half_outward(float x, bool up) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u >> 31, mag = u & 0x7fffffffu;
    const bool away = (up && !sign) || (!up && sign);          // round the magnitude up?
    const int e = (int)(mag >> 23) - 127 + 15;                 // biased binary16 exponent
    uint32_t h;
    if (mag == 0) h = 0;
    else if (e >= 31) h = away ? 0x7c00u : 0x7bffu;            // beyond the largest half: infinity or 65504
    else if (e <= 0) h = away ? 0x0400u : 0u;                  // below the smallest normal half: 2^-14 or zero
    else {
        h = ((uint32_t)e << 10) | ((mag >> 13) & 0x3ffu);
        if (away && (mag & 0x1fffu)) ++h;                      // a carry runs into the exponent, up to infinity
    }
    return (uint16_t)((sign << 15) | h);
}
// the value of a binary16 bit pattern (zero, normal or infinite: what half_outward produces)
MRT_HD float half_value(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) u = sign;
    else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// A triangle whose edges e1, e2 enclose less than a thousandth of a radian (or that has an edge of length zero):
// the determinant of its ray test is small against its rounding errors for EVERY ray, so nothing can be said
// about where such a test reports hits.  Groups holding one are never skipped.
MRT_HD bool triangle_is_sliver(f3 p0, f3 p1, f3 p2) {
    const f3 e1 = p1 - p0, e2 = p2 - p0, n = cross(e1, e2);
    const double nn = (double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z;
    const double a = (double)e1.x * e1.x + (double)e1.y * e1.y + (double)e1.z * e1.z, b = (double)e2.x * e2.x + (double)e2.y * e2.y + (double)e2.z * e2.z;
    return !(nn > 1e-6 * a * b);
}

// Does the point O lie within `pad` of the plane of the triangle {p0, p0 + e1, p0 + e2}?  A ray that starts there can
// run IN that plane, where every numerator of the triangle test is rounding noise and no box around the triangle can
// predict the test's verdict; the frame kernels therefore never skip a group of triangles for primary rays when the eye
// is near the plane of one of them.  Errs towards "near" (also for a NaN or a triangle without area).
MRT_HD bool point_near_plane(f3 O, f3 p0, f3 e1, f3 e2, float pad) {
    const f3 n = cross(e1, e2), d = O - p0;
    const double nn = (double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z;
    const double h = (double)d.x * n.x + (double)d.y * n.y + (double)d.z * n.z;      // distance times |n|
    return !(h * h > (double)pad * (double)pad * nn);
}
// the group box no ray misses (device_scene.h: triG): -infinity .. +infinity on every axis, as three pairs of binary16
#define MRT_GROUP_EVERYWHERE_X 0xfc00fc00u
#define MRT_GROUP_EVERYWHERE_Y 0x7c00fc00u
#define MRT_GROUP_EVERYWHERE_Z 0x7c007c00u

MRT_HD uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MRT_HD float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ---- host-side matrix helpers (camera set-up happens once per frame on the host) --------
inline float& el(m4& m, int col, int row) { return (&m.c[col].x)[row]; }
inline float el(const m4& m, int col, int row) { return (&m.c[col].x)[row]; }
inline m4 identity() { return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}}; }
// detail/type_mat4x4.inl:676-687 -- (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline f4 mul(const m4& m, f4 v) {
    f4 a0 = m.c[0] * v.x + m.c[1] * v.y;
    f4 a1 = m.c[2] * v.z + m.c[3] * v.w;
    return a0 + a1;
}
// detail/type_mat4x4.inl:753-775
inline m4 mul(const m4& a, const m4& b) {
    m4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = ((a.c[0] * b.c[j].x + a.c[1] * b.c[j].y) + a.c[2] * b.c[j].z) + a.c[3] * b.c[j].w;
    return r;
}
// detail/type_mat3x3.inl:587-597 applied to the upper-left 3x3 of a mat4
MRT_HD f3 mul3(const m4& m, f3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
m4 inverse(const m4& m);            // detail/type_mat4x4.inl:477-535
m4 look_at(f3 eye, f3 center, f3 up);  // gtc/matrix_transform.inl:416-441

}  // namespace mrt
