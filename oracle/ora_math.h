// TEST INFRASTRUCTURE ONLY -- CPU oracle for the ray/scene intersection path.
// Nothing under oracle/ is part of the shipped product; only tests/, the smoke check and
// the cpu_baseline leg of bench.py may build, load or call it.
//
// Restatement of the vector arithmetic the reference gets from its vendored glm 0.9.5.4.
// Every function cites the glm source it follows (paths relative to
// /root/reference/Project/lib/glm/).  Compile with -ffp-contract=off and without
// -ffast-math / -march=native: operation ORDER is part of the contract.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ora {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { vec4 c[4]; };  // column-major, c[i] = column i (glm tmat4x4::value[i])

// detail/func_common.inl:367-399 -- min/max are compare+select, NOT fminf/fmaxf
static inline float gmin(float x, float y) { return x < y ? x : y; }
static inline float gmax(float x, float y) { return x > y ? x : y; }
static inline int   gmin(int x, int y) { return x < y ? x : y; }
static inline int   gmax(int x, int y) { return x > y ? x : y; }
// detail/func_common.inl:403-415
static inline float gclamp(float x, float lo, float hi) { return gmin(hi, gmax(lo, x)); }

// detail/type_vec3.inl:540-616 -- component-wise, operand order as written
static inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
static inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
static inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
static inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
static inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
// detail/type_vec4.inl:602-626
static inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
static inline vec4 operator*(vec4 a, vec4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
static inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }
static inline vec4 operator*(float s, vec4 a) { return {s * a.x, s * a.y, s * a.z, s * a.w}; }

static inline vec3 vmin(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
static inline vec3 vmax(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
static inline vec4 vclamp(vec4 a, float lo, float hi) {
    return {gclamp(a.x, lo, hi), gclamp(a.y, lo, hi), gclamp(a.z, lo, hi), gclamp(a.w, lo, hi)};
}

// detail/func_geometric.inl:66-73 -- products first, then (x+y)+z
static inline float dot(vec3 a, vec3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
// detail/func_geometric.inl:215-228
static inline vec3 cross(vec3 x, vec3 y) {
    return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
}
// detail/func_geometric.inl:256-265 with detail/func_exponential.inl:226-229 (1.0f / sqrt)
static inline vec3 normalize(vec3 v) {
    float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
    float r = 1.0f / sqrtf(sqr);
    return v * r;
}

// detail/type_mat4x4.inl:67-75 -- default-constructed mat4 is the identity
static inline mat4 mat4_identity() {
    return {{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}}};
}
// detail/type_mat4x4.inl:676-687 -- (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
static inline vec4 mul(const mat4& m, vec4 v) {
    vec4 mul0 = m.c[0] * v.x, mul1 = m.c[1] * v.y;
    vec4 add0 = mul0 + mul1;
    vec4 mul2 = m.c[2] * v.z, mul3 = m.c[3] * v.w;
    vec4 add1 = mul2 + mul3;
    return add0 + add1;
}
// detail/type_mat4x4.inl:753-775 -- ((A0*b0 + A1*b1) + A2*b2) + A3*b3 per column
static inline mat4 mul(const mat4& a, const mat4& b) {
    mat4 r;
    for (int j = 0; j < 4; ++j)
        r.c[j] = ((a.c[0] * b.c[j].x + a.c[1] * b.c[j].y) + a.c[2] * b.c[j].z) + a.c[3] * b.c[j].w;
    return r;
}
static inline float& el(mat4& m, int col, int row) { return (&m.c[col].x)[row]; }
static inline float el(const mat4& m, int col, int row) { return (&m.c[col].x)[row]; }

// detail/type_mat4x4.inl:477-535 (compute_inverse<tmat4x4>)
static inline mat4 inverse(const mat4& m) {
#define M(c, r) el(m, c, r)
    float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    float Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    float Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    float Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    float Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    float Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    float Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    float Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    float Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    float Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    float Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    float Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    float Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    vec4 Fac0{Coef00, Coef00, Coef02, Coef03}, Fac1{Coef04, Coef04, Coef06, Coef07};
    vec4 Fac2{Coef08, Coef08, Coef10, Coef11}, Fac3{Coef12, Coef12, Coef14, Coef15};
    vec4 Fac4{Coef16, Coef16, Coef18, Coef19}, Fac5{Coef20, Coef20, Coef22, Coef23};
    vec4 Vec0{M(1,0), M(0,0), M(0,0), M(0,0)}, Vec1{M(1,1), M(0,1), M(0,1), M(0,1)};
    vec4 Vec2{M(1,2), M(0,2), M(0,2), M(0,2)}, Vec3{M(1,3), M(0,3), M(0,3), M(0,3)};
#undef M
    vec4 Inv0 = (Vec1 * Fac0 - Vec2 * Fac1) + Vec3 * Fac2;
    vec4 Inv1 = (Vec0 * Fac0 - Vec2 * Fac3) + Vec3 * Fac4;
    vec4 Inv2 = (Vec0 * Fac1 - Vec1 * Fac3) + Vec3 * Fac5;
    vec4 Inv3 = (Vec0 * Fac2 - Vec1 * Fac4) + Vec2 * Fac5;
    vec4 SignA{+1, -1, +1, -1}, SignB{-1, +1, -1, +1};
    mat4 Inverse{{Inv0 * SignA, Inv1 * SignB, Inv2 * SignA, Inv3 * SignB}};
    vec4 Row0{Inverse.c[0].x, Inverse.c[1].x, Inverse.c[2].x, Inverse.c[3].x};
    vec4 Dot0 = m.c[0] * Row0;
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    return {{Inverse.c[0] * OneOverDeterminant, Inverse.c[1] * OneOverDeterminant,
             Inverse.c[2] * OneOverDeterminant, Inverse.c[3] * OneOverDeterminant}};
}

// gtc/matrix_transform.inl:416-441 (right-handed lookAt)
static inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 R = mat4_identity();
    el(R, 0, 0) = s.x;  el(R, 1, 0) = s.y;  el(R, 2, 0) = s.z;
    el(R, 0, 1) = u.x;  el(R, 1, 1) = u.y;  el(R, 2, 1) = u.z;
    el(R, 0, 2) = -f.x; el(R, 1, 2) = -f.y; el(R, 2, 2) = -f.z;
    el(R, 3, 0) = -dot(s, eye);
    el(R, 3, 1) = -dot(u, eye);
    el(R, 3, 2) = dot(f, eye);
    return R;
}

static inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

}  // namespace ora
