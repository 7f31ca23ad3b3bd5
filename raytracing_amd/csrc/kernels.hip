// HIP kernels of the ray path for gfx950 (CDNA4, 64-lane wavefronts).
//
// Mapping: one wavefront = one 8x8 pixel tile = one 64-ray packet of the reference
// (raytracer.h:6 PACKETSIZE, game.h:6 TILESIZE); lane k is ray k of the packet, i.e. pixel
// (tile.x + k%8, tile.y + k/8) exactly as threads.cpp:196-211 numbers them.
//
// Three traversals, all bit-faithful to the reference's results:
//   traverse_packet   64-ray "first active ray" packet traversal (primary rays of Render),
//                     wave-uniform control flow, node/triangle data fetched once per wave;
//   traverse_nearest  independent per-lane nearest-hit traversal (every ray of the recursive
//                     shader), per-lane stack in LDS, inner-node and leaf phases kept apart;
//   occluded_wave     shadow rays of one wave traced together: the reference only uses
//                     "was anything hit", which does not depend on visiting order, so the 64
//                     rays share one work list and each lane keeps its own activity bit.
// Scene data comes either from global memory (render_tiles_global: any size, one wave per tile,
// wave-uniform reads become scalar loads; used for the packet path and for large scenes) or
// from an LDS copy held by a persistent 16-wave workgroup (render_tiles_resident: recursive
// shader on scenes that fit, whose per-lane traversal reads a different address in every lane).
//
// All arithmetic that decides a hit or a colour follows the reference's operation order;
// this file must be compiled with -ffp-contract=off and without fast-math.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>

#include "device_scene.h"
#include "mrt_math.h"

namespace mrt {

#define MRT_BACKGROUND f4{1.0f, 0.3f, 0.3f, 0.3f}   // raytracer.cpp:5
#define MRT_AMBIENT    f4{1.0f, 0.0f, 0.0f, 0.0f}   // raytracer.cpp:6
#define MRT_LDS __attribute__((address_space(3)))
#ifndef MRT_XCD_RUN
#define MRT_XCD_RUN 16
#endif
constexpr int kXcds = 8;                            // accelerator complex dies of the MI355X, each with its own L2
constexpr int kXcdRun = MRT_XCD_RUN;                // consecutive tiles given to one XCD
constexpr float kEpsilon = 1e-4f;                   // raytracer.cpp:7
constexpr int kMaxReflPasses = 5, kMaxRefrPasses = 5;   // raytracer.cpp:9-10

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));

#define MRT_LEAF_CHUNK 16       // triangles of a leaf a lane tests before the wave looks at inner nodes again (at most 32)
#define MRT_GROUP_MIN kGroupMinLeaf

struct Hit { int tri; float u, v; };

// Loop budget shared by the outer traversal loops of one tile (work-list loops, the per-lane
// traversal's outer loop, shading rounds).  No correct traversal comes near it (it is sized from
// the node and triangle counts); if a loop ever exhausts it, the loop ends, a code is left in
// counters[5] for the host to report, and the kernel still terminates.  The innermost loops are
// not charged: a leaf's triangle loop is bounded by its count and a descent by the depth of the
// tree, which the library builds itself, and a check there costs 10-17 % of the frame time.
struct Watch {
    int left;
    unsigned long long* flag;
#ifdef MRT_LOOP_COUNTS
    uint32_t wave_n[kProfSites] = {}, lane_n[kProfSites] = {};
    unsigned long long phase_t0 = 0;
    // shader clocks of a phase of the wave's work, added to site `site` as its wave count (lane count = calls)
    __device__ __forceinline__ void phase_begin() { phase_t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void phase_end(int site) {
        const unsigned long long dt = __builtin_readcyclecounter() - phase_t0;
        const int me = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (me == 0) { wave_n[site] += (uint32_t)(dt >> 4); ++lane_n[site]; }
    }
    // one tick per wave-level execution of a site (counted on the first active lane) and one per lane
    __device__ __forceinline__ void tick(int site) {
        const int me = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        if (__ffsll((long long)__ballot(1)) - 1 == me) ++wave_n[site];
        ++lane_n[site];
    }
#define MRT_TICK(wd, site) (wd).tick(site)
#define MRT_PHASE_BEGIN(wd) (wd).phase_begin()
#define MRT_PHASE_END(wd, site) (wd).phase_end(site)
#else
#define MRT_TICK(wd, site) ((void)0)
#define MRT_PHASE_BEGIN(wd) ((void)0)
#define MRT_PHASE_END(wd, site) ((void)0)
#endif
    __device__ __forceinline__ bool expired(unsigned long long code) {
        if (--left >= 0) return false;
        if (flag) atomicOr(flag, code);
        return true;
    }
};
// rays of a frame through a scene's second tree that were traced again through the reference's (hit_is_the_references);
// must be called by all lanes the surrounding branch left active
__device__ __forceinline__ void count_retraced(const Watch& wd, bool mine) {
    const unsigned long long m = __ballot(mine);
    if (wd.flag && m && (int)__lane_id() == __ffsll((long long)m) - 1) atomicAdd(wd.flag + (kRedoCountSlot - 5), (unsigned long long)__popcll(m));
}

// ---------------------------------------------------------------------------------------
// Scene accessors: the traversals are written once and instantiated for both homes of the
// node / triangle arrays.  Indices are in units of the array element.
// ---------------------------------------------------------------------------------------
// G: the kernel was launched for a scene with group boxes (triG != nullptr).  A compile-time choice: kernels for
// scenes without them carry none of the code.
template <bool G>
struct SceneGlobalT {
    static constexpr bool kPrefetchLeaf = false;
    static constexpr bool kEyeRelative = false;
    static constexpr bool kShadowGroups = true;      // the shared shadow traversal consults the group boxes
    const v4f* nodes; const v4f* triA; const v4f* triB; const v2f* triC;
    const v4u* triG;      // group boxes (device_scene.h)
    const v4f* triM;      // the same triangles as one 48-byte record each (device_scene.h): one address per test
    const uint2* place;   // triangle id -> {place in the tree's own leaf order, first place of its leaf} (device_scene.h)
    const v4f* leaf_box = nullptr;      // (SceneDev::leaf_box: set when these arrays are a scene's second tree)
    float reach = 0.0f;                 // (SceneDev::reach, with leaf_box)
    static constexpr bool kMergedTris = true;
    __device__ __forceinline__ bool has_groups() const { return G; }
    __device__ __forceinline__ v4u group(uint32_t g) const { return triG[g]; }
    __device__ __forceinline__ v4f node(uint32_t i) const { return nodes[i]; }
    // the two children of an inner node (slots L and L + 1): four records read through ONE pointer, so that the
    // compiler sees 64 consecutive bytes (one wide load) instead of four indices that might wrap
    __device__ __forceinline__ const v4f* siblings(uint32_t L) const { return nodes + 2 * (size_t)L; }
    __device__ __forceinline__ v4f tri_a(uint32_t k) const { return triA[k]; }
    __device__ __forceinline__ v4f tri_b(uint32_t k) const { return triB[k]; }
    __device__ __forceinline__ v2f tri_c(uint32_t k) const { return triC[k]; }
    __device__ __forceinline__ v2f tri_b_lo(uint32_t k) const { return reinterpret_cast<const v2f*>(triB)[2 * k]; }
    __device__ __forceinline__ v2f tri_b_hi(uint32_t k) const { return reinterpret_cast<const v2f*>(triB)[2 * k + 1]; }
    __device__ __forceinline__ uint32_t node_first(uint32_t n) const { return __float_as_uint(reinterpret_cast<const float*>(nodes)[8 * n + 3]); }
    __device__ __forceinline__ uint32_t node_count(uint32_t n) const { return __float_as_uint(reinterpret_cast<const float*>(nodes)[8 * n + 7]); }
};
// The scene as the frame's primary rays see it.  All primary rays start at the eye, so everything the packet
// traversal computes from the origin and the scene alone is the same for every ray of every tile:
// bound - O of the box tests (raytracer.cpp:1408-1409) and, of the triangle test (raytracer.cpp:287-356),
// tvec = O - p0, qvec = cross(tvec, e1) and the numerator dot(e2, qvec).  eye_relative_kernel evaluates them
// once per frame with the same operations in the same order; the packet traversal of the frame kernel reads
// them (scalar loads, like the nodes and triangles themselves) instead of issuing those subtractions,
// the cross product and the dot product in every wave:
//   eyeNodes[2i], [2i+1] = {min - O, bits(firstLeft)}, {max - O, bits(count)}
//   eyeTri[4k .. 4k+3] = {e1.xyz, e2.x}, {e2.yz, tvec.xy}, {tvec.z, dot(e2, qvec), qvec.xy}, {qvec.z, bits(triangle id), 0, 0}
//                        (one 64-byte record per triangle: one address, the first half decides "faces away")
//   eyeG[2g], [2g+1] = {lo - O, 0}, {hi - O, 0} of group box g (device_scene.h: triG)
template <bool G>
struct SceneGlobalEyeT : SceneGlobalT<G> {
    static constexpr bool kEyeRelative = true;
    const v4f* eyeNodes; const v4f* eyeTri;
    const v4f* eyeG;      // [2g], [2g+1] = the group box's corners minus O (nullptr: no group boxes)
};
// A scene's second tree as the packet path of the one-wave-per-tile kernel sees it (no group boxes), with the
// reference's own tree beside it (hit_is_the_references); GR: the reference's tree has group boxes.
template <bool GR>
struct SceneTwoTreesT : SceneGlobalEyeT<false> {
    SceneGlobalEyeT<GR> ref;
};
struct SceneLds {
    static constexpr bool kPrefetchLeaf = true;     // request triangle k+1 before testing triangle k
    static constexpr bool kEyeRelative = false;
    static constexpr bool kShadowGroups = true;
    MRT_LDS const v4f* nodes; MRT_LDS const v4f* triA; MRT_LDS const v4f* triB; MRT_LDS const v2f* triC;
    MRT_LDS const v4u* triG; bool groups;
    const uint2* place;   // (global memory: read only when two hits are equally far)
    const v4f* leaf_box;  // (global memory: SceneDev::leaf_box, or null)
    float reach;          // (SceneDev::reach, with leaf_box)
    SceneGlobalT<false> ref;      // with leaf_box: the reference's own tree, in global memory (SceneDev::ref_nodes ...)
    static constexpr bool kMergedTris = false;
    __device__ __forceinline__ bool has_groups() const { return groups; }
    __device__ __forceinline__ v4u group(uint32_t g) const { return triG[g]; }
    __device__ __forceinline__ v4f node(uint32_t i) const { return nodes[i]; }
    __device__ __forceinline__ MRT_LDS const v4f* siblings(uint32_t L) const { return nodes + 2 * L; }
    __device__ __forceinline__ v4f tri_a(uint32_t k) const { return triA[k]; }
    __device__ __forceinline__ v4f tri_b(uint32_t k) const { return triB[k]; }
    __device__ __forceinline__ v2f tri_c(uint32_t k) const { return triC[k]; }
    __device__ __forceinline__ v2f tri_b_lo(uint32_t k) const { return ((MRT_LDS const v2f*)triB)[2 * k]; }
    __device__ __forceinline__ v2f tri_b_hi(uint32_t k) const { return ((MRT_LDS const v2f*)triB)[2 * k + 1]; }
    __device__ __forceinline__ uint32_t node_first(uint32_t n) const { return __float_as_uint(((MRT_LDS const float*)nodes)[8 * n + 3]); }
    __device__ __forceinline__ uint32_t node_count(uint32_t n) const { return __float_as_uint(((MRT_LDS const float*)nodes)[8 * n + 7]); }
};

__device__ __forceinline__ f3 xyz(v4f v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ f3 xyz(float4 v) { return {v.x, v.y, v.z}; }
__device__ __forceinline__ uint32_t wbits(v4f v) { return __float_as_uint(v.w); }
__device__ __forceinline__ float lane_bcast(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

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
// Same terms when no operand can be a NaN: hardware min/max then agree with compare+select
// except for the sign of a zero, which none of the predicates below can see.
// The instructions are named directly: fminf / fmaxf make the compiler quiet possible signalling
// NaNs first (one extra v_max x,x per operand), which cannot occur here.
__device__ __forceinline__ float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ void slab_terms_finite(f3 tMin, f3 tMax, float& tNear, float& tFar) {
    tNear = hw_max3(hw_min(tMin.x, tMax.x), hw_min(tMin.y, tMax.y), hw_min(tMin.z, tMax.z));
    tFar = hw_min3(hw_max(tMin.x, tMax.x), hw_max(tMin.y, tMax.y), hw_max(tMin.z, tMax.z));
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

// Per-ray constants of the single-ray traversal, whose child tests divide by D
// (raytracer.cpp:1227, 1233).  r = 1/D lets the twelve divisions of an inner node be replaced
// by div_by_recip (mrt_math.h) when `shortcut` holds; otherwise true divisions are issued.
struct RayDiv {
    f3 O, D, r;
    bool shortcut;
};
__device__ __forceinline__ RayDiv make_ray_div(f3 O, f3 D, bool coords_ok) {
    RayDiv R;
    R.O = O; R.D = D;
    R.r = 1.0f / D;
    R.shortcut = coords_ok && div_shortcut_dir_ok(D.x) && div_shortcut_dir_ok(D.y) && div_shortcut_dir_ok(D.z) &&
                 div_shortcut_coord_ok(O.x) && div_shortcut_coord_ok(O.y) && div_shortcut_coord_ok(O.z);
    return R;
}
__device__ __forceinline__ f3 div3(f3 a, const RayDiv& R) {
    return {div_by_recip(a.x, R.D.x, R.r.x), div_by_recip(a.y, R.D.y, R.r.y), div_by_recip(a.z, R.D.z, R.r.z)};
}
// Both children of an inner node.  `shortcut` must be uniform over the lanes that call this.
__device__ __forceinline__ void slab_div_pair(v4f l0, v4f l1, v4f r0, v4f r1, const RayDiv& R, bool shortcut, float t,
                                              bool& leftN, bool& rightN, float& tNear, float& tNear2) {
    float tFar, tFar2;
    if (shortcut) {
        slab_terms_finite(div3(xyz(l0) - R.O, R), div3(xyz(l1) - R.O, R), tNear, tFar);
        slab_terms_finite(div3(xyz(r0) - R.O, R), div3(xyz(r1) - R.O, R), tNear2, tFar2);
    } else {
        slab_terms((xyz(l0) - R.O) / R.D, (xyz(l1) - R.O) / R.D, tNear, tFar);
        slab_terms((xyz(r0) - R.O) / R.D, (xyz(r1) - R.O) / R.D, tNear2, tFar2);
    }
    leftN = slab_pass(tNear, tFar, t);
    rightN = slab_pass(tNear2, tFar2, t);
}

// Both children of an inner node of a scene's SECOND tree (hit_is_the_references): its boxes stand back from their
// triangles by 2^-16 of the scene's largest coordinate and need not reproduce the reference's arithmetic -- only never
// dismiss a box that holds a hit -- so a quotient (b - O) / D is one fused multiply-add, b * r - ro with r = 1 / D and
// ro = O * r: off by 2^-24 (|O| |r| + |quotient|) at most, while the boxes stand back by 2^-16 of the scene's largest
// coordinate, times |r| in the quotient -- sixty times the error for an origin within SceneDev::reach (64 scene sizes).
// The caller sees to that (within_reach) and to r being an ordinary number (make_ray_second_tree).  12 operations for
// the twelve quotients instead of 84.
__device__ __forceinline__ void slab_mul_pair(v4f l0, v4f l1, v4f r0, v4f r1, const RayDiv& R, f3 ro, float t,
                                              bool& leftN, bool& rightN, float& tNear, float& tNear2) {
    float tFar, tFar2;
    auto q = [&](v4f b) { return f3{__builtin_fmaf(b.x, R.r.x, -ro.x), __builtin_fmaf(b.y, R.r.y, -ro.y), __builtin_fmaf(b.z, R.r.z, -ro.z)}; };
    slab_terms_finite(q(l0), q(l1), tNear, tFar);
    slab_terms_finite(q(r0), q(r1), tNear2, tFar2);
    leftN = slab_pass(tNear, tFar, t);
    rightN = slab_pass(tNear2, tFar2, t);
}

// The ray constants for that: a reciprocal that is an ordinary number on every axis.  Where 1 / D is infinite or beyond
// 2^60 (the ray runs along the axis' slabs: it moves less than 2^-52 scene sizes across them on its way through the
// scene) +-2^60 stands in: a box whose slab holds the origin, with the margin its faces stand back by, then spans
// parameters from below -2^44 to above 2^44 scene sizes on that axis, more than any hit's distance, and a slab that
// does not hold the origin can hold no hit of this ray.  No product overflows for coordinates up to 2^30 (the second
// tree is only kept for such scenes: mrt_scene_upload_lbvh_exact) and origins within SceneDev::reach.
__device__ __forceinline__ float second_tree_recip(float d) {
    const float r = 1.0f / d;
    return fabsf(r) <= 0x1p60f ? r : __builtin_copysignf(0x1p60f, d);
}
__device__ __forceinline__ RayDiv make_ray_second_tree(f3 O, f3 D) {
    RayDiv R;
    R.O = O; R.D = D;
    R.r = {second_tree_recip(D.x), second_tree_recip(D.y), second_tree_recip(D.z)};
    R.shortcut = true;
    return R;
}
// may a ray from O walk the second tree (slab_mul_pair)?
template <class SC>
__device__ __forceinline__ bool within_reach(const SC& sc, f3 O) { return fabsf(O.x) <= sc.reach && fabsf(O.y) <= sc.reach && fabsf(O.z) <= sc.reach; }

// The two child tests of a shadow ray (t = INFINITY, so `tNear < t` always holds) decided from
// interval bounds, for rays and scenes inside the ranges of the exact-division shortcut.
// q~ = a * RN(1/d) and the reference's q = RN(a / d) are within 2^-23 resp. 2^-24 of a/d
// relatively, so |q - q~| <= 2^-22 |q~|, and zero and sign are preserved exactly.  g(x) = x - e|x|
// and h(x) = x + e|x| are increasing and commute with min and max, so the reference's tNear lies in
// [g(tNear~), h(tNear~)] and its tFar in [g(tFar~), h(tFar~)] (e = 2^-21 leaves room for the
// rounding of g and h themselves), and tFar > 0 iff tFar~ > 0.  Returns false for a lane whose
// `tFar >= tNear` falls inside the bounds; the caller then evaluates slab_div_pair for the wave.
__device__ __forceinline__ bool slab_pair_shadow_interval(v4f l0, v4f l1, v4f r0, v4f r1, const RayDiv& R, bool& leftN, bool& rightN, float& nL, float& nR) {
    constexpr float e = 0x1p-21f;
    float fL, fR;
    slab_terms_finite((xyz(l0) - R.O) * R.r, (xyz(l1) - R.O) * R.r, nL, fL);
    slab_terms_finite((xyz(r0) - R.O) * R.r, (xyz(r1) - R.O) * R.r, nR, fR);
    const float nLlo = __builtin_fmaf(-e, fabsf(nL), nL), nLhi = __builtin_fmaf(e, fabsf(nL), nL);
    const float fLlo = __builtin_fmaf(-e, fabsf(fL), fL), fLhi = __builtin_fmaf(e, fabsf(fL), fL);
    const float nRlo = __builtin_fmaf(-e, fabsf(nR), nR), nRhi = __builtin_fmaf(e, fabsf(nR), nR);
    const float fRlo = __builtin_fmaf(-e, fabsf(fR), fR), fRhi = __builtin_fmaf(e, fabsf(fR), fR);
    leftN = (fLlo >= nLhi) && (fL > 0.0f);
    rightN = (fRlo >= nRhi) && (fR > 0.0f);
    const bool l_no = (fLhi < nLlo) || !(fL > 0.0f), r_no = (fRhi < nRlo) || !(fR > 0.0f);
    return (leftN || l_no) && (rightN || r_no);
}

// Two triangles hit at exactly the same distance.  The reference tests the triangles of a leaf in the order of its
// index array and keeps a hit only if it is strictly nearer (raytracer.cpp:1336-1344, 1452-1461): of two equally far
// ones the one that comes FIRST in the leaf is kept, and a hit as far as one kept from an earlier leaf is dropped.
// The device arrays hold the triangles of a long leaf in another order (device_scene.h), so the rule is applied by
// name: place[id] = {the triangle's place in the reference's order, the first place of its leaf}.
// place == nullptr: only "any hit" is asked.
// unsure (scenes walked through another tree than the reference's, see hit_is_the_references): the lane's flag, set when
// it has seen two candidates within 16 units in the last place of each other; null elsewhere.
struct Tie {
    const uint2* place;
    bool* unsure = nullptr;
};

// A scene may be walked through ANOTHER tree than the reference's (SceneDev::leaf_box; EXACT in the functions below):
// boxes that stand back a little (so that no triangle is hidden from a ray that passes its test), visited in any
// order, and dismissed by distance only when they begin clearly beyond the nearest hit so far (kExactCull).  Such a
// walk tests every triangle that is as near as the nearest one, so it finds the smallest distance T of all triangles
// whose test the ray passes, and the triangle W there.  The reference finds the same unless its own boxes, with their
// rounding, keep it from W or the order of its visits matters.  Neither is the case when
//   (a) no other triangle's hit lies within 16 units in the last place of T (tri_decide records such pairs), and
//   (b) the ray passes the box of W's LEAF in the reference's tree by the reference's own arithmetic (tFar >= tNear and
//       tFar > 0), entering it no later than 8 units in the last place after T, with no quotient 0/0:
// the boxes of W's ancestors contain the leaf's, and the slab quotients are monotone in the box's bounds under correct
// rounding, so by (b) every ancestor passes the two tests that do not involve the running distance; that distance is,
// until W is tested, above T by more than 16 units (a), so `tNear < t` holds on the whole path (b), W is tested, is
// strictly nearer than everything found before, and nothing found later is as near (a).  A ray that fails the check
// sets its lane's `unsure` flag and is traced once more, through the reference's own tree in global memory
// (shade_recursive), which by construction gives the reference's answer.
// A shadow ray only asks whether the reference reaches SOME triangle the ray hits (occluded_wave), and its running
// distance stays infinite until it does: the two tests of (b) decide that for every triangle by itself -- a leaf the
// ray misses by the reference's arithmetic is never entered, one it passes is reached through ancestors that pass too
// -- so a hit on a triangle of a leaf the ray misses is dropped and the walk goes on; only 0/0 leaves it undecided
// (then the lane's shadow ray too goes through the reference's tree: direct_light).
constexpr float kExactCull = 1.0f + 0x1p-10f;
// Test build (make retrace: libmrt_retrace.so): EVERY ray of a frame through a second tree counts as unsure, so the walks
// through the reference's tree that normally serve a handful of rays per frame render the whole frame.
#ifdef MRT_FORCE_RETRACE
constexpr bool kForceRetrace = true;
#else
constexpr bool kForceRetrace = false;
#endif
// undecided: a quotient of the leaf's box test is 0/0 (the ray runs inside a face of the box)
// PACKET: the box test as the packet traversal computes it (raytracer.cpp:1372-1376: products with the ray's reciprocal
// direction rD, passed in D's place; 0 * inf is its 0/0).  The argument is the same: a ray that passes the leaf's box
// passes its ancestors', so every node on the way is entered for a ray no later than this one, and the leaf's
// triangles are tested for every ray from that one on (raytracer.cpp:1445-1464).
template <bool PACKET = false>
__device__ __forceinline__ bool hit_is_the_references(const v4f* leaf_box, int tri, f3 O, f3 D, float T, bool nearest, bool& undecided) {
    const v4f lo = leaf_box[2 * (size_t)tri], hi = leaf_box[2 * (size_t)tri + 1];
    bool nan_x, nan_y, nan_z;
    if (PACKET) {
        nan_x = !(fabsf(D.x) < INFINITY) && (lo.x == O.x || hi.x == O.x); nan_y = !(fabsf(D.y) < INFINITY) && (lo.y == O.y || hi.y == O.y);
        nan_z = !(fabsf(D.z) < INFINITY) && (lo.z == O.z || hi.z == O.z);
    } else {
        nan_x = D.x == 0.0f && (lo.x == O.x || hi.x == O.x); nan_y = D.y == 0.0f && (lo.y == O.y || hi.y == O.y);
        nan_z = D.z == 0.0f && (lo.z == O.z || hi.z == O.z);
    }
    undecided = nan_x || nan_y || nan_z;
    float tNear, tFar;
    if (PACKET) slab_terms((xyz(lo) - O) * D, (xyz(hi) - O) * D, tNear, tFar);
    else slab_terms((xyz(lo) - O) / D, (xyz(hi) - O) / D, tNear, tFar);
    bool ok = !undecided && (tFar >= tNear) && (tFar > 0);
    if (nearest) ok = ok && (tNear <= T || (uint32_t)(__float_as_int(tNear) - __float_as_int(T)) <= 8u);
    return ok;
}
__device__ __forceinline__ bool tie_goes_to(const Tie& tie, uint32_t id, int kept) {
    if (!tie.place || kept < 0) return false;
    const uint2 mine = tie.place[id], theirs = tie.place[kept];
    return mine.y == theirs.y && mine.x < theirs.x;
}

// Second half of the triangle test: from the determinant and the three numerators to the running hit.
template <bool WAVE_REJECT>
__device__ __forceinline__ void tri_decide(float det, float un, float vn, float tn, uint32_t id, float& t, Hit& h, Watch& wd, const Tie& tie) {
    if (WAVE_REJECT) {
        constexpr float m = 1e-5f;      // far above the 2^-22 relative error of estimate * numerator
        const float rc = __builtin_amdgcn_rcpf(det);
        const float ua = un * rc, va = vn * rc, ta = tn * rc;
        bool clear_miss = (det < -kEpsilon) || (ua < -m) || (ua > 1.0f + m) || (va < -m) || (ua + va > 1.0f + 2.0f * m) ||
                          (ta < -m) || (ta > t * (1.0f + m));
        clear_miss = clear_miss && (fabsf(det) >= 1e-30f);   // tiny determinants: always decide exactly
        if (!__any(!clear_miss)) return;
    }
    MRT_TICK(wd, WAVE_REJECT ? 7 : 13);
    // The reference's chain of early returns with wave-uniform branches: `live` carries each lane's
    // own outcome, a stage is skipped only when no lane is left, and the running hit is updated by
    // selects at one place.
    bool live = !(det < -kEpsilon);
    if (!__any(live)) return;
    const float invDet = 1 / det;
    const float u = un * invDet;
    live = live && !(u < 0 || u > 1);
    if (!__any(live)) return;
    const float v = vn * invDet;
    live = live && !(v < 0 || u + v > 1);
    if (!__any(live)) return;
    const float T = tn * invDet;
    if (tie.unsure) *tie.unsure = *tie.unsure || (live && !(T < 0.0f) && (uint32_t)(__float_as_int(T) - __float_as_int(t)) + 16u <= 32u);
    bool nearer = T < t;
    if (live && T == t) nearer = tie_goes_to(tie, id, h.tri);
    live = live && !(T < 0.0f) && nearer;
    t = live ? T : t;
    h.tri = live ? (int)id : h.tri;
    h.u = live ? u : h.u;
    h.v = live ? v : h.v;
}

// ---------------------------------------------------------------------------------------
// Ray/triangle: Triangle::Intersect / Intersect2 (raytracer.cpp:287-356).
// WAVE_REJECT: the lanes calling this test the SAME triangle (packet leaves, shared shadow
// traversal).  The exact test needs a correctly rounded 1/det; a hardware reciprocal
// estimate is enough to see that a ray misses by a clear margin, and the exact division is
// issued only when some lane cannot be ruled out that way.
// ---------------------------------------------------------------------------------------
template <bool WAVE_REJECT>
__device__ __forceinline__ void tri_test(v4f A, v4f B, v2f C, f3 O, f3 D, float& t, Hit& h, Watch& wd, const Tie& tie) {
    const f3 e1 = {A.x, A.y, A.z}, e2 = {A.w, B.x, B.y}, p0 = {B.z, B.w, C.x};
    const f3 pvec = cross(D, e2);
    const float det = dot(e1, pvec);
    if (WAVE_REJECT) {
        if (!__any(!(det < -kEpsilon))) return;          // back-facing for every ray of the wave
    }
    MRT_TICK(wd, WAVE_REJECT ? 6 : 12);
    const f3 tvec = O - p0;
    const float un = dot(tvec, pvec);
    const f3 qvec = cross(tvec, e1);
    const float vn = dot(D, qvec);
    const float tn = dot(e2, qvec);
    tri_decide<WAVE_REJECT>(det, un, vn, tn, __float_as_uint(C.y), t, h, wd, tie);
}

// A triangle all calling lanes test together, read the way the scene's home allows: one record, or three arrays.
template <class SC>
__device__ __forceinline__ void tri_test_shared(const SC& sc, uint32_t k, f3 O, f3 D, float& t, Hit& h, Watch& wd, const Tie& tie) {
    if constexpr (SC::kMergedTris) {
        const v4f* rec = sc.triM + 3 * (size_t)k;
        const v4f C = rec[2];
        tri_test<true>(rec[0], rec[1], v2f{C.x, C.y}, O, D, t, h, wd, tie);
    } else {
        tri_test<true>(sc.tri_a(k), sc.tri_b(k), sc.tri_c(k), O, D, t, h, wd, tie);
    }
}

// The same test for a ray that starts at the eye, from the frame's precomputed record (SceneGlobalEyeT: eyeTri).
__device__ __forceinline__ void tri_test_eye(const v4f* rec, f3 D, float& t, Hit& h, Watch& wd, const Tie& tie) {
    const v4f A = rec[0], B = rec[1];
    const f3 e1 = {A.x, A.y, A.z}, e2 = {A.w, B.x, B.y};
    const f3 pvec = cross(D, e2);
    const float det = dot(e1, pvec);
    if (!__any(!(det < -kEpsilon))) return;
    MRT_TICK(wd, 6);
    const v4f C = rec[2], E = rec[3];
    const float un = dot(f3{B.z, B.w, C.x}, pvec);
    const float vn = dot(D, f3{C.z, C.w, E.x});
    tri_decide<true>(det, un, vn, C.y, __float_as_uint(E.y), t, h, wd, tie);
}

// ---------------------------------------------------------------------------------------
// A small wave-uniform stack kept in VGPR lanes: entry j is lane j of each register.
// ---------------------------------------------------------------------------------------
struct LaneStack1 {
    uint32_t a = 0;
    int sp = 0;
    __device__ __forceinline__ void push(int lane, uint32_t x) { a = (lane == sp) ? x : a; ++sp; }
    __device__ __forceinline__ uint32_t pop() { --sp; return (uint32_t)__builtin_amdgcn_readlane((int)a, sp); }
};
// Entries of the shared shadow traversal: a node as its two words {firstLeft, count} (what the traversal needs of the
// node itself, known from its parent's read: taking an entry is then one memory round trip, to the children or the
// triangles, instead of two) and the lanes whose rays reach it.
struct LaneStack4 {
    uint32_t a = 0, b = 0, c = 0, d = 0;
    int sp = 0;
    __device__ __forceinline__ void push(int lane, uint32_t first, uint32_t count, unsigned long long m) {
        const bool me = lane == sp;
        a = me ? first : a; b = me ? count : b; c = me ? (uint32_t)m : c; d = me ? (uint32_t)(m >> 32) : d;
        ++sp;
    }
    __device__ __forceinline__ void pop(uint32_t& first, uint32_t& count, unsigned long long& m) {
        --sp;
        first = (uint32_t)__builtin_amdgcn_readlane((int)a, sp);
        count = (uint32_t)__builtin_amdgcn_readlane((int)b, sp);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)c, sp), hi = (uint32_t)__builtin_amdgcn_readlane((int)d, sp);
        m = ((unsigned long long)hi << 32) | lo;
    }
};

// Can the ray O + s / r (r = 1/D, every component finite and non-zero) with 0 <= s <= t touch the group box G
// (device_scene.h: triG)?  Approximate arithmetic: the box's padding is four thousand times its rounding errors.
// Skipping a group is exact for every ray whose triangle tests are well conditioned: the test then places the hit
// within a ten-thousandth of the triangle's size, far inside the box's padding.  The exception is a ray that runs
// IN the plane of a triangle it passes beside (within about 1e-5 rad: harness in tests/native/check_groups.cpp): all
// numerators of its test are rounding noise, the reference's verdict for it is noise too, and a box cannot
// reproduce noise.  Rays that start on a surface meet that case with the surface's own coplanar neighbours, so a
// secondary or shadow ray within a thousandth of its surface's plane skips nothing.  Primary rays meet it when the
// eye lies in a triangle's plane: the frame then hands out the box no ray misses for that triangle's group
// (eye_grazes_group below).
__device__ __forceinline__ bool not_grazing(f3 D, f3 surface_normal) { return fabsf(dot(D, surface_normal)) >= 1e-3f; }

__device__ __forceinline__ void group_corners(v4u G, f3& lo, f3& hi) {
    const uint32_t w0 = G.x, w1 = G.y, w2 = G.z;      // (a bit cast taken straight from a vector element reads element 0)
    const v2h a = __builtin_bit_cast(v2h, w0), b = __builtin_bit_cast(v2h, w1), c = __builtin_bit_cast(v2h, w2);
    lo = {(float)a.x, (float)a.y, (float)b.x};
    hi = {(float)b.y, (float)c.x, (float)c.y};
}
// lo, hi: the box's corners relative to the ray's origin
__device__ __forceinline__ bool group_touched_rel(f3 lo, f3 hi, f3 r, float t) {
    const f3 t0 = lo * r, t1 = hi * r;
    const float tn = hw_max3(hw_min(t0.x, t1.x), hw_min(t0.y, t1.y), hw_min(t0.z, t1.z));
    const float tf = hw_min3(hw_max(t0.x, t1.x), hw_max(t0.y, t1.y), hw_max(t0.z, t1.z));
    return tn <= tf && tf >= 0.0f && tn <= t;
}
__device__ __forceinline__ bool group_touched(v4u G, f3 O, f3 r, float t) {
    f3 lo, hi;
    group_corners(G, lo, hi);
    return group_touched_rel(lo - O, hi - O, r, t);
}

// A leaf that all calling lanes walk together (packet leaves, the shared shadow traversal), group by group: the
// four triangles of a group are skipped when no calling lane's ray touches the group's box.  tri(k) tests triangle k
// and returns true when the walk can stop.  r_ok: this lane's reciprocal direction is an ordinary number on every
// axis (a lane without that never lets a group be skipped).  first, count: wave-uniform.
template <bool EYE, class SC, class TRI>
__device__ __forceinline__ void walk_leaf_by_groups(const SC& sc, uint32_t first, uint32_t count, f3 O, f3 r, bool r_ok, const float& t, Watch& wd, TRI&& tri) {
    const uint32_t end = first + count;
    for (uint32_t k = first; k < end;) {
        const uint32_t gend = ((k | 3u) + 1u) < end ? ((k | 3u) + 1u) : end;
        if (gend - k >= 2u) {
            MRT_TICK(wd, 21);
            bool touched;
            if constexpr (EYE) {
                const v4f* gbox = sc.eyeG + 2 * (size_t)(k >> 2);
                const v4f glo = gbox[0], ghi = gbox[1];
                touched = group_touched_rel(xyz(glo), xyz(ghi), r, t);
            } else {
                touched = group_touched(sc.group(k >> 2), O, r, t);
            }
            if (!__any(!r_ok || touched)) { k = gend; continue; }
        }
        for (; k < gend; ++k) if (tri(k)) return;
    }
}

// ---------------------------------------------------------------------------------------
// 64-ray packet traversal: BVH::TraverseRayPacket, BVHNode::TraversePacket and
// IntersectPrimitivesPacket (raytracer.cpp:1349-1464).  Must be called by all 64 lanes.
//
// The reference recursion visit(node, firstActive) is unrolled into a descent plus a work
// list of deferred "scan" tasks (child, f): find the first lane i >= f whose ray passes the
// child's box with its CURRENT t, then visit the child with firstActive = i.  Visiting an
// inner node loads both children once, orders them by lane i's entry distance, defers the
// far child (it must be re-tested after the near subtree has shortened t) and scans the near
// child at once from the terms already computed.  Visiting a leaf tests every lane >= i
// against every triangle of the leaf, in leaf order.
// ---------------------------------------------------------------------------------------
// EXACT: sc is another tree than the reference's (hit_is_the_references): boxes are dismissed by distance only when they
// begin clearly beyond a ray's nearest hit so far, and near-equal candidates set the lane's `unsure`.
template <bool FINITE, bool EXACT, class SC>
__device__ __forceinline__ void traverse_packet_impl(const SC& sc, f3 O, f3 D, f3 rD, float& t, Hit& h, int lane, Watch& wd, bool* unsure) {
    auto reach = [&]() { return EXACT ? t * kExactCull : t; };
    // box(i): node record i as the box tests want it; with the frame's eye-relative copy the subtraction of
    // the origin has already been done
    auto box = [&](uint32_t i) -> v4f {
        if constexpr (SC::kEyeRelative) return sc.eyeNodes[i]; else return sc.node(i);
    };
    auto terms = [&](f3 bmin, f3 bmax, float& tn, float& tf) {
        if constexpr (SC::kEyeRelative) {
            if (FINITE) slab_terms_finite(bmin * rD, bmax * rD, tn, tf);
            else slab_terms(bmin * rD, bmax * rD, tn, tf);
        } else {
            if (FINITE) slab_terms_finite((bmin - O) * rD, (bmax - O) * rD, tn, tf);
            else slab_terms((bmin - O) * rD, (bmax - O) * rD, tn, tf);
        }
    };
    // the group boxes of long leaves are tested with the reciprocal of D itself (a caller of the packet operator may
    // hand over any rD)
    const f3 rg = 1.0f / D;
    const bool rg_ok = fabsf(rg.x) < INFINITY && fabsf(rg.y) < INFINITY && fabsf(rg.z) < INFINITY;
    LaneStack1 work;
    work.push(lane, 0u);                       // scan(root, f = 0): raytracer.cpp:1351-1357
    while (work.sp > 0) {
        if (wd.expired(1)) break;
        MRT_TICK(wd, 8);
        const uint32_t e = work.pop();
        uint32_t node = e >> 6;
        const int f = (int)(e & 63u);
        v4f n0, n1;
        if constexpr (SC::kEyeRelative) { const v4f* p = sc.eyeNodes + 2 * (size_t)node; n0 = p[0]; n1 = p[1]; }
        else { n0 = box(2 * node); n1 = box(2 * node + 1); }
        float tn, tf;
        terms(xyz(n0), xyz(n1), tn, tf);
        const bool pass = slab_pass(tn, tf, reach());
        unsigned long long mask = __ballot(pass) & (~0ull << f);
        if (mask == 0) continue;
        int i = __ffsll((long long)mask) - 1;
        uint32_t first = wbits(n0), count = wbits(n1);
        for (;;) {                             // visit(node = {first,count}, firstActive = i)
            if (count != 0) {
                if (lane >= i) {
                    const Tie tie{sc.place, EXACT ? unsure : nullptr};
                    if (sc.has_groups() && count >= (uint32_t)MRT_GROUP_MIN) {
                        walk_leaf_by_groups<SC::kEyeRelative>(sc, first, count, O, rg, rg_ok, t, wd, [&](uint32_t k) {
                            MRT_TICK(wd, 10);
                            if constexpr (SC::kEyeRelative) tri_test_eye(sc.eyeTri + 4 * (size_t)k, D, t, h, wd, tie);
                            else tri_test_shared(sc, k, O, D, t, h, wd, tie);
                            return false;
                        });
                    } else if constexpr (SC::kEyeRelative) {
                        for (uint32_t k = first; k < first + count; ++k) {
                            MRT_TICK(wd, 10);
                            tri_test_eye(sc.eyeTri + 4 * (size_t)k, D, t, h, wd, tie);
                        }
                    } else if constexpr (SC::kPrefetchLeaf) {
                        v4f A = sc.tri_a(first), B = sc.tri_b(first);
                        v2f C = sc.tri_c(first);
                        const uint32_t last = first + count - 1;
                        for (uint32_t k = first; k <= last; ++k) {
                            MRT_TICK(wd, 10);
                            const uint32_t kn = k < last ? k + 1 : k;
                            const v4f An = sc.tri_a(kn), Bn = sc.tri_b(kn);
                            const v2f Cn = sc.tri_c(kn);
                            tri_test<true>(A, B, C, O, D, t, h, wd, tie);
                            A = An; B = Bn; C = Cn;
                        }
                    } else {
                        for (uint32_t k = first; k < first + count; ++k) {
                            MRT_TICK(wd, 10);
                            tri_test_shared(sc, k, O, D, t, h, wd, tie);
                        }
                    }
                }
                break;
            }
            MRT_TICK(wd, 9);
            const uint32_t L = first, R = first + 1;
            v4f l0, l1, r0, r1;
            if constexpr (SC::kEyeRelative) { const v4f* p = sc.eyeNodes + 2 * (size_t)L; l0 = p[0]; l1 = p[1]; r0 = p[2]; r1 = p[3]; }
            else { const auto p = sc.siblings(L); l0 = p[0]; l1 = p[1]; r0 = p[2]; r1 = p[3]; }
            float tnl, tfl, tnr, tfr;
            terms(xyz(l0), xyz(l1), tnl, tfl);
            terms(xyz(r0), xyz(r1), tnr, tfr);
            const bool right_first = lane_bcast(tnr, i) < lane_bcast(tnl, i);   // raytracer.cpp:1384
            work.push(lane, ((right_first ? L : R) << 6) | (uint32_t)i);
            const bool p = right_first ? slab_pass(tnr, tfr, reach()) : slab_pass(tnl, tfl, reach());
            mask = __ballot(p) & (~0ull << i);
            if (mask == 0) break;
            i = __ffsll((long long)mask) - 1;
            first = right_first ? wbits(r0) : wbits(l0);
            count = right_first ? wbits(r1) : wbits(l1);
        }
    }
}

template <bool EXACT = false, class SC>
__device__ __forceinline__ void traverse_packet(const SC& sc, int n_tris, f3 O, f3 D, f3 rD, float& t, Hit& h, int lane, Watch& wd, bool* unsure = nullptr) {
    if (n_tris == 0) return;
    // With every 1/D component finite, (bound - O) * rD is never a NaN, and hardware min/max
    // then give the predicates the same values as glm's compare+select (up to the sign of a
    // zero, which no comparison sees).  A wave holding an axis-parallel ray (1/0 = inf) runs the
    // compare+select instance of the traversal.
    const bool finite = !__any(!(fabsf(rD.x) < INFINITY && fabsf(rD.y) < INFINITY && fabsf(rD.z) < INFINITY));
    if (finite) traverse_packet_impl<true, EXACT>(sc, O, D, rD, t, h, lane, wd, unsure);
    else traverse_packet_impl<false, EXACT>(sc, O, D, rD, t, h, lane, wd, unsure);
}

// ---------------------------------------------------------------------------------------
// Single-ray nearest-hit traversal: BVH::Traverse + BVHNode::Traverse + IntersectPrimitives
// (raytracer.cpp:945-950, 1204-1264, 1328-1346).  Returns true when a hit was recorded.
// skip_ok: this ray may skip groups of a long leaf whose box it misses (see not_grazing).
// `stack` points at this lane's column of an LDS array laid out [entry][64 lanes].  Lanes
// first walk inner nodes until each holds a leaf, then all leaves are processed together.
// ---------------------------------------------------------------------------------------
// One step of a team's agreement on the nearest hit: the running candidate {t, j} (distance, member that found it)
// against the candidate of the lane the DPP pattern pairs this lane with; the nearer wins, the lower member on a tie
// (it tested the earlier triangle: raytracer.cpp:1336-1344 keeps the first of two equal distances).
// close (optional): set when the two candidates are different triangles within 16 units in the last place of each other.
template <int DPP>
__device__ __forceinline__ void team_min_step(float& t, int& j, bool* close = nullptr) {
    // (the members of a team are active together; a lane whose partner were not would see its own candidate)
    const int ti = __builtin_bit_cast(int, t);
    const float to = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(ti, ti, DPP, 0xf, 0xf, false));
    const int jo = __builtin_amdgcn_update_dpp(j, j, DPP, 0xf, 0xf, false);
    const bool take = (to < t) || (to == t && jo < j);
    if (close) *close = *close || (jo != j && (uint32_t)(__float_as_int(to) - __float_as_int(t)) + 16u <= 32u);
    t = take ? to : t;
    j = take ? jo : j;
}

// TEAM: the 16 lanes of every DPP row carry the SAME ray (render_tile); `lane` is then needed.  Such lanes take the
// same way through the tree; in a leaf member j tests triangles j, j + 16, ... of it, and the members agree on the
// nearest hit (the lowest triangle of equal distances) before any of them goes on.
// EXACT: sc is not the reference's tree (hit_is_the_references); unsure: set when the lane's result is not certain.
template <bool TEAM = false, bool EXACT = false, class SC, typename STP>
__device__ __forceinline__ bool traverse_nearest(const SC& sc, int n_tris, bool coords_ok, f3 O, f3 D, bool skip_ok, float& t, Hit& h, STP stack, Watch& wd,
                                                 int lane = 0, bool* unsure = nullptr) {
    const Tie tie{sc.place, EXACT ? unsure : nullptr};
    h.tri = -1;
    if (n_tris == 0) return false;
    const v4f root0 = sc.node(0), root1 = sc.node(1);
    {
        // root gate: the reference hands CheckBox the direction where it expects the reciprocal
        float tn;
        if (!slab_mul(xyz(root0), xyz(root1), O, D, t, tn)) return false;
    }
    MRT_TICK(wd, 2);
    const RayDiv R = EXACT ? make_ray_second_tree(O, D) : make_ray_div(O, D, coords_ok);      // (EXACT: the caller has checked within_reach)
    const f3 ro = R.O * R.r;      // (EXACT: slab_mul_pair)
    const bool shortcut = EXACT || __all(R.shortcut);
    // groups of a long leaf may be skipped for rays whose reciprocal direction is an ordinary number on every axis
    const bool may_skip = !EXACT && sc.has_groups() && skip_ok &&      // (a second tree has no group boxes)
                          fabsf(R.r.x) < INFINITY && fabsf(R.r.y) < INFINITY && fabsf(R.r.z) < INFINITY;
    int sp = 0;
    uint32_t first = wbits(root0), count = wbits(root1);
    bool alive = true;
    while (alive) {
        if (wd.expired(4)) break;
        while (alive && count == 0) {
            MRT_TICK(wd, 0);
            const uint32_t L = first, Rr = first + 1;
            const auto pair = sc.siblings(L);
            const v4f l0 = pair[0], l1 = pair[1], r0 = pair[2], r1 = pair[3];
            bool leftN, rightN;
            float tNear, tNear2;
            if constexpr (EXACT) slab_mul_pair(l0, l1, r0, r1, R, ro, t * kExactCull, leftN, rightN, tNear, tNear2);
            else slab_div_pair(l0, l1, r0, r1, R, shortcut, t, leftN, rightN, tNear, tNear2);
            // raytracer.cpp:1239-1262 as selects: go right when only the right child passes, or when
            // both pass and the right one is entered first (tNear2 < tNear); the other passing
            // child, if any, is deferred; with no passing child the next deferred node is taken.
            const bool both = leftN && rightN;
            const bool go_right = rightN && (!leftN || tNear2 < tNear);
            if (both) { stack[sp * 64] = go_right ? L : Rr; ++sp; }
            first = go_right ? wbits(r0) : wbits(l0);
            count = go_right ? wbits(r1) : wbits(l1);
            if (!(leftN || rightN)) {
                if (sp == 0) alive = false;
                else {
                    --sp;
                    const uint32_t n = stack[sp * 64];
                    first = sc.node_first(n); count = sc.node_count(n);
                }
            }
        }
        if (alive) {
            // A leaf of the reference's tree holds up to about a hundred triangles while most hold two or
            // three.  The lanes walk their leaves MRT_LEAF_CHUNK triangles at a time: a lane whose leaf is
            // longer keeps the rest for the next leaf phase, and the lanes that are done go on with inner nodes
            // in between instead of waiting for the longest leaf of the wave.  Per lane the sequence of tests
            // is unchanged.
            const uint32_t n_now = count < (uint32_t)MRT_LEAF_CHUNK ? count : (uint32_t)MRT_LEAF_CHUNK;
            // A ray that misses the padded box around four consecutive triangles cannot pass the test of any of
            // them (the test computes a point of the triangle to within rounding; only for a ray that lies IN a
            // triangle's plane is every numerator of the test rounding noise, and such a ray's outcome is noise in
            // the reference too).  todo: bit j = triangle first + j is still to be tested; the order of the tests
            // that remain is the leaf's.
            uint32_t todo = n_now >= 32u ? ~0u : (1u << n_now) - 1u;
            const bool long_leaf = may_skip && n_now >= (uint32_t)MRT_GROUP_MIN;
            if constexpr (TEAM) {
                // one triangle of the chunk per member
                static_assert(MRT_LEAF_CHUNK == 16, "a team is the 16 lanes of a DPP row");
                const uint32_t j = (uint32_t)lane & 15u;
                const int tri_before = h.tri;
                if (j < n_now) tri_test<false>(sc.tri_a(first + j), sc.tri_b(first + j), sc.tri_c(first + j), O, D, t, h, wd, tie);
                if (__any(h.tri != tri_before)) {   // a member has a new hit: nearer, or equally far and earlier in the reference's order
                    float tw = t;
                    // what decides between members: the distance, then the kept triangle's place in the reference's order
                    // (a member without a new hit holds the old one: equal in all, and never better than a new one).
                    // EXACT: the triangle's number stands in for the place; candidates that near each other make the
                    // ray unsure anyway (hit_is_the_references).
                    auto rank = [&](int tri) { return tri < 0 ? 0x7fffffff : (EXACT ? tri : (int)sc.place[tri].x); };
                    int jw = rank(h.tri);
                    team_min_step<0xB1>(tw, jw, EXACT ? unsure : nullptr);     // quad_perm [1,0,3,2]
                    team_min_step<0x4E>(tw, jw, EXACT ? unsure : nullptr);     // quad_perm [2,3,0,1]
                    team_min_step<0x141>(tw, jw, EXACT ? unsure : nullptr);    // row_half_mirror
                    team_min_step<0x140>(tw, jw, EXACT ? unsure : nullptr);    // row_mirror
                    // the members that hold the winning triangle are asked for it (they hold the same u and v: same arithmetic)
                    const int mine = rank(h.tri);
                    const unsigned long long holders = (__ballot(t == tw && mine == jw) >> (lane & 48)) & 0xffffull;
                    const int from = (lane & 48) | (__ffsll((long long)holders) - 1);
                    t = tw;
                    h.tri = __shfl(h.tri, from, 64); h.u = __shfl(h.u, from, 64); h.v = __shfl(h.v, from, 64);
                }
                todo = 0u;
            } else if (__any(long_leaf)) {
                const uint32_t shift = first & 3u, g0 = first >> 2;
                const uint32_t n_groups = long_leaf ? (shift + n_now + 3u) >> 2 : 0u;     // at most MRT_LEAF_CHUNK / 4 + 1
                unsigned long long keep = 0;
                for (uint32_t i = 0; i < n_groups; ++i) {
                    MRT_TICK(wd, 20);
                    if (group_touched(sc.group(g0 + i), O, R.r, EXACT ? t * kExactCull : t)) keep |= 0xfull << (4u * i);
                }
                if (long_leaf) todo &= (uint32_t)(keep >> shift);
            }
            while (todo) {
                MRT_TICK(wd, 1);
                const uint32_t k = first + (uint32_t)__builtin_ctz(todo);
                todo &= todo - 1u;
                tri_test<false>(sc.tri_a(k), sc.tri_b(k), sc.tri_c(k), O, D, t, h, wd, tie);
            }
            first += n_now; count -= n_now;
            if (count == 0) {
                if (sp == 0) alive = false;
                else {
                    --sp;
                    const uint32_t n = stack[sp * 64];
                    first = sc.node_first(n); count = sc.node_count(n);
                }
            }
        }
    }
    if constexpr (EXACT) {
        bool undecided;
        if (h.tri >= 0 && !hit_is_the_references(sc.leaf_box, h.tri, O, D, t, true, undecided)) *unsure = true;
        if constexpr (TEAM) *unsure = ((__ballot(*unsure) >> (lane & 48)) & 0xffffull) != 0;       // the members carry one ray
    }
    return h.tri >= 0;
}

// ---------------------------------------------------------------------------------------
// Shadow rays of one wave.  Raytracer::IsOccluded (raytracer.cpp:414-418) runs the nearest-hit
// traversal with t = INFINITY and only asks whether anything was hit.  While nothing has been
// hit t stays infinite, so the set of leaves the reference reaches is "every leaf whose
// ancestors' boxes the ray passes with t = INFINITY", whatever the visiting order; and as soon
// as one triangle of that set is hit the answer is fixed.  So: one shared work list of
// (node, 64-bit lane mask) entries; a lane belongs to an entry's mask iff its own ray passed
// every box test on the way (root gate included), and only those lanes test the leaf.
// Must be called by all 64 lanes; `take_part` selects the lanes that really have a ray.
// ---------------------------------------------------------------------------------------
// TEAM: the 16 lanes of every DPP row carry the same ray (render_tile) and share out the triangles of a leaf.
// EXACT: as in traverse_nearest; a triangle in a leaf of the reference's tree that the ray misses blocks nothing.
template <bool TEAM = false, bool EXACT = false, class SC>
__device__ __forceinline__ bool occluded_wave(const SC& sc, int n_tris, bool coords_ok, f3 O, f3 D, bool take_part, bool skip_ok, int lane, Watch& wd,
                                              bool* unsure = nullptr) {
    if (n_tris == 0) return false;
    const v4f root0 = sc.node(0), root1 = sc.node(1);
    float tn;
    const bool gate = take_part && slab_mul(xyz(root0), xyz(root1), O, D, INFINITY, tn);   // D, not 1/D (raytracer.cpp:947)
    unsigned long long m = __ballot(gate);
    if (m == 0) return false;
    const RayDiv R = EXACT ? make_ray_second_tree(O, D) : make_ray_div(O, D, coords_ok);      // (EXACT: the caller has checked within_reach)
    const f3 ro = R.O * R.r;      // (EXACT: slab_mul_pair)
    const bool shortcut = EXACT || !__any(take_part && !R.shortcut);
    const bool r_ok = skip_ok && fabsf(R.r.x) < INFINITY && fabsf(R.r.y) < INFINITY && fabsf(R.r.z) < INFINITY;
    float t = INFINITY;
    Hit h;
    h.tri = -1;
    // EXACT: a hit just found counts only if the reference reaches that triangle (hit_is_the_references)
    auto vet = [&]() {
        if constexpr (EXACT) {
            if (h.tri >= 0) {
                bool undecided;
                const bool counts = hit_is_the_references(sc.leaf_box, h.tri, O, D, t, false, undecided);
                if (undecided) *unsure = true;
                else if (!counts) { h.tri = -1; t = INFINITY; }
            }
        }
    };
    LaneStack4 work;
    // The entry in hand {first, count, m} (wave-uniform: scalar registers); an inner node goes on with one of its
    // children at once and defers only the other, so the list is touched once per node with two passing children
    // instead of three times per node.
    auto uniform = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    uint32_t first = uniform(wbits(root0)), count = uniform(wbits(root1));
    bool in_hand = true;
    unsigned long long found = 0;
    for (;;) {
        if (wd.expired(16)) break;
        if (!in_hand) {
            if (work.sp == 0) break;
            work.pop(first, count, m);
        }
        in_hand = false;
        m &= ~found;
        if (m == 0) continue;
        MRT_TICK(wd, 3);
        if (count != 0) {
            if constexpr (TEAM) {
                // member j of a team tests triangles j, j + 16, ... of the leaf; one member's hit is the team's answer
                for (uint32_t k0 = first; k0 < first + count; k0 += 16u) {
                    const uint32_t k = k0 + ((uint32_t)lane & 15u);
                    if (((m >> lane) & 1ull) && k < first + count && h.tri < 0) { tri_test<false>(sc.tri_a(k), sc.tri_b(k), sc.tri_c(k), O, D, t, h, wd, Tie{nullptr}); vet(); }
                    const unsigned long long hits = __ballot(h.tri >= 0);
                    if ((hits >> (lane & 48)) & 0xffffull) h.tri = 0;
                    if (!__any(h.tri < 0 && ((m >> lane) & 1ull))) break;
                }
            } else if ((m >> lane) & 1ull) {
                if (!EXACT && SC::kShadowGroups && sc.has_groups() && count >= (uint32_t)MRT_GROUP_MIN) {
                    walk_leaf_by_groups<false>(sc, first, count, O, R.r, r_ok, t, wd, [&](uint32_t k) {
                        MRT_TICK(wd, 5);
                        if (h.tri < 0) { tri_test_shared(sc, k, O, D, t, h, wd, Tie{nullptr}); vet(); }
                        return !__any(h.tri < 0);
                    });
                } else if constexpr (SC::kPrefetchLeaf) {
                    v4f A = sc.tri_a(first), B = sc.tri_b(first);
                    v2f C = sc.tri_c(first);
                    const uint32_t last = first + count - 1;
                    for (uint32_t k = first; k <= last; ++k) {
                        MRT_TICK(wd, 5);
                        const uint32_t kn = k < last ? k + 1 : k;
                        const v4f An = sc.tri_a(kn), Bn = sc.tri_b(kn);
                        const v2f Cn = sc.tri_c(kn);
                        if (h.tri < 0) { tri_test<true>(A, B, C, O, D, t, h, wd, Tie{nullptr}); vet(); }
                        if (!__any(h.tri < 0)) break;
                        A = An; B = Bn; C = Cn;
                    }
                } else {
                    for (uint32_t k = first; k < first + count; ++k) {
                        MRT_TICK(wd, 5);
                        if (h.tri < 0) { tri_test_shared(sc, k, O, D, t, h, wd, Tie{nullptr}); vet(); }
                        if (!__any(h.tri < 0)) break;
                    }
                }
            }
            found = __ballot(h.tri >= 0);
        } else {
            const auto pair = sc.siblings(first);
            const v4f l0 = pair[0], l1 = pair[1], r0 = pair[2], r1 = pair[3];
            MRT_TICK(wd, 4);
            bool leftN, rightN;
            float nl = 0.0f, nr = 0.0f;          // entry distances (estimates will do: they only order the two children)
            if constexpr (EXACT) {
                slab_mul_pair(l0, l1, r0, r1, R, ro, INFINITY, leftN, rightN, nl, nr);
            } else {
                const bool sure = shortcut && slab_pair_shadow_interval(l0, l1, r0, r1, R, leftN, rightN, nl, nr);
                if (__any(((m >> lane) & 1ull) && !sure)) {
                    MRT_TICK(wd, 15);
                    slab_div_pair(l0, l1, r0, r1, R, shortcut, INFINITY, leftN, rightN, nl, nr);
                }
            }
            const unsigned long long ml = __ballot(leftN) & m, mr = __ballot(rightN) & m;
            // (Any order gives the same answer.  Entering first the child that most of the rays reach first, the likelier
            // one to block them soon, was measured: 2.7 % slower.)
            if (ml) {
                if (mr) work.push(lane, wbits(r0), wbits(r1), mr);
                first = uniform(wbits(l0)); count = uniform(wbits(l1)); m = ml; in_hand = true;
            } else if (mr) {
                first = uniform(wbits(r0)); count = uniform(wbits(r1)); m = mr; in_hand = true;
            }
        }
    }
    if constexpr (EXACT && TEAM) *unsure = ((__ballot(*unsure) >> (lane & 48)) & 0xffffull) != 0;       // the members carry one ray
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
    uint32_t packed;    // the palette entry colorP was made from (triangles only)
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
    s.packed = S.pal[(size_t)tex * (32 * 256) + 15 * 256 + texel];
    s.colorP = pixel_to_vec(s.packed);
    return s;
}

// ---------------------------------------------------------------------------------------
// Analytic primitives: Sphere and Plane with the reference's Intersect / IsOccluded / HitNormal
// (raytracer.cpp:146-213).  The reference keeps a list of them (raytracer.h:243) but never traces it; here they
// join a frame like this: after the BVH has been searched every object is intersected in list order with the
// ray's current t (so the nearest of triangles and objects wins, a triangle on an exact tie) and is reported as
// id -2 - index; its surface is its HitNormal, its m_Diffuse as the colour and its own material scalars; a shadow
// ray is blocked if the BVH blocks it or any object's IsOccluded says so.  Kernels are instantiated with
// OBJ = true only for scenes that have objects, so frames without them run the code they always ran.
// Records: [3k] = {pos.xyz | plane normal, radius | dist}, [3k+1] = diffuse (x=A,y=R,z=G,w=B),
// [3k+2] = {refl, refr, rindex, bits(kind)}, kind 0 sphere, 1 plane.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void trace_objects(const SceneDev& S, f3 O, f3 D, float& t, Hit& h) {
    for (int k = 0; k < S.n_objects; ++k) {
        const float4 a = S.objects[3 * k];
        const f3 pos = xyz(a);
        bool hit = false;
        float tn = 0.0f;
        if (__float_as_uint(S.objects[3 * k + 2].w) == 0u) {      // Sphere::Intersect, raytracer.cpp:153-169
            const f3 c = pos - O;
            float tt = dot(c, D);
            if (!(tt < 0.0f)) {
                const f3 q = c - tt * D;
                const float p2 = dot(q, q);
                const float r2 = a.w * a.w;
                if (!(p2 > r2)) {
                    tt -= sqrtf(r2 - p2);
                    hit = tt < t;
                    tn = tt;
                }
            }
        } else {                                                    // Plane::Intersect, raytracer.cpp:189-202
            const float dnd = dot(pos, D);
            if (!(dnd == 0.0f)) {
                const float tt = -(dot(pos, O) + a.w) / dnd;
                if (!(tt < 0.0f)) { hit = tt < t; tn = tt; }
            }
        }
        if (hit) { t = tn; h.tri = -2 - k; h.u = 0.0f; h.v = 0.0f; }
    }
}
__device__ __forceinline__ bool objects_occlude(const SceneDev& S, f3 O, f3 D) {
    bool occluded = false;
    for (int k = 0; k < S.n_objects; ++k) {
        const float4 a = S.objects[3 * k];
        const f3 pos = xyz(a);
        if (__float_as_uint(S.objects[3 * k + 2].w) == 0u) {      // Sphere::IsOccluded, raytracer.cpp:170-180
            const f3 c = pos - O;
            const float tt = dot(c, D);
            if (!(tt < 0.0f)) {
                const f3 q = c - tt * D;
                occluded = occluded || (dot(q, q) < a.w * a.w);
            }
        } else {                                                    // Plane::IsOccluded, raytracer.cpp:203-213
            const float dnd = dot(pos, D);
            if (!(dnd == 0)) occluded = occluded || !(-(dot(pos, O) + a.w) / dnd < 0.0f);
        }
    }
    return occluded;
}
__device__ __forceinline__ Surf object_surface(const SceneDev& S, int k, f3 hit) {
    const float4 a = S.objects[3 * k], d = S.objects[3 * k + 1], m = S.objects[3 * k + 2];
    Surf s;
    s.normal = __float_as_uint(m.w) == 0u ? (hit - xyz(a)) / a.w : xyz(a);      // HitNormal, raytracer.cpp:149-152, 185-188
    s.refl = m.x; s.refr = m.y; s.rindex = m.z;
    s.colorP = {d.x, d.y, d.z, d.w};
    return s;
}

// Direct light of raytracer.cpp:551-570 / 604-623 for the lanes with `shade` set: one shadow
// ray per light, starting exactly on the surface (no epsilon).  Must be called by all 64 lanes.
// REMAKE_COLOUR: the surface colour is made again from the packed palette entry after every shadow traversal
// instead of being kept in four registers across it (the packet kernel is compiled for 64 registers).
// EXACT: see hit_is_the_references.
template <bool OBJ, bool REMAKE_COLOUR, bool TEAM = false, bool EXACT = false, class SC>
__device__ __forceinline__ void direct_light(const SC& sc, const SceneDev& S, bool shade, const Surf& s, f3 hit, f4& finalColor,
                                             int lane, uint32_t& n_shadow, Watch& wd) {
    const float lightStrength = shade ? 1.0f - (s.refr + s.refl) : 0.0f;
    const bool lit = shade && lightStrength > 0.0f;
    // the ambient term opens the sum; with REMAKE_COLOUR it is added when the first light's colour is at hand
    // (same operands, same order), so that nothing of the sum is alive during the first traversal either
    const bool ambient_later = REMAKE_COLOUR && S.n_lights > 0;
    if (lit && !ambient_later) finalColor = finalColor + (MRT_AMBIENT * s.colorP) * lightStrength;
    if (!__any(lit)) return;
    uint32_t packed = s.packed;
    for (int i = 0; i < S.n_lights; ++i) {
        const float4 lp = S.lights[2 * i];
        const f3 L = lit ? normalize(xyz(lp) - hit) : f3{0.0f, 0.0f, 1.0f};
        const bool off_plane = not_grazing(L, s.normal);
        bool unsure = false;           // (a shadow ray nearly in its surface's plane, or from too far away, goes straight to the reference's tree)
        if constexpr (EXACT) unsure = lit && (kForceRetrace || !(off_plane && within_reach(sc, hit)));
        bool occluded = occluded_wave<TEAM, EXACT>(sc, S.n_tris, S.coords_ok != 0, hit, L, lit && !unsure, off_plane, lane, wd, &unsure);
        if constexpr (EXACT) {
            if (__any(unsure)) {          // those lanes' shadow rays once more, through the reference's tree
                count_retraced(wd, unsure && (!TEAM || (lane & 15) == 0));
                const bool again = occluded_wave<false, false>(sc.ref, S.n_tris, S.ref_coords_ok != 0, hit, L, unsure, off_plane, lane, wd);
                occluded = unsure ? again : occluded;
            }
        }
        if constexpr (OBJ) { if (lit && !occluded) occluded = objects_occlude(S, hit, L); }
        const float4 lc = S.lights[2 * i + 1];         // read after the traversal: nothing to keep alive across it
        f4 colorP = s.colorP;
        if constexpr (REMAKE_COLOUR) {
            asm volatile("" : "+v"(packed));        // (keeps the compiler from holding on to the first conversion)
            colorP = pixel_to_vec(packed);
            if (lit && i == 0) finalColor = finalColor + (MRT_AMBIENT * colorP) * lightStrength;
        }
        if (lit) {
            ++n_shadow;
            if (!occluded) {
                float d = dot(s.normal, L);
                if (d < 0.0f) d = 0.0f;
                finalColor = finalColor + ((d * f4{lc.x, lc.y, lc.z, lc.w}) * colorP) * lightStrength;
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

// Raytracer::GetColorFromSphere (raytracer.cpp:575-664) for the 64 pixels of a tile, without
// recursion.  Every round, the lanes that hold a ray trace it (per-lane nearest hit), the wave
// then shades those hits together (shared shadow traversal), and each lane advances its own
// continuation: "maybe reflect, then maybe refract, then clamp and return to the caller frame".
// The two pass counters are shared by the whole ray tree of a pixel (the reference passes them
// by reference), so they are a budget of 5 + 5 secondary rays, which bounds the frame stack.
// Frames live in private memory; a frame waiting for its refraction ray only needs its colour
// so far and the Beer's-law factors, one waiting for its reflection ray needs the whole hit.
constexpr int kFrames = kMaxReflPasses + kMaxRefrPasses + 1;
constexpr int kFrameWords = 24;

struct ShadeState {
    f4 finalColor, colorP;
    f3 hit, normal, D;
    float t, refl, refr, rindex, curIndex;
    int stage;      // 0: reflection not tried yet, 1: refraction not tried yet, 2: done
};

template <bool OBJ, bool TEAM, bool EXACT, class SC, typename STP>
__device__ __forceinline__ f4 shade_recursive(const SC& sc, const SceneDev& S, f3 O, f3 D, STP stack, int lane, Hit& primary,
                                              uint32_t& n_shadow, uint32_t& n_secondary, Watch& wd) {
    float frames[kFrames][kFrameWords];
    int depth = 0, reflPass = 0, refrPass = 0;
    float curIndex = 1.0f, tmax = 100000000;
    f4 ret = MRT_BACKGROUND;
    bool tracing = true, first = true;
    bool skip_ok = true;        // the current ray may skip groups of long leaves (not_grazing)
    ShadeState F;
    while (__any(tracing)) {
        if (wd.expired(32)) break;
        MRT_TICK(wd, 11);
        Hit h;
        h.tri = -1; h.u = 0; h.v = 0;
        float t = tmax;
        bool isHit = false;
        MRT_PHASE_BEGIN(wd);
        // (EXACT: a ray nearly in the plane of the surface it starts from -- see not_grazing -- goes straight to the reference's tree)
        bool unsure = false;
        if constexpr (EXACT) unsure = tracing && (kForceRetrace || !(skip_ok && within_reach(sc, O)));
        if (tracing && !unsure) isHit = traverse_nearest<TEAM, EXACT>(sc, S.n_tris, S.coords_ok != 0, O, D, skip_ok, t, h, stack, wd, lane, &unsure);
        if constexpr (EXACT) {
            if (__any(unsure)) {          // not certainly the reference's result: BVH::Traverse itself on the reference's tree
                count_retraced(wd, unsure && (!TEAM || (lane & 15) == 0));
                if (unsure) {
                    t = tmax;
                    isHit = traverse_nearest<false, false>(sc.ref, S.n_tris, S.ref_coords_ok != 0, O, D, true, t, h, stack, wd);
                }
            }
        }
        if constexpr (OBJ) { if (tracing) { trace_objects(S, O, D, t, h); isHit = h.tri != -1; } }
        MRT_PHASE_END(wd, 16);
        if (tracing && first) { primary = h; first = false; }
        const bool shade = tracing && isHit;
        Surf s;
        s.normal = {0, 0, 1}; s.refl = 0; s.refr = 0; s.rindex = 1; s.colorP = {0, 0, 0, 0}; s.packed = 0;
        f3 hit = O;
        if (shade) {
            hit = O + D * t;
            if (!OBJ || h.tri >= 0) s = fetch_surface(S, h); else s = object_surface(S, -2 - h.tri, hit);
        }
        f4 finalColor = {0, 0, 0, 0};
        MRT_PHASE_BEGIN(wd);
        direct_light<OBJ, false, TEAM, EXACT>(sc, S, shade, s, hit, finalColor, lane, n_shadow, wd);
        MRT_PHASE_END(wd, 17);
        if (!tracing) continue;
        tracing = false;
        bool returning;
        if (!isHit) { ret = MRT_BACKGROUND; returning = true; }
        else {
            F.finalColor = finalColor; F.colorP = s.colorP; F.hit = hit; F.normal = s.normal; F.D = D; F.t = t;
            F.refl = s.refl; F.refr = s.refr; F.rindex = s.rindex; F.curIndex = curIndex; F.stage = 0;
            returning = false;
        }
        for (;;) {
            if (!returning) {
                if (F.stage == 0) {
                    F.stage = 1;
                    if (F.refl > 0.0f && reflPass < kMaxReflPasses) {      // raytracer.cpp:625-631
                        reflPass++;
                        ++n_secondary;
                        float* fr = frames[depth++];
                        fr[0] = F.finalColor.x; fr[1] = F.finalColor.y; fr[2] = F.finalColor.z; fr[3] = F.finalColor.w;
                        fr[4] = F.colorP.x; fr[5] = F.colorP.y; fr[6] = F.colorP.z; fr[7] = F.colorP.w;
                        fr[8] = F.hit.x; fr[9] = F.hit.y; fr[10] = F.hit.z;
                        fr[11] = F.normal.x; fr[12] = F.normal.y; fr[13] = F.normal.z;
                        fr[14] = F.D.x; fr[15] = F.D.y; fr[16] = F.D.z;
                        fr[17] = F.t; fr[18] = F.refl; fr[19] = F.refr; fr[20] = F.rindex; fr[21] = F.curIndex;
                        fr[23] = 1.0f;
                        O = F.hit; D = reflect(F.D, F.normal); tmax = INFINITY; curIndex = F.curIndex;
                        skip_ok = not_grazing(D, F.normal);
                        tracing = true;
                        break;
                    }
                }
                if (F.stage == 1) {
                    F.stage = 2;
                    if (F.refr > 0.0f && refrPass < kMaxRefrPasses) {      // raytracer.cpp:633-652
                        refrPass++;
                        const float rindex = F.rindex;
                        const float n = F.curIndex / rindex;
                        const float cosI = -dot(F.normal, F.D);
                        const float cosT2 = 1.0f - n * n * (1.0f - cosI * cosI);
                        if (cosT2 > 0.0f) {
                            const f3 T = (n * F.D) + (n * cosI - sqrtf(cosT2)) * F.normal;
                            const f4 absorbance = (F.colorP * 0.15f) * -F.t;
                            ++n_secondary;
                            float* fr = frames[depth++];
                            fr[0] = F.finalColor.x; fr[1] = F.finalColor.y; fr[2] = F.finalColor.z; fr[3] = F.finalColor.w;
                            fr[4] = expf(absorbance.x); fr[5] = expf(absorbance.y); fr[6] = expf(absorbance.z);
                            fr[23] = 2.0f;
                            O = F.hit + T * kEpsilon; D = T; tmax = INFINITY; curIndex = rindex;
                            skip_ok = not_grazing(D, F.normal);
                            tracing = true;
                            break;
                        }
                    }
                }
                ret = clamp01(F.finalColor);
            }
            if (depth == 0) break;          // the primary ray's colour is in `ret`
            const float* fr = frames[--depth];
            F.finalColor = {fr[0], fr[1], fr[2], fr[3]};
            if (fr[23] == 2.0f) {
                F.finalColor = F.finalColor + ret * f4{fr[4], fr[5], fr[6], 1.0f};
                F.stage = 2;
            } else {
                F.colorP = {fr[4], fr[5], fr[6], fr[7]};
                F.hit = {fr[8], fr[9], fr[10]}; F.normal = {fr[11], fr[12], fr[13]}; F.D = {fr[14], fr[15], fr[16]};
                F.t = fr[17]; F.refl = fr[18]; F.refr = fr[19]; F.rindex = fr[20]; F.curIndex = fr[21];
                F.finalColor = F.finalColor + (ret * F.colorP) * F.refl;
                F.stage = 1;
            }
            returning = false;
        }
    }
    return ret;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------
// One tile: ray generation (threads.cpp:189-211), trace, shade, pixel pack + store
// (threads.cpp:215-224).  MODE 0 = packet path, 1 = recursive shader.  All 64 lanes call it.
// sub: 0 = the whole tile; 1 ... 16 (recursive shader without a sample grid only) = one sixteenth of it, the 2x2 pixels
// at column 2 * ((sub - 1) % 4), row 2 * ((sub - 1) / 4) of the tile.  A pixel's ray tree depends on nothing but its
// own ray (raytracer.cpp:575-664), so sixteen such calls write the tile's pixels.
// ---------------------------------------------------------------------------------------
// EXACT (recursive shader, whole tiles): sc is another tree than the reference's (hit_is_the_references).
template <int MODE, bool OBJ, bool TEAM = false, bool EXACT = false, class SC, typename STP>
__device__ __forceinline__ void render_tile(const SC& sc, const SceneDev& S, const FrameDev& F, int tile, int lane, STP stack, int sub = 0) {
    // A sixteenth of a tile is four pixels for 64 lanes: every pixel gets a TEAM of 16 lanes (one DPP row) that all
    // carry the pixel's ray.  They walk the tree side by side (same ray, same decisions) and share out the triangles of
    // a leaf among themselves (traverse_nearest), which is what shortens the walk of a single ray tree; member 0 of a
    // team writes the pixel and counts its rays.
    constexpr bool team = TEAM;          // (instantiated apart: whole tiles run the code they always ran; sub != 0 exactly when TEAM)
    auto pixel_of = [&](int l) {       // the tile's pixel (0 ... 63, row-major) lane l works for
        return team ? (2 * ((sub - 1) >> 2) + (l >> 5)) * 8 + 2 * ((sub - 1) & 3) + ((l >> 4) & 1) : l;
    };
    const bool writer = !team || (lane & 15) == 0;
    int pl = pixel_of(lane);
    const int ltr = tile / F.tiles_x, tx = tile - ltr * F.tiles_x;   // ltr: tile row within this launch
    int ty = F.tile_y0 + ltr;
    if (F.n_parts > 1) {
        const int lr = ltr / F.grid_g, sub = ltr - lr * F.grid_g;  // output tile row within this launch, traced row inside it
        const int blk = lr / F.part_block;
        ty = ((blk * F.n_parts + F.part) * F.part_block + (lr - blk * F.part_block)) * F.grid_g + sub;
    }
    int x = tx * 8 + (pl & 7);
    const int y = ty * 8 + (pl >> 3);

    const f3 p0 = {F.p0[0], F.p0[1], F.p0[2]}, d10 = {F.d10[0], F.d10[1], F.d10[2]}, d20 = {F.d20[0], F.d20[1], F.d20[2]};
    const f3 O = {F.cam[0], F.cam[1], F.cam[2]};
    const float u = (float)x * F.inv_w;
    const float v = (float)y * F.inv_h;
    const f3 planepos = (p0 + u * d10) + v * d20;
    const f3 D = normalize(planepos - O);

    uint32_t n_shadow = 0, n_secondary = 0;
    Watch wd{watch_budget(S.n_node_slots, S.n_tris), F.counters ? F.counters + 5 : nullptr};
    Hit h;
    h.tri = -1; h.u = 0; h.v = 0;
    f4 color;
    if (MODE == 0) {
        const f3 rD = 1.0f / D;
        float t = 100000000;
        MRT_PHASE_BEGIN(wd);
        if constexpr (EXACT) {
            bool unsure = kForceRetrace;
            // (the second tree's boxes take a reciprocal that is an ordinary number on every axis, see make_ray_second_tree:
            // one instance of the walk for all packets; the camera is within SceneDev::reach, the launcher's condition)
            const f3 rD2 = {second_tree_recip(D.x), second_tree_recip(D.y), second_tree_recip(D.z)};
            if (S.n_tris != 0) traverse_packet_impl<true, true>(sc, O, D, rD2, t, h, lane, wd, &unsure);
            bool undecided;
            if (h.tri >= 0 && !hit_is_the_references<true>(sc.leaf_box, h.tri, O, rD, t, true, undecided)) unsure = true;
            if (__any(unsure)) {          // the packet once more, through the reference's tree: what a ray of a packet tests depends on the others
                count_retraced(wd, unsure);
                t = 100000000; h.tri = -1; h.u = 0; h.v = 0;
                traverse_packet(sc.ref, S.n_tris, O, D, rD, t, h, lane, wd);
            }
        } else {
            traverse_packet(sc, S.n_tris, O, D, rD, t, h, lane, wd);
        }
        MRT_PHASE_END(wd, 16);
        if constexpr (OBJ) trace_objects(S, O, D, t, h);
        const bool shade = h.tri != -1;
        Surf s;
        s.normal = {0, 0, 1}; s.refl = 0; s.refr = 0; s.rindex = 1; s.colorP = {0, 0, 0, 0}; s.packed = 0;
        f3 hit = O;
        if (shade) {
            hit = O + D * t;
            if (!OBJ || h.tri >= 0) s = fetch_surface(S, h); else s = object_surface(S, -2 - h.tri, hit);
        }
        f4 finalColor = {0, 0, 0, 0};
        MRT_PHASE_BEGIN(wd);
        direct_light<OBJ, !OBJ, false, EXACT>(sc, S, shade, s, hit, finalColor, lane, n_shadow, wd);
        MRT_PHASE_END(wd, 17);
        color = shade ? clamp01(finalColor) : MRT_BACKGROUND;
    } else {
        color = shade_recursive<OBJ, TEAM, EXACT>(sc, S, O, D, stack, lane, h, n_shadow, n_secondary, wd);
        if (!writer) { n_shadow = 0; n_secondary = 0; }
    }
    const uint32_t px = vec_to_pixel(color);

    // where the pixel goes: made again from the tile and the lane rather than carried through the traversals (in the
    // 64-register packet kernel the carried value was spilled to private memory, one store per lane and tile)
    {
        int lane_again = lane;
        asm volatile("" : "+v"(lane_again));
        pl = pixel_of(lane_again);
        x = tx * 8 + (pl & 7);
    }
    const int y_store = (F.compact ? ltr : ty) * 8 + (pl >> 3);    // row of the traced image as laid out in fb / hit_ids
    if (F.hit_ids && writer) F.hit_ids[(size_t)y_store * F.width + x] = h.tri;
    const int g = F.grid_g;
    if (g == 1) {
        if (writer) F.fb[(size_t)y_store * F.pitch + x] = px;
    } else {
        // box filter over g x g blocks of QUANTISED pixels, rounded down per channel
        uint32_t ar = ((px >> 24) << 16) | ((px >> 16) & 255u);      // A | R   (16 bits each)
        uint32_t gb = (((px >> 8) & 255u) << 16) | (px & 255u);      // G | B
        for (int o = 1; o < g; o <<= 1) {
            ar += __shfl_xor(ar, o, 64);      gb += __shfl_xor(gb, o, 64);
            ar += __shfl_xor(ar, o * 8, 64);  gb += __shfl_xor(gb, o * 8, 64);
        }
        if (((lane & 7) % g) == 0 && ((lane >> 3) % g) == 0) {
            const int sh = 2 * (__ffs(g) - 1);          // g is 2, 4 or 8: dividing by g * g is a shift
            const uint32_t out = (((ar >> 16) >> sh) << 24) | (((ar & 0xffffu) >> sh) << 16) | (((gb >> 16) >> sh) << 8) | ((gb & 0xffffu) >> sh);
            F.fb[(size_t)(y_store / g) * F.pitch + (x / g)] = out;
        }
    }
#ifdef MRT_LOOP_COUNTS
    for (int i = 0; i < kProfSites; ++i) {
        const uint32_t a = wave_sum(wd.wave_n[i]), b = wave_sum(wd.lane_n[i]);
        if (lane == 0 && b) { atomicAdd(&F.counters[kProfBase + 2 * i], (unsigned long long)a); atomicAdd(&F.counters[kProfBase + 2 * i + 1], (unsigned long long)b); }
    }
#endif
    if (F.stats) {
        const uint32_t hits = (uint32_t)__popcll(__ballot(writer && h.tri != -1));
        const uint32_t sh = wave_sum(n_shadow), se = wave_sum(n_secondary);
        if (lane == 0) {
            unsigned long long* slot = F.counters + kStatBase + 16 * (tile % kStatSlots);
            // Counter updates are executed at the memory side (the XCDs' L2s do not share lines), so a tile
            // sends as few as it can: one word carries the tile count (low half) and the primary hits (high
            // half); shadow rays are sent as the difference from "one per primary hit and light", which is zero
            // for a tile of the packet path on opaque-or-lit surfaces, and secondary rays only when there are any.
            atomicAdd(&slot[0], (sub <= 1 ? 1ull : 0ull) | ((unsigned long long)hits << 32));      // (a tile in sixteenths is counted once)
            const long long extra_shadow = (long long)sh - (long long)hits * S.n_lights;
            if (extra_shadow) atomicAdd(&slot[1], (unsigned long long)extra_shadow);
            if (se) atomicAdd(&slot[2], (unsigned long long)se);
        }
    }
}

// ---------------------------------------------------------------------------------------
// BVH-depth view (raytracer.cpp:581-586 with BVH::TraverseDepth :951-954 and BVHNode::TraverseDepth :1265-1326,
// TEST 0 branch): a pixel shows how many nodes its primary ray's traversal enters.  That traversal is the
// nearest-hit traversal with three differences: no root gate, a child box passes only with tFar > tNear
// (strictly), and every node entered counts.  Per lane, true divisions, one node or one leaf per turn: a debug
// view, written for clarity.
// ---------------------------------------------------------------------------------------
template <class SC, typename STP>
__device__ __forceinline__ int traverse_depth(const SC& sc, int n_tris, f3 O, f3 D, float& t, Hit& h, STP stack, Watch& wd) {
    h.tri = -1;
    if (n_tris == 0) return 0;
    int depth = 1, sp = 0;                     // the root is entered
    uint32_t first = sc.node_first(0), count = sc.node_count(0);
    bool alive = true;
    while (alive) {
        if (wd.expired(64)) break;
        bool take_next = false;
        if (count == 0) {
            const uint32_t L = first, R = first + 1;
            const v4f l0 = sc.node(2 * L), l1 = sc.node(2 * L + 1), r0 = sc.node(2 * R), r1 = sc.node(2 * R + 1);
            float tNear, tFar, tNear2, tFar2;
            slab_terms((xyz(l0) - O) / D, (xyz(l1) - O) / D, tNear, tFar);
            slab_terms((xyz(r0) - O) / D, (xyz(r1) - O) / D, tNear2, tFar2);
            const bool leftN = (tFar > tNear) && (tNear < t) && (tFar > 0);
            const bool rightN = (tFar2 > tNear2) && (tNear2 < t) && (tFar2 > 0);
            const bool go_right = rightN && (!leftN || tNear2 < tNear);
            if (leftN && rightN) { stack[sp * 64] = go_right ? L : R; ++sp; }
            if (leftN || rightN) {
                first = go_right ? wbits(r0) : wbits(l0);
                count = go_right ? wbits(r1) : wbits(l1);
                ++depth;
            } else take_next = true;
        } else {
            for (uint32_t k = first; k < first + count; ++k) tri_test<false>(sc.tri_a(k), sc.tri_b(k), sc.tri_c(k), O, D, t, h, wd, Tie{sc.place});
            take_next = true;
        }
        if (take_next) {
            if (sp == 0) alive = false;
            else {
                --sp;
                const uint32_t n = stack[sp * 64];
                first = sc.node_first(n); count = sc.node_count(n);
                ++depth;
            }
        }
    }
    return depth;
}

template <bool G>
__device__ __forceinline__ SceneGlobalT<G> global_scene(const SceneDev& S) {
    return SceneGlobalT<G>{reinterpret_cast<const v4f*>(S.nodes), reinterpret_cast<const v4f*>(S.triA),
                       reinterpret_cast<const v4f*>(S.triB), reinterpret_cast<const v2f*>(S.triC),
                       reinterpret_cast<const v4u*>(S.triG), reinterpret_cast<const v4f*>(S.triM), S.place};
}

// Scene in global memory: one 64-thread workgroup (one wave) per tile.  SD = stack entries.
template <int MODE, int SD, bool OBJ, bool G>
#ifndef MRT_PACKET_WAVES
#define MRT_PACKET_WAVES 8      // waves per SIMD the packet kernel is compiled for (register budget 512 / n)
#endif
#ifndef MRT_RECURSIVE_WAVES
#define MRT_RECURSIVE_WAVES 4
#endif
__global__ __launch_bounds__(64, MODE == 0 ? MRT_PACKET_WAVES : MRT_RECURSIVE_WAVES) void render_tiles_global(SceneDev S, FrameDev F) {
    __shared__ uint32_t lds_stack[MODE == 0 ? 64 : SD * 64];      // only the single-ray traversal keeps a per-lane stack
    const int lane = threadIdx.x;
    // Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has
    // its own L2.  Tiles are dealt to the XCDs in runs of kXcdRun consecutive tiles, whose rays share
    // nodes and triangles, instead of one tile at a time.  (Giving each XCD one contiguous eighth of
    // the frame was measured twice as slow: sky and floor regions cost very different amounts.)
    const int b = (int)blockIdx.x, q = b / kXcds;
    const int tile = ((q / kXcdRun) * kXcds + b % kXcds) * kXcdRun + q % kXcdRun;
    if (tile >= F.n_tiles) return;
    if constexpr (MODE == 0) {
        const SceneGlobalEyeT<G> sc{global_scene<G>(S), reinterpret_cast<const v4f*>(F.eye_nodes), reinterpret_cast<const v4f*>(F.eye_tri),
                                    reinterpret_cast<const v4f*>(F.eye_g)};
        render_tile<MODE, OBJ>(sc, S, F, tile, lane, (MRT_LDS uint32_t*)lds_stack + lane);
    } else {
        render_tile<MODE, OBJ>(global_scene<G>(S), S, F, tile, lane, (MRT_LDS uint32_t*)lds_stack + lane);
    }
}

// The packet path through a scene's second tree (S.nodes ... the second tree's, S.leaf_box and S.ref_* set; F.eye_*2 its
// eye-relative copy, F.eye_* the reference tree's): render_tiles_global<0> with both trees at hand.
template <bool GR>
__global__ __launch_bounds__(64, MRT_PACKET_WAVES) void render_tiles_two_trees_packet(SceneDev S, FrameDev F) {
    __shared__ uint32_t lds_stack[64];
    const int lane = threadIdx.x;
    const int b = (int)blockIdx.x, q = b / kXcds;
    const int tile = ((q / kXcdRun) * kXcds + b % kXcds) * kXcdRun + q % kXcdRun;       // (as in render_tiles_global)
    if (tile >= F.n_tiles) return;
    SceneTwoTreesT<GR> sc;
    sc.nodes = reinterpret_cast<const v4f*>(S.nodes); sc.triA = reinterpret_cast<const v4f*>(S.triA); sc.triB = reinterpret_cast<const v4f*>(S.triB);
    sc.triC = reinterpret_cast<const v2f*>(S.triC); sc.triG = nullptr; sc.triM = reinterpret_cast<const v4f*>(S.triM); sc.place = nullptr;
    sc.leaf_box = reinterpret_cast<const v4f*>(S.leaf_box); sc.reach = S.reach;
    sc.eyeNodes = reinterpret_cast<const v4f*>(F.eye_nodes2); sc.eyeTri = reinterpret_cast<const v4f*>(F.eye_tri2); sc.eyeG = nullptr;
    sc.ref.nodes = reinterpret_cast<const v4f*>(S.ref_nodes); sc.ref.triA = reinterpret_cast<const v4f*>(S.ref_triA); sc.ref.triB = reinterpret_cast<const v4f*>(S.ref_triB);
    sc.ref.triC = reinterpret_cast<const v2f*>(S.ref_triC); sc.ref.triG = reinterpret_cast<const v4u*>(S.ref_triG); sc.ref.triM = reinterpret_cast<const v4f*>(S.ref_triM);
    sc.ref.place = S.ref_place;
    sc.ref.eyeNodes = reinterpret_cast<const v4f*>(F.eye_nodes); sc.ref.eyeTri = reinterpret_cast<const v4f*>(F.eye_tri); sc.ref.eyeG = reinterpret_cast<const v4f*>(F.eye_g);
    render_tile<0, false, false, true>(sc, S, F, tile, lane, (MRT_LDS uint32_t*)lds_stack + lane);
}

// The frame's eye-relative copy of the nodes and triangles (SceneGlobalEyeT): one thread per node record and
// per triangle, the operations of the box and triangle tests in their order.
// Is the eye within `pad` of the plane of one of the (up to four) triangles of group g?  Primary rays from such an eye
// can graze that triangle, and a grazing test's verdict is rounding noise that no box predicts (see not_grazing), so
// the frame gives the group the box no ray misses: the reference tests every ray against every triangle of a leaf
// (raytracer.cpp:1445-1464, 1328-1346), and so do the frame's primary rays for this group.
__device__ __forceinline__ bool eye_grazes_group(const SceneDev& S, uint32_t g, f3 O, float pad) {
    bool near = false;
    for (uint32_t k = 4u * g; k < 4u * g + 4u && k < (uint32_t)S.n_tris; ++k) {
        const float4 A = S.triA[k], B = S.triB[k];
        const float2 C = S.triC[k];
        near = near || point_near_plane(O, f3{B.z, B.w, C.x}, f3{A.x, A.y, A.z}, f3{A.w, B.x, B.y}, pad);
    }
    return near;
}

// The group boxes as the primary rays of a frame of the recursive shader may use them (the per-lane traversal reads
// boxes in the scene's own binary16 form): a copy of triG in which the groups the eye grazes are the box no ray misses.
__global__ __launch_bounds__(256) void eye_groups_kernel(SceneDev S, float ox, float oy, float oz, float pad, uint4* out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ((uint32_t)S.n_tris + 3u) / 4u) return;
    out[g] = eye_grazes_group(S, g, f3{ox, oy, oz}, pad) ? make_uint4(MRT_GROUP_EVERYWHERE_X, MRT_GROUP_EVERYWHERE_Y, MRT_GROUP_EVERYWHERE_Z, 0u)
                                                          : S.triG[g];
}

__global__ __launch_bounds__(256) void eye_relative_kernel(SceneDev S, float ox, float oy, float oz, float pad, float4* eye_nodes, float4* eye_tri, float4* eye_g) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const f3 O = {ox, oy, oz};
    if (i < 2u * (uint32_t)S.n_node_slots) {
        const float4 n = S.nodes[i];
        const f3 d = f3{n.x, n.y, n.z} - O;
        eye_nodes[i] = make_float4(d.x, d.y, d.z, n.w);
    }
    if (i < (uint32_t)S.n_tris) {
        const float4 A = S.triA[i], B = S.triB[i];
        const float2 C = S.triC[i];
        const f3 e1 = {A.x, A.y, A.z}, e2 = {A.w, B.x, B.y}, p0 = {B.z, B.w, C.x};
        const f3 tvec = O - p0;
        const f3 qvec = cross(tvec, e1);
        const float tn = dot(e2, qvec);
        eye_tri[4 * (size_t)i] = A;
        eye_tri[4 * (size_t)i + 1] = make_float4(B.x, B.y, tvec.x, tvec.y);
        eye_tri[4 * (size_t)i + 2] = make_float4(tvec.z, tn, qvec.x, qvec.y);
        eye_tri[4 * (size_t)i + 3] = make_float4(qvec.z, C.y, 0.0f, 0.0f);
    }
    if (S.triG && i < ((uint32_t)S.n_tris + 3u) / 4u) {
        const uint4 g = S.triG[i];
        f3 lo, hi;
        group_corners(v4u{g.x, g.y, g.z, g.w}, lo, hi);
        lo = lo - O; hi = hi - O;      // rounding: a ten-millionth of the operands, the padding is 2^-12 of the scene
        if (eye_grazes_group(S, i, O, pad)) { lo = {-INFINITY, -INFINITY, -INFINITY}; hi = {INFINITY, INFINITY, INFINITY}; }
        eye_g[2 * i] = make_float4(lo.x, lo.y, lo.z, 0.0f);
        eye_g[2 * i + 1] = make_float4(hi.x, hi.y, hi.z, 0.0f);
    }
}

// First half of the depth view: one wave per tile writes every pixel's node count into the frame and folds the
// frame's maximum into counters[6]; depth_colour_kernel then turns counts into colours.  The reference divides by
// a running maximum that its worker threads update without synchronisation and that survives from frame to frame
// (raytracer.cpp:523, 584); once a static view has been rendered that value is the frame's maximum, used here.
template <int SD>
__global__ __launch_bounds__(64) void depth_tiles_kernel(SceneDev S, FrameDev F) {
    __shared__ uint32_t lds_stack[SD * 64];
    const int lane = threadIdx.x, tile = (int)blockIdx.x;
    if (tile >= F.n_tiles) return;
    const int ty = tile / F.tiles_x, tx = tile - ty * F.tiles_x;
    const int x = tx * 8 + (lane & 7), y = ty * 8 + (lane >> 3);
    const f3 p0 = {F.p0[0], F.p0[1], F.p0[2]}, d10 = {F.d10[0], F.d10[1], F.d10[2]}, d20 = {F.d20[0], F.d20[1], F.d20[2]};
    const f3 O = {F.cam[0], F.cam[1], F.cam[2]};
    const float u = (float)x * F.inv_w;
    const float v = (float)y * F.inv_h;
    const f3 planepos = (p0 + u * d10) + v * d20;
    const f3 D = normalize(planepos - O);
    Watch wd{watch_budget(S.n_node_slots, S.n_tris), F.counters + 5};
    float t = 100000000;
    Hit h;
    h.u = 0; h.v = 0;
    const int depth = traverse_depth(global_scene<false>(S), S.n_tris, O, D, t, h, (MRT_LDS uint32_t*)lds_stack + lane, wd);
    F.fb[(size_t)y * F.pitch + x] = (uint32_t)depth;
    if (F.hit_ids) F.hit_ids[(size_t)y * F.width + x] = h.tri;
    int m = depth;
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    const unsigned long long hits = (unsigned long long)__popcll(__ballot(h.tri >= 0));
    if (lane == 0) {
        atomicMax(&F.counters[6], (unsigned long long)m);
        atomicAdd(&F.counters[kStatBase + 16 * (tile % kStatSlots)], 1ull | (hits << 32));     // primary rays and hits, as every frame kernel counts them
    }
}
__global__ __launch_bounds__(256) void depth_colour_kernel(uint32_t* fb, int pitch, int W, int H, const unsigned long long* max_depth) {
    const int x = (int)(blockIdx.x * 256 + threadIdx.x), y = (int)blockIdx.y;
    if (x >= W || y >= H) return;
    const int maxDepth = (int)*max_depth, depth = (int)fb[(size_t)y * pitch + x];
    // (a scene without triangles: every count is 0, and 0 of 0 is shown as 0 rather than evaluated)
    const float part = maxDepth > 0 ? (float)depth / maxDepth : 0.0f;
    fb[(size_t)y * pitch + x] = vec_to_pixel(clamp01(f4{1.0f, part, 1.0f - part, 0.0f}));
}
hipError_t launch_depth_view(const SceneDev& S, const FrameDev& F, hipStream_t stream) {
    if (F.n_tiles <= 0) return hipSuccess;
    const dim3 grid((unsigned)F.n_tiles), block(64);
    if (F.stack_entries <= 16) hipLaunchKernelGGL((depth_tiles_kernel<16>), grid, block, 0, stream, S, F);
    else if (F.stack_entries <= 32) hipLaunchKernelGGL((depth_tiles_kernel<32>), grid, block, 0, stream, S, F);
    else hipLaunchKernelGGL((depth_tiles_kernel<kStackDepth>), grid, block, 0, stream, S, F);
    hipLaunchKernelGGL(depth_colour_kernel, dim3((unsigned)((F.width + 255) / 256), (unsigned)F.height), dim3(256), 0, stream,
                       F.fb, F.pitch, F.width, F.height, (const unsigned long long*)(F.counters + 6));
    return hipGetLastError();
}

// Scene resident in LDS: persistent workgroups of 16 waves copy nodes and triangles into LDS
// once, then every wave pulls tile numbers from a shared counter until none are left -- the
// GPU form of the reference's tile queue (threads.cpp:158-169: workers decrement one counter).
// LDS: nodes | triA | triB | triC | per-lane stacks (16 bit entries: a resident scene has
// fewer than 65536 node slots).
// EXACT: S is the scene's second tree (SceneDev::leaf_box; render_tile)
template <int MODE, bool EXACT = false>
#ifndef MRT_RESIDENT_WAVES
#define MRT_RESIDENT_WAVES 4
#endif
__global__ __launch_bounds__(kResidentThreads, MRT_RESIDENT_WAVES) void render_tiles_resident(SceneDev S, FrameDev F) {
    extern __shared__ __align__(16) unsigned char smem[];
#ifdef MRT_LOOP_COUNTS
    const unsigned long long t_enter = __builtin_readcyclecounter();
    unsigned long long t_tiles = 0;
#endif
#ifdef MRT_WAVE_LIVES
    const unsigned long long lives_enter = __builtin_readcyclecounter();
    unsigned long long lives_tiles = 0;
#endif
    const uint32_t n_node_v4 = 2u * (uint32_t)S.n_node_slots, n_tri = (uint32_t)S.n_tris;
    MRT_LDS v4f* l_nodes = (MRT_LDS v4f*)smem;
    MRT_LDS v4f* l_a = l_nodes + n_node_v4;
    MRT_LDS v4f* l_b = l_a + n_tri;
    MRT_LDS v2f* l_c = (MRT_LDS v2f*)(l_b + n_tri);
    MRT_LDS v4u* l_g = (MRT_LDS v4u*)(l_c + n_tri + (n_tri & 1u));
    const uint32_t n_groups = S.triG ? (n_tri + 3u) / 4u : 0u;       // (a scene without group boxes leaves their room to nodes and stacks)
    MRT_LDS uint16_t* l_stacks = (MRT_LDS uint16_t*)(l_g + n_groups);
    {
        const v4f* gn = reinterpret_cast<const v4f*>(S.nodes);
        const v4f* ga = reinterpret_cast<const v4f*>(S.triA);
        const v4f* gb = reinterpret_cast<const v4f*>(S.triB);
        const v2f* gc = reinterpret_cast<const v2f*>(S.triC);
        for (uint32_t i = threadIdx.x; i < n_node_v4; i += blockDim.x) l_nodes[i] = gn[i];
        for (uint32_t i = threadIdx.x; i < n_tri; i += blockDim.x) { l_a[i] = ga[i]; l_b[i] = gb[i]; l_c[i] = gc[i]; }
        if (S.triG) {
            const v4u* gg = reinterpret_cast<const v4u*>(S.triG);
            for (uint32_t i = threadIdx.x; i < n_groups; i += blockDim.x) l_g[i] = gg[i];
        }
    }
    __syncthreads();
    const SceneLds sc{l_nodes, l_a, l_b, l_c, l_g, S.triG != nullptr, S.place, reinterpret_cast<const v4f*>(S.leaf_box), S.reach,
                      SceneGlobalT<false>{reinterpret_cast<const v4f*>(S.ref_nodes), reinterpret_cast<const v4f*>(S.ref_triA), reinterpret_cast<const v4f*>(S.ref_triB),
                                          reinterpret_cast<const v2f*>(S.ref_triC), nullptr, reinterpret_cast<const v4f*>(S.ref_triM), S.ref_place}};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef MRT_LOOP_COUNTS
    const unsigned long long t_ready = __builtin_readcyclecounter();
#endif
    MRT_LDS uint16_t* stack = l_stacks + (size_t)wave * F.stack_entries * 64 + lane;
    // Tile queue.  One shared counter would serialise every pull of every wave on one address
    // (tens of thousands of returning atomics per frame); instead the tiles are dealt round-robin
    // into F.n_queues queues, each with its own counter on its own cache line.  Workgroup b draws
    // from queue b % n_queues only and its waves leave when that queue runs dry; the launcher never
    // makes more queues than workgroups, so every queue has a workgroup that drains it and all tiles
    // are drawn exactly once whatever the placement of the workgroups.  (Letting a wave move on to
    // the other queues before it leaves cost 2 % of the headline frame: the last wave of the frame
    // ended with one failed pull per queue, back to back, each a round trip to memory.)
    // (Measured and dropped: pulling the next tile while the current one renders, which hides the pull's
    // round trip to the L2 -- 8 % slower, because with the expensive tiles handed out first a tile reserved
    // early waits behind an expensive one.)
    // What the queues hand out: the frame's tiles, or (F.tile_order) the entries of the list tile_order_kernel made:
    // tile | sub << 24, a whole tile or one sixteenth of an expensive one; counters[kOrderCountSlot] holds their number.
    const int n_queues = F.n_queues;
    const int q = (int)(blockIdx.x % (unsigned)n_queues);
    const long long n_pos = F.tile_order ? (long long)F.counters[kOrderCountSlot] : (long long)F.n_tiles;
    for (long long pulls = 0; pulls <= n_pos; ++pulls) {   // bounded: no wave can draw more
        unsigned long long next = 0;
        if (lane == 0) next = atomicAdd(&F.counters[kQueueBase + 16 * q], 1ull);
        const int idx = __builtin_amdgcn_readfirstlane((int)next);
        const long long pos = (long long)idx * n_queues + q;
        if (pos >= n_pos) break;
        const uint32_t entry = F.tile_order ? F.tile_order[pos] : (uint32_t)pos;
        const int tile = (int)(entry & 0xffffffu), sub = (int)(entry >> 24);
        const unsigned long long t0 = __builtin_readcyclecounter();
        if constexpr (EXACT) {
            if (sub != 0) render_tile<MODE, false, true, true>(sc, S, F, tile, lane, stack, sub);
            else render_tile<MODE, false, false, true>(sc, S, F, tile, lane, stack);
        } else if (MODE == 1 && sub != 0) render_tile<MODE, false, MODE == 1>(sc, S, F, tile, lane, stack, sub);
        else render_tile<MODE, false>(sc, S, F, tile, lane, stack);
        if (F.tile_cost && lane == 0) {
            const unsigned long long dt = __builtin_readcyclecounter() - t0;
            const uint32_t c = dt > 0x3fffffffull ? 0x3fffffffu : (uint32_t)dt;
            // a whole tile's clocks, or (bit 31) those of a tile's slowest sixteenth: see tile_keys_kernel
            if (sub == 0) F.tile_cost[tile] = c;
            else atomicMax(&F.tile_cost[tile], c | 0x80000000u);
        }
#ifdef MRT_LOOP_COUNTS
        t_tiles += __builtin_readcyclecounter() - t0;
#endif
#ifdef MRT_WAVE_LIVES
        ++lives_tiles;
#endif
    }
#ifdef MRT_WAVE_LIVES
    if ((threadIdx.x & 63) == 0) {
        const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (w < (uint32_t)kLivesWaves) {
            unsigned long long* rec = F.counters + kLivesBase + 3 * w;
            rec[0] = lives_enter; rec[1] = __builtin_readcyclecounter(); rec[2] = lives_tiles;
        }
    }
#endif
#ifdef MRT_LOOP_COUNTS
    if (lane == 0) {       // sites 18, 19: clocks until the scene copy is done | in tiles; the wave's whole life | 1
        const unsigned long long t_exit = __builtin_readcyclecounter();
        atomicAdd(&F.counters[kProfBase + 2 * 18], t_ready - t_enter);
        atomicAdd(&F.counters[kProfBase + 2 * 18 + 1], t_tiles);
        atomicAdd(&F.counters[kProfBase + 2 * 19], t_exit - t_enter);
        atomicAdd(&F.counters[kProfBase + 2 * 19 + 1], 1ull);
        atomicMax(&F.counters[kProfBase + 2 * 14], t_exit - t_enter);      // site 14: longest wave life | latest first-tile start
        atomicMax(&F.counters[kProfBase + 2 * 14 + 1], t_ready - t_enter);
    }
#endif
}

// ---------------------------------------------------------------------------------------
// order[] = the tiles sorted by decreasing cost of the previous launch, to 1/8 octave: a counting
// sort over 256 keys in two small kernels (the order inside a key is arbitrary).  A frame ends
// when its last tile ends, and a tile of the recursive shader can take ten times the average:
// started last it leaves the rest of the chip idle, started first it overlaps with everything
// else.  Any permutation renders the same frame; the costs only steer the order.
//   tile_keys_kernel:  packed[i] = key << 24 | rank of tile i among the tiles with its key
//                      (hist[key] must be zero on entry and holds the key's tile count afterwards)
//   tile_order_kernel: order[first[key] + rank] = i, first = exclusive prefix sum of hist
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cost_key(uint32_t cost) {
    const uint32_t b = __float_as_uint((float)cost) >> 20;            // exponent and 3 significand bits
    const uint32_t k = b > (127u << 3) ? b - (127u << 3) : 0u;         // cost 1 -> 0, cost 2^32 -> 256
    return 255u - (k > 255u ? 255u : k);                                // expensive first
}
// cost[i]: what the previous launch left for tile i -- its shader clocks if it was rendered whole, or, with bit 31 set,
// the clocks of its slowest sixteenth.  A sixteenth takes 0.4 to 0.9 of its tile's time, differently from tile to tile,
// so state[i] keeps the tile's estimate (bits 8-31, units of 64 clocks) together with the ratio sixteenth : whole
// (bits 0-7, in 1/256; 0 = not known) as measured in the first frame that split the tile, when the whole tile's cost
// was known from the frame before; while the tile stays split, estimate = slowest sixteenth / ratio.  (One fixed ratio
// for all tiles made the tiles it flattered fall under the splitting threshold and come back every other frame.)
// While the camera moves, what was expensive in the previous frame is expensive a few tiles further on in this one.
// tile_spread_kernel (two passes: along the rows, then along the columns) gives every tile the largest previous cost
// within `radius` tiles of it, and tile_keys_kernel orders a tile by the larger of its own cost and `share` of that: a
// cheap tile started early costs nothing, an expensive one started late costs the frame its tail.  The splitting of
// expensive tiles goes by the same rank (tile_order_kernel).  FROM_COST: the input is the launch's raw cost words (a whole tile's clocks, or
// with bit 31 the slowest sixteenth's: taken as it is, the smaller figure) -- else values of the first pass.
template <bool FROM_COST>
__global__ __launch_bounds__(256) void tile_spread_kernel(const uint32_t* in, uint32_t* out, int tiles_x, int n, int radius, int step) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int x = i % tiles_x, row0 = i - x;
    uint32_t best = 0u;
    for (int d = -radius; d <= radius; ++d) {
        int j;
        if (step == 1) { const int xx = x + d; if (xx < 0 || xx >= tiles_x) continue; j = row0 + xx; }
        else { j = i + d * tiles_x; if (j < 0) continue; }
        if (j >= n) continue;
        uint32_t v = in[j];
        if (FROM_COST) v &= 0x7fffffffu;
        best = v > best ? v : best;
    }
    out[i] = best;
}
__global__ __launch_bounds__(1024) void tile_keys_kernel(uint32_t* cost, uint32_t* state, uint32_t* packed, uint32_t* hist, int n,
                                                         const uint32_t* nearby, float share) {
    __shared__ uint32_t count[256], base[256];
    const int t = threadIdx.x, i = (int)blockIdx.x * 1024 + t;
    if (t < 256) count[t] = 0;
    __syncthreads();
    uint32_t key = 0u;
    if (i < n) {
        const uint32_t c = cost[i];
        uint32_t whole = c, ratio = 0u;
        if (c >> 31) {
            const uint32_t part = c & 0x7fffffffu, st = state[i], before = (st >> 8) << 6;
            ratio = st & 255u;
            if (ratio == 0u) {
                const unsigned long long r = ((unsigned long long)part << 8) / (before ? before : 1u);
                ratio = r < 64u ? 64u : (r > 255u ? 255u : (uint32_t)r);
            }
            const unsigned long long w = ((unsigned long long)part << 8) / ratio;
            whole = w > 0x3fffffffull ? 0x3fffffffu : (uint32_t)w;
        }
        state[i] = ((whole >> 6) << 8) | ratio;
        cost[i] = 0u;            // the sixteenths of a tile leave their maximum (atomicMax)
        uint32_t rank_by = whole;
        if (nearby) { const uint32_t around = (uint32_t)(share * (float)nearby[i]); rank_by = around > whole ? around : whole; }
        key = cost_key(rank_by);
    }
    const uint32_t local = i < n ? atomicAdd(&count[key], 1u) : 0u;
    __syncthreads();
    if (t < 256 && count[t]) base[t] = atomicAdd(&hist[t], count[t]);
    __syncthreads();
    if (i < n) packed[i] = (key << 24) | (base[key] + local);
}
// Splitting the most expensive tiles.  A frame cannot end before its most expensive tile has, and a tile is one wave
// that walks its pixels' ray trees one ray after the other: below about a million pixels the frame takes as long as
// that one tile while most of the chip idles.  A wave with only a 2x2 block of the tile's pixels live finishes in about
// 0.6 of the time (the longest ray tree still has to be walked), so the tiles within 6/8 octave of the most expensive one
// are handed out as sixteen entries -- if they also cost more than twice what a wave has to do on average anyway (a large
// frame is bound by the sum of the work, which splitting multiplies by ten for the tiles it splits), and only as many of
// them, most expensive first, as the chip has time for while the slowest sixteenth runs (without that limit the extra
// waves slow every tile down, more tiles look expensive, and the number of split tiles runs away: measured).
//   n_waves: waves the coming launch will run; capacity: entries `order` can hold; n_entries: where the length of the list goes
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* packed, const uint32_t* hist, uint32_t* order, int n,
                                                          int allow_split, int n_waves, float budget_frac, uint32_t split_keys, uint32_t capacity, unsigned long long* n_entries) {
    __shared__ uint32_t first[256], mult[256];
    const int t = threadIdx.x, i = (int)blockIdx.x * 1024 + t;
    if (t < 64) {            // one wave: four keys per lane
        uint32_t c[4], m[4];
        float work = 0.0f;
        uint32_t kmin = 1024u;
        for (int j = 3; j >= 0; --j) {
            c[j] = hist[4 * t + j];
            work += (float)c[j] * exp2f((float)(255 - (4 * t + j)) * 0.125f);       // cost_key: one key per 1/8 octave, expensive first
            if (c[j]) kmin = (uint32_t)(4 * t + j);
        }
        for (int o = 32; o > 0; o >>= 1) {
            work += __shfl_xor(work, o, 64);
            const uint32_t other = (uint32_t)__shfl_xor((int)kmin, o, 64);
            kmin = other < kmin ? other : kmin;
        }
        // work of the keys up to and including each of this lane's (keys are ordered expensive first)
        float mine = 0.0f, upto[4];
        for (int j = 0; j < 4; ++j) { mine += (float)c[j] * exp2f((float)(255 - (4 * t + j)) * 0.125f); upto[j] = mine; }
        float before = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(before, o, 64);
            if (t >= o) before += up;
        }
        before -= mine;
        const float per_wave = work / (float)(n_waves > 0 ? n_waves : 1);
        // what the chip can do in the time the slowest sixteenth of the most expensive tile will take (0.64 of that tile),
        // at the 60 % to which waves of such different lengths can be packed
        const float budget = budget_frac * (float)n_waves * exp2f((float)(255 - (int)(kmin < 256u ? kmin : 255u)) * 0.125f);
        uint32_t plain = 0, split = 0;
        for (int j = 0; j < 4; ++j) {
            const uint32_t k = (uint32_t)(4 * t + j);
            // sixteen waves at 0.64 of the tile's time each: ten times the tile's work, nine more than it costs whole
            const bool s16 = allow_split && k - kmin < split_keys && exp2f((float)(255 - (int)k) * 0.125f) > 2.0f * per_wave &&
                             work + 9.0f * (before + upto[j]) <= budget;
            m[j] = s16 ? 16u : 1u;
            plain += c[j]; split += c[j] * m[j];
        }
        uint32_t incl_p = plain, incl_s = split;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up_p = __shfl_up(incl_p, o, 64), up_s = __shfl_up(incl_s, o, 64);
            if (t >= o) { incl_p += up_p; incl_s += up_s; }
        }
        const uint32_t total_s = (uint32_t)__shfl((int)incl_s, 63, 64);
        const bool fits = total_s <= capacity;                     // otherwise nothing is split this time
        uint32_t at = fits ? incl_s - split : incl_p - plain;
        for (int j = 0; j < 4; ++j) {
            const uint32_t mj = fits ? m[j] : 1u;
            first[4 * t + j] = at; mult[4 * t + j] = mj;
            at += c[j] * mj;
        }
        if (blockIdx.x == 0 && t == 63) *n_entries = fits ? total_s : (uint32_t)n;
    }
    __syncthreads();
    if (i < n) {
        const uint32_t p = packed[i], k = p >> 24, m = mult[k];
        const uint32_t at = first[k] + (p & 0xffffffu) * m;
        if (m == 1u) order[at] = (uint32_t)i;
        else for (uint32_t j = 0; j < m; ++j) order[at + j] = (uint32_t)i | ((j + 1u) << 24);
    }
}
// n < 2^24.  cost, state, packed: n words each.  hist: 256 words, zero on entry.  order: `capacity` entries; n_entries receives how many were written (n, or
// more when allow_split let some tiles be listed as sixteenths).
// spread_a, spread_b: n words each, or null; with them and radius > 0 the order takes the costs of the tiles within
// `radius` of a tile into account (tile_spread_kernel; tiles_x: tiles per row).
hipError_t launch_order_tiles(uint32_t* cost, uint32_t* state, uint32_t* packed, uint32_t* hist, uint32_t* order, int n, bool allow_split, int n_waves,
                              float budget_frac, int split_keys, size_t capacity, unsigned long long* n_entries,
                              uint32_t* spread_a, uint32_t* spread_b, int tiles_x, int radius, float share, hipStream_t stream) {
    const dim3 grid((unsigned)((n + 1023) / 1024)), block(1024);
    const bool spread = spread_a && spread_b && radius > 0 && tiles_x > 0;
    if (spread) {
        const dim3 g256((unsigned)((n + 255) / 256)), b256(256);
        hipLaunchKernelGGL((tile_spread_kernel<true>), g256, b256, 0, stream, (const uint32_t*)cost, spread_a, tiles_x, n, radius, 1);
        hipLaunchKernelGGL((tile_spread_kernel<false>), g256, b256, 0, stream, (const uint32_t*)spread_a, spread_b, tiles_x, n, radius, 0);
    }
    hipLaunchKernelGGL(tile_keys_kernel, grid, block, 0, stream, cost, state, packed, hist, n, spread ? (const uint32_t*)spread_b : (const uint32_t*)nullptr, share);
    hipLaunchKernelGGL(tile_order_kernel, grid, block, 0, stream, (const uint32_t*)packed, (const uint32_t*)hist, order, n, allow_split ? 1 : 0, n_waves,
                       budget_frac, (uint32_t)split_keys, (uint32_t)std::min<size_t>(capacity, 0xffffffffu), n_entries);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Operator-level kernels on caller-supplied rays
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void trace_packets_kernel(SceneDev S, int n_packets, const float* O, const float* D,
                                                           const float* rD, float* t, int32_t* tri, float* u, float* v,
                                                           unsigned long long* flag) {
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= n_packets) return;
    const size_t r = (size_t)p * 64 + lane;
    const f3 o = {O[3 * r], O[3 * r + 1], O[3 * r + 2]};
    const f3 d = {D[3 * r], D[3 * r + 1], D[3 * r + 2]};
    const f3 rd = {rD[3 * r], rD[3 * r + 1], rD[3 * r + 2]};
    float tt = t[r];
    Hit h;
    h.tri = -1; h.u = 0; h.v = 0;
    Watch wd{watch_budget(S.n_node_slots, S.n_tris), flag};
    traverse_packet(global_scene<false>(S), S.n_tris, o, d, rd, tt, h, lane, wd);
    t[r] = tt; tri[r] = h.tri; u[r] = h.u; v[r] = h.v;
}

template <int SD>
__global__ __launch_bounds__(64) void trace_rays_kernel(SceneDev S, int n_rays, const float* O, const float* D,
                                                        float* t, int32_t* tri, float* u, float* v, unsigned long long* flag) {
    __shared__ uint32_t lds_stack[SD * 64];
    const int lane = threadIdx.x;
    const size_t r = (size_t)blockIdx.x * 64 + lane;
    if (r >= (size_t)n_rays) return;
    const f3 o = {O[3 * r], O[3 * r + 1], O[3 * r + 2]};
    const f3 d = {D[3 * r], D[3 * r + 1], D[3 * r + 2]};
    float tt = t[r];
    Hit h;
    h.u = 0; h.v = 0;
    Watch wd{watch_budget(S.n_node_slots, S.n_tris), flag};
    traverse_nearest(global_scene<false>(S), S.n_tris, S.coords_ok != 0, o, d, true, tt, h, (MRT_LDS uint32_t*)lds_stack + lane, wd);
    t[r] = tt; tri[r] = h.tri; u[r] = h.u; v[r] = h.v;
}

// ---------------------------------------------------------------------------------------
// Host-callable launchers (used by the C ABI in mrt_api.cpp)
// ---------------------------------------------------------------------------------------
size_t resident_scene_bytes(int n_node_slots, int n_tris, bool groups) {
    return (size_t)n_node_slots * 32 + (size_t)n_tris * 40 + ((n_tris & 1) ? 8 : 0) + (groups ? (size_t)((n_tris + 3) / 4) * 16 : 0);
}
size_t resident_lds_bytes(int n_node_slots, int n_tris, int stack_entries, bool groups) {
    return resident_scene_bytes(n_node_slots, n_tris, groups) + (size_t)(kResidentThreads / 64) * stack_entries * 64 * sizeof(uint16_t);
}

template <int MODE>
static hipError_t launch_global_mode(const SceneDev& S, const FrameDev& F, hipStream_t stream) {
    const int group = kXcds * kXcdRun;
    const dim3 grid((unsigned)((F.n_tiles + group - 1) / group * group)), block(64);
    const bool g = S.triG != nullptr;
    auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, block, 0, stream, S, F); };
    if (S.n_objects > 0) g ? go(render_tiles_global<MODE, kStackDepth, true, true>) : go(render_tiles_global<MODE, kStackDepth, true, false>);
    else if (F.stack_entries <= 16) g ? go(render_tiles_global<MODE, 16, false, true>) : go(render_tiles_global<MODE, 16, false, false>);
    else if (F.stack_entries <= 32) g ? go(render_tiles_global<MODE, 32, false, true>) : go(render_tiles_global<MODE, 32, false, false>);
    else g ? go(render_tiles_global<MODE, kStackDepth, false, true>) : go(render_tiles_global<MODE, kStackDepth, false, false>);
    return hipGetLastError();
}

// Workgroups and waves of a launch of the persistent kernel.  With a tile order the list may hold sixteen entries for
// a tile, so even a frame of few tiles gets every CU.
int resident_blocks(int n_tiles, int n_cus, int threads, bool ordered) {
    const int waves = std::max(64, std::min(kResidentThreads, threads / 64 * 64)) / 64;
    const long long entries = ordered ? 16ll * n_tiles : (long long)n_tiles;
    return (int)std::max<long long>(1, std::min<long long>(n_cus, (entries + waves - 1) / waves));
}
int resident_waves(int n_tiles, int n_cus, int threads, bool ordered) {
    return resident_blocks(n_tiles, n_cus, threads, ordered) * (std::max(64, std::min(kResidentThreads, threads / 64 * 64)) / 64);
}

template <int MODE, bool EXACT = false>
static hipError_t launch_resident_mode(const SceneDev& S, const FrameDev& F, int n_cus, int threads, hipStream_t stream) {
    const size_t lds = resident_lds_bytes(S.n_node_slots, S.n_tris, F.stack_entries, S.triG != nullptr);
    const int blocks = resident_blocks(F.n_tiles, n_cus, threads, F.tile_order != nullptr);
    threads = std::max(64, std::min(kResidentThreads, threads / 64 * 64));
    FrameDev Fq = F;
    Fq.n_queues = std::min(kTileQueues, blocks);          // every queue needs a workgroup of its own: nobody else drains it
    hipLaunchKernelGGL((render_tiles_resident<MODE, EXACT>), dim3((unsigned)blocks), dim3((unsigned)threads), lds, stream, S, Fq);
    return hipGetLastError();
}

// Once per device, with that device current (mrt_create): lets the persistent kernels use a whole CU's LDS.
// Done at set-up so that a frame launch is nothing but launches.
hipError_t prepare_render_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_tiles_resident<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&render_tiles_resident<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&render_tiles_resident<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
}

// resident: use the LDS-resident persistent kernel (the caller has checked that the scene fits)
// resident_threads: workgroup size of that kernel (kResidentThreads unless a diagnostic build says otherwise)
hipError_t launch_render(const SceneDev& S, const FrameDev& F, int mode, bool resident, int n_cus, int resident_threads, hipStream_t stream) {
    if (F.n_tiles <= 0) return hipSuccess;
    if (resident) return mode == 0 ? launch_resident_mode<0>(S, F, n_cus, resident_threads, stream) : launch_resident_mode<1>(S, F, n_cus, resident_threads, stream);
    return mode == 0 ? launch_global_mode<0>(S, F, stream) : launch_global_mode<1>(S, F, stream);
}

// A frame through a scene's second tree (S.leaf_box and S.ref_* set, S.nodes ... the second tree's; no analytic
// primitives).  Recursive shader: the persistent kernel, the second tree in LDS, F.stack_entries enough for either tree.
// Packet path: one wave per tile, both trees in global memory with their eye-relative copies (F.eye_*2, F.eye_*).
hipError_t launch_render_second_tree(const SceneDev& S, const FrameDev& F, int mode, int n_cus, int resident_threads, hipStream_t stream) {
    if (F.n_tiles <= 0) return hipSuccess;
    if (mode == 1) return launch_resident_mode<1, true>(S, F, n_cus, resident_threads, stream);
    const int group = kXcds * kXcdRun;
    const dim3 grid((unsigned)((F.n_tiles + group - 1) / group * group)), block(64);
    if (S.ref_triG) hipLaunchKernelGGL((render_tiles_two_trees_packet<true>), grid, block, 0, stream, S, F);
    else hipLaunchKernelGGL((render_tiles_two_trees_packet<false>), grid, block, 0, stream, S, F);
    return hipGetLastError();
}

// group_pad: the padding of the scene's group boxes (mrt_scene_upload), also the distance from a triangle's plane
// within which the eye counts as grazing it
hipError_t launch_eye_relative(const SceneDev& S, const float eye[3], float group_pad, float4* eye_nodes, float4* eye_tri, float4* eye_g, hipStream_t stream) {
    const uint32_t n = std::max(2u * (uint32_t)S.n_node_slots, (uint32_t)S.n_tris);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(eye_relative_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, S, eye[0], eye[1], eye[2], group_pad, eye_nodes, eye_tri, eye_g);
    return hipGetLastError();
}

// S.triG must be set; out: one uint4 per group of four leaf-order slots
hipError_t launch_eye_groups(const SceneDev& S, const float eye[3], float group_pad, uint4* out, hipStream_t stream) {
    const uint32_t n = ((uint32_t)S.n_tris + 3u) / 4u;
    if (n == 0 || !S.triG) return hipSuccess;
    hipLaunchKernelGGL(eye_groups_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, S, eye[0], eye[1], eye[2], group_pad, out);
    return hipGetLastError();
}

// flag: one device word, zero on entry; a traversal that exhausts its loop budget leaves a code in it
hipError_t launch_trace_packets(const SceneDev& S, int n_packets, const float* O, const float* D, const float* rD,
                                float* t, int32_t* tri, float* u, float* v, unsigned long long* flag, hipStream_t stream) {
    if (n_packets <= 0) return hipSuccess;
    hipLaunchKernelGGL(trace_packets_kernel, dim3((unsigned)n_packets), dim3(64), 0, stream, S, n_packets, O, D, rD, t, tri, u, v, flag);
    return hipGetLastError();
}

hipError_t launch_trace_rays(const SceneDev& S, int n_rays, int stack_entries, const float* O, const float* D,
                             float* t, int32_t* tri, float* u, float* v, unsigned long long* flag, hipStream_t stream) {
    if (n_rays <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_rays + 63) / 64)), block(64);
    if (stack_entries <= 16) hipLaunchKernelGGL((trace_rays_kernel<16>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v, flag);
    else if (stack_entries <= 32) hipLaunchKernelGGL((trace_rays_kernel<32>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v, flag);
    else hipLaunchKernelGGL((trace_rays_kernel<kStackDepth>), grid, block, 0, stream, S, n_rays, O, D, t, tri, u, v, flag);
    return hipGetLastError();
}

}  // namespace mrt
