// TEST INFRASTRUCTURE ONLY -- CPU oracle for the ray/scene intersection path.
// Nothing under oracle/ is part of the shipped product; only tests/, the smoke check and
// the cpu_baseline leg of bench.py may build, load or call it.
//
// A from-scratch CPU restatement of the reference's algorithm for: BVH build, 64-ray packet
// traversal, scalar traversal, direct-light shading, the recursive reflect/refract shader,
// the 8x8-tile worker pool, and the scan-line rasterizer.  Each function cites the reference
// file:line it follows (paths relative to /root/reference/Project/).  The reference itself
// cannot be compiled in this environment without stand-ins for windows.h / FreeImage / SDL
// headers, so this restatement is pinned against the known answers the survey recorded from
// the reference's own code (tests/golden/known_answers.json), not against a live build.
//
// Build: g++ -std=c++17 -O2 -ffp-contract=off (no -ffast-math, no -march=native).
#include "ora_math.h"

#include <atomic>
#include <cassert>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <vector>

namespace ora {

// ------------------------------------------------------------------------------------------
// Scene description (scenedata.h:143-167 Mesh, surface.h:41-55 Surface8)
// ------------------------------------------------------------------------------------------
struct Texture {
    int w = 0, h = 0;
    std::vector<uint8_t> idx;       // w*h palette indices
    std::vector<uint32_t> pal;      // 32 levels x 256 entries, 0x00RRGGBB
    std::string name;
};
struct Mesh {
    int verts = 0, tris = 0;
    std::vector<vec3> pos, norm;    // per vertex
    std::vector<vec2> uv;           // per vertex
    std::vector<vec3> N;            // per triangle
    std::vector<int> tri;           // 3 per triangle
    int texture = -1;
    vec3 bounds[2];
    float refl = 0, refr = 0, rindex = 1;
};
struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Texture> textures;
};
struct Light { vec3 pos; vec4 color; };

// Work counters (optional).  A "slab test" is one ray-vs-box evaluation, a "tri test" one
// Triangle::Intersect/Intersect2 call.
struct Counters {
    uint64_t packet_box = 0, packet_tri = 0, packet_nodes = 0, packet_leaves = 0;
    uint64_t scalar_box = 0, scalar_tri = 0, scalar_nodes = 0, scalar_rays = 0;
    uint64_t primary_rays = 0, shadow_rays = 0, secondary_rays = 0;
    uint64_t packet_trifetch = 0;      // triangles of the leaves a packet visits, counted once per visit (not per ray)
    void add(const Counters& o) {
        packet_box += o.packet_box; packet_tri += o.packet_tri; packet_nodes += o.packet_nodes;
        packet_leaves += o.packet_leaves; scalar_box += o.scalar_box; scalar_tri += o.scalar_tri;
        scalar_nodes += o.scalar_nodes; scalar_rays += o.scalar_rays; primary_rays += o.primary_rays;
        shadow_rays += o.shadow_rays; secondary_rays += o.secondary_rays; packet_trifetch += o.packet_trifetch;
    }
};
static thread_local Counters* g_ctr = nullptr;

// ------------------------------------------------------------------------------------------
// Triangle (raytracer.h:151-190) and BVH containers (raytracer.h:327-362)
// ------------------------------------------------------------------------------------------
struct Triangle {
    vec3 m_Pos;                   // centroid
    float refl, refr, rindex;
    vec3 normal, p0, p1, p2;
    vec2 uv0, uv1, uv2;
    int tex;
};
struct Node { vec3 mn, mx; uint32_t firstLeft, count; };
static_assert(sizeof(Node) == 32, "node layout");

struct BVH {
    std::vector<Triangle> tris;
    std::vector<uint32_t> idx;
    std::vector<Node> nodes;
    uint32_t nTris = 0, nNodes = 0, poolPtr = 0;
    int maxDepth = 0;
    const Scene* scene = nullptr;
};

// raytracer.h:154-172 Triangle constructor
static Triangle make_triangle(const Mesh& m, int i, float refl, float refr, float rindex) {
    Triangle t;
    t.refl = refl; t.refr = refr; t.rindex = rindex;
    t.normal = m.N[i];
    t.p0 = m.pos[m.tri[i * 3]];
    t.p1 = m.pos[m.tri[i * 3 + 1]];
    t.p2 = m.pos[m.tri[i * 3 + 2]];
    t.m_Pos = ((t.p0 + t.p1) + t.p2) / 3.0f;
    t.uv0 = m.uv[m.tri[i * 3]];
    t.uv1 = m.uv[m.tri[i * 3 + 1]];
    t.uv2 = m.uv[m.tri[i * 3 + 2]];
    t.tex = m.texture;
    return t;
}

static inline float axis_of(const vec3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// raytracer.h:278-293 AABB over an index range
static void range_bounds(const BVH& b, uint32_t start, uint32_t count, vec3& mn, vec3& mx) {
    mn = {INFINITY, INFINITY, INFINITY};
    mx = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = start; i < start + count; ++i) {
        const Triangle& t = b.tris[b.idx[i]];
        mn.x = gmin(gmin(t.p0.x, gmin(t.p1.x, t.p2.x)), mn.x);
        mn.y = gmin(gmin(t.p0.y, gmin(t.p1.y, t.p2.y)), mn.y);
        mn.z = gmin(gmin(t.p0.z, gmin(t.p1.z, t.p2.z)), mn.z);
        mx.x = gmax(gmax(t.p0.x, gmax(t.p1.x, t.p2.x)), mx.x);
        mx.y = gmax(gmax(t.p0.y, gmax(t.p1.y, t.p2.y)), mx.y);
        mx.z = gmax(gmax(t.p0.z, gmax(t.p1.z, t.p2.z)), mx.z);
    }
}
// raytracer.h:303-307
static float surface_area(const vec3& mn, const vec3& mx) {
    vec3 d = mx - mn;
    return (d.z * d.x + d.z * d.y + d.x * d.y) * 2;
}

// raytracer.cpp:767-874 SortX/SortY/SortZ -- recursive Hoare quicksort on centroid[axis]
static void sort_axis(const BVH& b, uint32_t* ia, int axis, int start, int end) {
    int L = start, R = end;
    float center = axis_of(b.tris[ia[(L + R) / 2]].m_Pos, axis);
    while (L < R) {
        while (axis_of(b.tris[ia[L]].m_Pos, axis) < center) L++;
        while (axis_of(b.tris[ia[R]].m_Pos, axis) > center) R--;
        if (L <= R) {
            uint32_t left = ia[L];
            ia[L] = ia[R];
            ia[R] = left;
            L++;
            R--;
        }
    }
    if (start < R) sort_axis(b, ia, axis, start, R);
    if (L < end) sort_axis(b, ia, axis, L, end);
}

// raytracer.cpp:956-1203 BVHNode::Subdivide
static void subdivide(BVH& b, uint32_t self, int depth) {
    b.maxDepth = gmax(b.maxDepth, depth);
    {
        Node& n = b.nodes[self];
        range_bounds(b, n.firstLeft, n.count, n.mn, n.mx);
    }
    const uint32_t firstLeft = b.nodes[self].firstLeft;
    const uint32_t count = b.nodes[self].count;
    if (count < 3) return;  // primPerNode = 3 (raytracer.h:353)
    b.poolPtr++;
    const uint32_t leftNode = b.poolPtr;
    b.poolPtr++;
    const uint32_t rightNode = b.poolPtr;

    const vec3 mn = b.nodes[self].mn, mx = b.nodes[self].mx;
    float d[3] = {mx.x - mn.x, mx.y - mn.y, mx.z - mn.z};
    int splitIndexAxis[3] = {0, 0, 0};
    float initCost = surface_area(mn, mx) * count;
    float bestCost = surface_area(mn, mx) * count;
    int bestAxis = 0;
    uint32_t* ia = b.idx.data();
    for (int axis = 0; axis < 3; ++axis) {   // X, Y, Z blocks at :984-1043, :1045-1103, :1105-1163
        float posStep = d[axis] / 10.0f;
        sort_axis(b, ia, axis, (int)firstLeft, (int)(firstLeft + count - 1));
        float splitPoint = posStep + axis_of(mn, axis);
        splitPoint += posStep;
        int startIndex = (int)firstLeft;
        int bestSplitIndex = startIndex;
        for (int i = 0; i < 8; ++i) {
            for (int iTriangle = startIndex; (uint32_t)iTriangle < startIndex + count; ++iTriangle) {
                const Triangle& t = b.tris[ia[iTriangle]];
                if (axis_of(t.m_Pos, axis) > splitPoint) {
                    float leftVolume, rightVolume;
                    int leftCount = iTriangle - startIndex;
                    if (leftCount == 0) leftVolume = 0;
                    else {
                        vec3 a, c;
                        range_bounds(b, startIndex, leftCount, a, c);
                        leftVolume = surface_area(a, c);
                    }
                    int rightCount = (int)(count - leftCount);
                    if (rightCount == 0) rightVolume = 0;
                    else {
                        vec3 a, c;
                        range_bounds(b, iTriangle, rightCount, a, c);
                        rightVolume = surface_area(a, c);
                    }
                    float cost = leftVolume * leftCount + rightVolume * rightCount;
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestSplitIndex = iTriangle - 1;   // off by one vs the evaluated partition (:1033)
                        bestAxis = axis;
                    }
                    break;
                }
            }
            splitPoint += posStep;
        }
        splitIndexAxis[axis] = bestSplitIndex;
    }
    sort_axis(b, ia, bestAxis, (int)firstLeft, (int)(firstLeft + count - 1));   // :1165-1179
    int splitIndex = splitIndexAxis[bestAxis];
    if (initCost == bestCost) { b.poolPtr -= 2; return; }
    b.nodes[leftNode].firstLeft = firstLeft;
    b.nodes[leftNode].count = (uint32_t)(splitIndex - (int)firstLeft);
    b.nodes[rightNode].firstLeft = (uint32_t)splitIndex;
    b.nodes[rightNode].count = count - (uint32_t)(splitIndex - (int)firstLeft);
    if (b.nodes[leftNode].count == 0 || b.nodes[rightNode].count == 0) { b.poolPtr -= 2; return; }
    b.nodes[self].firstLeft = b.poolPtr - 1;
    b.nodes[self].count = 0;
    subdivide(b, leftNode, depth + 1);
    subdivide(b, rightNode, depth + 1);
}

// raytracer.cpp:897-944 BVH::BVH.  The reference hard-codes the material per mesh index
// (mesh 0: refl 0, refr 0.3, index 1.5; others 0, 0, 1.0); here the caller supplies them on
// the Mesh so other scenes can be expressed, and default_materials() applies that rule.
static void default_materials(Scene& s) {
    for (size_t i = 0; i < s.meshes.size(); ++i) {
        Mesh& m = s.meshes[i];
        if (i > 0) { m.refl = 0.0f; m.refr = 0.0f; m.rindex = 1.0f; }
        else       { m.refl = 0.0f; m.refr = 0.3f; m.rindex = 1.5f; }
    }
}
static BVH* build_bvh(const Scene& s) {
    BVH* b = new BVH();
    b->scene = &s;
    b->nTris = 0;
    for (auto& m : s.meshes) b->nTris += m.tris;
    b->tris.resize(b->nTris);
    b->idx.resize(b->nTris);
    uint32_t k = 0;
    for (auto& m : s.meshes)
        for (int j = 0; j < m.tris; ++j) {
            b->tris[k] = make_triangle(m, j, m.refl, m.refr, m.rindex);
            b->idx[k] = k;
            k++;
        }
    b->nNodes = b->nTris * 2 + 1;
    // poolPtr starts at 2 and children are taken at ++poolPtr, so slot nNodes can be touched
    b->nodes.assign(b->nNodes + 3, Node{{0, 0, 0}, {0, 0, 0}, 0, 0});
    b->poolPtr = 2;
    b->maxDepth = 0;
    b->nodes[0].firstLeft = 0;
    b->nodes[0].count = b->nTris;
    if (b->nTris > 0) subdivide(*b, 0, 0);
    return b;
}

// ------------------------------------------------------------------------------------------
// Intersection (raytracer.cpp:287-356, 876-883)
// ------------------------------------------------------------------------------------------
struct Ray { vec3 O, D; float t; };
struct HitRec { int tri; float u, v; };  // tri = index into BVH::tris, -1 = none

// raytracer.cpp:322-356 Triangle::Intersect2 (identical arithmetic to Intersect :287-321)
static inline bool tri_intersect(const Triangle& tr, const vec3& D, const vec3& O, float& t, float& a_U, float& a_V) {
    vec3 v0v1 = tr.p1 - tr.p0;
    vec3 v0v2 = tr.p2 - tr.p0;
    vec3 pvec = cross(D, v0v2);
    float det = dot(v0v1, pvec);
    float u, v;
    if (det < -1e-4f) return false;
    float invDet = 1 / det;
    vec3 tvec = O - tr.p0;
    u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    vec3 qvec = cross(tvec, v0v1);
    v = dot(D, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    float T = dot(v0v2, qvec) * invDet;
    if (T < 0.0f) return false;
    else if (T < t) { t = T; a_U = u; a_V = v; return true; }
    return false;
}

// predicate shared by raytracer.cpp:876-883, 1227-1237, 1389-1393
static inline bool slab_mul(const vec3& bmin, const vec3& bmax, const vec3& O, const vec3& rD, float t, float* tNearOut = nullptr) {
    vec3 tMin = (bmin - O) * rD, tMax = (bmax - O) * rD;
    vec3 t1 = vmin(tMin, tMax), t2 = vmax(tMin, tMax);
    float tNear = gmax(gmax(t1.x, t1.y), t1.z);
    float tFar = gmin(gmin(t2.x, t2.y), t2.z);
    if (tNearOut) *tNearOut = tNear;
    return (tFar >= tNear) && (tNear < t) && (tFar > 0);
}
static inline bool slab_div(const vec3& bmin, const vec3& bmax, const vec3& O, const vec3& D, float t, float& tNear) {
    vec3 tMin = (bmin - O) / D, tMax = (bmax - O) / D;
    vec3 t1 = vmin(tMin, tMax), t2 = vmax(tMin, tMax);
    tNear = gmax(gmax(t1.x, t1.y), t1.z);
    float tFar = gmin(gmin(t2.x, t2.y), t2.z);
    return (tFar >= tNear) && (tNear < t) && (tFar > 0);
}

// raytracer.cpp:1328-1346 BVHNode::IntersectPrimitives
static int leaf_scalar(const BVH& b, const Node& n, Ray& r, HitRec& res) {
    float u, v;
    bool hit = false;
    for (uint32_t i = n.firstLeft; i < n.firstLeft + n.count; ++i) {
        uint32_t ti = b.idx[i];
        if (g_ctr) g_ctr->scalar_tri++;
        if (tri_intersect(b.tris[ti], r.D, r.O, r.t, u, v)) {
            res.tri = (int)ti; res.u = u; res.v = v;
            hit = true;
        }
    }
    return hit;
}
// raytracer.cpp:1204-1264 BVHNode::Traverse (TEST 0 branch)
static int traverse_scalar(const BVH& b, uint32_t ni, Ray& r, HitRec& res) {
    const Node& n = b.nodes[ni];
    if (g_ctr) g_ctr->scalar_nodes++;
    if (n.count != 0) return leaf_scalar(b, n, r, res);
    const uint32_t li = n.firstLeft, ri = li + 1;
    const Node& left = b.nodes[li];
    const Node& right = b.nodes[ri];
    if (g_ctr) g_ctr->scalar_box += 2;
    float tNear, tNear2;
    bool leftN = slab_div(left.mn, left.mx, r.O, r.D, r.t, tNear);
    bool rightN = slab_div(right.mn, right.mx, r.O, r.D, r.t, tNear2);
    if (leftN) {
        if (rightN) {
            if (tNear2 < tNear) {
                int hit = traverse_scalar(b, ri, r, res);
                hit += traverse_scalar(b, li, r, res);
                return hit;
            }
            int hit = traverse_scalar(b, li, r, res);
            hit += traverse_scalar(b, ri, r, res);
            return hit;
        }
        return traverse_scalar(b, li, r, res);
    } else if (rightN) {
        return traverse_scalar(b, ri, r, res);
    }
    return 0;
}
// raytracer.cpp:945-950 BVH::Traverse -- the root gate passes D where CheckBox expects 1/D
static int bvh_traverse(const BVH& b, Ray& r, HitRec& res) {
    if (g_ctr) g_ctr->scalar_rays++;
    if (b.nTris == 0) return 0;
    if (slab_mul(b.nodes[0].mn, b.nodes[0].mx, r.O, r.D, r.t))
        return traverse_scalar(b, 0, r, res);
    return 0;
}

// raytracer.cpp:1265-1326 BVHNode::TraverseDepth (TEST 0 branch) and :951-954 BVH::TraverseDepth: the nearest-hit
// traversal of the BVH-depth view.  It differs from BVHNode::Traverse in three ways: no root gate, `depth` counts
// every node entered, and a child box passes only with tFar > tNear (strictly).
static int traverse_depth(const BVH& b, uint32_t ni, Ray& r, int& depth, HitRec& res) {
    const Node& n = b.nodes[ni];
    depth++;
    if (n.count != 0) return leaf_scalar(b, n, r, res);
    const uint32_t li = n.firstLeft, ri = li + 1;
    auto passes = [&](const Node& c, float& tNear) {
        vec3 tMin = (c.mn - r.O) / r.D, tMax = (c.mx - r.O) / r.D;
        vec3 t1 = vmin(tMin, tMax), t2 = vmax(tMin, tMax);
        tNear = gmax(gmax(t1.x, t1.y), t1.z);
        float tFar = gmin(gmin(t2.x, t2.y), t2.z);
        return (tFar > tNear) && (tNear < r.t) && (tFar > 0);
    };
    float tNear, tNear2;
    const bool leftN = passes(b.nodes[li], tNear), rightN = passes(b.nodes[ri], tNear2);
    if (leftN) {
        if (rightN) {
            if (tNear2 < tNear) {
                int hit = traverse_depth(b, ri, r, depth, res);
                hit += traverse_depth(b, li, r, depth, res);
                return hit;
            }
            int hit = traverse_depth(b, li, r, depth, res);
            hit += traverse_depth(b, ri, r, depth, res);
            return hit;
        }
        return traverse_depth(b, li, r, depth, res);
    } else if (rightN) {
        return traverse_depth(b, ri, r, depth, res);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Analytic primitives (raytracer.h:60-123, raytracer.cpp:146-213): Sphere and Plane, with the reference's
// Intersect / IsOccluded / HitNormal.  The reference keeps a list of them (raytracer.h:243) but never traces it
// ("ATM all objects are triangles", raytracer.cpp:594); how they join a frame is defined in trace_objects below.
// ------------------------------------------------------------------------------------------
struct Object {
    int kind;            // 0 sphere, 1 plane
    vec3 pos;            // centre | plane normal ("Plane normal is stored in pos")
    float param;         // radius | m_Dist
    vec4 diffuse;        // m_Diffuse
    float refl, refr, rindex;
};
static inline bool object_intersect(const Object& o, Ray& a_Ray) {
    if (o.kind == 0) {                              // raytracer.cpp:153-169
        vec3 c = o.pos - a_Ray.O;
        float t = dot(c, a_Ray.D);
        if (t < 0.0f) return false;
        vec3 q = c - t * a_Ray.D;
        float p2 = dot(q, q);
        float r2 = o.param * o.param;
        if (p2 > r2) return false;
        t -= sqrtf(r2 - p2);
        if (t < a_Ray.t) { a_Ray.t = t; return true; }
        return false;
    }
    float dnd = dot(o.pos, a_Ray.D);                // raytracer.cpp:189-202
    if (dnd == 0.0f) return false;
    float t = -(dot(o.pos, a_Ray.O) + o.param) / dnd;
    if (t < 0.0f) return false;
    if (t < a_Ray.t) { a_Ray.t = t; return true; }
    return false;
}
static inline bool object_occludes(const Object& o, const Ray& a_Ray) {
    if (o.kind == 0) {                              // raytracer.cpp:170-180
        vec3 c = o.pos - a_Ray.O;
        float t = dot(c, a_Ray.D);
        if (t < 0.0f) return false;
        vec3 q = c - t * a_Ray.D;
        float p2 = dot(q, q);
        float r2 = o.param * o.param;
        return p2 < r2;
    }
    float dnd = dot(o.pos, a_Ray.D);                // raytracer.cpp:203-213
    if (dnd == 0) return false;
    float t = -(dot(o.pos, a_Ray.O) + o.param) / dnd;
    return !(t < 0.0f);
}
static inline vec3 object_normal(const Object& o, const vec3& a_Pos) {
    if (o.kind == 0) return (a_Pos - o.pos) / o.param;   // raytracer.cpp:149-152
    return o.pos;                                        // raytracer.cpp:185-188
}

// ------------------------------------------------------------------------------------------
// Packet traversal (raytracer.h:207-212, raytracer.cpp:1349-1464)
// ------------------------------------------------------------------------------------------
constexpr int PACKETSIZE = 64;
constexpr int TILESIZE = 8;
struct RayPacket { vec3 O[PACKETSIZE], D[PACKETSIZE], rD[PACKETSIZE]; float t[PACKETSIZE]; };
struct HitPacket { int tri[PACKETSIZE]; float u[PACKETSIZE], v[PACKETSIZE]; };

// raytracer.cpp:1445-1464
static void leaf_packet(const BVH& b, const Node& n, RayPacket& p, HitPacket& res, int firstActive) {
    if (g_ctr) { g_ctr->packet_leaves++; g_ctr->packet_trifetch += n.count; }
    for (int iRay = firstActive; iRay < PACKETSIZE; ++iRay) {
        float u, v;
        for (uint32_t i = n.firstLeft; i < n.firstLeft + n.count; ++i) {
            uint32_t ti = b.idx[i];
            if (g_ctr) g_ctr->packet_tri++;
            if (tri_intersect(b.tris[ti], p.D[iRay], p.O[iRay], p.t[iRay], u, v)) {
                res.tri[iRay] = (int)ti; res.u[iRay] = u; res.v[iRay] = v;
            }
        }
    }
}
static void traverse_packet(const BVH& b, uint32_t ni, RayPacket& p, HitPacket& res, int firstActive);
// one of the four scan loops at raytracer.cpp:1387-1398, 1399-1411, 1416-1428, 1430-1441
static inline void scan_child(const BVH& b, uint32_t ci, RayPacket& p, HitPacket& res, int firstActive) {
    const Node& c = b.nodes[ci];
    for (int i = firstActive; i < PACKETSIZE; ++i) {
        if (g_ctr) g_ctr->packet_box++;
        if (slab_mul(c.mn, c.mx, p.O[i], p.rD[i], p.t[i])) {
            traverse_packet(b, ci, p, res, i);
            break;
        }
    }
}
// raytracer.cpp:1360-1443
static void traverse_packet(const BVH& b, uint32_t ni, RayPacket& p, HitPacket& res, int firstActive) {
    const Node& n = b.nodes[ni];
    if (g_ctr) g_ctr->packet_nodes++;
    if (n.count != 0) { leaf_packet(b, n, p, res, firstActive); return; }
    const uint32_t li = n.firstLeft, ri = li + 1;
    float tNear, tNear2;
    slab_mul(b.nodes[li].mn, b.nodes[li].mx, p.O[firstActive], p.rD[firstActive], 0.0f, &tNear);
    slab_mul(b.nodes[ri].mn, b.nodes[ri].mx, p.O[firstActive], p.rD[firstActive], 0.0f, &tNear2);
    if (tNear2 < tNear) {
        scan_child(b, ri, p, res, firstActive);
        scan_child(b, li, p, res, firstActive);
    } else {
        scan_child(b, li, p, res, firstActive);
        scan_child(b, ri, p, res, firstActive);
    }
}
// raytracer.cpp:1349-1358
static void bvh_traverse_packet(const BVH& b, RayPacket& p, HitPacket& res) {
    if (b.nTris == 0) return;
    for (int i = 0; i < PACKETSIZE; ++i) {
        if (g_ctr) g_ctr->packet_box++;
        if (slab_mul(b.nodes[0].mn, b.nodes[0].mx, p.O[i], p.rD[i], p.t[i])) {
            traverse_packet(b, 0, p, res, i);
            return;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Shading (raytracer.cpp:28-45, 60-78, 362-377, 524-664)
// ------------------------------------------------------------------------------------------
static const vec4 BACKGROUND = {1.0f, 0.3f, 0.3f, 0.3f};
static const vec4 AMBIENT = {1.0f, 0.0f, 0.0f, 0.0f};
constexpr float EPSILON = 1e-4f;
constexpr int MAXREFLPASSES = 5, MAXREFRPASSES = 5;

// float -> unsigned conversion as x86-64 compilers emit it for (unsigned int)(float): convert
// to a 64-bit signed integer (truncating) and keep the low 32 bits; NaN/out-of-range give 0.
static inline uint32_t f2u(float f) {
    if (!(f > -9.2e18f && f < 9.2e18f)) return 0u;
    return (uint32_t)(int64_t)f;
}
// raytracer.cpp:28-36 / threads.cpp:179-187
static inline uint32_t vec_to_pixel(vec4 c) {
    uint32_t r = 0;
    r += f2u(c.x * 255) << 24;
    r += f2u(c.y * 255) << 16;
    r += f2u(c.z * 255) << 8;
    r += f2u(c.w * 255);
    return r;
}
// raytracer.cpp:37-45
static inline vec4 pixel_to_vec(uint32_t c) {
    vec4 r;
    r.x = ((float)((c & 0xff000000u) >> 24)) / 255.0f;
    r.y = ((float)((c & 0x00ff0000u) >> 16)) / 255.0f;
    r.z = ((float)((c & 0x0000ff00u) >> 8)) / 255.0f;
    r.w = ((float)((c & 0x000000ffu))) / 255.0f;
    return r;
}
// raytracer.cpp:362-377 Triangle::GetColor.  The reference never applies umask/vmask, so uv
// outside [0,1) reads out of the row or out of the buffer (undefined there); here the linear
// index is clamped into the buffer, which is also what the device path does.
static inline vec4 tri_color(const BVH& b, const Triangle& t, float a_U, float a_V) {
    vec2 uvPos = (t.uv0 + a_U * (t.uv1 - t.uv0)) + a_V * (t.uv2 - t.uv0);
    const Texture& tx = b.scene->textures[t.tex];
    int xBuffer = (int)(uvPos.x * tx.w);
    int yBuffer = (int)(uvPos.y * tx.h);
    long li = (long)yBuffer * tx.w + xBuffer;
    long n = (long)tx.w * tx.h;
    if (li < 0) li = 0;
    if (li >= n) li = n - 1;
    return pixel_to_vec(tx.pal[15 * 256 + tx.idx[li]]);
}
// raytracer.cpp:60-78
static inline vec3 reflect(vec3 I, vec3 N) {
    float nl = dot(I, N);
    vec3 projN = nl * N;
    vec3 tan = I - projN;
    return tan - projN;
}

struct Tracer {
    const BVH* bvh;
    std::vector<Light> lights;
    std::vector<Object> objects;
};

// How the analytic primitives join a frame (this build's definition; the reference has none):
//  - after the BVH has been searched, every object is intersected in list order with the ray's current t
//    (Object::Intersect shortens t only for a nearer hit), so the nearest of triangles and objects wins and a
//    triangle wins an exact tie; the hit is reported as id -2 - (object index);
//  - an object's surface is its HitNormal, its m_Diffuse as the colour and its own m_Refl / m_Refr / m_RIndex;
//  - a shadow ray is blocked if the BVH blocks it or any object's IsOccluded says so.
static inline int trace_objects(const Tracer& T, Ray& r) {
    int which = -1;
    for (size_t k = 0; k < T.objects.size(); ++k)
        if (object_intersect(T.objects[k], r)) which = (int)k;
    return which;
}
struct SurfaceMaterial { float refl, refr, rindex; };

// raytracer.cpp:551-570 / 604-623 direct light for one surface point
static inline void direct_light(const Tracer& T, const SurfaceMaterial& tri, vec3 hit, vec3 normal, vec4 colorP, vec4& finalColor) {
    float lightStrength = 1.0f - (tri.refr + tri.refl);
    if (lightStrength > 0.0f) {
        finalColor = finalColor + (AMBIENT * colorP) * lightStrength;
        for (size_t i = 0; i < T.lights.size(); ++i) {
            const Light& light = T.lights[i];
            Ray lightRay{hit, normalize(light.pos - hit), INFINITY};
            HitRec r{-1, 0, 0};
            if (g_ctr) g_ctr->shadow_rays++;
            bool occluded = bvh_traverse(*T.bvh, lightRay, r) != 0;   // raytracer.cpp:414-418 IsOccluded
            for (size_t k = 0; k < T.objects.size() && !occluded; ++k) occluded = object_occludes(T.objects[k], lightRay);
            if (!occluded) {
                float d = dot(normal, lightRay.D);
                if (d < 0.0f) d = 0.0f;
                finalColor = finalColor + ((d * light.color) * colorP) * lightStrength;
            }
        }
    }
}

// raytracer.cpp:524-574 TraceRayPacket
static void trace_packet(const Tracer& T, RayPacket& p, vec4* color, HitPacket& res) {
    for (int i = 0; i < PACKETSIZE; ++i) res.tri[i] = -1;
    bvh_traverse_packet(*T.bvh, p, res);
    for (int i = 0; i < PACKETSIZE; ++i) {
        if (!T.objects.empty()) {
            Ray r{p.O[i], p.D[i], p.t[i]};
            const int k = trace_objects(T, r);
            if (k >= 0) { res.tri[i] = -2 - k; res.u[i] = 0; res.v[i] = 0; p.t[i] = r.t; }
        }
        if (res.tri[i] == -1) { color[i] = BACKGROUND; continue; }
        vec3 hit = p.O[i] + p.D[i] * p.t[i];
        vec3 normal;
        vec4 colorP;
        SurfaceMaterial mat;
        if (res.tri[i] >= 0) {
            const Triangle& tri = T.bvh->tris[res.tri[i]];
            normal = tri.normal;
            colorP = tri_color(*T.bvh, tri, res.u[i], res.v[i]);
            mat = {tri.refl, tri.refr, tri.rindex};
        } else {
            const Object& o = T.objects[-2 - res.tri[i]];
            normal = object_normal(o, hit);
            colorP = o.diffuse;
            mat = {o.refl, o.refr, o.rindex};
        }
        vec4 finalColor = {0, 0, 0, 0};
        direct_light(T, mat, hit, normal, colorP, finalColor);
        color[i] = vclamp(finalColor, 0.0f, 1.0f);
    }
}

// raytracer.cpp:575-664 GetColorFromSphere (traverseDepth branch omitted: debug view)
static vec4 shade_recursive(const Tracer& T, Ray& a_Ray, int& a_ReflPass, int& a_RefrPass, float a_RIndex, HitRec* firstHit) {
    HitRec res{-1, 0, 0};
    int isHit = bvh_traverse(*T.bvh, a_Ray, res);
    if (!T.objects.empty()) {
        const int k = trace_objects(T, a_Ray);
        if (k >= 0) { res = HitRec{-2 - k, 0, 0}; isHit = 1; }
    }
    if (firstHit) *firstHit = res;
    if (!isHit) return BACKGROUND;
    vec3 hit = a_Ray.O + a_Ray.D * a_Ray.t;
    vec3 normal;
    vec4 colorP;
    SurfaceMaterial tri;
    if (res.tri >= 0) {
        const Triangle& t = T.bvh->tris[res.tri];
        normal = t.normal;
        colorP = tri_color(*T.bvh, t, res.u, res.v);
        tri = {t.refl, t.refr, t.rindex};
    } else {
        const Object& o = T.objects[-2 - res.tri];
        normal = object_normal(o, hit);
        colorP = o.diffuse;
        tri = {o.refl, o.refr, o.rindex};
    }
    vec4 finalColor = {0, 0, 0, 0};
    direct_light(T, tri, hit, normal, colorP, finalColor);
    if (tri.refl > 0.0f && a_ReflPass < MAXREFLPASSES) {
        a_ReflPass++;
        Ray reflRay{hit, reflect(a_Ray.D, normal), INFINITY};
        if (g_ctr) g_ctr->secondary_rays++;
        vec4 reflColor = shade_recursive(T, reflRay, a_ReflPass, a_RefrPass, a_RIndex, nullptr);
        finalColor = finalColor + (reflColor * colorP) * tri.refl;
    }
    if (tri.refr > 0.0f && a_RefrPass < MAXREFRPASSES) {
        a_RefrPass++;
        float rindex = tri.rindex;
        float n = a_RIndex / rindex;
        float cosI = -dot(normal, a_Ray.D);
        float cosT2 = 1.0f - n * n * (1.0f - cosI * cosI);
        if (cosT2 > 0.0f) {
            vec3 Tdir = (n * a_Ray.D) + (n * cosI - sqrtf(cosT2)) * normal;
            Ray refrRay{hit + Tdir * EPSILON, Tdir, INFINITY};
            if (g_ctr) g_ctr->secondary_rays++;
            vec4 refrColor = shade_recursive(T, refrRay, a_ReflPass, a_RefrPass, rindex, nullptr);
            vec4 absorbance = (colorP * 0.15f) * -a_Ray.t;
            vec4 transparency = {expf(absorbance.x), expf(absorbance.y), expf(absorbance.z), 1.0f};
            finalColor = finalColor + refrColor * transparency;
        }
    }
    return vclamp(finalColor, 0.0f, 1.0f);
}

// ------------------------------------------------------------------------------------------
// Frame: view plane (raytracer.cpp:420-434), tile worker pool (threads.cpp:158-225)
// ------------------------------------------------------------------------------------------
struct Frame {
    int W, H, pitch;
    vec3 p0, p1, p2, camPos;
    float invW, invH;
    uint32_t* fb;
    int32_t* hit_ids;   // optional, W*H, -1 = miss
    float* tuv;         // optional, 3*W*H (t,u,v of the primary hit)
    int mode;           // 0 = packet + direct light, 1 = recursive shader per pixel
};

static void setup_view(Frame& f, const mat4& cam, int W, int H) {
    float fov = 1.8f;
    float aspect = (float)H / (float)W;   // game.h:5 SCRASPECT
    vec4 a = mul(cam, vec4{-1, aspect, -fov, 1});
    vec4 b = mul(cam, vec4{1, aspect, -fov, 1});
    vec4 c = mul(cam, vec4{-1, -aspect, -fov, 1});
    f.p0 = {a.x, a.y, a.z}; f.p1 = {b.x, b.y, b.z}; f.p2 = {c.x, c.y, c.z};
    f.invH = 1.0f / (float)H;
    f.invW = 1.0f / (float)W;
    f.camPos = {cam.c[3].x, cam.c[3].y, cam.c[3].z};   // scenedata.h:130
    f.W = W; f.H = H;
}

// threads.cpp:196-211 ray generation for pixel (x,y) of a (possibly super-sampled) image
static inline void gen_ray(const Frame& f, int x, int y, float invW, float invH, vec3& O, vec3& D, vec3& rD) {
    float u = (float)x * invW;
    float v = (float)y * invH;
    vec3 planepos = (f.p0 + u * (f.p1 - f.p0)) + v * (f.p2 - f.p0);
    D = normalize(planepos - f.camPos);
    rD = 1.0f / D;
    O = f.camPos;
}

// threads.cpp:189-225 RenderTile (one 8x8 tile = one 64-ray packet)
static void render_tile(const Tracer& T, const Frame& f, int startx, int starty) {
    vec4 color[PACKETSIZE];
    HitPacket res;
    RayPacket p;
    int i = 0;
    for (int y = starty; y < starty + TILESIZE; ++y)
        for (int x = startx; x < startx + TILESIZE; ++x) {
            gen_ray(f, x, y, f.invW, f.invH, p.O[i], p.D[i], p.rD[i]);
            p.t[i] = 100000000;
            i++;
        }
    if (g_ctr) g_ctr->primary_rays += PACKETSIZE;
    if (f.mode == 0) {
        trace_packet(T, p, color, res);
    } else if (f.mode == 2) {
        // BVH-depth view, first half (raytracer.cpp:581-586): the number of nodes the ray's traversal enters, kept as a
        // bit pattern in the colour's first channel until the frame's maximum is known (ora_render)
        for (i = 0; i < PACKETSIZE; ++i) {
            Ray r{p.O[i], p.D[i], p.t[i]};
            int depth = 0;
            HitRec h{-1, 0, 0};
            if (T.bvh->nTris != 0) traverse_depth(*T.bvh, 0, r, depth, h);
            res.tri[i] = h.tri; res.u[i] = h.u; res.v[i] = h.v; p.t[i] = r.t;
            f.fb[(startx + i % TILESIZE) + (starty + i / TILESIZE) * f.pitch] = (uint32_t)depth;
        }
    } else {
        // the per-pixel loop the reference keeps at raytracer.cpp:474-489 (commented out there)
        for (i = 0; i < PACKETSIZE; ++i) {
            Ray r{p.O[i], p.D[i], p.t[i]};
            int refl = 0, refr = 0;
            HitRec h;
            color[i] = shade_recursive(T, r, refl, refr, 1.0f, &h);
            res.tri[i] = h.tri; res.u[i] = h.u; res.v[i] = h.v; p.t[i] = r.t;
        }
    }
    i = 0;
    for (int y = starty; y < starty + TILESIZE; ++y)
        for (int x = startx; x < startx + TILESIZE; ++x) {
            if (f.mode != 2) f.fb[x + y * f.pitch] = vec_to_pixel(color[i]);
            if (f.hit_ids) f.hit_ids[x + (size_t)y * f.W] = res.tri[i];
            if (f.tuv) {
                float* o = f.tuv + 3 * ((size_t)x + (size_t)y * f.W);
                o[0] = p.t[i]; o[1] = res.tri[i] < 0 ? 0.0f : res.u[i]; o[2] = res.tri[i] < 0 ? 0.0f : res.v[i];
            }
            i++;
        }
}

// threads.cpp:101-169 -- a pool of parked workers; each Render wakes them all and every worker
// pulls tile numbers by decrementing one shared counter until it drops below 1.
// CPUs this process may run on, one per physical core first (the lowest-numbered hardware thread of each core),
// then the remaining hardware threads: pinning worker i to entry i keeps the first workers on distinct cores.
static std::vector<int> cpus_one_per_core_first() {
    std::vector<int> firsts, others;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return firsts;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &set)) continue;
        char path[128];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
        int lowest = c;
        if (FILE* f = fopen(path, "r")) {
            if (fscanf(f, "%d", &lowest) != 1) lowest = c;
            fclose(f);
        }
        (lowest == c ? firsts : others).push_back(c);
    }
    firsts.insert(firsts.end(), others.begin(), others.end());
    return firsts;
}

class TilePool {
public:
    // pin: worker i stays on the i-th CPU of cpus_one_per_core_first() (a timing aid; results do not depend on it)
    explicit TilePool(int n, bool pin = false) : nthreads_(n) {
        const std::vector<int> cpus = pin ? cpus_one_per_core_first() : std::vector<int>();
        for (int i = 0; i < n; ++i) {
            workers_.emplace_back([this, i] { loop(i); });
            if (!cpus.empty()) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[i % cpus.size()], &one);
                if (pthread_setaffinity_np(workers_.back().native_handle(), sizeof one, &one) == 0) ++pinned_;
            }
        }
        distinct_cores_ = 0;
        if (!cpus.empty()) {
            // entries before the first repeated core are distinct cores
            std::vector<int> firsts = cpus_one_per_core_first();
            int n_first = 0;
            for (int c : firsts) {
                char path[128];
                snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
                int lowest = c;
                if (FILE* f = fopen(path, "r")) { if (fscanf(f, "%d", &lowest) != 1) lowest = c; fclose(f); }
                if (lowest == c) ++n_first;
            }
            distinct_cores_ = n < n_first ? n : n_first;
        }
    }
    int pinned() const { return pinned_; }
    int distinct_cores() const { return distinct_cores_; }
    ~TilePool() {
        { std::lock_guard<std::mutex> l(m_); quit_ = true; gen_++; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return nthreads_; }
    void run(const Tracer& T, const Frame& f, int y_begin, int y_end, Counters* ctr) {
        tiles_.clear();
        for (int y = y_begin; y < y_end; y += TILESIZE)
            for (int x = 0; x < f.W; x += TILESIZE) tiles_.push_back({x, y});   // raytracer.cpp:436-444
        waiting_.store((long)tiles_.size());
        T_ = &T; f_ = &f; ctr_ = ctr;
        per_thread_.assign(nthreads_, Counters());
        { std::lock_guard<std::mutex> l(m_); done_ = 0; gen_++; }
        cv_.notify_all();
        std::unique_lock<std::mutex> l(m_);
        cv_done_.wait(l, [this] { return done_ == nthreads_; });
        if (ctr) for (auto& c : per_thread_) ctr->add(c);
    }
private:
    void loop(int id) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (quit_) return;
            }
            g_ctr = ctr_ ? &per_thread_[id] : nullptr;
            for (;;) {                       // threads.cpp:158-169 RenderJob::Main
                if (waiting_.load() < 1) break;
                long w = --waiting_;
                if (w >= 0) render_tile(*T_, *f_, tiles_[w].x, tiles_[w].y);
            }
            g_ctr = nullptr;
            { std::lock_guard<std::mutex> l(m_); done_++; }
            cv_done_.notify_one();
        }
    }
    struct XY { int x, y; };
    int nthreads_, pinned_ = 0, distinct_cores_ = 0;
    std::vector<std::thread> workers_;
    std::vector<XY> tiles_;
    std::atomic<long> waiting_{0};
    const Tracer* T_ = nullptr; const Frame* f_ = nullptr; Counters* ctr_ = nullptr;
    std::vector<Counters> per_thread_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    uint64_t gen_ = 0;
    int done_ = 0;
    bool quit_ = false;
};

// ------------------------------------------------------------------------------------------
// Rasterizer (rasterizer.cpp:21-184)
// ------------------------------------------------------------------------------------------
struct Raster {
    int W, H;
    // work of the last frame as the reference algorithm does it: span pixels visited (the depth test of
    // rasterizer.cpp:177 runs once per such fragment) and triangles that reached the span fill
    uint64_t fragments = 0, fragments_drawn = 0, triangles_drawn = 0;
    vec4 frustum[5];
    std::vector<float> zbuffer;
    std::vector<float> xleft, xright, uleft, uright, vleft, vright, zleft, zright;
};
// rasterizer.cpp:21-46
static Raster* raster_init(int W, int H) {
    Raster* r = new Raster();
    r->W = W; r->H = H;
    r->xleft.assign(H, (float)(W - 1)); r->xright.assign(H, 0.0f);
    r->uleft.assign(H, 0); r->uright.assign(H, 0); r->vleft.assign(H, 0); r->vright.assign(H, 0);
    r->zleft.assign(H, 0); r->zright.assign(H, 0);
    r->zbuffer.assign((size_t)W * H, 0.0f);
    float C = -1.0f, x1 = 0.5f, x2 = W - 1.5f, y1 = 0.5f, y2 = H - 1.5f;
    vec3 p0 = {0, 0, 0};
    vec3 p1 = {((x1 - W / 2) * C) / W, ((y1 - H / 2) * C) / W, 1.0f};
    vec3 p2 = {((x2 - W / 2) * C) / W, ((y1 - H / 2) * C) / W, 1.0f};
    vec3 p3 = {((x2 - W / 2) * C) / W, ((y2 - H / 2) * C) / W, 1.0f};
    vec3 p4 = {((x1 - W / 2) * C) / W, ((y2 - H / 2) * C) / W, 1.0f};
    auto plane = [](vec3 n) { return vec4{n.x, n.y, n.z, 0}; };
    r->frustum[0] = {0, 0, -1, 0.2f};
    r->frustum[1] = plane(normalize(cross(p1 - p0, p4 - p1)));
    r->frustum[2] = plane(normalize(cross(p2 - p0, p1 - p2)));
    r->frustum[3] = plane(normalize(cross(p3 - p0, p2 - p3)));
    r->frustum[4] = plane(normalize(cross(p4 - p0, p3 - p4)));
    return r;
}
static inline vec3 xyz(vec4 v) { return {v.x, v.y, v.z}; }
// lib/glm/detail/type_mat3x3.inl:587-597 with mat3(mat4) = upper-left 3x3
static inline vec3 mul3(const mat4& m, vec3 v) {
    return {m.c[0].x * v.x + m.c[1].x * v.y + m.c[2].x * v.z,
            m.c[0].y * v.x + m.c[1].y * v.y + m.c[2].y * v.z,
            m.c[0].z * v.x + m.c[1].z * v.y + m.c[2].z * v.z};
}
// rasterizer.cpp:89-184 RenderMesh.  tri_ids (optional) receives, per written pixel, the global
// triangle index (mesh-major) of the triangle that last wrote it.
static void raster_mesh(Raster& R, const Scene& S, const mat4& transform, const Mesh& mesh, int tri_base,
                        uint32_t* fb, int32_t* tri_ids) {
    const int W = R.W, H = R.H;
    vec3 c[8];
    for (int i = 0; i < 8; i++)
        c[i] = xyz(mul(transform, vec4{mesh.bounds[i & 1].x, mesh.bounds[(i >> 1) & 1].y, mesh.bounds[i >> 2].z, 1}));
    for (int i, p = 0; p < 5; p++) {
        for (i = 0; i < 8; i++) if ((dot(xyz(R.frustum[p]), c[i]) - R.frustum[p].w) > 0) break;
        if (i == 8) return;
    }
    std::vector<vec3> tpos(mesh.verts);
    for (int i = 0; i < mesh.verts; i++) tpos[i] = xyz(mul(transform, vec4{mesh.pos[i].x, mesh.pos[i].y, mesh.pos[i].z, 1}));
    if (mesh.texture < 0) return;
    const Texture& tx = S.textures[mesh.texture];
    const uint8_t* src = tx.idx.data();
    float* zbuffer = R.zbuffer.data(), f;
    const float tw = (float)tx.w, th = (float)tx.h;
    const int umask = (int)tw - 1, vmask = (int)th - 1;
    for (int i = 0; i < mesh.tris; i++) {
        vec3 Nt = mul3(transform, mesh.N[i]);
        if (dot(tpos[mesh.tri[i * 3 + 0]], Nt) > 0) continue;
        vec3 cpos[2][8], *pos;
        vec2 cuv[2][8], *tuv;
        int nin = 3, nout = 0, from = 0, to = 1, miny = H - 1, maxy = 0, h;
        for (int v = 0; v < 3; v++) cpos[0][v] = tpos[mesh.tri[i * 3 + v]], cuv[0][v] = mesh.uv[mesh.tri[i * 3 + v]];
        for (int p = 0; p < 2; p++, from = 1 - from, to = 1 - to, nin = nout, nout = 0) for (int v = 0; v < nin; v++) {
            const vec3 A = cpos[from][v], B = cpos[from][(v + 1) % nin];
            const vec2 Auv = cuv[from][v], Buv = cuv[from][(v + 1) % nin];
            const vec4 plane = R.frustum[p];
            const float t1 = dot(xyz(plane), A) - plane.w, t2 = dot(xyz(plane), B) - plane.w;
            if ((t1 < 0) && (t2 >= 0)) {
                f = t1 / (t1 - t2);
                cuv[to][nout] = Auv + f * (Buv - Auv); cpos[to][nout++] = A + f * (B - A);
                cuv[to][nout] = Buv; cpos[to][nout++] = B;
            } else if ((t1 >= 0) && (t2 >= 0)) { cuv[to][nout] = Buv; cpos[to][nout++] = B; }
            else if ((t1 >= 0) && (t2 < 0)) {
                f = t1 / (t1 - t2);
                cuv[to][nout] = Auv + f * (Buv - Auv); cpos[to][nout++] = A + f * (B - A);
            }
        }
        if (nin == 0) continue;
        const uint32_t* pal = tx.pal.data() + 256 * (int)(gmax(0.0f, Nt.z) * (32 - 1));
        pos = cpos[from]; tuv = cuv[from];
        for (int v = 0; v < nin; v++) {
            pos[v].x = ((pos[v].x * W) / -pos[v].z) + W / 2;
            pos[v].y = ((pos[v].y * W) / pos[v].z) + H / 2;
        }
        for (int j = 0; j < nin; j++) {
            int vert0 = j, vert1 = (j + 1) % nin;
            if (pos[vert0].y > pos[vert1].y) h = vert0, vert0 = vert1, vert1 = h;
            const float y0 = pos[vert0].y, y1 = pos[vert1].y, rydiff = 1.0f / (y1 - y0);
            if ((y0 == y1) || (y0 >= H) || (y1 < 1)) continue;
            const int iy0 = gmax(1, (int)y0 + 1), iy1 = gmin(H - 2, (int)y1);
            float x0 = pos[vert0].x, dx = (pos[vert1].x - x0) * rydiff;
            float z0 = 1.0f / pos[vert0].z, z1 = 1.0f / pos[vert1].z, dz = (z1 - z0) * rydiff;
            float u0 = tuv[vert0].x * z0, du = (tuv[vert1].x * z1 - u0) * rydiff;
            float v0 = tuv[vert0].y * z0, dv = (tuv[vert1].y * z1 - v0) * rydiff;
            const float ff = (float)iy0 - y0;
            x0 += dx * ff; u0 += du * ff; v0 += dv * ff; z0 += dz * ff;
            for (int y = iy0; y <= iy1; y++) {
                if (x0 < R.xleft[y]) R.xleft[y] = x0, R.uleft[y] = u0, R.vleft[y] = v0, R.zleft[y] = z0;
                if (x0 > R.xright[y]) R.xright[y] = x0, R.uright[y] = u0, R.vright[y] = v0, R.zright[y] = z0;
                x0 += dx; u0 += du; v0 += dv; z0 += dz;
            }
            miny = gmin(miny, iy0); maxy = gmax(maxy, iy1);
        }
        R.triangles_drawn++;
        for (int y = miny; y <= maxy; R.xleft[y] = (float)(W - 1), R.xright[y++] = 0) {
            float x0 = R.xleft[y], x1 = R.xright[y], rxdiff = 1.0f / (x1 - x0);
            float u0 = R.uleft[y], du = (R.uright[y] - u0) * rxdiff;
            float v0 = R.vleft[y], dv = (R.vright[y] - v0) * rxdiff;
            float z0 = R.zleft[y], dz = (R.zright[y] - z0) * rxdiff;
            const int ix0 = (int)x0 + 1, ix1 = gmin(W - 2, (int)x1);
            const float ff = (float)ix0 - x0;
            u0 += ff * du; v0 += ff * dv; z0 += ff * dz;
            uint32_t* dest = fb + (size_t)y * W;
            float* zbuf = zbuffer + (size_t)y * W;
            if (ix1 >= ix0) R.fragments += (uint64_t)(ix1 - ix0 + 1);
            for (int x = ix0; x <= ix1; x++, u0 += du, v0 += dv, z0 += dz) {
                if (z0 >= zbuf[x]) continue;
                R.fragments_drawn++;
                const float z = 1.0f / z0;
                const int u = (int)(u0 * z * tw) & umask, v = (int)(v0 * z * th) & vmask;
                dest[x] = pal[src[u + v * (umask + 1)]]; zbuf[x] = z0;
                if (tri_ids) tri_ids[(size_t)y * W + x] = tri_base + i;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// OBJ / MTL / .bin readers (scenedata.cpp:77-111, 121-271; surface.cpp:39-59)
// ------------------------------------------------------------------------------------------
static bool load_bin_texture(const std::string& image_file, Texture& t) {
    // surface.cpp:41-59: replace the extension after the last '.' by ".bin"
    std::string bin = image_file;
    size_t dot_ = bin.find('.', 1);
    size_t last = std::string::npos;
    while (dot_ != std::string::npos) { last = dot_; dot_ = bin.find('.', dot_ + 1); }
    if (last != std::string::npos) bin.erase(last);
    bin += ".bin";
    FILE* f = fopen(bin.c_str(), "rb");
    if (!f) return false;
    int32_t w = 0, h = 0;
    bool ok = fread(&w, 4, 1, f) == 1 && fread(&h, 4, 1, f) == 1 && w > 0 && h > 0;
    if (ok) {
        t.w = w; t.h = h;
        t.idx.resize((size_t)w * h);
        t.pal.resize(32 * 256);
        ok = fread(t.idx.data(), (size_t)w * h, 1, f) == 1 && fread(t.pal.data(), 4, 32 * 256, f) == 32 * 256;
    }
    fclose(f);
    t.name = image_file;
    return ok;
}

struct Material { std::string name; int texture = -1; };

static std::string extract_path(const std::string& file) {   // scenedata.cpp:39-49
    size_t s = file.find_last_of("/\\");
    return s == std::string::npos ? std::string() : file.substr(0, s + 1);
}

static void load_mtl(Scene& S, std::vector<Material>& mats, const std::string& scenePath, const char* file) {
    std::string fname = scenePath;                     // scenedata.cpp:80-82 ("assets/" + "/" + file)
    if (!fname.empty()) fname += "/";
    fname += file;
    FILE* f = fopen(fname.c_str(), "r");
    if (!f) return;
    int current = -1;
    char line[1024], cmd[256], matName[128];
    while (!feof(f)) {
        line[0] = 0;
        if (!fgets(line, 1022, f)) line[0] = 0;
        if (!line[0]) continue;
        if (line[strlen(line) - 1] < 32) line[strlen(line) - 1] = 0;
        cmd[0] = 0;
        sscanf(line, "%255s", cmd);
        if (!strcasecmp(cmd, "newmtl")) {
            matName[0] = 0;
            sscanf(line + strlen(cmd), "%127s", matName);
            mats.push_back(Material{matName, -1});
            current = (int)mats.size() - 1;
        }
        if (strcasecmp(cmd, "map_Kd")) continue;
        char* tname = strstr(line, " ");
        if (!tname) continue; else tname++;
        std::string tfile = scenePath + "textures/" + tname;
        int ti = -1;
        for (size_t i = 0; i < S.textures.size(); ++i) if (S.textures[i].name == tfile) ti = (int)i;
        if (ti < 0) {
            Texture t;
            if (!load_bin_texture(tfile, t)) { fprintf(stderr, "oracle: texture cache missing for %s\n", tfile.c_str()); continue; }
            S.textures.push_back(std::move(t));
            ti = (int)S.textures.size() - 1;
        }
        if (current >= 0) mats[current].texture = ti;
    }
    fclose(f);
}

// scenedata.cpp:121-271 Scene::LoadOBJ (scene-graph nodes are not kept: every node transform in
// the default scene is the identity, see raster_render)
static bool load_obj(Scene& S, const char* file, float scale) {
    struct UniqueVertex { int vertex, normal, uv, next, subid, idx; };
    auto UV = [](int v, int n = -1, int t = -1) { return UniqueVertex{v, n, t, -1, 0, 0}; };
    FILE* f = fopen(file, "r");
    if (!f) return false;
    std::string scenePath = extract_path(file);
    std::vector<Material> mats;
    std::vector<vec3> vlist, nlist, vlist_, nlist_;
    std::vector<vec2> uvlist, uvlist_;
    std::vector<unsigned> index, index_;
    std::vector<UniqueVertex> unique;
    int subID = 1, formata;
    int current = -1, nextMesh = -1;    // indices into `pending`
    struct Pending { int material; };
    std::vector<Pending> pending;
    while (!feof(f)) {
        current = nextMesh; subID++; formata = -1;
        bool hasUV = false;
        char line[2048], tmp[2048];
        while (!feof(f)) {
            line[0] = 0;
            if (!fgets(line, 1023, f)) line[0] = 0;
            if (!strncasecmp(line, "mtllib", 6)) {
                sscanf(line + 7, "%2047s", tmp);
                load_mtl(S, mats, scenePath, tmp);
                formata = -1; hasUV = false;
            }
            if (!strncasecmp(line, "usemtl", 6)) {
                sscanf(line + 7, "%2047s", tmp);
                int mi = -1;
                for (size_t i = 0; i < mats.size(); ++i) if (mats[i].name == tmp) { mi = (int)i; break; }
                pending.push_back(Pending{mi});
                nextMesh = (int)pending.size() - 1;
                subID++;
                break;
            }
            if (line[0] == 'g') formata = -1;
            if (line[0] == 'v') {
                if (line[1] == ' ') {
                    vec3 vertex = {0, 0, 0};
                    sscanf(line + 2, "%f %f %f", &vertex.x, &vertex.y, &vertex.z);
                    vlist.push_back(vertex * scale);
                    unique.push_back(UV((int)vlist.size() - 1));
                } else if (line[1] == 't') {
                    vec2 uv = {0, 0};
                    sscanf(line + 3, "%f %f", &uv.x, &uv.y);
                    uv.y = 1.0f - uv.y;
                    uvlist.push_back(uv);
                } else if (line[1] == 'n') {
                    vec3 normal = {0, 0, 0};
                    sscanf(line + 3, "%f %f %f", &normal.x, &normal.y, &normal.z);
                    nlist.push_back(normal);
                }
            }
            if (line[0] != 'f') continue;
            if (formata == -1) {
                formata = 0;
                for (int i = 0; i < (int)strlen(line); i++) if (line[i] == '/' && line[i + 1] == '/') formata = 1;
            }
            int v[3] = {0, 0, 0}, n[3] = {0, 0, 0}, t[3] = {0, 0, 0};
            if (formata) sscanf(line + 2, "%i//%i %i//%i %i//%i", &v[0], &n[0], &v[1], &n[1], &v[2], &n[2]);
            sscanf(line + 2, "%i/%i/%i %i/%i/%i %i/%i/%i", &v[0], &t[0], &n[0], &v[1], &t[1], &n[1], &v[2], &t[2], &n[2]);
            for (int i = 0; i < 3; i++) {
                int vidx = v[i] - 1, idx = vidx, lastIdx = idx, nidx = n[i] - 1, uvidx = t[i] - 1;
                if (uvidx > -1) hasUV = true;
                if (vidx < 0 || vidx >= (int)unique.size()) continue;   // malformed face: the reference would read out of bounds
                bool emitted = false;
                do {
                    UniqueVertex& u = unique[idx];
                    if (u.subid != subID) {
                        u.subid = subID; u.next = -1; u.vertex = vidx; u.normal = nidx; u.uv = uvidx;
                        index.push_back(idx);
                        emitted = true;
                        break;
                    } else if ((u.normal == nidx) && (u.uv == uvidx)) {
                        index.push_back(idx);
                        emitted = true;
                        break;
                    }
                    lastIdx = idx; idx = u.next;
                } while (idx > -1);
                if (emitted) continue;
                unsigned newIdx = (unsigned)unique.size();
                unique[lastIdx].next = (int)newIdx;
                index.push_back(newIdx);
                unique.push_back(UV(vidx, nidx, uvidx));
                unique[newIdx].subid = subID;
            }
        }
        if (current < 0) continue; else subID++;
        vlist_.clear(); nlist_.clear(); uvlist_.clear(); index_.clear();
        for (size_t i = 0; i < index.size(); i++) {
            UniqueVertex& u = unique[index[i]];
            if (u.subid == subID) index_.push_back(u.idx);
            else {
                vlist_.push_back(vlist[u.vertex]);
                nlist_.push_back((u.normal >= 0 && u.normal < (int)nlist.size()) ? nlist[u.normal] : vec3{0, 0, 0});
                if (hasUV) uvlist_.push_back((u.uv >= 0 && u.uv < (int)uvlist.size()) ? uvlist[u.uv] : vec2{0, 1});
                else uvlist_.push_back(vec2{0, 1});
                index_.push_back((unsigned)vlist_.size() - 1);
                u.idx = (int)vlist_.size() - 1; u.subid = subID;
            }
        }
        Mesh m;
        int nv = m.verts = (int)vlist_.size(), nt = m.tris = (int)index_.size() / 3;
        m.pos = vlist_; m.uv = uvlist_; m.norm = nlist_;
        m.tri.assign(index_.begin(), index_.begin() + nt * 3);
        m.N.resize(nt);
        m.texture = pending[current].material >= 0 ? mats[pending[current].material].texture : -1;
        for (int i = 0; i < nt; i++) {
            vec3 v0 = vlist_[index_[i * 3 + 0]], v1 = vlist_[index_[i * 3 + 1]], v2 = vlist_[index_[i * 3 + 2]];
            m.N[i] = normalize(cross(v1 - v0, v2 - v0));
            if (dot(m.N[i], nlist_[index_[i * 3 + 1]]) < 0) m.N[i] = m.N[i] * -1.0f;
        }
        vec3& bmin = m.bounds[0], &bmax = m.bounds[1];
        bmin = {1e30f, 1e30f, 1e30f}; bmax = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < nv; i++) {
            bmin.x = gmin(bmin.x, m.pos[i].x); bmax.x = gmax(bmax.x, m.pos[i].x);
            bmin.y = gmin(bmin.y, m.pos[i].y); bmax.y = gmax(bmax.y, m.pos[i].y);
            bmin.z = gmin(bmin.z, m.pos[i].z); bmax.z = gmax(bmax.z, m.pos[i].z);
        }
        S.meshes.push_back(std::move(m));
        while (unique.size() > vlist.size()) unique.pop_back();
        index.clear();
    }
    fclose(f);
    return true;
}

// hashes used by the recorded known answers
static inline uint64_t fold(uint64_t h, uint64_t v) { return h * 1099511628211ull ^ v; }

struct Context {
    Scene scene;
    BVH* bvh = nullptr;
    Tracer tracer;
    TilePool* pool = nullptr;
    Raster* raster = nullptr;
    Counters counters;
};

}  // namespace ora

// ==========================================================================================
// C entry points for the test harness (ctypes)
// ==========================================================================================
using namespace ora;
extern "C" {

Context* ora_create() {
    Context* c = new Context();
    c->tracer.lights.push_back(Light{{0, 5, 0}, {1, 1, 1, 1}});   // raytracer.cpp:396-399
    return c;
}
void ora_destroy(Context* c) {
    if (!c) return;
    delete c->pool; delete c->bvh; delete c->raster; delete c;
}
int ora_load_obj(Context* c, const char* path, float scale) { return load_obj(c->scene, path, scale) ? 0 : -1; }

int ora_add_texture(Context* c, int w, int h, const uint8_t* idx, const uint32_t* pal32x256) {
    Texture t; t.w = w; t.h = h;
    t.idx.assign(idx, idx + (size_t)w * h);
    t.pal.assign(pal32x256, pal32x256 + 32 * 256);
    c->scene.textures.push_back(std::move(t));
    return (int)c->scene.textures.size() - 1;
}
int ora_add_mesh(Context* c, int verts, int tris, const float* pos, const float* uv, const float* norm,
                 const float* triN, const int32_t* tri, int texture, const float* bounds6) {
    Mesh m; m.verts = verts; m.tris = tris;
    m.pos.resize(verts); m.uv.resize(verts); m.norm.resize(verts); m.N.resize(tris);
    memcpy(m.pos.data(), pos, sizeof(vec3) * verts);
    memcpy(m.uv.data(), uv, sizeof(vec2) * verts);
    if (norm) memcpy(m.norm.data(), norm, sizeof(vec3) * verts);
    memcpy(m.N.data(), triN, sizeof(vec3) * tris);
    m.tri.assign(tri, tri + 3 * (size_t)tris);
    m.texture = texture;
    if (bounds6) { m.bounds[0] = {bounds6[0], bounds6[1], bounds6[2]}; m.bounds[1] = {bounds6[3], bounds6[4], bounds6[5]}; }
    else {
        m.bounds[0] = {1e30f, 1e30f, 1e30f}; m.bounds[1] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < verts; i++) {
            m.bounds[0] = vmin(m.bounds[0], m.pos[i]); m.bounds[1] = vmax(m.bounds[1], m.pos[i]);
        }
    }
    c->scene.meshes.push_back(std::move(m));
    return (int)c->scene.meshes.size() - 1;
}
void ora_set_material(Context* c, int mesh, float refl, float refr, float rindex) {
    Mesh& m = c->scene.meshes[mesh]; m.refl = refl; m.refr = refr; m.rindex = rindex;
}
void ora_default_materials(Context* c) { default_materials(c->scene); }
int ora_n_meshes(Context* c) { return (int)c->scene.meshes.size(); }
int ora_n_textures(Context* c) { return (int)c->scene.textures.size(); }
void ora_mesh_info(Context* c, int i, int32_t* verts, int32_t* tris, int32_t* texture, float* bounds6) {
    const Mesh& m = c->scene.meshes[i];
    *verts = m.verts; *tris = m.tris; *texture = m.texture;
    memcpy(bounds6, &m.bounds[0], 24);
}
void ora_mesh_data(Context* c, int i, float* pos, float* uv, float* norm, float* triN, int32_t* tri) {
    const Mesh& m = c->scene.meshes[i];
    memcpy(pos, m.pos.data(), sizeof(vec3) * m.verts);
    memcpy(uv, m.uv.data(), sizeof(vec2) * m.verts);
    memcpy(norm, m.norm.data(), sizeof(vec3) * m.verts);
    memcpy(triN, m.N.data(), sizeof(vec3) * m.tris);
    memcpy(tri, m.tri.data(), sizeof(int) * 3 * m.tris);
}
void ora_texture_info(Context* c, int i, int32_t* w, int32_t* h) { *w = c->scene.textures[i].w; *h = c->scene.textures[i].h; }
void ora_texture_data(Context* c, int i, uint8_t* idx, uint32_t* pal) {
    const Texture& t = c->scene.textures[i];
    memcpy(idx, t.idx.data(), t.idx.size());
    memcpy(pal, t.pal.data(), 4 * 32 * 256);
}
void ora_set_lights(Context* c, const float* pos_rgba7, int n) {
    c->tracer.lights.clear();
    for (int i = 0; i < n; ++i) {
        const float* p = pos_rgba7 + 7 * i;
        c->tracer.lights.push_back(Light{{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}});
    }
}

int ora_build_bvh(Context* c) {
    delete c->bvh;
    c->bvh = build_bvh(c->scene);
    c->tracer.bvh = c->bvh;
    return 0;
}
// info[0..4] = nTris, nNodes(alloc), poolPtr, maxDepth, node slots valid (poolPtr+1)
void ora_bvh_info(Context* c, int64_t* info) {
    info[0] = c->bvh->nTris; info[1] = c->bvh->nNodes; info[2] = c->bvh->poolPtr; info[3] = c->bvh->maxDepth;
    info[4] = (int64_t)c->bvh->poolPtr + 1;
}
void ora_bvh_arrays(Context* c, float* node_minmax6, uint32_t* node_first_count2, uint32_t* tri_idx) {
    const BVH& b = *c->bvh;
    for (uint32_t i = 0; i <= b.poolPtr && i < b.nodes.size(); ++i) {
        memcpy(node_minmax6 + 6 * i, &b.nodes[i].mn, 24);
        node_first_count2[2 * i] = b.nodes[i].firstLeft;
        node_first_count2[2 * i + 1] = b.nodes[i].count;
    }
    memcpy(tri_idx, b.idx.data(), 4 * (size_t)b.nTris);
}
uint64_t ora_idxhash(Context* c) {
    uint64_t h = 0;
    for (uint32_t i = 0; i < c->bvh->nTris; ++i) h = fold(h, c->bvh->idx[i]);
    return h;
}
uint64_t ora_nodehash(Context* c) {
    const BVH& b = *c->bvh;
    uint64_t h = 0;
    for (uint32_t i = 0; i <= b.poolPtr; ++i) {
        if (i == 1 || i == 2) continue;
        const Node& n = b.nodes[i];
        h = fold(h, n.firstLeft); h = fold(h, n.count); h = fold(h, fbits(n.mn.x)); h = fold(h, fbits(n.mx.z));
    }
    return h;
}
uint64_t ora_fbhash(const uint32_t* fb, int64_t n) {
    uint64_t h = 0;
    for (int64_t i = 0; i < n; ++i) h = fold(h, fb[i]);
    return h;
}
// the same fold continued from h0: the hash of a frame from its row bands, top to bottom
uint64_t ora_fbhash_from(uint64_t h0, const uint32_t* fb, int64_t n) {
    uint64_t h = h0;
    for (int64_t i = 0; i < n; ++i) h = fold(h, fb[i]);
    return h;
}
// sum over packets (tile order y-major, x-minor), k = ray index inside the 8x8 tile
uint64_t ora_idsum(const int32_t* hit_ids, int W, int H) {
    uint64_t s = 0;
    for (int ty = 0; ty < H; ty += 8) for (int tx = 0; tx < W; tx += 8)
        for (int k = 0; k < 64; ++k) {
            int id = hit_ids[(tx + k % 8) + (ty + k / 8) * W];
            if (id >= 0) s += (uint64_t)id * (uint64_t)(k + 1);
        }
    return s;
}

void ora_lookat(const float* pos3, const float* target3, float* out16) {
    // scenedata.h:129-134: SetPosition on the current transform, then LookAt = inverse(lookAt(pos, target, (0,1,0)))
    mat4 m = inverse(lookAt({pos3[0], pos3[1], pos3[2]}, {target3[0], target3[1], target3[2]}, {0, 1, 0}));
    memcpy(out16, &m, 64);
}
void ora_mat4_inverse(const float* in16, float* out16) { mat4 m; memcpy(&m, in16, 64); m = inverse(m); memcpy(out16, &m, 64); }
void ora_view_plane(const float* cam16, int W, int H, float* out11) {
    mat4 m; memcpy(&m, cam16, 64);
    Frame f; setup_view(f, m, W, H);
    memcpy(out11, &f.p0, 12); memcpy(out11 + 3, &f.p1, 12); memcpy(out11 + 6, &f.p2, 12);
    out11[9] = f.invW; out11[10] = f.invH;
}

void ora_set_threads(Context* c, int n) {
    if (c->pool && c->pool->size() == n && c->pool->pinned() == 0) return;
    delete c->pool;
    c->pool = new TilePool(n);
}
// The same pool with every worker pinned to its own CPU, distinct physical cores first (sched affinity of the
// worker threads only: the calling thread and the rest of the process keep theirs).  Returns the number of
// distinct physical cores the workers sit on; 0 if pinning was not possible.
int ora_set_threads_pinned(Context* c, int n) {
    delete c->pool;
    c->pool = new TilePool(n, true);
    return c->pool->pinned() == n ? c->pool->distinct_cores() : 0;
}
// mode 0 = packet path (Raytracer::Render), 1 = recursive shader per pixel.  Rows [y_begin,y_end)
// only (multiples of 8).  spp_grid g > 1: the frame is rendered by the reference algorithm at
// (W*g) x (H*g) and each output pixel is the per-channel mean (rounded down) of its g x g block
// of quantised pixels; hit_ids/tuv then describe the (W*g) x (H*g) image.
// counters_out: 12 x uint64 or NULL.
int ora_render(Context* c, const float* cam16, int W, int H, int mode, int spp_grid, int y_begin, int y_end,
               uint32_t* fb, int pitch, int32_t* hit_ids, float* tuv, uint64_t* counters_out) {
    const int g = spp_grid < 1 ? 1 : spp_grid;
    if (W % 8 || H % 8 || y_begin % 8 || y_end % 8 || !c->bvh) return -1;
    if (!c->pool) c->pool = new TilePool(1);
    mat4 m; memcpy(&m, cam16, 64);
    Frame f; setup_view(f, m, W * g, H * g);
    std::vector<uint32_t> big;
    if (g > 1) { big.assign((size_t)W * g * H * g, 0u); f.fb = big.data(); f.pitch = W * g; }
    else { f.fb = fb; f.pitch = pitch; }
    f.hit_ids = hit_ids; f.tuv = tuv; f.mode = mode;
    Counters ctr;
    c->pool->run(c->tracer, f, y_begin * g, y_end * g, counters_out ? &ctr : nullptr);
    if (g > 1) {
        const uint32_t n = (uint32_t)(g * g);
        for (int y = y_begin; y < y_end; ++y) for (int x = 0; x < W; ++x) {
            uint32_t a[4] = {0, 0, 0, 0};
            for (int j = 0; j < g; ++j) for (int i = 0; i < g; ++i) {
                uint32_t px = big[(size_t)(y * g + j) * W * g + (x * g + i)];
                a[0] += px >> 24; a[1] += (px >> 16) & 255; a[2] += (px >> 8) & 255; a[3] += px & 255;
            }
            fb[x + (size_t)y * pitch] = ((a[0] / n) << 24) | ((a[1] / n) << 16) | ((a[2] / n) << 8) | (a[3] / n);
        }
    }
    if (mode == 2) {
        // second half of the depth view: colour = clamp((1, depth / maxDepth, 1 - depth / maxDepth, 0)).  The reference
        // divides by a running maximum that its worker threads update without synchronisation and that survives from
        // frame to frame (raytracer.cpp:523, 584); its value once a static view has been rendered is the frame's maximum,
        // which is what is used here for every pixel.
        int maxDepth = 0;
        for (int y = y_begin; y < y_end; ++y) for (int x = 0; x < W; ++x) maxDepth = std::max(maxDepth, (int)fb[x + (size_t)y * pitch]);
        for (int y = y_begin; y < y_end; ++y) for (int x = 0; x < W; ++x) {
            const int depth = (int)fb[x + (size_t)y * pitch];
            // (a scene without triangles has depth 0 everywhere: the reference would divide 0 by 0, shown here as 0)
            const float part = maxDepth > 0 ? (float)depth / maxDepth : 0.0f;
            const vec4 col = vclamp(vec4{1, part, 1.0f - part, 0}, 0.0f, 1.0f);
            fb[x + (size_t)y * pitch] = vec_to_pixel(col);
        }
    }
    if (counters_out) memcpy(counters_out, &ctr, sizeof(ctr));
    return 0;
}
// the analytic primitives of the frame: n records of 13 floats {kind, pos.xyz, radius | dist, diffuse.xyzw, refl, refr, rindex}
void ora_set_objects(Context* c, const float* rec13, int n) {
    c->tracer.objects.clear();
    for (int i = 0; i < n; ++i) {
        const float* p = rec13 + 13 * i;
        c->tracer.objects.push_back(Object{(int)p[0], {p[1], p[2], p[3]}, p[4], {p[5], p[6], p[7], p[8]}, p[9], p[10], p[11]});
    }
}
// single rays, for known-answer checks: kind 0 = packet-of-one semantics is not defined, so
// kind 0 = scalar traversal from the root WITHOUT the root gate, 1 = BVH::Traverse (with gate)
int ora_trace_ray(Context* c, const float* O3, const float* D3, float tmax, int kind, float* t_out, float* uv_out) {
    Ray r{{O3[0], O3[1], O3[2]}, {D3[0], D3[1], D3[2]}, tmax};
    HitRec h{-1, 0, 0};
    if (kind == 1) bvh_traverse(*c->bvh, r, h); else traverse_scalar(*c->bvh, 0, r, h);
    *t_out = r.t; uv_out[0] = h.u; uv_out[1] = h.v;
    return h.tri;
}
// batch forms of the two traversal operators, for parity tests on caller-made rays
void ora_trace_packets(Context* c, int n_packets, const float* O, const float* D, const float* rD, float* t,
                       int32_t* tri, float* u, float* v) {
    for (int p = 0; p < n_packets; ++p) {
        RayPacket pk; HitPacket res;
        const size_t b = (size_t)p * PACKETSIZE;
        memcpy(pk.O, O + 3 * b, sizeof(pk.O)); memcpy(pk.D, D + 3 * b, sizeof(pk.D));
        memcpy(pk.rD, rD + 3 * b, sizeof(pk.rD)); memcpy(pk.t, t + b, sizeof(pk.t));
        for (int i = 0; i < PACKETSIZE; ++i) { res.tri[i] = -1; res.u[i] = 0; res.v[i] = 0; }
        bvh_traverse_packet(*c->bvh, pk, res);
        memcpy(t + b, pk.t, sizeof(pk.t));
        for (int i = 0; i < PACKETSIZE; ++i) { tri[b + i] = res.tri[i]; u[b + i] = res.u[i]; v[b + i] = res.v[i]; }
    }
}
void ora_trace_rays(Context* c, int n, const float* O, const float* D, float* t, int32_t* tri, float* u, float* v) {
    for (int i = 0; i < n; ++i) {
        Ray r{{O[3 * i], O[3 * i + 1], O[3 * i + 2]}, {D[3 * i], D[3 * i + 1], D[3 * i + 2]}, t[i]};
        HitRec h{-1, 0, 0};
        bvh_traverse(*c->bvh, r, h);
        t[i] = r.t; tri[i] = h.tri; u[i] = h.u; v[i] = h.v;
    }
}
// brute force over all triangles in storage order (cross-check of the traversal)
int ora_trace_ray_brute(Context* c, const float* O3, const float* D3, float tmax, float* t_out, float* uv_out) {
    vec3 O{O3[0], O3[1], O3[2]}, D{D3[0], D3[1], D3[2]};
    float t = tmax, u = 0, v = 0; int best = -1;
    for (uint32_t i = 0; i < c->bvh->nTris; ++i) {
        float uu, vv;
        if (tri_intersect(c->bvh->tris[i], D, O, t, uu, vv)) { best = (int)i; u = uu; v = vv; }
    }
    *t_out = t; uv_out[0] = u; uv_out[1] = v;
    return best;
}
void ora_gen_ray(const float* cam16, int W, int H, int x, int y, float* D3, float* rD3) {
    mat4 m; memcpy(&m, cam16, 64);
    Frame f; setup_view(f, m, W, H);
    vec3 O, D, rD; gen_ray(f, x, y, f.invW, f.invH, O, D, rD);
    memcpy(D3, &D, 12); memcpy(rD3, &rD, 12);
}

int ora_raster_render(Context* c, const float* cam16, int W, int H, uint32_t* fb, float* zbuf_out, int32_t* tri_ids) {
    if (!c->raster || c->raster->W != W || c->raster->H != H) { delete c->raster; c->raster = raster_init(W, H); }
    Raster& R = *c->raster;
    std::fill(R.zbuffer.begin(), R.zbuffer.end(), 0.0f);           // rasterizer.cpp:55
    R.fragments = R.fragments_drawn = R.triangles_drawn = 0;
    mat4 cam; memcpy(&cam, cam16, 64);
    mat4 inv = inverse(cam);                                        // rasterizer.cpp:57
    // rasterizer.cpp:65-72: scene root -> OBJ root -> group -> mesh, all identity local transforms
    mat4 I = mat4_identity();
    mat4 M = mul(mul(mul(mul(inv, I), I), I), I);
    int base = 0;
    for (auto& m : c->scene.meshes) { raster_mesh(R, c->scene, M, m, base, fb, tri_ids); base += m.tris; }
    if (zbuf_out) memcpy(zbuf_out, R.zbuffer.data(), sizeof(float) * (size_t)W * H);
    return 0;
}

// out[0] span pixels visited, out[1] of them passing the depth test, out[2] triangles that reached the span fill -- last ora_raster_render
void ora_raster_work(Context* c, uint64_t* out) {
    out[0] = c->raster ? c->raster->fragments : 0;
    out[1] = c->raster ? c->raster->fragments_drawn : 0;
    out[2] = c->raster ? c->raster->triangles_drawn : 0;
}

}  // extern "C"
