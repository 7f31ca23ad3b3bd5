// The group boxes of long leaves (device_scene.h: triG) must never hide a triangle the ray/triangle test would hit.
// 1. half_outward: binary16 bounds that contain the value, on zero / normal numbers / infinities only.
// 2. Random groups of four triangles, boxed and padded the way mrt_scene_upload does it, against rays chosen to be
//    awkward (aimed at vertices, edge points and points just outside the triangles, starting on a triangle, along
//    the axes, far away): whenever the triangle test of raytracer.cpp:287-356 (same operations, -ffp-contract=off)
//    reports a hit, the box test the kernels use (group_touched, restated here in plain C++) must say "touched".
//    Rays are aimed at points of the triangles from generic origins, so none of them runs inside a triangle's plane
//    (the case kernels.hip: not_grazing describes); groups holding a sliver get the box no ray misses.
// 3. Rays that graze a wall (see below): hits hidden without the eye-in-plane guard are counted, none may be with it.
// usage: check_groups [rays]
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "mrt_math.h"
using namespace mrt;

static uint64_t s = 0x2545F4914F6CDD1Dull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float uni(float a, float b) { return a + (b - a) * (float)((rnd() >> 11) * (1.0 / 9007199254740992.0)); }

static bool tri_hit(f3 p0, f3 p1, f3 p2, f3 O, f3 D, float tmax) {
    const f3 e1 = p1 - p0, e2 = p2 - p0;
    const f3 pvec = cross(D, e2);
    const float det = dot(e1, pvec);
    if (det < -0.000001f) return false;
    const float invDet = 1 / det;
    const f3 tvec = O - p0;
    const float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(D, qvec) * invDet;
    if (v < 0 || u + v > 1) return false;
    const float T = dot(e2, qvec) * invDet;
    return !(T < 0.0f) && T < tmax;
}
static bool touched(const uint16_t h[6], f3 O, f3 r, float t) {
    const f3 lo = {half_value(h[0]), half_value(h[1]), half_value(h[2])}, hi = {half_value(h[3]), half_value(h[4]), half_value(h[5])};
    const f3 t0 = (lo - O) * r, t1 = (hi - O) * r;
    const float tn = std::max(std::max(std::min(t0.x, t1.x), std::min(t0.y, t1.y)), std::min(t0.z, t1.z));
    const float tf = std::min(std::min(std::max(t0.x, t1.x), std::max(t0.y, t1.y)), std::max(t0.z, t1.z));
    return tn <= tf && tf >= 0.0f && tn <= t;
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 4000000;
    long bad = 0;
    for (uint32_t u = 0; u < 0x7f800000u; u += 4099) for (int sg = 0; sg < 2; ++sg) {
        const uint32_t b = u | (sg ? 0x80000000u : 0u);
        float x; memcpy(&x, &b, 4);
        const uint16_t l = half_outward(x, false), h = half_outward(x, true);
        const bool shape_ok = (((l >> 10) & 31) != 0 || (l & 0x3ff) == 0) && (((h >> 10) & 31) != 0 || (h & 0x3ff) == 0);
        if (!(half_value(l) <= x && x <= half_value(h)) || !shape_ok) { if (bad < 5) printf("half_outward(%a)\n", x); ++bad; }
    }
    long hits = 0, skipped_tests = 0, tests = 0, slivers = 0;
    for (long n = 0; n < N;) {
        const float scale = (rnd() % 4 == 0) ? uni(0.05f, 2000.0f) : 20.0f;      // the scene's largest coordinate
        const float pad = scale * 0x1p-12f;
        f3 tri[4][3];
        const f3 c = {uni(-scale, scale), uni(-scale, scale), uni(-scale, scale)};
        const float ext = uni(0.001f, 0.3f) * scale;
        const int flat = (int)(rnd() % 4);                                         // 0..2: a wall in that plane
        for (auto& t : tri) for (auto& p : t) {
            p = {c.x + uni(-ext, ext), c.y + uni(-ext, ext), c.z + uni(-ext, ext)};
            if (flat == 0) p.x = c.x; if (flat == 1) p.y = c.y; if (flat == 2) p.z = c.z;
            p = {std::min(std::max(p.x, -scale), scale), std::min(std::max(p.y, -scale), scale), std::min(std::max(p.z, -scale), scale)};
        }
        f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
        for (auto& t : tri) for (auto& p : t) {
            lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
            hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
        }
        uint16_t h[6] = {half_outward(lo.x - pad, false), half_outward(lo.y - pad, false), half_outward(lo.z - pad, false),
                         half_outward(hi.x + pad, true), half_outward(hi.y + pad, true), half_outward(hi.z + pad, true)};
        for (auto& t : tri)
            if (triangle_is_sliver(t[0], t[1], t[2])) { h[0] = h[1] = h[2] = 0xfc00; h[3] = h[4] = h[5] = 0x7c00; ++slivers; }    // as mrt_scene_upload does
        for (int k = 0; k < 64; ++k, ++n) {
            const auto& T = tri[rnd() % 4];
            float a = uni(0, 1), b = uni(0, 1);
            switch (rnd() % 6) {                       // where the ray is aimed, in the triangle's own coordinates
                case 0: a = 0; b = uni(0, 1); break;                      // an edge
                case 1: a = (float)(rnd() % 2); b = 0; break;            // a vertex
                case 2: a = uni(-1e-5f, 1e-5f); b = uni(0, 1); break;    // a hair beside an edge
                case 3: if (a + b > 1) { a = 1 - a; b = 1 - b; } break;  // inside
                case 4: a = uni(0, 1); b = 1 - a + uni(-1e-5f, 1e-5f); break;
                default: break;                                           // anywhere in the parallelogram
            }
            const f3 target = T[0] + a * (T[1] - T[0]) + b * (T[2] - T[0]);
            f3 O;
            switch (rnd() % 5) {
                case 0: O = {uni(-scale, scale), uni(-scale, scale), uni(-scale, scale)}; break;
                case 1: O = tri[rnd() % 4][rnd() % 3]; break;                                           // on a vertex of the group
                case 2: { const auto& S = tri[rnd() % 4]; const float p = uni(0, 1), q = uni(0, 1 - p); O = S[0] + p * (S[1] - S[0]) + q * (S[2] - S[0]); break; }   // on a triangle
                case 3: O = {target.x, target.y, uni(-scale, scale)}; break;                            // straight along z
                default: O = {uni(-scale, scale) * 50, uni(-scale, scale) * 50, uni(-scale, scale) * 50}; break;  // far away
            }
            const f3 d = target - O;
            if (d.x == 0 && d.y == 0 && d.z == 0) continue;
            const f3 D = normalize(d);
            const f3 r = 1.0f / D;
            if (!(fabsf(r.x) < INFINITY && fabsf(r.y) < INFINITY && fabsf(r.z) < INFINITY)) continue;   // such rays never skip a group
            const float tmax = (rnd() % 3 == 0) ? uni(0, 3 * scale) : INFINITY;
            const bool box = touched(h, O, r, tmax);
            ++tests; skipped_tests += !box;
            for (auto& t : tri) {
                if (!tri_hit(t[0], t[1], t[2], O, D, tmax)) continue;
                ++hits;
                if (!box) { if (bad < 5) printf("hidden hit: O %a %a %a D %a %a %a\n", O.x, O.y, O.z, D.x, D.y, D.z); ++bad; }
            }
        }
    }
    printf("rays=%ld skipped=%ld hits=%ld slivers=%ld hidden=%ld\n", tests, skipped_tests, hits, slivers, bad);

    // 3. Grazing rays: four triangles of one wall (coplanar, axis-parallel or slanted), the ray's origin a few float
    //    steps beside the wall's plane, the ray running along the wall and meeting its plane at 1e-7 ... 1e-3 rad.  All
    //    numerators of the triangle test are rounding noise for such rays, so a "hit" may be reported far from where
    //    the ray really crosses the plane, outside the group's box: counted as `grazing_hidden`.  The frame kernels
    //    therefore give a group the box no ray misses whenever the eye is within the padding of the plane of one of
    //    its triangles (point_near_plane, the function the kernels call); with that guard nothing may be hidden.
    long g_tests = 0, g_hits = 0, g_hidden = 0, g_hidden_guarded = 0, g_guarded = 0;
    for (long n = 0; n < N;) {
        const float scale = 20.0f, pad = scale * 0x1p-12f;
        f3 tri[4][3];
        const f3 c = {uni(-scale, scale), uni(-scale, scale), uni(-scale, scale)};
        // the wall's plane through c: normal nrm, in-plane axes a and b
        f3 nrm = (rnd() % 3) ? f3{1, 0, 0} : normalize(f3{uni(-1, 1), uni(-1, 1), uni(-1, 1)});
        if (rnd() % 3 == 0) nrm = {0, 0, 1};
        f3 a = normalize(cross(nrm, fabsf(nrm.y) < 0.9f ? f3{0, 1, 0} : f3{1, 0, 0})), b = cross(nrm, a);
        const float ext = uni(0.5f, 6.0f);
        for (auto& t : tri) {
            const float cu = uni(-ext, ext), cv = uni(-ext, ext), sz = uni(0.2f, 3.0f);
            for (auto& p : t) { const float pu = cu + uni(-sz, sz), pv = cv + uni(-sz, sz); p = c + pu * a + pv * b; }
        }
        f3 lo = {INFINITY, INFINITY, INFINITY}, hi = {-INFINITY, -INFINITY, -INFINITY};
        bool sliver = false;
        for (auto& t : tri) {
            sliver = sliver || triangle_is_sliver(t[0], t[1], t[2]);
            for (auto& p : t) {
                lo = {std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z)};
                hi = {std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z)};
            }
        }
        if (sliver) continue;
        const uint16_t h[6] = {half_outward(lo.x - pad, false), half_outward(lo.y - pad, false), half_outward(lo.z - pad, false),
                               half_outward(hi.x + pad, true), half_outward(hi.y + pad, true), half_outward(hi.z + pad, true)};
        for (int k = 0; k < 256; ++k, ++n) {
            // origin: a point of the plane beside or among the triangles, moved off the plane by a few float steps
            const float ou = uni(-3 * ext, 3 * ext), ov = uni(-3 * ext, 3 * ext);
            f3 O = c + ou * a + ov * b;
            const float off = (rnd() % 4 == 0) ? 0.0f : uni(-1.0f, 1.0f) * powf(10.0f, uni(-7.0f, -1.0f));    // inside and outside the guard's reach
            O = O + off * nrm;
            const float ang = uni(0, 6.2831853f), eps = ((rnd() & 1) ? 1.0f : -1.0f) * powf(10.0f, uni(-7.5f, -3.0f));
            const f3 D = normalize((cosf(ang) * a + sinf(ang) * b) + eps * nrm);
            const f3 r = 1.0f / D;
            if (!(fabsf(r.x) < INFINITY && fabsf(r.y) < INFINITY && fabsf(r.z) < INFINITY)) continue;
            const float tmax = (rnd() % 3 == 0) ? uni(0, 3 * scale) : INFINITY;
            const bool box = touched(h, O, r, tmax);
            bool guard = false;
            for (auto& t : tri) guard = guard || point_near_plane(O, t[0], t[1] - t[0], t[2] - t[0], pad);
            ++g_tests; g_guarded += guard;
            for (auto& t : tri) {
                if (!tri_hit(t[0], t[1], t[2], O, D, tmax)) continue;
                ++g_hits;
                if (!box) { ++g_hidden; if (!guard) { if (g_hidden_guarded < 5) printf("hidden from a guarded group: O %a %a %a D %a %a %a\n", O.x, O.y, O.z, D.x, D.y, D.z); ++g_hidden_guarded; } }
            }
        }
    }
    printf("grazing: rays=%ld guarded=%ld hits=%ld grazing_hidden=%ld hidden_with_guard=%ld\n", g_tests, g_guarded, g_hits, g_hidden, g_hidden_guarded);
    return (bad || g_hidden_guarded) ? 1 : 0;
}
