// Host-side camera math: the 4x4 inverse and lookAt with the operation order of the glm
// version the reference vendors, so that a caller who replays Camera::SetPosition +
// Camera::LookAt (scenedata.h:129-134) gets a bit-identical camera-to-world matrix.
#include "mrt_math.h"
#include "../../include/mrt.h"

namespace mrt {

// cofactor expansion in the order of detail/type_mat4x4.inl:477-535
m4 inverse(const m4& m) {
    auto a = [&](int c, int r) { return el(m, c, r); };
    // 2x2 sub-determinants, named by the two columns and two rows they use
    const float s2233 = a(2,2) * a(3,3) - a(3,2) * a(2,3);
    const float s1233 = a(1,2) * a(3,3) - a(3,2) * a(1,3);
    const float s1223 = a(1,2) * a(2,3) - a(2,2) * a(1,3);
    const float s2133 = a(2,1) * a(3,3) - a(3,1) * a(2,3);
    const float s1133 = a(1,1) * a(3,3) - a(3,1) * a(1,3);
    const float s1123 = a(1,1) * a(2,3) - a(2,1) * a(1,3);
    const float s2132 = a(2,1) * a(3,2) - a(3,1) * a(2,2);
    const float s1132 = a(1,1) * a(3,2) - a(3,1) * a(1,2);
    const float s1122 = a(1,1) * a(2,2) - a(2,1) * a(1,2);
    const float s2033 = a(2,0) * a(3,3) - a(3,0) * a(2,3);
    const float s1033 = a(1,0) * a(3,3) - a(3,0) * a(1,3);
    const float s1023 = a(1,0) * a(2,3) - a(2,0) * a(1,3);
    const float s2032 = a(2,0) * a(3,2) - a(3,0) * a(2,2);
    const float s1032 = a(1,0) * a(3,2) - a(3,0) * a(1,2);
    const float s1022 = a(1,0) * a(2,2) - a(2,0) * a(1,2);
    const float s2031 = a(2,0) * a(3,1) - a(3,0) * a(2,1);
    const float s1031 = a(1,0) * a(3,1) - a(3,0) * a(1,1);
    const float s1021 = a(1,0) * a(2,1) - a(2,0) * a(1,1);
    const f4 F0{s2233, s2233, s1233, s1223}, F1{s2133, s2133, s1133, s1123}, F2{s2132, s2132, s1132, s1122};
    const f4 F3{s2033, s2033, s1033, s1023}, F4{s2032, s2032, s1032, s1022}, F5{s2031, s2031, s1031, s1021};
    const f4 V0{a(1,0), a(0,0), a(0,0), a(0,0)}, V1{a(1,1), a(0,1), a(0,1), a(0,1)};
    const f4 V2{a(1,2), a(0,2), a(0,2), a(0,2)}, V3{a(1,3), a(0,3), a(0,3), a(0,3)};
    const f4 plus_minus{+1, -1, +1, -1}, minus_plus{-1, +1, -1, +1};
    m4 adj;
    adj.c[0] = ((V1 * F0 - V2 * F1) + V3 * F2) * plus_minus;
    adj.c[1] = ((V0 * F0 - V2 * F3) + V3 * F4) * minus_plus;
    adj.c[2] = ((V0 * F1 - V1 * F3) + V3 * F5) * plus_minus;
    adj.c[3] = ((V0 * F2 - V1 * F4) + V2 * F5) * minus_plus;
    const f4 row0{adj.c[0].x, adj.c[1].x, adj.c[2].x, adj.c[3].x};
    const f4 d = m.c[0] * row0;
    const float det = (d.x + d.y) + (d.z + d.w);
    const float inv_det = 1.0f / det;
    return {{adj.c[0] * inv_det, adj.c[1] * inv_det, adj.c[2] * inv_det, adj.c[3] * inv_det}};
}

// gtc/matrix_transform.inl:416-441
m4 look_at(f3 eye, f3 center, f3 up) {
    const f3 f = normalize(center - eye);
    const f3 s = normalize(cross(f, up));
    const f3 u = cross(s, f);
    m4 r = identity();
    el(r, 0, 0) = s.x;  el(r, 1, 0) = s.y;  el(r, 2, 0) = s.z;
    el(r, 0, 1) = u.x;  el(r, 1, 1) = u.y;  el(r, 2, 1) = u.z;
    el(r, 0, 2) = -f.x; el(r, 1, 2) = -f.y; el(r, 2, 2) = -f.z;
    el(r, 3, 0) = -dot(s, eye);
    el(r, 3, 1) = -dot(u, eye);
    el(r, 3, 2) = dot(f, eye);
    return r;
}

}  // namespace mrt

extern "C" void mrt_camera_look_at(const float pos[3], const float target[3], float out[16]) {
    mrt::m4 m = mrt::inverse(mrt::look_at({pos[0], pos[1], pos[2]}, {target[0], target[1], target[2]}, {0, 1, 0}));
    memcpy(out, &m, 64);
}
