// TEST INFRASTRUCTURE ONLY.  Reference-side probe: compiled (by oracle/Makefile, target `ref`)
// against the glm 0.9.5.4 headers that the reference vendors under
// /root/reference/Project/lib/glm, where they lie.  It prints, as IEEE-754 bit patterns, the
// results of the glm calls the ray path depends on, so that the oracle's restatement of that
// arithmetic (ora_math.h) can be checked against the real library.  The output is committed
// as tests/golden/glm_vectors.json by tests/golden/make_glm_vectors.py.
//
// Input (stdin), one request per line:
//   pose  px py pz tx ty tz W H   -> camera = inverse(lookAt(p, t, (0,1,0))) as Camera::LookAt does
//                                    (scenedata.h:134), its inverse, and the three view-plane
//                                    corners of Raytracer::Render (raytracer.cpp:424-426)
//   vec   seed n                  -> n pseudo-random vec3 pairs: dot, cross, normalize, min, max
#define GLM_FORCE_RADIANS
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <cstdint>
#include <cstdio>
#include <cstring>

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static void pm(const char* tag, const glm::mat4& m) {
    printf("%s", tag);
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) printf(" %08x", bits(m[c][r]));
    printf("\n");
}
static void pv(const char* tag, const glm::vec3& v) { printf("%s %08x %08x %08x\n", tag, bits(v.x), bits(v.y), bits(v.z)); }

int main() {
    char line[512];
    while (fgets(line, sizeof line, stdin)) {
        if (!strncmp(line, "pose", 4)) {
            float p[3], t[3]; int W, H;
            if (sscanf(line + 4, "%f %f %f %f %f %f %d %d", p, p + 1, p + 2, t, t + 1, t + 2, &W, &H) != 8) continue;
            glm::mat4 cam;                                   // identity in this glm version
            cam[3][0] = p[0]; cam[3][1] = p[1]; cam[3][2] = p[2];
            glm::vec3 pos(cam[3][0], cam[3][1], cam[3][2]);
            cam = glm::inverse(glm::lookAt(pos, glm::vec3(t[0], t[1], t[2]), glm::vec3(0, 1, 0)));
            pm("cam", cam);
            pm("inv", glm::inverse(cam));
            float aspect = (float)H / (float)W, fov = 1.8f;
            pv("p0", glm::vec3(cam * glm::vec4(glm::vec3(-1, aspect, -fov), 1)));
            pv("p1", glm::vec3(cam * glm::vec4(glm::vec3(1, aspect, -fov), 1)));
            pv("p2", glm::vec3(cam * glm::vec4(glm::vec3(-1, -aspect, -fov), 1)));
        } else if (!strncmp(line, "vec", 3)) {
            unsigned s; int n;
            if (sscanf(line + 3, "%u %d", &s, &n) != 2) continue;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) * 40.0f - 20.0f; };
            for (int i = 0; i < n; ++i) {
                glm::vec3 a(rnd(), rnd(), rnd()), b(rnd(), rnd(), rnd());
                printf("dot %08x\n", bits(glm::dot(a, b)));
                pv("cross", glm::cross(a, b));
                pv("norm", glm::normalize(a));
                pv("min", glm::min(a, b));
                pv("max", glm::max(a, b));
                pv("div", a / b);
                pv("rdiv", 1.0f / a);
                glm::vec4 c = glm::clamp(glm::vec4(a, b.x) * 0.1f, 0.0f, 1.0f);
                printf("clamp %08x %08x %08x %08x\n", bits(c.x), bits(c.y), bits(c.z), bits(c.w));
            }
        }
    }
    return 0;
}
