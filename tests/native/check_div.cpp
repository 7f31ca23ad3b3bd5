// Host-side check of mrt::div_by_recip (raytracing_amd/csrc/mrt_math.h) against IEEE division:
// random and adversarial operands inside the documented ranges must give bit-identical
// quotients.  Prints "mismatches=<n>"; exit status 0 iff n == 0.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../raytracing_amd/csrc/mrt_math.h"

static uint64_t s = 88172645463325252ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float mk(uint32_t sign, int e, uint32_t m) {
    uint32_t u = (sign << 31) | ((uint32_t)(e + 127) << 23) | (m & 0x7fffffu);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 20000000;
    long bad = 0, tested = 0;
    for (long i = 0; i < N; ++i) {
        const uint64_t x = rnd(), y = rnd();
        // |a| in [2^-50, 2^31], |d| in [2^-40, 2^12]; mantissas biased towards the extremes one time in eight
        uint32_t ma = (uint32_t)(y & 0x7fffffu), md = (uint32_t)((y >> 23) & 0x7fffffu);
        if ((x >> 20) % 8 == 0) md = ((x >> 24) & 1) ? 0x7ffffeu - (uint32_t)((y >> 50) % 8) : (uint32_t)((y >> 50) % 16);
        if ((x >> 28) % 8 == 0) ma = ((x >> 32) & 1) ? 0x7fffffu - (uint32_t)((y >> 54) % 8) : (uint32_t)((y >> 54) % 16);
        float a = mk((uint32_t)(x >> 63), (int)(x % 81) - 50, ma);
        float d = mk((uint32_t)((x >> 62) & 1), (int)((x >> 8) % 52) - 40, md);
        if ((x >> 36) % 64 == 0) a = ((x >> 42) & 1) ? 0.0f : -0.0f;
        if (!mrt::div_shortcut_dir_ok(d)) continue;
        const float r = 1.0f / d;
        const float want = a / d, got = mrt::div_by_recip(a, d, r);
        ++tested;
        // a zero quotient may differ in sign (no box predicate can see it); everything else is bit-exact
        if (memcmp(&want, &got, 4) && !(want == 0.0f && got == 0.0f)) {
            if (bad < 5) printf("a=%a d=%a want=%a got=%a\n", a, d, want, got);
            ++bad;
        }
        // the bound the interval form of the shadow rays' box test rests on (kernels.hip,
        // slab_pair_shadow_interval): the one-multiply estimate a * RN(1/d) is within 2^-22 of the
        // exact quotient relatively, has its sign, and is zero exactly when it is
        const float est = a * r;
        const double err = (double)want - (double)est, lim = 0x1p-22 * (est < 0 ? -(double)est : (double)est);
        if ((err < 0 ? -err : err) > lim || (want == 0.0f) != (est == 0.0f) || (want < 0.0f) != (est < 0.0f)) {
            if (bad < 5) printf("estimate: a=%a d=%a want=%a est=%a\n", a, d, want, est);
            ++bad;
        }
    }
    // range predicates
    if (mrt::div_shortcut_dir_ok(0.0f) || mrt::div_shortcut_dir_ok(mk(0, 0, 0x7fffffu)) || mrt::div_shortcut_dir_ok(1e-20f) ||
        !mrt::div_shortcut_dir_ok(-0.5f) || !mrt::div_shortcut_coord_ok(0.0f) || !mrt::div_shortcut_coord_ok(-17.8125f) ||
        mrt::div_shortcut_coord_ok(1e-12f) || mrt::div_shortcut_coord_ok(4e9f) || mrt::div_shortcut_coord_ok(__builtin_nanf("")))
        ++bad;
    // loop budget of the traversal kernels: 64-bit arithmetic, saturated -- positive and monotone up to the largest
    // scene the upload accepts (2^26 node slots, 2^30 triangles); the 32-bit product wrapped near 3.7 M triangles
    {
        int prev = 0;
        const long long tris[] = {0, 1, 2120, 1000000, 3700000, 4000000, 33554430, 1073741823};
        for (long long n : tris) {
            const long long slots = 2 * n + 3 > (1ll << 26) ? (1ll << 26) : 2 * n + 3;
            const int b = mrt::watch_budget((int)slots, (int)n);
            const long long exact = 64 * (4 * slots + n + 64);
            if (b <= 0 || b < prev || (exact <= 0x7fffffffll ? b != (int)exact : b != 0x7fffffff)) {
                printf("watch_budget(%lld, %lld) = %d\n", slots, n, b);
                ++bad;
            }
            prev = b;
        }
    }
    printf("tested=%ld mismatches=%ld\n", tested, bad);
    return bad == 0 ? 0 : 1;
}
