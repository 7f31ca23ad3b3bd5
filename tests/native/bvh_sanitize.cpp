// Host BVH builder under AddressSanitizer / UBSan on random, tie-heavy, flat and degenerate inputs, with a
// structural check (every triangle in exactly one reachable leaf).  Built and run by tests/test_host_logic.py.
#include "bvh_builder.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
using namespace mrt;
int main(int argc, char** argv) {
    const bool big = argc > 1;      // one scene large enough for the builder to use its thread pool (run under TSan)
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (int trial = 0; trial < (big ? 2 : 40); ++trial) {
        int n = big ? 90000 : (trial < 5 ? trial : (int)(rnd() * 3000) + 1);
        std::vector<float> pos(9 * n), uv(6 * n, 0.5f), N(3 * n, 0.0f);
        std::vector<int32_t> tri(3 * n);
        for (int i = 0; i < n; ++i) {
            float c[3] = {rnd() * 40 - 20, rnd() * 4 - 2, -rnd() * 40 - 3};
            int mode = trial % 4;
            for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) {
                float v = c[a] + (rnd() - 0.5f);
                if (mode == 1) v = floorf(v);                 // many ties
                if (mode == 2 && a == 1) v = 0.0f;            // flat
                if (mode == 3) v = c[a];                      // degenerate triangles
                pos[9 * i + 3 * k + a] = v;
            }
            tri[3 * i] = 3 * i; tri[3 * i + 1] = 3 * i + 1; tri[3 * i + 2] = 3 * i + 2;
            N[3 * i + 2] = 1.0f;
        }
        mrt_mesh m{pos.data(), uv.data(), N.data(), tri.data(), 3 * n, n, 0, 0.f, 0.f, 1.f};
        HostBVH b; std::string err;
        if (!flatten_meshes(&m, 1, 1, false, b.tris, err)) { printf("flatten: %s\n", err.c_str()); return 1; }
        build_bvh(b);
        // structural check: every triangle appears exactly once in the leaves reachable from the root
        std::vector<int> seen(n, 0); std::vector<uint32_t> st{0}; size_t leaves = 0;
        while (!st.empty() && n > 0) { uint32_t k = st.back(); st.pop_back();
            if (b.count[k] == 0) { st.push_back(b.first[k]); st.push_back(b.first[k] + 1); }
            else { ++leaves; for (uint32_t j = 0; j < b.count[k]; ++j) seen[b.idx[b.first[k] + j]]++; } }
        for (int i = 0; i < n; ++i) if (seen[i] != 1) { printf("trial %d: triangle %d seen %d times\n", trial, i, seen[i]); return 1; }
        printf("trial %d n=%d pool=%u depth=%d leaves=%zu\n", trial, n, b.pool_ptr, b.max_depth, leaves);
    }
    return 0;
}
