// parallel_for for the host-side array loops of a scene upload (flattening, packing): splits [0, n) into
// contiguous chunks over up to 16 threads; small ranges run on the calling thread.
#pragma once
#include <stddef.h>
#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace mrt {

inline int host_threads() {
    int n = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("MRT_BUILD_THREADS")) n = std::max(1, std::min(64, atoi(e)));
    return n;
}

template <class F>
inline void parallel_for(size_t n, F&& body, size_t min_chunk = 32768) {     // body(begin, end)
    const size_t want = std::min<size_t>((size_t)host_threads(), (n + min_chunk - 1) / min_chunk);
    if (want <= 1) { body((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n + want - 1) / want;
    for (size_t t = 1; t < want; ++t) {
        const size_t b = t * chunk, e = std::min(n, b + chunk);
        if (b < e) th.emplace_back([&body, b, e] { body(b, e); });
    }
    body((size_t)0, std::min(n, chunk));
    for (std::thread& x : th) x.join();
}

}  // namespace mrt
