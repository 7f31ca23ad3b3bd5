// Micro-benchmark: issue rate of unpacked vs packed fp32 VALU operations on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/valu_rate.hip -o scripts/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const v2f ss = {s, s};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] * s + 0.25f;                         // mul + add (unfused)
            if (MODE == 1) a[i] = __builtin_fmaf(a[i], s, 0.25f);           // fma
            if (MODE == 2) p[i] = p[i] * ss + v2f{0.25f, 0.25f};            // packed mul + packed add
            if (MODE == 3) p[i] = __builtin_elementwise_fma(p[i], ss, v2f{0.25f, 0.25f});
            if (MODE == 4) a[i] = fminf(a[i] * s, 3.0f);                    // mul + min
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, int flops_per_op) {
    float* d;
    const int blocks = 256 * 8, threads = 256, iters = 20000;
    hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 0.999f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * threads * iters * 8 * flops_per_op;
    printf("%-28s %8.3f ms  %7.2f Tflop/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    run<0>("mul+add (2 instr)", 2);
    run<1>("fma", 2);
    run<2>("pk_mul+pk_add (2 instr)", 4);
    run<3>("pk_fma", 4);
    run<4>("mul+min (2 instr)", 2);
    return 0;
}
