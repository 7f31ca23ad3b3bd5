// Micro-benchmark: how much independent work one wave needs to keep the fp32 VALU busy at low occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float a[CH];
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16 / CH; ++j)
#pragma unroll
            for (int i = 0; i < CH; ++i) a[i] = a[i] * s + 0.25f;
    }
    float r = 0;
    for (int i = 0; i < CH; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CH>
void run(int waves_per_simd) {
    float* d;
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;
    (void)hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, d, 100, 0.999f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * threads * iters * 16 * 2;
    printf("waves/SIMD %d  independent chains %d : %7.2f Tflop/s\n", waves_per_simd, CH, ops / (ms * 1e-3) / 1e12);
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 3, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
