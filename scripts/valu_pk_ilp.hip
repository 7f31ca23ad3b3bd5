// Micro-benchmark: time per dependent fp32 VALU instruction, plain against packed (v_pk_mul_f32 / v_pk_add_f32),
// for 1/2/4 independent chains per wave at 1..8 waves per SIMD.  Tells whether packing two values into one
// instruction shortens a latency-bound instruction stream on gfx950.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o raytracing_amd/variants/valu_pk_ilp scripts/valu_pk_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int CH, bool PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
    float r = 0;
    if (PK) {
        v2f a[CH];
        const v2f ss = {s, s}, cc = {0.25f, 0.25f};
        for (int i = 0; i < CH; ++i) a[i] = v2f{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16 / CH; ++j)
#pragma unroll
                for (int i = 0; i < CH; ++i) { a[i] = a[i] * ss; a[i] = a[i] + cc; }
        }
        for (int i = 0; i < CH; ++i) r += a[i].x + a[i].y;
    } else {
        float a[CH];
        for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16 / CH; ++j)
#pragma unroll
                for (int i = 0; i < CH; ++i) { a[i] = a[i] * s; a[i] = a[i] + 0.25f; }
        }
        for (int i = 0; i < CH; ++i) r += a[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int CH, bool PK>
void run(int waves_per_simd) {
    float* d;
    const int blocks = 256 * waves_per_simd, threads = 256, iters = 20000;
    (void)hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CH, PK>), dim3(blocks), dim3(threads), 0, 0, d, 100, 0.999f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, PK>), dim3(blocks), dim3(threads), 0, 0, d, iters, 0.999f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 32;      // 16 mul + 16 add per iteration
    printf("%s chains %d waves/SIMD %d: %.2f ns per instruction per wave, %.1f clocks at 2.4 GHz; SIMD issues one every %.2f clocks\n",
           PK ? "packed" : "plain ", CH, waves_per_simd, ms * 1e6 / instr_per_wave, ms * 1e6 / instr_per_wave * 2.4,
           ms * 1e6 / instr_per_wave * 2.4 / waves_per_simd);
    (void)hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<1, false>(w); run<1, true>(w);
        run<2, false>(w); run<2, true>(w);
        run<4, false>(w); run<4, true>(w);
    }
    return 0;
}
