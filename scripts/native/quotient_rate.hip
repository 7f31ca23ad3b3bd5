// Micro-benchmark: throughput of two exact forms of q = RN(a / d) for a divisor whose reciprocal is known, on gfx950.
//   MODE 0: fp32, one multiply and two fused residual corrections (5 operations; mrt::div_by_recip)
//   MODE 1: through fp64: (float)((double)a * R) with R = RN64(1 / (double)d)  (convert, multiply, convert)
//   MODE 2: plain fp32 multiply by the reciprocal (1 operation; not exact, the issue-rate yardstick)
//   MODE 3: IEEE fp32 division
// Also checks that forms 0, 1 and 3 agree bit for bit on the operands it runs through.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/native/quotient_rate.hip -o scripts/native/quotient_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float d0, unsigned* mismatches) {
    float a[8], acc[8];
    for (int i = 0; i < 8; ++i) { a[i] = (threadIdx.x + 1) * 0.37f + i * 1.3f + blockIdx.x * 1e-3f; acc[i] = 0; }
    const float d = d0 + threadIdx.x * 0.0137f;
    const float r = 1.0f / d;
    const double R = 1.0 / (double)d;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float q;
            if (MODE == 0) {
                const float q0 = a[i] * r;
                const float e1 = __builtin_fmaf(-d, q0, a[i]);
                const float q1 = __builtin_fmaf(e1, r, q0);
                const float e2 = __builtin_fmaf(-d, q1, a[i]);
                q = __builtin_fmaf(e2, r, q1);
            } else if (MODE == 1) {
                q = (float)((double)a[i] * R);
            } else if (MODE == 2) {
                q = a[i] * r;
            } else if (MODE == 3) {
                q = a[i] / d;
            } else {   // check
                const float q0 = a[i] * r;
                const float e1 = __builtin_fmaf(-d, q0, a[i]);
                const float q1 = __builtin_fmaf(e1, r, q0);
                const float e2 = __builtin_fmaf(-d, q1, a[i]);
                const float qa = __builtin_fmaf(e2, r, q1);
                const float qb = (float)((double)a[i] * R);
                q = a[i] / d;
                bad += (__float_as_uint(qa) != __float_as_uint(q)) + (__float_as_uint(qb) != __float_as_uint(q));
            }
            acc[i] += q;
            a[i] = a[i] * 1.0001f + 0.013f;       // next operand (2 more operations, the same in every mode)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (MODE == 4 && bad) atomicAdd(mismatches, bad);
}

template <int MODE>
void run(const char* name) {
    float* d;
    unsigned* mm;
    const int blocks = 256 * 8, threads = 256, iters = MODE == 4 ? 2000 : 10000;
    hipMalloc(&d, blocks * threads * 4);
    hipMalloc(&mm, 4);
    hipMemset(mm, 0, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 0.731f, mm);
    hipMemset(mm, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.731f, mm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned h = 0;
    hipMemcpy(&h, mm, 4, hipMemcpyDeviceToHost);
    const double q = (double)blocks * threads * iters * 8;
    if (MODE == 4) printf("%-44s mismatches %u of %.3g quotients\n", name, h, 2 * q);
    else printf("%-44s %8.3f ms  %7.2f Gquot/s\n", name, ms, q / (ms * 1e-3) / 1e9);
    hipFree(d); hipFree(mm);
}

int main() {
    run<2>("a * r (not exact; + 3 ops of loop)");
    run<0>("fp32 two-correction form (5 ops)");
    run<1>("through fp64 (cvt, mul_f64, cvt)");
    run<3>("IEEE fp32 division");
    run<4>("agreement of the three exact forms");
    return 0;
}
