// Exhaustive comparison over all 2^32 float bit patterns of IEEE 1/d (the compiler's correctly
// rounded division) with refinements of the hardware reciprocal estimate.  A tuning aid:
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o gpurun_out/recip_exhaustive scripts/native/recip_exhaustive.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>

__global__ void sweep(unsigned long long* out) {
    // out[0]: mismatches of one refinement step inside the guarded range, out[1]: of two steps,
    // out[2]: values inside the range, out[3..6]: examples
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t bits = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad1 = 0, bad2 = 0, inr = 0;
    for (uint64_t i = bits; i < (1ull << 32); i += stride) {
        const float d = __uint_as_float((uint32_t)i);
        const float a = fabsf(d);
        if (!(a >= 0x1p-100f && a <= 0x1p100f)) continue;
        ++inr;
        const float exact = 1.0f / d;
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float e0 = __builtin_fmaf(-d, r0, 1.0f);
        const float r1 = __builtin_fmaf(r0, e0, r0);
        const float e1 = __builtin_fmaf(-d, r1, 1.0f);
        const float r2 = __builtin_fmaf(r1, e1, r1);
        if (__float_as_uint(r1) != __float_as_uint(exact)) { if (!bad1) out[3] = i; ++bad1; }
        if (__float_as_uint(r2) != __float_as_uint(exact)) { if (!bad2) out[4] = i; ++bad2; }
    }
    atomicAdd(&out[0], bad1); atomicAdd(&out[1], bad2); atomicAdd(&out[2], inr);
}

int main() {
    unsigned long long* d; unsigned long long h[8] = {0};
    hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
    sweep<<<4096, 256>>>(d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("in range %llu  mismatches: one step %llu (e.g. 0x%08llx)  two steps %llu (e.g. 0x%08llx)\n", h[2], h[0], h[3], h[1], h[4]);
    return 0;
}
