// When do single-wave "helper" workgroups of a second kernel get onto CUs that already hold a 1024-thread
// workgroup with 140 KB of LDS?  Varies the big kernel's VGPR count (128 / 96), scratch use of either kernel
// and the launch order.  A tuning aid (measured on gfx950: never when neither kernel uses scratch memory; at
// 96 VGPRs and with scratch in at least one of them the helpers run beside the big workgroups, in either order).
//   hipcc -O3 --offload-arch=gfx950 -o raytracing_amd/variants/coresidency_helpers scripts/native/coresidency_helpers.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int VG, bool SCR>
__global__ __launch_bounds__(1024) void big(unsigned long long ticks, float* out, int idx) {
    extern __shared__ float lds[];
    if (VG == 96) asm volatile("" ::: "v95"); else asm volatile("" ::: "v127");
    float loc[64];
    if (SCR) { for (int i = 0; i < 64; ++i) loc[i] = i * 0.5f + idx; }
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    float acc = lds[(threadIdx.x * 7) & 1023];
    while (wall_clock64() - t0 < ticks) { acc = acc * 0.999f + 0.001f; if (SCR) acc += loc[(int)acc & 63]; }
    if (acc == 12345.0f) out[0] = acc;
}
template <bool SCR>
__global__ __launch_bounds__(64) void small(float* buf, int n, unsigned long long ticks, int idx) {
    __shared__ float s[1024];
    asm volatile("" ::: "v119");
    float loc[64];
    if (SCR) { for (int i = 0; i < 64; ++i) loc[i] = i * 0.25f + idx; }
    s[threadIdx.x] = 1.0f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    float acc = s[threadIdx.x];
    while (wall_clock64() - t0 < ticks) { acc = acc * 0.999f + 0.001f; if (SCR) acc += loc[(int)acc & 63]; }
    buf[(blockIdx.x * 64 + threadIdx.x) % n] = acc;
}
#include <unistd.h>
template <int VG, bool SCR, bool SSCR>
int run(int cus, const char* name, bool small_first) {
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&big<VG, SCR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    float *out, *buf;
    const int n = 1 << 20;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&buf, n * 4));
    hipEvent_t e0, ea, eb;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    hipLaunchKernelGGL((big<VG, SCR>), dim3(cus), dim3(1024), lds, a, 1000, out, 0);
    hipLaunchKernelGGL((small<SSCR>), dim3(16), dim3(64), 0, b, buf, n, 1000, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); CK(hipStreamSynchronize(0));
    CK(hipEventRecord(e0, a));
    CK(hipStreamSynchronize(a));
    if (small_first) {
        hipLaunchKernelGGL((small<SSCR>), dim3(cus * 4), dim3(64), 0, b, buf, n, 20000, 0);
        CK(hipEventRecord(eb, b));
        usleep(50);
        hipLaunchKernelGGL((big<VG, SCR>), dim3(cus), dim3(1024), lds, a, 100000, out, 0);
        CK(hipEventRecord(ea, a));
    } else {
        hipLaunchKernelGGL((big<VG, SCR>), dim3(cus), dim3(1024), lds, a, 100000, out, 0);
        CK(hipEventRecord(ea, a));
        usleep(50);
        hipLaunchKernelGGL((small<SSCR>), dim3(cus * 4), dim3(64), 0, b, buf, n, 20000, 0);
        CK(hipEventRecord(eb, b));
    }
    CK(hipDeviceSynchronize());
    float ta, tb;
    CK(hipEventElapsedTime(&ta, e0, ea)); CK(hipEventElapsedTime(&tb, e0, eb));
    printf("%-44s big (1 ms) done at %.3f ms, helpers (0.2 ms) done at %.3f ms\n", name, ta, tb);
    return 0;
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    run<128, false, false>(cus, "big 128 no scr, small no scr, big first", false);
    run<96, false, false>(cus, "big 96 no scr, small no scr, big first", false);
    run<96, false, false>(cus, "big 96 no scr, small no scr, small first", true);
    run<96, true, false>(cus, "big 96 scr, small no scr, big first", false);
    run<96, false, true>(cus, "big 96 no scr, small scr, big first", false);
    run<96, true, true>(cus, "big 96 scr, small scr, big first", false);
    run<96, true, true>(cus, "big 96 scr, small scr, small first", true);
    return 0;
}
