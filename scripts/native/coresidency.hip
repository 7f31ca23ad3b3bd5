// Can a small kernel on a second stream (standing in for an RCCL send/recv kernel) start while
// persistent workgroups that take a CU's whole LDS are running back to back on the first stream?
//   hipcc -O3 --offload-arch=gfx950 -o raytracing_amd/variants/coresidency scripts/native/coresidency.hip
// usage: coresidency [big workgroups, default = number of CUs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void big(unsigned long long ticks, float* out) {
    extern __shared__ float lds[];
    asm volatile("" ::: "v127");          // take the whole register budget of a 1024-thread workgroup, as the render kernel does
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    float acc = lds[(threadIdx.x * 7) & 1023];
    while (wall_clock64() - t0 < ticks) acc = acc * 0.999f + 0.001f;       // bounded: the clock always advances
    if (acc == 12345.0f) out[0] = acc;
}
__global__ __launch_bounds__(256) void small(float* buf, int n) {
    __shared__ float s[2048];
    s[threadIdx.x] = 1.0f;
    __syncthreads();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) buf[i] += s[threadIdx.x];
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int wgs = argc > 1 ? atoi(argv[1]) : cus;
    const size_t lds = 140 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    float *out, *buf;
    const int n = 2 << 20;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    const unsigned long long ticks = 100000;          // 1 ms at the 100 MHz wall clock
    const int frames = 12;
    std::vector<hipEvent_t> ev(frames + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    hipEvent_t s0, s1;
    CK(hipEventCreate(&s0)); CK(hipEventCreate(&s1));
    hipLaunchKernelGGL(big, dim3(wgs), dim3(1024), lds, a, ticks / 10, out);     // warm-up
    hipLaunchKernelGGL(small, dim3(16), dim3(256), 0, b, buf, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev[0], a));
    for (int f = 0; f < frames; ++f) {
        hipLaunchKernelGGL(big, dim3(wgs), dim3(1024), lds, a, ticks, out);
        CK(hipEventRecord(ev[f + 1], a));
        if (f == 2) {        // the small kernel becomes ready when frame 2 has ended, as a send after a frame does
            CK(hipStreamWaitEvent(b, ev[3], 0));
            CK(hipEventRecord(s0, b));
            hipLaunchKernelGGL(small, dim3(16), dim3(256), 0, b, buf, n);
            CK(hipEventRecord(s1, b));
        }
    }
    CK(hipDeviceSynchronize());
    float t_end3, t_small, t_all;
    CK(hipEventElapsedTime(&t_end3, ev[0], ev[3]));
    CK(hipEventElapsedTime(&t_small, ev[0], s1));
    CK(hipEventElapsedTime(&t_all, ev[0], ev[frames]));
    printf("%d CUs, %d big workgroups: frame 2 ends at %.3f ms, the small kernel is done at %.3f ms (%.3f ms later), %d frames take %.3f ms\n",
           cus, wgs, t_end3, t_small, t_small - t_end3, frames, t_all);
    return 0;
}
