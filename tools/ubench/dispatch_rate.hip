// Workgroup dispatch cost: empty-ish kernels with the tile kernel's launch shape
// (16384 x 256 threads, ~20 KB LDS, ~100 VGPRs, 2 KB of kernel arguments).
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { double d[240]; };  // 1920 B of kernarg
template <int VG>
__global__ __launch_bounds__(256) void k_empty(unsigned* out, Big big, int n)
{
    extern __shared__ unsigned lds[];
    unsigned v[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) v[i] = threadIdx.x * (i + 1);
    if (n == 12345) {  // never true: keeps the registers and LDS alive
#pragma unroll
        for (int i = 0; i < VG; i++) lds[(threadIdx.x + i) & 1023] += v[i];
        __syncthreads();
        out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + (unsigned)big.d[threadIdx.x % 240];
    }
}
__global__ __launch_bounds__(256) void k_small(unsigned* out, int n)
{
    if (n == 12345) out[blockIdx.x * 256 + threadIdx.x] = 1;
}
int main()
{
    unsigned* out; hipMalloc(&out, 1 << 26);
    Big big{}; 
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 50; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-48s %.2f us per launch\n", name, ms * 1000 / 50);
    };
    for (int wgs : {4096, 16384, 65536}) {
        printf("grid = %d workgroups x 256 threads\n", wgs);
        time("  minimal kernel, no LDS", [&] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, 0, out, 1); });
        time("  ~100 VGPRs, 0 LDS, 2 KB kernarg", [&] { hipLaunchKernelGGL(k_empty<96>, dim3(wgs), dim3(256), 0, 0, out, big, 1); });
        time("  ~100 VGPRs, 20 KB LDS, 2 KB kernarg", [&] { hipLaunchKernelGGL(k_empty<96>, dim3(wgs), dim3(256), 20480, 0, out, big, 1); });
        time("  ~32 VGPRs, 20 KB LDS, 2 KB kernarg", [&] { hipLaunchKernelGGL(k_empty<24>, dim3(wgs), dim3(256), 20480, 0, out, big, 1); });
        time("  ~100 VGPRs, 40 KB LDS, 2 KB kernarg", [&] { hipLaunchKernelGGL(k_empty<96>, dim3(wgs), dim3(256), 40960, 0, out, big, 1); });
    }
    return 0;
}
