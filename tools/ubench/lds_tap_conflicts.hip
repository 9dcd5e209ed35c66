// LDS bank behaviour of the tile kernel's tap reads: a wave = 4 output rows x 16 lanes, every lane
// reads ds_read2_b32 (dwords a, a+1) with a = row * pitch + 4 * lx + koff.
//   koff = k            : all lanes read their pixel k at the same time (stride-4 pattern)
//   koff = (k + ly) & 3 : rows staggered by one pixel
// Reports LDS cycles per wave-instruction (one wave per SIMD, then 4).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* out, int pitch, int stagger, int iters, int rowsep)
{
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    unsigned acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int koff = stagger ? (kk + ly) & 3 : kk;
            const int a = (ly * rowsep) * pitch + 4 * lx + koff + (it & 7) * 4;
            acc += lds[a] ^ lds[a + 1];
            acc += lds[a + pitch] ^ lds[a + pitch + 1];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(t1 - t0);
}
int main()
{
    unsigned* out; hipMalloc(&out, 1 << 24);
    const int iters = 2000;
    for (int wgs : {256, 1024}) {
        printf("%d workgroups (%d wave(s) per SIMD)\n", wgs, wgs / 256);
        for (int pitch : {72, 74, 76, 80, 68, 66, 64}) {
            for (int stagger = 0; stagger < 2; stagger++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, pitch, stagger, iters, 1);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, pitch, stagger, iters, 1);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                // per CU: (wgs/256) workgroups x 4 waves x iters x 8 ds_read2 instructions
                const double instr_per_cu = (double)(wgs / 256) * 4 * iters * 8;
                printf("  pitch %3d stagger %d: %.2f LDS cycles per ds_read2_b32 wave-instruction (per CU, 2.4 GHz)\n", pitch, stagger,
                       ms * 1e-3 * 2.4e9 / instr_per_cu);
            }
        }
    }
    return 0;
}
