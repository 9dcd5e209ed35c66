// LDS bank behaviour of the bilinear tap gather for two lane -> pixel mappings of a wave (64 lanes x 4
// pixels = a 64 x 4 block of the output tile), with realistic NON-integer source strides:
//   A (kernel today): lane = (lx 0..15, ly 0..3), pixel k -> output (4 lx + k, ly)
//   B               : lane = (lx 0..31, ly 0..1), pixel k -> output (2 lx + (k & 1), 2 ly + (k >> 1))
// source pixel of output (x, y): ix = floor(X0 + a x + b y), iy = floor(Y0 + c x + d y), (a b; c d) =
// scale * rotation(angle); every lane reads ds_read2_b32 (iy * pitch + ix, +1) and the same one row down.
// Prints cycles per tap-row wave-instruction per CU (8 waves per SIMD resident; conflict-free: ds_read2_b32 4, ds_read2_b64 8).
// hipcc --offload-arch=gfx950 -O3 -o lds_tap_mapping lds_tap_mapping.hip && ./lds_tap_mapping
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ __launch_bounds__(256) void k(unsigned* out, int pitch, int mapping, float a, float b, float c, float d, int iters)
{
    __shared__ unsigned lds[4096 + 512];  // 18 KB: 8 workgroups = 8 waves per SIMD resident (a 64 KB array left 2 per SIMD and measured latency)
    for (int i = threadIdx.x; i < 4096 + 512; i += 256)
        lds[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int addr[4];
    for (int kk = 0; kk < 4; kk++) {
        int x, y;
        if (mapping == 5 || mapping == 6)      // F / G: 16-lane group = 4 lane columns x 4 rows (F: ds_read2_b32, G: pair cells ds_read2_b64)
            x = 4 * ((lane & 3) + 4 * (lane >> 4)) + kk, y = ((lane >> 2) & 3) + 4 * wave;
        else if (mapping == 7 || mapping == 8)  // H / I: pixel k of lane l = column 16 k + l: a row's lanes read ADJACENT source pixels (H b32, I b64)
            x = 16 * kk + (lane & 15), y = (lane >> 4) + 4 * wave;
        else if (mapping != 1)
            x = 4 * (lane & 15) + kk, y = (lane >> 4) + 4 * wave;
        else
            x = 2 * (lane & 31) + (kk & 1), y = 2 * (lane >> 5) + (kk >> 1) + 4 * wave;
        const int ix = (int)floorf(40.3f + a * x + b * y), iy = (int)floorf(30.6f + c * x + d * y);
        addr[kk] = (iy * pitch + ix) & 4095;
    }
    unsigned acc = 0;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int p = addr[kk] + (it & 3);
            if (mapping == 4) {  // E: as D, the two adjacent cells as ONE ds_read_b128 at an 8-byte-aligned address
                const int pp = p & 2047;
                typedef unsigned __attribute__((ext_vector_type(4))) u4;
                u4 u, v;
                const unsigned a0 = (unsigned)__builtin_amdgcn_readfirstlane(0) + (unsigned)(pp * 8), a1 = a0 + (unsigned)(pitch * 8);  // (lds[] is the only LDS object: offset 0)
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(u), "=&v"(v) : "v"(a0), "v"(a1) : "memory");
                acc += (u.x ^ u.z ^ v.x ^ v.z) + (u.y ^ u.w ^ v.y ^ v.w);
                continue;
            }
            if (mapping == 3 || mapping == 6 || mapping == 8) {  // D: both eyes interleaved per pixel (8 bytes), one ds_read2_b64 per tap row for both
                const unsigned long long* q = (const unsigned long long*)lds;
                const int pp = p & 2047;
                const unsigned long long u0 = q[pp], u1 = q[pp + 1], v0 = q[pp + pitch], v1 = q[pp + pitch + 1];
                acc += (unsigned)(u0 ^ u1 ^ v0 ^ v1) + (unsigned)((u0 ^ u1 ^ v0 ^ v1) >> 32);
                continue;
            }
            if (mapping == 2) {  // C: mapping A with 8-byte-aligned pairs read as ds_read_b64 (needs two copies of the box)
                const uint2 u = *(const uint2*)&lds[p & ~1], v = *(const uint2*)&lds[(p + pitch) & ~1];
                acc += u.x ^ u.y, acc += v.x ^ v.y;
                continue;
            }
            acc += lds[p] ^ lds[p + 1];
            acc += lds[p + pitch] ^ lds[p + pitch + 1];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc + (t1 == t0 ? 1u : 0u);
}

int main()
{
    unsigned* out;
    (void)hipMalloc(&out, 1 << 24);
    const int iters = 2000, wgs = 2048;
    for (float scale : {1.0f, 0.85f, 0.7f})
        for (float deg : {0.f, 10.f, 25.f, 45.f})
            for (int pitch : {76, 77, 81}) {
                printf("scale %.2f angle %2.0f pitch %d:", scale, deg, pitch);
                for (int mapping = 0; mapping < 9; mapping++) {
                    const float r = deg * 3.14159265f / 180, a = scale * cosf(r), b = -scale * sinf(r), c = scale * sinf(r), d = scale * cosf(r);
                    hipEvent_t e0, e1;
                    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
                    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, pitch, mapping, a, b, c, d, iters);
                    (void)hipDeviceSynchronize();
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, pitch, mapping, a, b, c, d, iters);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    const double instr_per_cu = (double)(wgs / 256) * 4 * iters * 8;
                    printf("  %c %.2f", "ABCDEFGHI"[mapping], ms * 1e-3 * 2.4e9 / instr_per_cu);
                    if (false) {
                        unsigned h[256];
                        (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                        unsigned cs = 0;
                        for (int q = 0; q < 256; q++) cs += h[q] * (q + 1);
                        printf(" [%08x]", cs);
                    }
                }
                printf("\n");
            }
    return 0;
}
