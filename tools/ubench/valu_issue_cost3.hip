// VALU issue cost, exact instructions via inline asm (no compiler packing / fusing):
// one wave per SIMD (and 2, 4) runs REP x 8 independent copies of one instruction;
// prints cycles per wave-instruction per SIMD at the measured shader clock (s_memrealtime-free:
// wall clock x nominal 2.4 GHz, and clock64() of wave 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define ASM8(S)                                                                                                       \
    asm volatile(S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b))
template <int KIND>
__global__ void k(unsigned* out, const unsigned* in, int iters)
{
    unsigned a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63];
    unsigned r0 = in[2], r1 = in[3], r2 = in[4], r3 = in[5];
    double d0 = a, d1 = b;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            // each S uses a different destination among r0..r3 round-robin via the operand numbering
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\nv_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4\nv_add_u32 %0, %0, %5\nv_add_u32 %1, %1, %5\nv_add_u32 %2, %2, %5\nv_add_u32 %3, %3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %4, %5\nv_perm_b32 %1, %1, %4, %5\nv_perm_b32 %2, %2, %4, %5\nv_perm_b32 %3, %3, %4, %5\nv_perm_b32 %0, %0, %4, %5\nv_perm_b32 %1, %1, %4, %5\nv_perm_b32 %2, %2, %4, %5\nv_perm_b32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 3) asm volatile("v_dot4_u32_u8 %0, %0, %4, %5\nv_dot4_u32_u8 %1, %1, %4, %5\nv_dot4_u32_u8 %2, %2, %4, %5\nv_dot4_u32_u8 %3, %3, %4, %5\nv_dot4_u32_u8 %0, %0, %4, %5\nv_dot4_u32_u8 %1, %1, %4, %5\nv_dot4_u32_u8 %2, %2, %4, %5\nv_dot4_u32_u8 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_mad_u32_u24 %0, %0, %4, %5\nv_mad_u32_u24 %1, %1, %4, %5\nv_mad_u32_u24 %2, %2, %4, %5\nv_mad_u32_u24 %3, %3, %4, %5\nv_mad_u32_u24 %0, %0, %4, %5\nv_mad_u32_u24 %1, %1, %4, %5\nv_mad_u32_u24 %2, %2, %4, %5\nv_mad_u32_u24 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 5) asm volatile("v_cvt_f32_ubyte0 %0, %4\nv_cvt_f32_ubyte1 %1, %4\nv_cvt_f32_ubyte2 %2, %5\nv_cvt_f32_ubyte3 %3, %5\nv_cvt_f32_ubyte0 %0, %5\nv_cvt_f32_ubyte1 %1, %5\nv_cvt_f32_ubyte2 %2, %4\nv_cvt_f32_ubyte3 %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 6) asm volatile("v_cvt_pk_u8_f32 %0, %4, 0, %0\nv_cvt_pk_u8_f32 %1, %4, 1, %1\nv_cvt_pk_u8_f32 %2, %4, 2, %2\nv_cvt_pk_u8_f32 %3, %4, 3, %3\nv_cvt_pk_u8_f32 %0, %5, 1, %0\nv_cvt_pk_u8_f32 %1, %5, 2, %1\nv_cvt_pk_u8_f32 %2, %5, 0, %2\nv_cvt_pk_u8_f32 %3, %5, 1, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 7) asm volatile("v_dot2_u32_u16 %0, %0, %4, %5\nv_dot2_u32_u16 %1, %1, %4, %5\nv_dot2_u32_u16 %2, %2, %4, %5\nv_dot2_u32_u16 %3, %3, %4, %5\nv_dot2_u32_u16 %0, %0, %4, %5\nv_dot2_u32_u16 %1, %1, %4, %5\nv_dot2_u32_u16 %2, %2, %4, %5\nv_dot2_u32_u16 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 8) asm volatile("v_lshl_or_b32 %0, %0, 3, %5\nv_lshl_or_b32 %1, %1, 3, %5\nv_lshl_or_b32 %2, %2, 3, %5\nv_lshl_or_b32 %3, %3, 3, %5\nv_lshl_or_b32 %0, %0, 3, %5\nv_lshl_or_b32 %1, %1, 3, %5\nv_lshl_or_b32 %2, %2, 3, %5\nv_lshl_or_b32 %3, %3, 3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 9) asm volatile("v_and_b32 %0, %0, %4\nv_lshrrev_b32 %1, 3, %1\nv_and_b32 %2, %2, %4\nv_lshrrev_b32 %3, 5, %3\nv_or_b32 %0, %0, %5\nv_lshlrev_b32 %1, 3, %1\nv_or_b32 %2, %2, %5\nv_lshlrev_b32 %3, 5, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 10) asm volatile("v_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1\nv_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(d1));
            if (KIND == 11) asm volatile("v_mad_u16 %0, %0, %4, %5\nv_mad_u16 %1, %1, %4, %5\nv_mad_u16 %2, %2, %4, %5\nv_mad_u16 %3, %3, %4, %5\nv_pk_mad_u16 %0, %0, %4, %5\nv_pk_mad_u16 %1, %1, %4, %5\nv_pk_mad_u16 %2, %2, %4, %5\nv_pk_mad_u16 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 12) asm volatile("v_mul_u32_u24 %0, %0, %4\nv_mul_u32_u24 %1, %1, %4\nv_mul_u32_u24 %2, %2, %4\nv_mul_u32_u24 %3, %3, %4\nv_mul_u32_u24 %0, %0, %5\nv_mul_u32_u24 %1, %1, %5\nv_mul_u32_u24 %2, %2, %5\nv_mul_u32_u24 %3, %3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 13) asm volatile("v_bfe_u32 %0, %0, 3, 8\nv_bfe_u32 %1, %1, 3, 8\nv_bfe_u32 %2, %2, 3, 8\nv_bfe_u32 %3, %3, 3, 8\nv_bfe_u32 %0, %4, 3, 8\nv_bfe_u32 %1, %5, 3, 8\nv_bfe_u32 %2, %4, 3, 8\nv_bfe_u32 %3, %5, 3, 8" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 14) asm volatile("v_mul_f32 %0, %0, %4\nv_mul_f32 %1, %1, %4\nv_mul_f32 %2, %2, %4\nv_mul_f32 %3, %3, %4\nv_add_f32 %0, %0, %5\nv_add_f32 %1, %1, %5\nv_add_f32 %2, %2, %5\nv_add_f32 %3, %3, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            if (KIND == 15) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc\nv_cndmask_b32 %0, %0, %5, vcc\nv_cndmask_b32 %1, %1, %5, vcc\nv_cndmask_b32 %2, %2, %5, vcc\nv_cndmask_b32 %3, %3, %5, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + (unsigned)d0;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(t1 - t0);
}
template <int KIND>
__global__ void kpk(unsigned* out, const unsigned* in, int iters)
{
    typedef float __attribute__((ext_vector_type(2))) f2;
    f2 a = {(float)in[threadIdx.x & 63], 1.5f}, b = {0.001f, 0.002f};
    f2 r0 = {1, 2}, r1 = {3, 4}, r2 = {5, 6}, r3 = {7, 8};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\nv_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(r0.x + r1.y + r2.x + r3.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(t1 - t0);
}
int main()
{
    unsigned *out, *in;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096);
    std::vector<unsigned> h(1024, 3); hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[] = {"v_fma_f32", "v_add_u32", "v_perm_b32", "v_dot4_u32_u8", "v_mad_u32_u24", "v_cvt_f32_ubyteN", "v_cvt_pk_u8_f32", "v_dot2_u32_u16", "v_lshl_or_b32", "and/shift/or mix", "v_fma_f64", "v_mad_u16+v_pk_mad_u16", "v_mul_u32_u24", "v_bfe_u32", "v_mul_f32+v_add_f32", "v_cndmask_b32", "v_pk_fma_f32"};
    const int iters = 256;
    for (int wps = 1; wps <= 4; wps *= 2) {
        printf("waves/SIMD=%d\n", wps);
        for (int kind = 0; kind < 17; kind++) {
            dim3 grid(256 * wps), block(256);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, grid, block, 0, 0, out, in, iters); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
                    case 16: hipLaunchKernelGGL(kpk<0>, grid, block, 0, 0, out, in, iters); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
            printf("  %-24s wall*2.4GHz/instr/SIMD %.2f   clock64/instr (wave 0) %.2f\n", names[kind], ms * 1e-3 * 2.4e9 / ((double)iters * REP * wps), (double)cyc / (iters * REP));
        }
    }
    return 0;
}
