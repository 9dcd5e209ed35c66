#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 256
typedef unsigned short __attribute__((ext_vector_type(2))) us2;
template <int KIND>
__global__ void k(unsigned* out, const unsigned* in, int iters)
{
    unsigned a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63], c0 = in[2], c1 = in[3], c2 = in[4], c3 = in[5];
    float f0 = a, f1 = b, f2 = c0, f3 = c1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (KIND == 0) { asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(c0) : "v"(c0), "v"(a));
                             asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(c1) : "v"(c1), "v"(b));
                             asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(c2) : "v"(c2), "v"(a));
                             asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(c3) : "v"(c3), "v"(b)); }
            if (KIND == 1) { asm volatile("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(c0) : "v"(c0), "v"(a)); asm volatile("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(c1) : "v"(c1), "v"(b));
                             asm volatile("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(c2) : "v"(c2), "v"(a)); asm volatile("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(c3) : "v"(c3), "v"(b)); }
            if (KIND == 2) { asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f0) : "v"(c0)); asm volatile("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f1) : "v"(c1));
                             asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f2) : "v"(c2)); asm volatile("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f3) : "v"(c3)); c0 += (unsigned)f0; }
            if (KIND == 3) { asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(c0) : "v"(c0), "v"(a)); asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(c1) : "v"(c1), "v"(b));
                             asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(c2) : "v"(c2), "v"(a)); asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(c3) : "v"(c3), "v"(b)); }
            if (KIND == 4) { asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(c0) : "v"(c0), "v"(a), "v"(b)); asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(c1) : "v"(c1), "v"(b), "v"(a));
                             asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(c2) : "v"(c2), "v"(a), "v"(b)); asm volatile("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(c3) : "v"(c3), "v"(b), "v"(a)); }
            if (KIND == 5) { asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(c0) : "v"(c0)); asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(c1) : "v"(c1));
                             asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(c2) : "v"(c2)); asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(c3) : "v"(c3)); }
            if (KIND == 6) { asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(c0) : "v"(c0), "v"(a)); asm volatile("v_lshrrev_b32_e32 %0, 3, %1" : "=v"(c1) : "v"(c1));
                             asm volatile("v_and_b32_e32 %0, %1, %2" : "=v"(c2) : "v"(c2), "v"(a)); asm volatile("v_add_u32_e32 %0, %1, %2" : "=v"(c3) : "v"(c3), "v"(b)); }
            if (KIND == 7) { asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c0) : "v"(c0), "v"(a), "v"(b)); asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c1) : "v"(c1), "v"(b), "v"(a));
                             asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c2) : "v"(c2), "v"(a), "v"(b)); asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(c3) : "v"(c3), "v"(b), "v"(a)); }
            if (KIND == 8) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(f0), "v"(f1), "v"(f2)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(f1), "v"(f2), "v"(f3));
                             asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f2) : "v"(f2), "v"(f3), "v"(f0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(f3), "v"(f0), "v"(f1)); }
            if (KIND == 9) { asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f2)); asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f1) : "v"(f2), "v"(f3));
                             asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f2) : "v"(f3), "v"(f0)); asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f3) : "v"(f0), "v"(f1)); }
            if (KIND == 10) { asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_2" : "+v"(c0) : "v"(a), "v"(b));
                              asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0" : "+v"(c1) : "v"(b), "v"(a));
                              asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_1" : "+v"(c2) : "v"(a), "v"(b));
                              asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_3" : "+v"(c3) : "v"(b), "v"(a)); }
            if (KIND == 11) { asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f0) : "v"((double)1.5)); asm volatile("v_rndne_f32_e32 %0, %1" : "=v"(f1) : "v"(f1));
                              asm volatile("v_cvt_i32_f32_e32 %0, %1" : "=v"(c2) : "v"(f2)); asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(f3) : "v"(f3), "v"(f0), "v"(f1)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + (unsigned)(f0 + f1 + f2 + f3);
}
int main()
{
    unsigned *out, *in;
    hipMalloc(&out, 1 << 24); hipMalloc(&in, 4096);
    std::vector<unsigned> h(1024, 3); hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[] = {"mul_u24_sdwa", "mul_u24_e32", "cvt_f32_ubyte", "lshl_or", "pk_mad_u16", "bfe_u32", "add/shr/and", "mad_u32_u24", "fma_f32(3src)", "fmac_f32", "add_sdwa_dstbyte", "cvt/rndne/cvt/med3"};
    const int iters = 64, wps = 8;
    for (int kind = 0; kind < 12; kind++) {
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
            switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, grid, block, 0, 0, out, in, iters); break;
                C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11)
            }
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s %.2f cycles/instr/SIMD (at 2.4 GHz nominal)\n", names[kind], ms * 1e-3 * 2.4e9 / ((double)iters * REP * wps));
    }
    return 0;
}
