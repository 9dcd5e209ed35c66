// VALU issue-cost microbenchmark: N waves per SIMD each run a long unrolled stream of one
// instruction kind with 4 independent chains; report cycles per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP 256
template <int KIND>
__global__ void k(unsigned* out, double* dout, const unsigned* in, int iters)
{
    unsigned a = in[threadIdx.x & 63], b = in[(threadIdx.x + 1) & 63], c0 = in[2], c1 = in[3], c2 = in[4], c3 = in[5];
    double d0 = a * 1e-3, d1 = b * 1e-3, d2 = c0 * 1e-3, d3 = c1 * 1e-3, m = 1.0000001, ad = 1e-9;
    float f0 = a, f1 = b, f2 = c0, f3 = c1;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (KIND == 0) { d0 = fma(d0, m, ad); d1 = fma(d1, m, ad); d2 = fma(d2, m, ad); d3 = fma(d3, m, ad); }
            if (KIND == 1) { f0 = fmaf(f0, 1.0001f, 1e-3f); f1 = fmaf(f1, 1.0001f, 1e-3f); f2 = fmaf(f2, 1.0001f, 1e-3f); f3 = fmaf(f3, 1.0001f, 1e-3f); }
            if (KIND == 2) { c0 = __builtin_amdgcn_perm(c0, a, 0x07020500u + r); c1 = __builtin_amdgcn_perm(c1, b, 0x01060304u + r); c2 = __builtin_amdgcn_perm(c2, a, 0x02030405u + r); c3 = __builtin_amdgcn_perm(c3, b, 0x00010607u + r); }
            if (KIND == 3) { c0 = __builtin_amdgcn_udot4(c0, a, c0, false); c1 = __builtin_amdgcn_udot4(c1, b, c1, false); c2 = __builtin_amdgcn_udot4(c2, a, c2, false); c3 = __builtin_amdgcn_udot4(c3, b, c3, false); }
            if (KIND == 4) { c0 = __umul24(c0, a) + b; c1 = __umul24(c1, b) + a; c2 = __umul24(c2, a) + b; c3 = __umul24(c3, b) + a; }
            if (KIND == 5) { c0 = c0 * a; c1 = c1 * b; c2 = c2 * a; c3 = c3 * b; }  // v_mul_lo_u32
            if (KIND == 6) { d0 = d0 * m; d1 = d1 * m; d2 = d2 * m; d3 = d3 * m; }
            if (KIND == 7) { f0 = (float)d0 + f0; d0 = d0 + (double)f0; }   // cvt f64<->f32 mix (4 instrs)
            if (KIND == 8) { c0 = (c0 >> 3) ^ a; c1 = (c1 >> 5) ^ b; c2 = (c2 >> 7) ^ a; c3 = (c3 >> 9) ^ b; }  // 2 int ops each
            if (KIND == 9) { d0 = __builtin_amdgcn_rsq(d0 + 1.0); d1 = __builtin_amdgcn_rsq(d1 + 1.0); }  // rsq + add
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + (unsigned)(t1 - t0);
    dout[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned)(t1 - t0);
}
int main()
{
    unsigned *out, *in; double* dout;
    hipMalloc(&out, 1 << 24); hipMalloc(&dout, 1 << 25); hipMalloc(&in, 4096);
    std::vector<unsigned> h(1024, 3); hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const char* names[] = {"v_fma_f64", "v_fma_f32", "v_perm_b32", "v_dot4_u32_u8", "v_mad_u32_u24", "v_mul_lo_u32", "v_mul_f64", "cvt mix(4)", "shift+xor(2)", "rsq_f64+add(2)"};
    const int iters = 64;
    for (int wps = 1; wps <= 8; wps *= 2) {  // waves per SIMD
        printf("waves/SIMD=%d:", wps);
        for (int kind = 0; kind < 10; kind++) {
            dim3 grid(256 * wps), block(256);  // 4 waves per block -> one per SIMD per block
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&]() {
                switch (kind) {
#define C(K) case K: hipLaunchKernelGGL(k<K>, grid, block, 0, 0, out, dout, in, iters); break;
                    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned cyc; hipMemcpy(&cyc, out, 4, hipMemcpyDeviceToHost);
            // instructions per wave: iters*REP (kind 7: REP/4*4.. treat as REP), per SIMD: wps waves
            double per_instr_wall = ms * 1e-3 * 2.4e9 / ((double)iters * REP * wps);
            printf("  %s %.2f(w0 %.2f)", names[kind], per_instr_wall, (double)cyc / (iters * REP));
        }
        printf("\n");
    }
    return 0;
}
