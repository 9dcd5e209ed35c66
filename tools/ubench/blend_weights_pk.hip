// Checks the packed form of the bilinear blend weights (v_pk_mad_u16 with clamp + v_pk_mul_lo_u16,
// tile_device.hpp::blend_weights) against the scalar form for all 32 x 32 fractions.
// hipcc --offload-arch=gfx950 -O3 -o blend_weights_pk blend_weights_pk.hip && ./blend_weights_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k(uint32_t* out)
{
    const uint32_t fq = threadIdx.x & 31, fr = threadIdx.x >> 5;
    // reference form
    const uint32_t wxp = (32u - fq) | (fq << 16);
    const uint32_t ra = wxp * (32u - fr), rb = wxp * fr;
    const uint32_t wa0 = (ra << 6) - ((ra >> 10) & 1u), wb0 = rb << 6;
    // packed form
    const uint32_t wx64 = fq * 0x3FFFC0u + 2048u;
    const uint32_t frc = 32u - fr;
    uint32_t wa, wb;
    asm("v_pk_mad_u16 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(wa) : "v"(wx64), "v"(frc));
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(wb) : "v"(wx64), "v"(fr));
    out[threadIdx.x * 4 + 0] = wa0, out[threadIdx.x * 4 + 1] = wa, out[threadIdx.x * 4 + 2] = wb0, out[threadIdx.x * 4 + 3] = wb;
}

int main()
{
    uint32_t* d;
    hipMalloc(&d, 1024 * 16);
    k<<<1, 1024>>>(d);
    static uint32_t h[4096];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++)
        if (h[4 * i] != h[4 * i + 1] || h[4 * i + 2] != h[4 * i + 3]) {
            if (bad++ < 8)
                printf("fq=%d fr=%d wa %08x %08x wb %08x %08x\n", i & 31, i >> 5, h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
        }
    printf("%d mismatches of 1024\n", bad);
    return bad != 0;
}
