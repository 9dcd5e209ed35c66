// What each resource of an 8 x 8 (Lanczos4) pair sampler costs on its own, at C4's size (65536 tile visits of 4 waves x 4
// pixel slots; times in ms, box and weights as in lanczos_pair_forms.hip form B):
//   valu     : the 216 v_dot4 + epilogue of a pixel pair, operands from registers
//   w-lane   : the weight entry of a pixel (128 B) as 8 global_load_dwordx4 per LANE (every lane its own entry: 64 lines per
//              instruction) -- what the kernel does today
//   w-coop   : the same bytes fetched 8 lanes per entry (lane l reads row l % 8 of the entry of lane 8 j + l / 8: 8 lines
//              per instruction), not redistributed
//   w-coop-t : ... and redistributed through LDS (8 ds_write_b128 + 8 ds_read_b128 per lane, entry pitch 144 B)
//   lds-b64  : 48 dword-aligned ds_read_b64 per pixel pair (byte planes kept in 4 byte-shifted copies)
//   lds-b96  : 48 dword-aligned 12-byte reads (one copy; the window is cut out with 2 v_alignbyte per read)
//   lds-b96a : ... with those 96 v_alignbyte_b32
// hipcc --offload-arch=gfx950 -O3 -o lanczos_parts lanczos_parts.hip && ./lanczos_parts
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int BW = 96, BH = 32, PLANE = 128, ROWB = 6 * PLANE + 32;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2v;
typedef uint32_t __attribute__((ext_vector_type(3))) u32x3v;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4v;
typedef const __attribute__((address_space(1))) u32x4v* glb_u128_ptr;
typedef __attribute__((address_space(3))) uint8_t* lds_byte_ptr;

enum { VALU, W_LANE, W_COOP, W_COOP_T, LDS_B64, LDS_B96, LDS_B96A, NPART };

template <int PART>
__global__ __launch_bounds__(256) void k(uint32_t* out, const u32x4v* tab, float a, float b, float c, float d, int iters)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[BH * ROWB / 4 + 64];
    __shared__ __attribute__((aligned(16))) uint32_t wst[PART == W_COOP_T ? 4 * 64 * 36 : 4];  // 144 B per lane
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < BH * ROWB / 4 + 64; i += 256)
        lds[i] = (uint32_t)i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const float X0 = 6.0f + (float)((blockIdx.x * 37 + it * 11) & 31) * 0.031f, Y0 = 6.0f + (float)((blockIdx.x * 13 + it * 7) & 31) * 0.029f;
#pragma unroll 1
        for (int kk = 0; kk < 4; kk++) {
            const int x = 4 * lx + kk, y = ly;
            const int sx = (int)rintf(32.0f * (X0 + a * x + b * (y - 8))), sy = (int)rintf(32.0f * (Y0 + 8 + c * (x - 32) + d * y));
            int ix = (sx >> 5) - 3, iy = (sy >> 5) - 3;
            ix = min(max(ix, 0), BW - 12), iy = min(max(iy, 0), BH - 8);
            const uint32_t e = (uint32_t)((sy & 31) * 32 + (sx & 31));
            const glb_u128_ptr w = (glb_u128_ptr)tab + e * 8;
            if (PART == VALU) {
                int lo[6], hi[6];
#pragma unroll
                for (int p = 0; p < 6; p++)
                    lo[p] = 1 << 14, hi[p] = 0;
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const uint32_t w0 = e * 3 + r, w1 = e * 5 + r, w2 = e * 7 + r, w3 = e * 11 + r;
#pragma unroll
                    for (int p = 0; p < 6; p++) {
                        const uint32_t d0 = (uint32_t)ix * (r + 3) + p, d1 = (uint32_t)iy * (r + 5) + p;
                        lo[p] = __builtin_amdgcn_sdot4((int)d0, (int)w0, lo[p], false);
                        lo[p] = __builtin_amdgcn_sdot4((int)d1, (int)w1, lo[p], false);
                        hi[p] = __builtin_amdgcn_sdot4((int)d0, (int)w2, hi[p], false);
                        hi[p] = __builtin_amdgcn_sdot4((int)d1, (int)w3, hi[p], false);
                        if (r == 3 || r == 4) {
                            hi[p] = __builtin_amdgcn_sdot4((int)d0, 0x01000000, hi[p], false);
                            hi[p] = __builtin_amdgcn_sdot4((int)d1, 0x00000001, hi[p], false);
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < 6; p++)
                    acc += (uint32_t)min(max((lo[p] + (hi[p] << 8)) >> 15, 0), 255) << (8 * (p % 3));
            } else if (PART == W_LANE) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const u32x4v v = w[r];
                    acc += v.x ^ v.y ^ v.z ^ v.w;
                }
            } else if (PART == W_COOP || PART == W_COOP_T) {
                u32x4v v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t ej = (uint32_t)__builtin_amdgcn_ds_bpermute((8 * j + (lane >> 3)) * 4, (int)e);
                    v[j] = ((glb_u128_ptr)tab + ej * 8)[lane & 7];
                }
                if (PART == W_COOP) {
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
                } else {
                    // lane l holds row l % 8 of the entry of lane 8 j + l / 8: to that lane's slot, then everyone reads its own
                    u32x4v* ws = (u32x4v*)(wst + wave * 64 * 36);
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        ws[(8 * j + (lane >> 3)) * 9 + (lane & 7)] = v[j];
                    // (a wave is its own producer and consumer: no barrier, the LDS queue is in order)
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const u32x4v q = ws[lane * 9 + r];
                        acc += q.x ^ q.y ^ q.z ^ q.w;
                    }
                }
            } else {
                const lds_byte_ptr p0 = (lds_byte_ptr)lds + iy * ROWB + (ix & ~3);
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int p = 0; p < 6; p++) {
                        if (PART == LDS_B64) {
                            struct __attribute__((packed, aligned(4))) U64 {
                                u32x2v v;
                            };
                            const u32x2v q = ((const __attribute__((address_space(3))) U64*)(p0 + r * ROWB + p * PLANE))->v;
                            acc += q.x ^ q.y;
                        } else {
                            struct __attribute__((packed, aligned(4))) U96 {
                                u32x3v v;
                            };
                            const u32x3v q = ((const __attribute__((address_space(3))) U96*)(p0 + r * ROWB + p * PLANE))->v;
                            if (PART == LDS_B96)
                                acc += q.x ^ q.y ^ q.z;
                            else
                                acc += __builtin_amdgcn_alignbyte(q.y, q.x, (uint32_t)ix) ^ __builtin_amdgcn_alignbyte(q.z, q.y, (uint32_t)ix);
                        }
                    }
            }
        }
    }
    out[(size_t)blockIdx.x * 256 + tid] = acc;
}

template <int PART>
static float run(uint32_t* out, const u32x4v* tab, float a, float b, float c, float d)
{
    const int wgs = 4096, iters = 16;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PART>, dim3(wgs), dim3(256), 0, 0, out, tab, a, b, c, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<PART>, dim3(wgs), dim3(256), 0, 0, out, tab, a, b, c, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    std::vector<uint32_t> tab(1024 * 32);
    for (size_t i = 0; i < tab.size(); i++)
        tab[i] = (uint32_t)i * 2246822519u;
    uint32_t *dT, *out;
    (void)hipMalloc(&dT, tab.size() * 4), (void)hipMalloc(&out, (size_t)4096 * 256 * 4);
    (void)hipMemcpy(dT, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    for (float deg : {0.f, 20.f, 45.f})
        for (float scale : {1.0f, 0.8f}) {
            const float r = deg * 3.14159265f / 180, a = scale * cosf(r), b = -scale * sinf(r), c = scale * sinf(r), d = scale * cosf(r);
            const u32x4v* t = (const u32x4v*)dT;
            printf("angle %2.0f scale %.1f: valu %.3f  w-lane %.3f  w-coop %.3f  w-coop-t %.3f  lds-b64 %.3f  lds-b96 %.3f  lds-b96a %.3f\n", deg, scale,
                   run<VALU>(out, t, a, b, c, d), run<W_LANE>(out, t, a, b, c, d), run<W_COOP>(out, t, a, b, c, d), run<W_COOP_T>(out, t, a, b, c, d),
                   run<LDS_B64>(out, t, a, b, c, d), run<LDS_B96>(out, t, a, b, c, d), run<LDS_B96A>(out, t, a, b, c, d));
        }
    return 0;
}
