// Staging a BGR source box into LDS as BGRx (4 bytes per pixel), three ways:
//   regs : global_load_dwordx3 (4 px per lane) + 3 v_perm + ds_write_b128     -- what tile_device.hpp stage_load + stage_store do
//   dma3 : global_load_lds_dword with per-lane UNALIGNED byte addresses 3 * px  -- LDS-DMA does the expansion:
//          lane i of a wave-instruction fetches bytes [3 i, 3 i + 4) = B G R (B') into LDS dword i
//   dma4 : the same instruction with aligned addresses 4 * px (how much the misalignment costs)
// Each workgroup stages `tiles` boxes of ROWS x 64 px, reads a few dwords back (so nothing is dead) and
// the first workgroup dumps its last box for the host check.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int ROWS = 32, BW = 64;
typedef __attribute__((address_space(3))) void* lds_void;

template <int MODE>
__global__ __launch_bounds__(256) void k_stage(const unsigned char* __restrict__ src, int pitch, int h, int w, int tiles, unsigned* out, unsigned* dump, int hot)
{
    __shared__ __attribute__((aligned(16))) unsigned box[ROWS * BW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned acc = 0;
    for (int t = 0; t < tiles; t++) {
        const unsigned tile = hot ? blockIdx.x * tiles + (unsigned)(tiles - 1) : blockIdx.x * tiles + t;  // hot: the same box every time (L1 hits)
        const int x0 = (int)((tile * 64u) % (unsigned)(w - 2 * BW)) & ~3, y0 = (int)((tile * 7u) % (unsigned)(h - ROWS));
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < ROWS * BW / 4 / 256; q++) {
                const int ch = tid + q * 256, r = ch / (BW / 4), col = ch % (BW / 4);
                const unsigned* p = (const unsigned*)(src + (size_t)(y0 + r) * pitch + (x0 + 4 * col) * 3);
                const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
                uint4 o;
                o.x = w0 & 0xffffffu;
                o.y = __builtin_amdgcn_perm(w1, w0, 0x0c050403u);
                o.z = __builtin_amdgcn_perm(w2, w1, 0x0c040302u);
                o.w = w2 >> 8;
                *(uint4*)(box + r * BW + col * 4) = o;
            }
        } else {
            const int step = MODE == 1 ? 3 : 4;
#pragma unroll
            for (int q = 0; q < ROWS / 4; q++) {
                const int r = wave + 4 * q;  // wave-uniform row
                const unsigned char* g = src + (size_t)(y0 + r) * pitch + x0 * 3 + lane * step;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void)(box + r * BW), 4, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        acc += box[(tid * 5 + t) & (ROWS * BW - 1)] & 0xffffffu;
        if (dump && blockIdx.x == 0 && t == tiles - 1)
            for (int i = tid; i < ROWS * BW; i += 256)
                dump[i] = box[i];
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc;
}

int main()
{
    const int h = 4096, w = 4096, pitch = w * 3;
    std::vector<unsigned char> hs((size_t)h * pitch + 64);
    for (size_t i = 0; i < hs.size(); i++) hs[i] = (unsigned char)(i * 2654435761u >> 13);
    unsigned char* src; hipMalloc(&src, hs.size()); hipMemcpy(src, hs.data(), hs.size(), hipMemcpyHostToDevice);
    const int wgs = 256 * 8, tiles = 32;
    unsigned *out, *dump; hipMalloc(&out, wgs * 256 * 4); hipMalloc(&dump, ROWS * BW * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int step, int hot) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, pitch, h, w, tiles, out, dump, hot);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, pitch, h, w, tiles, out, (unsigned*)nullptr, hot);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        const double px = (double)wgs * tiles * ROWS * BW;
        std::vector<unsigned> hd(ROWS * BW);
        hipMemcpy(hd.data(), dump, hd.size() * 4, hipMemcpyDeviceToHost);
        // host check of workgroup 0's last box
        const unsigned tile = 0 * tiles + tiles - 1;
        const int x0 = (int)((tile * 64u) % (unsigned)(w - 2 * BW)) & ~3, y0 = (int)((tile * 7u) % (unsigned)(h - ROWS));
        long bad = 0;
        for (int r = 0; r < ROWS; r++)
            for (int i = 0; i < BW; i++) {
                const unsigned char* p = &hs[(size_t)(y0 + r) * pitch + x0 * 3 + i * step];
                const unsigned want = p[0] | p[1] << 8 | p[2] << 16;
                bad += (hd[r * BW + i] & 0xffffffu) != want;
            }
        printf("%-6s %8.3f ms  %7.1f Gpx/s staged  (%5.2f TB/s of source bytes)  mismatching dwords in the checked box: %ld\n", name, ms,
               px / ms / 1e6, px * 3 / ms / 1e9, bad);
    };
    for (int hot = 0; hot < 2; hot++) {
        printf(hot ? "-- the same box every iteration (L1 hits: instruction / LDS-write rate)\n" : "-- a new box every iteration (L2 / HBM)\n");
        run("regs", k_stage<0>, 3, hot);
        run("dma3", k_stage<1>, 3, hot);
        run("dma4", k_stage<2>, 4, hot);
    }
    return 0;
}
