// Building blocks of a "raw" pair kernel: the source box goes to LDS as it is in memory (packed BGR, 12 bytes per 4-pixel
// chunk, lane-linear) by LDS-DMA -- no staging registers, so coordinates can be computed while the box is in flight -- and the
// bilinear taps (6 bytes at byte offset 3 * px) are cut out of dword-aligned LDS reads.
//   A. staging rate: global_load_dwordx3 + 3 v_perm + ds_write_b128 (tile_device.hpp) vs global_load_lds_dwordx3 (12 B per
//      lane, 768 B per wave-instruction) vs global_load_lds_dwordx4
//   B. three dwords at a 4-byte-aligned LDS address: ds_read_b96 vs ds_read2_b32 + ds_read_b32; two dwords: ds_read_b64 at a
//      4- (not 8-) byte-aligned address vs ds_read2_b32
// hipcc --offload-arch=gfx950 -O3 -o dma_raw_forms dma_raw_forms.hip && ./dma_raw_forms
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
constexpr int ROWS = 24, CPR = 20;  // box: 24 rows x 20 chunks (80 px) = 480 chunks, 5760 B raw
typedef __attribute__((address_space(3))) void* lds_void;
typedef unsigned __attribute__((ext_vector_type(2))) u2;
typedef unsigned __attribute__((ext_vector_type(3))) u3;

template <int MODE>  // 0 regs -> BGRx cells, 1 dma x3 raw, 2 dma x4 raw (16 B per lane)
__global__ __launch_bounds__(256) void k_stage(const unsigned char* __restrict__ src, int pitch, int h, int w, int tiles, unsigned* out, unsigned* dump, int hot)
{
    __shared__ __attribute__((aligned(16))) unsigned box[ROWS * CPR * 4 + 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned acc = 0;
    for (int t = 0; t < tiles; t++) {
        const unsigned tile = hot ? blockIdx.x * tiles + (unsigned)(tiles - 1) : blockIdx.x * tiles + t;
        const int x0 = (int)((tile * 64u) % (unsigned)(w - 256)) & ~3, y0 = (int)((tile * 7u) % (unsigned)(h - ROWS));
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ch = tid + q * 256;
                if (ch < ROWS * CPR) {
                    const int r = ch / CPR, col = ch % CPR;
                    const unsigned* p = (const unsigned*)(src + (size_t)(y0 + r) * pitch + (x0 + 4 * col) * 3);
                    const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
                    uint4 o;
                    o.x = w0 & 0xffffffu, o.y = __builtin_amdgcn_perm(w1, w0, 0x0c050403u), o.z = __builtin_amdgcn_perm(w2, w1, 0x0c040302u), o.w = w2 >> 8;
                    *(uint4*)(box + ch * 4) = o;
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ch = tid + q * 256;
                if (ch < ROWS * CPR) {
                    const int r = ch / CPR, col = ch % CPR;
                    const unsigned char* g = src + (size_t)(y0 + r) * pitch + (x0 + 4 * col) * 3;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void)(box + (q * 256 + wave * 64) * 3), 12, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // 16 B per lane: the raw box as ROWS x (CPR * 12 = 240 B = 15 lanes) -- rows are padded to 16 lanes
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int u = tid + q * 256;  // 16-byte unit
                if (u < ROWS * 16) {
                    const int r = u >> 4, col = u & 15;
                    const unsigned char* g = src + (size_t)(y0 + r) * pitch + x0 * 3 + col * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void)(box + (q * 256 + wave * 64) * 4), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        acc += box[(tid * 5 + t) % (ROWS * CPR * 3)];
        if (dump && blockIdx.x == 0 && t == tiles - 1)
            for (int i = tid; i < ROWS * CPR * 4; i += 256)
                dump[i] = box[i];
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc;
}

// B: LDS read forms; every lane reads at dword index base + 3 * lane-ish (the raw gather's addresses), xor of the data
template <int FORM>
__global__ __launch_bounds__(256) void k_read(unsigned* out, int iters, int stride_b, int misalign)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[4096 + 64];
    for (int i = threadIdx.x; i < 4096 + 64; i += 256)
        lds[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds + (((unsigned)(lane * stride_b) & ~3u) + (unsigned)misalign * 4u);
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned a = a0 + (unsigned)((it & 63) * 240);
        if (FORM == 0) {
            u3 v;
            asm volatile("ds_read_b96 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x ^ v.y ^ v.z;
        } else if (FORM == 1) {
            u2 v;
            unsigned z;
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read_b32 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(z) : "v"(a) : "memory");
            acc += v.x ^ v.y ^ z;
        } else if (FORM == 2) {
            u2 v;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x ^ v.y;
        } else {
            u2 v;
            asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x ^ v.y;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    const int h = 4096, w = 4096, pitch = w * 3;
    std::vector<unsigned char> hs((size_t)h * pitch + 64);
    for (size_t i = 0; i < hs.size(); i++)
        hs[i] = (unsigned char)(i * 2654435761u >> 13);
    unsigned char* src;
    (void)hipMalloc(&src, hs.size());
    (void)hipMemcpy(src, hs.data(), hs.size(), hipMemcpyHostToDevice);
    const int wgs = 256 * 8, tiles = 32;
    unsigned *out, *dump;
    (void)hipMalloc(&out, wgs * 256 * 4);
    (void)hipMalloc(&dump, ROWS * CPR * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int mode, int hot) {
        for (int i = 0; i < 3; i++)
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, pitch, h, w, tiles, out, dump, hot);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; i++)
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, src, pitch, h, w, tiles, out, (unsigned*)nullptr, hot);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= 10;
        std::vector<unsigned> hd(ROWS * CPR * 4);
        (void)hipMemcpy(hd.data(), dump, hd.size() * 4, hipMemcpyDeviceToHost);
        const unsigned tile = tiles - 1;
        const int x0 = (int)((tile * 64u) % (unsigned)(w - 256)) & ~3, y0 = (int)((tile * 7u) % (unsigned)(h - ROWS));
        long bad = 0;
        const unsigned char* lb = (const unsigned char*)hd.data();
        for (int r = 0; r < ROWS; r++)
            for (int i = 0; i < CPR * 4; i++) {
                const unsigned char* p = &hs[(size_t)(y0 + r) * pitch + (x0 + i) * 3];
                for (int c = 0; c < 3; c++) {
                    const unsigned char got = mode == 0 ? lb[(r * CPR * 4 + i) * 4 + c] : mode == 1 ? lb[(r * CPR * 4 + i) * 3 + c] : lb[r * 256 + i * 3 + c];
                    bad += got != p[c];
                }
            }
        printf("%-6s %8.3f ms  %6.2f TB/s of source bytes  mismatching bytes in the checked box: %ld\n", name, ms,
               (double)wgs * tiles * ROWS * CPR * 12 / ms / 1e9, bad);
    };
    for (int hot = 0; hot < 2; hot++) {
        printf(hot ? "-- the same box every iteration (L1 hits)\n" : "-- a new box every iteration (L2 / HBM)\n");
        run("regs", k_stage<0>, 0, hot);
        run("dma12", k_stage<1>, 1, hot);
        run("dma16", k_stage<2>, 2, hot);
    }
    auto readrate = [&](const char* name, auto kern, int stride_b, int mis) {
        const int iters = 4000, n = 2048;
        hipLaunchKernelGGL(kern, dim3(n), dim3(256), 0, 0, out, iters, stride_b, mis);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(n), dim3(256), 0, 0, out, iters, stride_b, mis);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %s %.2f", name, ms * 1e-3 * 2.4e9 / ((double)(n / 256) * 4 * iters));
    };
    {  // misaligned forms must read the same data as the dword forms
        std::vector<unsigned> ha(256), hb(256);
        long bad = 0;
        for (int mis = 0; mis < 4; mis++) {
            hipLaunchKernelGGL(k_read<0>, dim3(1), dim3(256), 0, 0, out, 100, 9, mis);
            (void)hipMemcpy(ha.data(), out, 1024, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_read<1>, dim3(1), dim3(256), 0, 0, out, 100, 9, mis);
            (void)hipMemcpy(hb.data(), out, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; i++) bad += ha[i] != hb[i];
            hipLaunchKernelGGL(k_read<2>, dim3(1), dim3(256), 0, 0, out, 100, 9, mis);
            (void)hipMemcpy(ha.data(), out, 1024, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_read<3>, dim3(1), dim3(256), 0, 0, out, 100, 9, mis);
            (void)hipMemcpy(hb.data(), out, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 256; i++) bad += ha[i] != hb[i];
        }
        printf("ds_read_b96 / ds_read_b64 at 4-byte-aligned addresses vs dword reads: %ld mismatches\n", bad);
    }
    for (int stride_b : {12, 9, 6})
        for (int mis : {0, 1, 2, 3}) {
            printf("lane stride %2d B, address = 4 x %d (mod 16): cycles per wave-read per CU:", stride_b, mis);
            readrate("b96", k_read<0>, stride_b, mis);
            readrate("read2_b32+b32", k_read<1>, stride_b, mis);
            readrate("b64", k_read<2>, stride_b, mis);
            readrate("read2_b32", k_read<3>, stride_b, mis);
            printf("\n");
        }
    return 0;
}
