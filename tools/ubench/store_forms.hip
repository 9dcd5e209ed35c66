// How fast can a CU write the remap kernels' output pattern, and does the store FORM matter?
// Every 256-thread workgroup writes one 64 x 16 pixel tile of a (H, W, 3) uint8 image (192 contiguous bytes per row,
// 16 rows a row pitch apart), tiles in XCD-friendly row-major order, non-temporal.  Forms:
//   x3   : global_store_dwordx3, 16 lanes per row (what the tile kernels do, tile_device.hpp: lane = 4 pixels = 12 bytes)
//   x4   : global_store_dwordx4, 12 lanes per row (the same bytes, 48 of 64 lanes active)
//   x4w  : global_store_dwordx4, all 64 lanes: a wave writes 5.33 rows (rows split at 16-byte granularity)
//   x1   : 3 x global_store_dword per lane (what an unmerged packing would give)
// Also `load` forms for a 64 x 16 box read (dwordx3 / dwordx4), to price the staging side.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_forms store_forms.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned __attribute__((ext_vector_type(3))) u3;
typedef unsigned __attribute__((ext_vector_type(4))) u4;

template <int FORM>
__global__ __launch_bounds__(256) void k_store(unsigned char* dst, int pitch, int tiles_x, unsigned seed)
{
    const int tid = threadIdx.x, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    unsigned char* tile = dst + (size_t)ty * 16 * pitch + (size_t)tx * 192;
    const unsigned v = seed + tid;
    if (FORM == 0) {  // x3: lane -> (row = tid / 16, 12 bytes at 12 * (tid % 16))
        u3* p = (u3*)(tile + (size_t)(tid >> 4) * pitch + (tid & 15) * 12);
        __builtin_nontemporal_store(u3{v, v + 1, v + 2}, p);
    } else if (FORM == 1) {  // x4, 12 lanes per row
        const int row = tid >> 4, l = tid & 15;
        if (l < 12)
            __builtin_nontemporal_store(u4{v, v + 1, v + 2, v + 3}, (u4*)(tile + (size_t)row * pitch + l * 16));
    } else if (FORM == 2) {  // x4w: 192 active lanes of 256 cover 16 rows x 12 slots, waves fully packed
        if (tid < 192) {
            const int row = tid / 12, l = tid % 12;
            __builtin_nontemporal_store(u4{v, v + 1, v + 2, v + 3}, (u4*)(tile + (size_t)row * pitch + l * 16));
        }
    } else {  // x1
        unsigned* p = (unsigned*)(tile + (size_t)(tid >> 4) * pitch + (tid & 15) * 12);
        __builtin_nontemporal_store(v, p), __builtin_nontemporal_store(v + 1, p + 1), __builtin_nontemporal_store(v + 2, p + 2);
    }
}

template <int FORM>
__global__ __launch_bounds__(256) void k_load(const unsigned char* src, int pitch, int tiles_x, unsigned* out)
{
    const int tid = threadIdx.x, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const unsigned char* tile = src + (size_t)ty * 16 * pitch + (size_t)tx * 192;
    unsigned acc = 0;
    if (FORM == 0) {
        const u3 v = *(const u3*)(tile + (size_t)(tid >> 4) * pitch + (tid & 15) * 12);
        acc = v.x ^ v.y ^ v.z;
    } else {
        const int row = tid >> 4, l = tid & 15;
        if (l < 12) {
            const u4 v = *(const u4*)(tile + (size_t)row * pitch + l * 16);
            acc = v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}

int main()
{
    const int W = 8192, H = 4096, pitch = W * 3, tiles_x = W / 64, ntiles = tiles_x * (H / 16);
    const size_t bytes = (size_t)H * pitch;
    unsigned char* buf[4];
    for (auto& b : buf) hipMalloc(&b, bytes), hipMemset(b, 1, bytes);
    unsigned* out; hipMalloc(&out, ntiles * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; i++) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 200; i++) launch(i);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 200;
        printf("%-34s %7.2f us per 100.7 MB image   %5.2f TB/s   %5.1f B/clk/CU at 2.4 GHz\n", name, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    };
    time("store x3 (16 lanes x 12 B per row)", [&](int i) { hipLaunchKernelGGL(k_store<0>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, (unsigned)i); });
    time("store x4 (12 lanes x 16 B per row)", [&](int i) { hipLaunchKernelGGL(k_store<1>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, (unsigned)i); });
    time("store x4, waves packed", [&](int i) { hipLaunchKernelGGL(k_store<2>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, (unsigned)i); });
    time("store 3 x dword", [&](int i) { hipLaunchKernelGGL(k_store<3>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, (unsigned)i); });
    time("load x3", [&](int i) { hipLaunchKernelGGL(k_load<0>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, out); });
    time("load x4 (12 lanes per row)", [&](int i) { hipLaunchKernelGGL(k_load<1>, dim3(ntiles), dim3(256), 0, 0, buf[i & 3], pitch, tiles_x, out); });
    return 0;
}
