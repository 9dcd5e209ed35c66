// kernels.hpp -- declarations shared by kernels.hip (device code) and plan.hip (host C ABI).
#pragma once

#include <vector>

#include "v1c_core.hpp"

namespace v1c {

constexpr int kPX = 4;       // output pixels per lane (12 bytes = 3 dword stores for BGR)
constexpr int kBlockX = 64;  // lanes per row segment (= one wave)
constexpr int kBlockY = 4;   // rows (waves) per workgroup

enum { MODE_LITERAL = 0, MODE_RAY = 1, MODE_LUT = 2, MODE_FIXUP = 3 };

// Everything a launch needs besides the units; passed by value as a kernel argument.
struct KernelCtx {
    Geom g;
    const v1c_chain* chain;   // device copy of the lowered chain (literal / fix-up modes)
    RayParams ray;            // ray mode tables
    const short* itab;        // cubic / lanczos4 weight table (device), else null
    uint32_t* tile_flags;     // one word per (unit slot, tile): set by MODE_RAY, consumed by MODE_FIXUP
    const float* xmap;        // MODE_LUT
    const float* ymap;
    int64_t map_pitch;        // bytes
};

int tiles_per_unit(const Geom& g);
hipError_t launch_remap(int mode, const KernelCtx& c, const UnitArgs& ua, int n_units, hipStream_t stream);
hipError_t launch_get_map(int mode, const KernelCtx& c, const UnitArgs& u, float* xmap, float* ymap, int64_t pitch,
                          hipStream_t stream);

// hot configuration (CN = 3, INTER_LINEAR, BORDER_CONSTANT, ray mode): kernels_tile.hip
bool tile_kernel_supports(const Geom& g);
size_t tile_box_bytes(const Geom& g);
hipError_t launch_tile_boxes(const KernelCtx& c, void* boxes, bool shared_entry, hipStream_t stream);
int tile_half_dwords(const void* host_boxes, size_t n_tiles);
hipError_t launch_ray_lin3_tile(const KernelCtx& c, const UnitArgs& ua, int n_units, bool use_rot, const void* boxes, int half_dwords,
                                bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half,
                                int strip_len, hipStream_t stream);
int tile_xcd_strips(const void* host_boxes, const Geom& g, int half_dwords, int lean_half);
int tile_lean_half_dwords(int half_dwords);
std::vector<uint32_t> tile_rest_list(const void* host_boxes, const Geom& g, int half_dwords);

// merge=True of apply_lr (remapper.py:485-497)
hipError_t launch_anaglyph(const uint8_t* left, int64_t left_pitch, const uint8_t* right, int64_t right_pitch, int h, int w,
                           double* out, int64_t out_pitch, hipStream_t stream);

}  // namespace v1c
