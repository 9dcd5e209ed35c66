// kernels.hpp -- declarations shared by kernels.hip (device code) and plan.hip (host C ABI).
#pragma once

#include <cstdlib>
#include <vector>

#include "v1c_core.hpp"

namespace v1c {

// A/B, coverage and diagnostic switches (environment variables, read once per process: V1C_UPB, V1C_DISABLE_*,
// V1C_XCD_STRIPS, V1C_*_CAP, V1C_DEBUG ...; tools/README.md) exist only in the -DV1C_TUNING build
// (libvr180remap_tuning.so, `make tuning`): the shipped library reads no environment.
#ifdef V1C_TUNING
inline const char* tuning_env(const char* name) { return std::getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif

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
// k_ray_lin_cn (grayscale / BGRA, bilinear): same plan-time boxes (launch_tile_boxes); `kb` = tile_cn_box_kb(); every source and its
// pitch dword-aligned, no unit overriding the rotation, one table entry per lane (shared_entry)
bool cn_kernel_supports(const Geom& g);
int tile_cn_box_kb(const void* host_boxes, const Geom& g);
hipError_t launch_ray_lin_cn(const KernelCtx& c, const UnitArgs& ua, int n_units, bool use_rot, const void* boxes, int kb, hipStream_t stream);
size_t tile_box_bytes(const Geom& g);
hipError_t launch_tile_boxes(const KernelCtx& c, void* boxes, bool shared_entry, hipStream_t stream, int mirror_h = 0);
// apply_lr pairs of unrotated chains: a tile and its mirror image about the equator from one set of coordinates
// (k_ray_lin3_pair_mirror); `host_mboxes` = boxes of the mirrored bands (launch_tile_boxes with mirror_h).
// tile_mirror_rest: the tiles (ty << 16 | tx) that launch leaves to the pair kernel; false: no mirror launch for this plan
// `raw_nwp` > 0: for k_ray_lin3_pair_mirror_raw (boxes by LDS-DMA, packed BGR in LDS, box buffers of raw_nwp KB:
// tile_mirror_raw_passes) instead of k_ray_lin3_pair_mirror
// `full_rows`: for k_ray_lin3_pair_mirror_raw proper (tile rows 0 .. TY / 2, boxes of up to two buffers: mirror_raw_fit), `n_eyes` = 2
// (apply_lr's pair) or 1 (a single image: its own list)
bool tile_mirror_rest(const void* host_boxes, const void* host_mboxes, const Geom& g, int half_dwords, int mirror_h,
                      std::vector<uint32_t>& rest, int raw_nwp, bool full_rows = false, int n_eyes = 2);
int tile_mirror_raw_passes(const void* host_boxes, const void* host_mboxes, const Geom& g, int permille = 980, int max_kb = 12);
// `pipe_tab` > 0: k_ray_lin3_pair_mirror_pipe (two tile rows per workgroup, the second pair's boxes requested while the first is
// sampled), tile_mirror_pipe_tab() table entries per pair in LDS
hipError_t launch_ray_lin3_pair_mirror(const KernelCtx& c, const UnitArgs& ua, const void* boxes, const void* mboxes, int half_dwords,
                                       int mirror_h, const uint32_t* rest_list, int n_rest, int raw_nwp, int pipe_tab, hipStream_t stream,
                                       int n_eyes = 2,   // n_eyes = 1 (raw_nwp > 0): a single image through the same workgroups
                                       int seq_kb = 0);  // > 0 (pairs): k_ray_lin3_pair_mirror_seq, two box buffers of seq_kb KB, the eyes one after the other
int tile_mirror_pipe_tab(const void* host_boxes, const void* host_mboxes, const Geom& g, int raw_nwp);
int tile_half_dwords(const void* host_boxes, size_t n_tiles);
// `coords_bounded` (launches without boxes): the host has bounded |32 x|, |32 y| < 2^21 for every pixel of every unit
// (radial_table_g_bound): k_ray_lin3_rot_pair_raw evaluates its speculative coordinates without the clamps of the cvRound trick
hipError_t launch_ray_lin3_tile(const KernelCtx& c, const UnitArgs& ua, int n_units, bool use_rot, const void* boxes, int half_dwords,
                                bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half,
                                int strip_len, int lean_raw_nwp, hipStream_t stream, bool coords_bounded = false);
int tile_xcd_strips(const void* host_boxes, const Geom& g, int half_dwords, int lean_half);
int tile_lean_half_dwords(int half_dwords);
// `raw_nwp` > 0: batches through k_ray_lin3_batch_lean_raw (boxes by LDS-DMA, buffers of raw_nwp KB: tile_lean_raw_passes)
std::vector<uint32_t> tile_rest_list(const void* host_boxes, const Geom& g, int half_dwords, int raw_nwp);
int tile_lean_raw_passes(const void* host_boxes, const Geom& g);

// merge=True of apply_lr (remapper.py:485-497)
hipError_t launch_anaglyph(const uint8_t* left, int64_t left_pitch, const uint8_t* right, int64_t right_pitch, int h, int w,
                           double* out, int64_t out_pitch, hipStream_t stream);

}  // namespace v1c
