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

// Everything a launch needs besides the units and its own scalars: launch-invariant.  The plan keeps a copy in device memory
// (v1c_plan::ctx_dev) that the tile kernels read through a constant-address-space reference -- scalar loads where a value is
// used; the generic kernels (kernels.hip) still take it by value.
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

// ---- argument block of the tile kernels (kernels_tile.hip, kernels_mirror.hip, kernels_cn.hip): their ONE by-value kernel argument ----
// Until round 3 every tile kernel took KernelCtx (330 bytes) and UnitArgs (16 units, 1.8 KB) by value.  Kernel-argument loads are
// invariant and dereferenceable, so the compiler hoisted them to the kernel entry and kept them live: 98 of the 102 scalar
// registers pinned, up to 28 of them spilled into vector lanes (two instantiations into scratch), and 16 units per launch at most.
// Now the launch-invariant part is plan-resident (`ctx`), the units are read through a pointer -- the launch's own copy in this block
// for up to kInlineUnits units (the runtime copies kernel arguments for free; no upload in front of a pair's 47 us launch) or a slot
// of the plan's device ring for longer batches (any number of units per launch) -- and the kernels read everything through
// constant-address-space references: scalar loads that stay where the value is used.
constexpr int kInlineUnits = 16;
struct TileBox;
struct TileArgs {
    const KernelCtx* ctx;        // plan-resident copy (device memory)
    const DevUnit* units;        // null: the launch's units are `inl` below; else n_units records in device memory
    const TileBox* boxes;        // plan-time tile boxes (null: units that override the rotation reduce theirs in the kernel)
    const TileBox* mboxes;       // mirror launches: (tile box, box of the band that mirrors it) pairs, 64 bytes per tile
                                 // (tuning build, register-staged k_ray_lin3_pair_mirror: the plain array of band boxes)
    const uint32_t* rest_list;   // tiles (ty << 16 | tx) a launch's fast path leaves to its general code / the tile list of a LIST launch
    uint32_t* tile_flags;        // one word per (unit slot, tile) for the fix-up pass behind this launch; null: proven unnecessary
    int n_units, upb;            // units of the launch, units per workgroup group
    int half_dwords, n_rest;     // LDS dwords per box buffer of the general pair / batch code; entries of rest_list
    unsigned tiles_x_magic, strip_len, strip_magic, rest_rows;  // xcd_tile(); rows of workgroups in front of the grid that serve rest_list
    int mirror_h, kb, tiles_x, same_rot;  // kb: box buffer KB of the LDS-DMA kernels (bytes for k_ray_lin3_rot_pair_raw); same_rot: launches
                                          // without plan-time boxes whose units all carry ONE rotation (v1c_plan_run_auto): pairs share coordinates
    // copies of what a workgroup needs for its FIRST vector loads (the row / column table entries of its tile): with the pointers here
    // those loads go out one scalar round trip earlier, next to the reads of the plan's context instead of behind them
    const double *col_s, *col_c, *col_h, *row_s, *row_c, *row_h;  // = ctx->ray.*
    int dst_w, dst_h, pad2[2];                                     // = ctx->g.*
    DevUnit inl[kInlineUnits];
};

// the units of one launch as the host launchers see them
struct LaunchUnits {
    const DevUnit* host;  // n records
    const DevUnit* dev;   // their device copy (ring slot), or null: n <= kInlineUnits, they travel in the kernel arguments
    int n;
};

int tiles_per_unit(const Geom& g);
hipError_t launch_remap(int mode, const KernelCtx& c, const UnitArgs& ua, int n_units, hipStream_t stream);
hipError_t launch_get_map(int mode, const KernelCtx& c, const UnitArgs& u, float* xmap, float* ymap, int64_t pitch,
                          hipStream_t stream);

// ---- tile kernels (kernels_tile.hip, kernels_mirror.hip, kernels_cn.hip; building blocks in tile_device.hpp).  Every launcher takes the plan's context twice -- `c`, the host copy its decisions read, and
// `cdev`, the device copy the kernels read --, the launch's units and `flags`: the plan's tile-flag words when a fix-up pass follows
// the launch, null when the host has proven it unnecessary (the kernels then write none). ----
// copy `n` unit records into device memory (a ring slot) with launches of their own: stream-ordered and graph-capturable
hipError_t launch_put_units(DevUnit* dst, const DevUnit* host, int n, hipStream_t stream);
// hot configuration (CN = 3, INTER_LINEAR, BORDER_CONSTANT, ray mode)
bool tile_kernel_supports(const Geom& g);
// k_ray_lin_cn (grayscale / BGRA, bilinear): same plan-time boxes (launch_tile_boxes); `kb` = tile_cn_box_kb(); every source and its
// pitch dword-aligned, no unit overriding the rotation, one table entry per lane (shared_entry)
bool cn_kernel_supports(const Geom& g);
int tile_cn_box_kb(const void* host_boxes, const Geom& g);
hipError_t launch_ray_lin_cn(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const void* boxes,
                             int kb, hipStream_t stream);
size_t tile_box_bytes(const Geom& g);
hipError_t launch_tile_boxes(const KernelCtx& c, const KernelCtx* cdev, void* boxes, bool shared_entry, hipStream_t stream, int mirror_h = 0);
// apply_lr pairs of unrotated chains: a tile and its mirror image about the equator from one set of coordinates;
// `host_mboxes` = boxes of the mirrored bands (launch_tile_boxes with mirror_h).
// tile_mirror_rest: the tiles (ty << 16 | tx) that launch leaves to the general pair code; false: no mirror launch for this plan.
// `raw_nwp` > 0: for the LDS-DMA kernels (packed BGR boxes in LDS, buffers of raw_nwp KB: tile_mirror_raw_passes), whose workgroups
// serve tile rows 0 .. TY / 2 (`full_rows`) and `n_eyes` = 2 (apply_lr's pair) or 1 (a single image: its own list) units;
// raw_nwp == 0 (tuning build): for the register-staged k_ray_lin3_pair_mirror
bool tile_mirror_rest(const void* host_boxes, const void* host_mboxes, const Geom& g, int half_dwords, int mirror_h,
                      std::vector<uint32_t>& rest, int raw_nwp, bool full_rows = false, int n_eyes = 2);
int tile_mirror_raw_passes(const void* host_boxes, const void* host_mboxes, const Geom& g, int permille = 980, int max_kb = 12);
// lu.n = 2: a pair -- k_ray_lin3_pair_mirror_seq (seq_kb > 0: two box buffers of seq_kb KB, the eyes one after the other); 1: a single
// image through the one-eye instantiation of k_ray_lin3_pair_mirror_raw (raw_nwp KB per box).  Tuning build: seq_kb == 0 selects the
// four-buffer / register-staged A/B partners
hipError_t launch_ray_lin3_pair_mirror(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, const void* boxes,
                                       const void* mboxes, int half_dwords, int mirror_h, const uint32_t* rest_list, int n_rest, int raw_nwp,
                                       hipStream_t stream, int seq_kb);
int tile_half_dwords(const void* host_boxes, size_t n_tiles, int max_chunks, bool occupancy_classes);
// `coords_bounded` (launches without boxes): the host has bounded |32 x|, |32 y| < 2^21 for every pixel of every unit
// (radial_table_g_bound): k_ray_lin3_rot_pair_raw evaluates its speculative coordinates without the clamps of the cvRound trick
hipError_t launch_ray_lin3_tile(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const void* boxes,
                                int half_dwords, bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half,
                                int strip_len, int lean_raw_nwp, hipStream_t stream, bool coords_bounded = false, int* kind = nullptr,
                                bool same_rot = false);
int tile_xcd_strips(const void* host_boxes, const Geom& g, int half_dwords, int lean_half);
int tile_lean_half_dwords(int half_dwords);
// `raw_nwp` > 0: batches through k_ray_lin3_batch_lean_raw (boxes by LDS-DMA, buffers of raw_nwp KB: tile_lean_raw_passes)
std::vector<uint32_t> tile_rest_list(const void* host_boxes, const Geom& g, int half_dwords, int raw_nwp);
int tile_lean_raw_passes(const void* host_boxes, const Geom& g);

// merge=True of apply_lr (remapper.py:485-497)
hipError_t launch_anaglyph(const uint8_t* left, int64_t left_pitch, const uint8_t* right, int64_t right_pitch, int h, int w,
                           double* out, int64_t out_pitch, hipStream_t stream);

// get_radius() (transformer.py:108-140) on the device: out[0] = radius, out[1] = 0 / 1 (no black border)
hipError_t launch_get_radius(const uint8_t* img, int h, int w, int64_t pitch, int cn, int threshold, double* out, hipStream_t stream);

// v1c_plan_run_auto: the Denormalize scale of the device context `ctx_dev` from n (radius, status) pairs in device memory
hipError_t launch_patch_radius(KernelCtx* ctx_dev, const double* rad_dev, int n, double r_limit, double cx32, double cy32, hipStream_t stream);
// get_radius of up to kInlineUnits images of one geometry + the patch above in one launch (v1c_plan_run_auto_images): `line[k]` = first
// pixel of image k's centre row (w > h) or centre column (transformer.py:126-129), `step[k]` bytes from pixel to pixel along it, `n`
// pixels.  `scratch_dev`: kAutoScratchInts plan-resident ints, [0, 16) = 0x7fffffff, [16, 32) = -1, [32] = 0 (auto_scratch_init) before the
// first launch; every launch leaves them so.
constexpr int kAutoBlocks = 64;
constexpr int kAutoScratchInts = 2 * kInlineUnits + 1;
struct AutoLines {
    const uint8_t* line[kInlineUnits];
    int64_t step[kInlineUnits];
    int n, cn, threshold, count;
};
inline void auto_scratch_init(int* host)
{
    for (int k = 0; k < kInlineUnits; k++)
        host[k] = 0x7fffffff, host[kInlineUnits + k] = -1;
    host[2 * kInlineUnits] = 0;
}
hipError_t launch_auto_radius(KernelCtx* ctx_dev, int* scratch_dev, const AutoLines& im, double r_limit, double cx32, double cy32, hipStream_t stream);

}  // namespace v1c
