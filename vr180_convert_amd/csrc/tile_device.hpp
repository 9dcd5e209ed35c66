// tile_device.hpp -- device building blocks of the LDS-tiled kernels and the small host helpers their launchers share.
// Included by kernels_tile.hip (plan-time boxes, the general tile kernel, batches, per-unit rotations), kernels_mirror.hip (pairs and
// single images of unrotated chains: a tile and its mirror image per workgroup) and kernels_cn.hip (grayscale / BGRA; the unit-ring
// upload).  Round 4 split the former 3 600-line kernels_tile.hip along those lines; nothing here is a kernel.
//
// Why tiles: rocprof counters on a gather-from-global version show the vector L1 (TCP) as the
// limit -- it retires about one 64-byte access per clock, a wave's 64 unaligned 8-byte gathers
// cost ~100 accesses per load instruction (no coalescing) and a dwordx4 table read 16.  Here
//   * a workgroup owns a 64 x 16 output tile (lane = 4 px of one row, wave = 64 px x 4 rows);
//     tiles are dealt to the XCDs in contiguous runs so that shared halo rows hit one L2;
//   * the bounding box of the tile's source taps is copied from HBM to LDS with 12-byte,
//     row-contiguous, dword-aligned loads (one wave load = 768 contiguous bytes = 256 pixels for
//     ~13 L1 accesses) and kept 4 bytes per pixel in LDS (BGRx: three v_perm_b32 expand 4 pixels
//     into one ds_write_b128), so a 2x2 cell is two ALIGNED ds_read2_b32 (unaligned 8-byte LDS
//     reads of packed BGR measured ~60 stall cycles each);
//   * the box of every tile, the slice of the radial table it uses and whether all of its pixels
//     are valid and inside the source ("interior") are computed ONCE per plan by k_tile_boxes (the
//     map does not depend on the pixels), so the hot kernel starts its staging loads before any
//     coordinate math and has no reduction; units that override the rotation (per-frame
//     calibration) use the BOXES = 0 variant, which reduces the box in-kernel with DPP mins;
//   * the coordinates of a tile are computed once and shared by every unit of the launch that
//     uses the same map (the reference computes ONE map per apply() call): the two eyes of a pair
//     (PAIR = 1: straight-line code, 6 waves per SIMD) or up to 8 frames of a batch (loop with a
//     register prefetch of the unit after next);
//   * each lane reads ONE 64-byte radial-table entry for its 4 pixels; the plan proves from the
//     validity levels of the table (radial_fit.hpp) that this is enough (OWN = 0) or keeps the
//     per-pixel fallback compiled in (OWN = 1).
// Tiles whose box does not fit the LDS budget (strong rotation / minification) gather from global
// memory; pixels with taps outside the source go through the generic border-aware sampler;
// pixels outside the radial table's domain are left to the fix-up launch (kernels.hip MODE_FIXUP).
// Compile-time switches for A/B measurements: V1C_TILE_W (64), V1C_UPB (8), V1C_XCD_SWIZZLE (1),
// V1C_STAMPS (per-phase cycle counters); run-time: V1C_UPB=<n>.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

// V1C_PRIO (bits): s_setprio 3 from a workgroup's first instruction until its box / table loads are requested (back to 0 behind them) --
// 1: bicubic / Lanczos4 tile kernels, 2: bilinear tile kernels, 4: mirror kernels, 8: batch kernel.  A workgroup that has just started
// is the youngest on its SIMD: without the priority its few load instructions queue behind the older workgroups' tap arithmetic and
// the memory latency starts late.  Round 5, one box, interleaved (profiles/r05g_prio/): C4 1.256 -> 1.204 ms, C1L 0.0912 -> 0.0880, C2R
// 0.0677 -> 0.0653, P3 0.0502 -> 0.0481, C2NN 0.0530 -> 0.0519, C3 0.1741 -> 0.1727, C2 / C1 / C1S / P1 / P2 / C2L within 0.6 %.  Measured and
// not kept: the kernels without plan-time boxes at priority through coordinates and box reduction (C5 +3 %), priority around the
// mirror pair kernel's second-eye requests and the batch ring's refills (C2 +1.5 %, C1 +2 %), priority 1 for the mirror kernel's
// gather + store phases (C1 +9 %) or for its coordinates (C1 +3 %, P1 -3.5 %, C2 +1 %), the gray / BGRA kernels (+-0.3 %), the K x K
// kernels keeping the priority through their coordinates (C4 +2.6 %).
#ifndef V1C_PRIO
#define V1C_PRIO 15
#endif

namespace v1c {


#ifndef V1C_TILE_W
#define V1C_TILE_W 64
#endif
// ---- how the kernels see their arguments (kernels.hpp: TileArgs) ----
// Constant address space: a load through these is a scalar load (s_load) whatever stores or asm statements surround it -- the
// fence-less vmcnt protocol of the LDS-DMA kernels must never see a compiler-made VECTOR load behind a request -- and, unlike a load
// of a by-value kernel argument, it is not speculated to the kernel entry: the value occupies scalar registers from its use on.
#define V1C_CONST __attribute__((address_space(4)))
typedef const V1C_CONST KernelCtx& ctx_cref;
typedef const V1C_CONST Geom& geom_cref;
typedef const V1C_CONST RayParams& ray_cref;
typedef const V1C_CONST DevUnit* units_cptr;
typedef const V1C_CONST TileArgs& args_cref;
typedef const V1C_CONST double* rot_cptr;  // 9 doubles, row-major

// the kernel's own argument block (its one by-value parameter: offset 0 of the kernel-argument segment)
template <int OFFSET = 0>
__device__ __forceinline__ args_cref kernel_args()
{
    return *(const V1C_CONST TileArgs*)((const V1C_CONST uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() + OFFSET);
}

// ---- the preloaded head of the mirror launches ----
// gfx950 can hand a kernel its first kernel-argument dwords in scalar registers at wave start (14 of them next to the kernel-argument
// pointer; -mllvm -amdgpu-kernarg-preload-count=16: csrc/Makefile).  The mirror kernels -- the launches that ARE one chain of dependent
// memory round trips (config 1: one round of workgroups) -- take what their first loads need that way, as eight scalar parameters in front
// of the argument block: the (tile, band) box pairs, the tile-order constants, the destination size, the base of the plan's six row /
// column tables (one buffer: plan.hip), the plan's context and the box buffer size / mirror row.  Box pair, row / column values, the
// context's constants and the pair's unit records (in the block at a known offset) are then ALL requested at once: the fast path's scalar
// chain is one round trip instead of three.  (A struct parameter cannot be preloaded, hence the packing: 11 dwords.)
#define V1C_MIRROR_HEAD                                                                                                                  \
    const TileBox *pairs, unsigned tiles_x_magic, unsigned gx_rest /* tiles_x | rest_rows << 16 */,                                      \
        unsigned rows_strip /* rows of tile pairs | XCD strip rows (0 / 2) << 16 */, unsigned dst_wh /* dst_w | dst_h << 16 */,          \
        const double *rowcol_tables, const KernelCtx *ctxp, unsigned kb_mh /* box buffer KB | mirror row << 16 */
constexpr int kMirrorHeadBytes = 48;
struct DstSize {
    int dst_w, dst_h;
};
struct RowColTabs {
    const double *col_s, *col_c, *col_h, *row_s, *row_c, *row_h;
};
// the kernel-argument segment of a mirror kernel as the ABI lays it out (natural alignment, declaration order): the argument block
// behind the head (tests/test_resource_budget.py checks the same offset in the code object's metadata)
struct MirrorKernargs {
    const TileBox* pairs;
    unsigned tiles_x_magic, gx_rest, rows_strip, dst_wh;
    const double* rowcol_tables;
    const KernelCtx* ctxp;
    unsigned kb_mh;
    TileArgs args;
};
static_assert(offsetof(MirrorKernargs, args) == kMirrorHeadBytes, "kernel_args<kMirrorHeadBytes>() must point at the TileArgs block");
// the six tables of a plan from the base of their buffer (plan.hip: col_s | col_c | col_h of wpad entries, row_s | row_c | row_h of dst_h)
__device__ __forceinline__ RowColTabs rowcol_tables_at(const double* base, int dst_w, int dst_h)
{
    const int wpad = (dst_w + 3) & ~3;
    return RowColTabs{base, base + wpad, base + 2 * wpad, base + 3 * wpad, base + 3 * wpad + dst_h, base + 3 * wpad + 2 * dst_h};
}
__device__ __forceinline__ ctx_cref args_ctx(args_cref a)
{
    return *(const V1C_CONST KernelCtx*)a.ctx;
}
__device__ __forceinline__ units_cptr args_units(args_cref a)
{
    return a.units ? (units_cptr)a.units : (units_cptr)a.inl;
}
// ---- scalar-load clauses ----
// A constant-address-space load sits where its value is first used -- right for everything a slow path reads, wrong for what every
// workgroup needs in its prologue: a chain of five or six dependent scalar loads, each a round trip to L2 or HBM (the argument block
// is fresh memory every launch) in front of the first box request, and more of them behind the first barrier (the Denormalize constants
// of the coordinates).  A launch that fits the machine in one round of workgroups (config 1: 1056 of 1792 slots) IS that chain.  The
// touch_* helpers read a group of values at one point (the empty asm statement needs them in registers), so the compiler issues their
// loads there as one clause and waits once; later reads of the same fields are the same loads (common subexpressions: the address
// space is constant).  V1C_TOUCH=0 (A/B builds): every load where it is used.
#ifndef V1C_TOUCH
#define V1C_TOUCH 1
#endif
// everything of the argument block's header a fast path reads (and the grid size: a hidden kernel argument)
__device__ __forceinline__ void touch_args(args_cref a)
{
#if V1C_TOUCH
    asm volatile("" ::"s"(a.ctx), "s"(a.units), "s"(a.boxes), "s"(a.mboxes), "s"(a.n_units), "s"(a.upb), "s"(a.tiles_x_magic), "s"(a.strip_len),
                 "s"(a.strip_magic), "s"(a.rest_rows), "s"(a.mirror_h), "s"(a.kb), "s"(gridDim.x), "s"(gridDim.y), "s"(a.col_s), "s"(a.col_c),
                 "s"(a.col_h), "s"(a.row_s), "s"(a.row_c), "s"(a.row_h), "s"(a.dst_w), "s"(a.dst_h));
#endif
}
// the second round: the plan's constants of the coordinate evaluation, the table pointers of the prologue and the pointers / pitches of
// the first NU units of the workgroup, all in ONE statement (two statements are two waits)
template <int ROT, int NU>
__device__ __forceinline__ void touch_plan_and_units(ctx_cref c, units_cptr U, int z0, int z1)
{
#if V1C_TOUCH
    ray_cref P = c.ray;
    if (NU == 2)
        asm volatile("" ::"s"(P.rx32), "s"(P.ry32), "s"(P.cx32), "s"(P.cy32), "s"(P.inv_step), "s"(P.inv_step_f), "s"(P.n_int), "s"(P.radial),
                     "s"(P.radial_m), "s"(c.g.src_h), "s"(c.g.src_w), "s"(U[z0].src), "s"(U[z0].dst), "s"(U[z0].src_pitch), "s"(U[z0].dst_pitch),
                     "s"(U[z1].src), "s"(U[z1].dst), "s"(U[z1].src_pitch), "s"(U[z1].dst_pitch));
    else
        asm volatile("" ::"s"(P.rx32), "s"(P.ry32), "s"(P.cx32), "s"(P.cy32), "s"(P.inv_step), "s"(P.inv_step_f), "s"(P.n_int), "s"(P.radial),
                     "s"(P.radial_m), "s"(c.g.src_h), "s"(c.g.src_w), "s"(U[z0].src), "s"(U[z0].dst), "s"(U[z0].src_pitch), "s"(U[z0].dst_pitch));
#endif
}

// the table a tile's slice is cut from.  (Two loaded VALUES and a select: left alone the compiler selects between the two ADDRESSES and
// loads through the result -- a dependent scalar load, i.e. one more round trip, in front of the slice's request.)
__device__ __forceinline__ const double* radial_table(ray_cref P, bool mpoly)
{
    const double* m = P.radial_m;
    const double* r = P.radial;
    asm volatile("" : "+s"(m), "+s"(r));
    return mpoly ? m : r;
}

// a generic copy of the geometry for the border-aware samplers of v1c_core.hpp (slow paths only)
__device__ __forceinline__ Geom geom_copy(geom_cref g)
{
    Geom r;
    r.src_h = g.src_h, r.src_w = g.src_w, r.dst_h = g.dst_h, r.dst_w = g.dst_w;
    r.cn = g.cn, r.interp = g.interp, r.border = g.border;
#pragma unroll
    for (int k = 0; k < 4; k++)
        r.cval[k] = g.cval[k];
    return r;
}

constexpr int kTW = V1C_TILE_W;            // output tile width (px); height = threads / kLanesX
constexpr int kLanesX = kTW / 4;           // lanes per tile row (4 px each)
#ifndef V1C_BOX_KB
#define V1C_BOX_KB 24
#endif
#ifndef V1C_XCD_SWIZZLE
#define V1C_XCD_SWIZZLE 1
#endif
#ifndef V1C_PAIR_WAVES
#define V1C_PAIR_WAVES 6  // waves per SIMD the pair kernel (bilinear, no rotation, OWN = 0) is compiled for
#endif
#ifndef V1C_NOBOX_WAVES
#define V1C_NOBOX_WAVES 1  // waves per SIMD the bilinear kernel without plan-time boxes (per-unit rotations) is compiled for
#endif
#ifndef V1C_CHUNK_MAP_FP32
#define V1C_CHUNK_MAP_FP32 1  // chunk -> (row, column) of the staging map by an fp32 reciprocal (0: integer magic multiply)
#endif
#ifndef V1C_LEAN_RING
#define V1C_LEAN_RING 2  // box buffers of k_ray_lin3_batch_lean_raw (2 or 3: the boxes of 1 or 2 units in flight; C3: 0.1807 / 0.1852 ms,
                         // 7 / 5 workgroups per CU)
#endif
#ifndef V1C_RAW_WAVES
#define V1C_RAW_WAVES 5  // waves per SIMD k_ray_lin3_pair_mirror_raw is compiled for
#endif
#ifndef V1C_LEAN_WAVES
#define V1C_LEAN_WAVES 6  // waves per SIMD the lean batch kernel is compiled for (no rotation, OWN = 0)
#endif
#ifndef V1C_UPB
#define V1C_UPB 8
#endif
constexpr int kBoxBytes = V1C_BOX_KB * 1024;  // LDS budget for the source box (4 B per source pixel)
constexpr int kMaxCpr = 64;                // 4-pixel chunks per box row (magic division bound)
// a thread stages up to 4 chunks: 1024 chunks per 256-thread workgroup, 2048 per 512-thread one

// wave-wide signed min via DPP (no LDS traffic): after the six steps lane 63 holds the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_step(int v)
{
    // lanes / rows without a source get the identity of min
    return min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false));
}

__device__ __forceinline__ int wave_min_to_lane63(int v)
{
    v = dpp_min_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_min_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_min_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_min_step<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row = row minimum
    v = dpp_min_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_min_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return v;
}

// (sx, sy) = cv2's fixed-point coordinates cvRound(32 x), cvRound(32 y)
// (bit 24 of the result: the pixel is to be left untouched -- BORDER_TRANSPARENT with a footprint that leaves the source)
__device__ __forceinline__ uint32_t slow_pixel_linear3_t(const uint8_t* src, int64_t pitch, int h, int w, geom_cref gc, int sx, int sy)
{
    const Geom g = geom_copy(gc);
    uint8_t px[3] = {0, 0, 0};
    const Image im{src, pitch, h, w};
    const bool st = sample_linear_t<3>(im, g, taps_from_fixed(sx, sy), px);
    return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | (st ? 0u : 1u << 24);
}

// the same for INTER_NEAREST (the NN = 1 kernels: fixed point 32 * saturate_cast<short>(cvRound(x)), so sx >> 5 is remapNearest's pixel):
// remapNearest's border rules -- BORDER_TRANSPARENT leaves a pixel untouched when the pixel ITSELF lies outside, not when its bilinear
// footprint does
__device__ __forceinline__ uint32_t slow_pixel_nearest3_t(const uint8_t* src, int64_t pitch, int h, int w, geom_cref gc, int sx, int sy)
{
    const Geom g = geom_copy(gc);
    uint8_t px[3] = {0, 0, 0};
    const Image im{src, pitch, h, w};
    const bool st = sample_nearest<3>(im, g, (float)(sx >> 5), (float)(sy >> 5), px);  // (|sx >> 5| <= 2^15: exact in fp32)
    return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | (st ? 0u : 1u << 24);
}

struct u128 {
    uint32_t x, y, z, w;
};

// source box of one output tile: pixels [x0, x0 + 4*cpr) x rows [y0, y0 + nrows); cpr == 0: no
// pixel of the tile has its 2x2 cell inside the source
// idx0 / nidx: range of radial-table entries the tile's in-table pixels use (nidx == 0: unknown)
struct TileBox {
    int x0, y0, cpr, nrows;
    int idx0, nidx;
    int interior;  // 1: every pixel of the tile has valid coordinates and its whole footprint inside the source
    int magic;     // ceil(2^20 / cpr): floor(ch / cpr) == (ch * magic) >> 20 for ch < 16k, cpr <= 64 (chunk map)
};

// (plan-time data, never written by a remap launch: read through the constant address space -- scalar loads; a plain pointer that
// itself came out of memory would make them vector loads)
__device__ __forceinline__ TileBox load_tile_box(const TileBox* boxes, int tile)
{
    typedef int __attribute__((ext_vector_type(4))) i32x4;
    const V1C_CONST i32x4* bp = (const V1C_CONST i32x4*)(boxes + __builtin_amdgcn_readfirstlane(tile));
    const i32x4 b0 = bp[0], b1 = bp[1];
    TileBox b;
    b.x0 = b0.x, b.y0 = b0.y, b.cpr = b0.z, b.nrows = b0.w, b.idx0 = b1.x, b.nidx = b1.y, b.interior = b1.z, b.magic = b1.w;
    return b;
}

// the mirror launches' (tile box, band box) pair: one 64-byte entry, one scalar load
struct TileBoxPair {
    TileBox b, q;
};
__device__ __forceinline__ TileBoxPair load_tile_box_pair(const TileBox* pairs, int tile)
{
    typedef int __attribute__((ext_vector_type(16))) i32x16;
    const i32x16 v = *(const V1C_CONST i32x16*)(pairs + 2 * __builtin_amdgcn_readfirstlane(tile));
    TileBoxPair r;
    r.b.x0 = v[0], r.b.y0 = v[1], r.b.cpr = v[2], r.b.nrows = v[3], r.b.idx0 = v[4], r.b.nidx = v[5], r.b.interior = v[6], r.b.magic = v[7];
    r.q.x0 = v[8], r.q.y0 = v[9], r.q.cpr = v[10], r.q.nrows = v[11], r.q.idx0 = v[12], r.q.nidx = v[13], r.q.interior = v[14], r.q.magic = v[15];
    return r;
}

// ceil(2^20 / cpr) for cpr = 1 .. kMaxCpr (a wave-uniform table read instead of an integer division,
// which the compiler expands to ~25 vector instructions)
struct ChunkMagicLut {
    int v[kMaxCpr + 1];
    constexpr ChunkMagicLut() : v{}
    {
        for (int q = 1; q <= kMaxCpr; q++)
            v[q] = (int)(((1u << 20) + (unsigned)q - 1u) / (unsigned)q);
    }
};
__device__ const ChunkMagicLut kChunkMagic{};

#ifndef V1C_KXK_OWN_LANES
#define V1C_KXK_OWN_LANES 0  // 1: bicubic / Lanczos4 pairs gather in the coordinates' lane -> pixel mapping (A/B builds)
#endif
constexpr int kKxkExchangeBytes = 4096;  // K x K pair path: 4 waves x (4 rows x 64 columns) dwords behind the box buffers
constexpr int kTabSlice = 64;  // radial-table entries a workgroup may keep in LDS (4 KB)

// Tiles the lean batch kernel takes (k_ray_lin3_batch_lean), as far as the plan can tell: interior, table
// slice and box fit.  One definition for the kernel's own test and for the host's list of the
// remaining tiles (tile_rest_list), which the general kernel then serves.
// (a box of up to TWO buffers is served too, single-buffered: see the lean path in shared_map_tile)
__host__ __device__ inline bool lean_static_ok(int cpr, int nrows, int nidx, int interior, int half_dwords)
{
    return interior != 0 && nidx > 0 && nidx <= kTabSlice && cpr > 0 && cpr <= kMaxCpr && nrows * cpr <= 1024 &&
           nrows * (cpr * 4 + 4) <= 2 * half_dwords;
}
constexpr int kUnitsPerBlock = V1C_UPB;  // units sharing the map that one workgroup serves (BOXES = 1)

struct LaneCoords {
    int idx_lo, idx_hi;        // range of table entries of the lane's in-table pixels (k_tile_boxes)
    int sx[kPX], sy[kPX];      // cv2's fixed point: cvRound(32 x)
    int sy2[kPX];              // MIRROR: cvRound(32 y) of the same columns in the row mirrored about the equator
    unsigned ok;               // coordinate valid (inside the radial table's domain), bit per pixel
    unsigned inside;           // ... and the whole 2x2 cell (plus 8 readable bytes) inside the source
};

// ---- coordinates of a lane's 4 pixels: identical operations to ray_eval() (v1c_core.hpp) ----
struct RowCol {
    double sl, cl, hl;            // row: sin / cos / 1-cos of the latitude
    double slon[kPX], qlon[kPX];  // columns: sin(lon) and 1-cos(lon) (no rotation) or cos(lon) (rotation)
};

// `P`: where the six table pointers are read from -- the plan's context (c.ray) or the launch's argument block, which carries
// copies so that these loads need not wait for the context (kernels.hpp: TileArgs)
template <int ROT, typename Tables>
__device__ __forceinline__ void load_rowcol(const Tables& P, int xc, int jc, RowCol& rc)
{
    rc.sl = P.row_s[jc], rc.cl = P.row_c[jc], rc.hl = P.row_h[jc];
    const double* __restrict__ ps = P.col_s + xc;
    // (ROT = 2, the general modes: the third column table -- 1 - cos(lon) / xn^2, or the cosine itself for lat_x plans: gen_vector)
    const double* __restrict__ pq = (ROT == 1 ? P.col_c : P.col_h) + xc;
#pragma unroll
    for (int k = 0; k < kPX; k++)
        rc.slon[k] = ps[k], rc.qlon[k] = pq[k];
}

// K = taps per axis: 2 (bilinear), 4 (bicubic), 8 (Lanczos4); top-left tap at ix - (K/2 - 1).
// `tab` = radial table (global memory, or the tile's slice in LDS starting at entry `tab0`).
// OWN = 0: the plan proved that pixel 1's entry is valid for all 4 pixels of every lane
// (plan.hip: shared_entry), so the per-pixel fallback is not compiled in.
// INTERIOR = 1: the plan found every pixel of this tile valid and inside (TileBox::interior, same
// arithmetic): the validity / inside tests are skipped.
// MPOLY = 1 (OWN = 0 only): `tab` is the table of polynomials in m (RayParams::radial_m): the
// interval index comes from fp32 arithmetic (an fp32 square root for w-tables) and
// G = poly6(m - m_c) -- no fp64 root, no index conversion in fp64.  The
// plan flags the tiles where every entry a lane can pick is valid at the level the lanes need.
// MIRROR = 1 (no rotation, INTERIOR = 1): also the y coordinate of the lane's 4 columns in the output row mirrored
// about the equator (row 2 * norm_cy - j): there sin(lat) changes sign and nothing else does (the host's row tables
// are exactly antisymmetric / symmetric), so m, G and x are the same numbers and y32' = fma(G, -ky, cy32) -- bit for
// bit what the mirrored row's own evaluation gives.
// NN = 1: INTER_NEAREST through the bilinear machinery.  cv2's remapNearest reads the ONE pixel (cvRound(x), cvRound(y)) of the float32
// coordinates (half to even, saturated to short); the kernels' fixed point is cvRound(32 x), so the lane's coordinates become
// 32 * cvRound(x): fractions zero, for which the bilinear blend returns its top-left tap exactly ((65535 p + 32768) >> 16 == p,
// (1024 p + 512) >> 10 == p in the border-aware sampler, whose other three taps -- pixels or border values -- carry weight 0, and whose
// top-left tap follows borderInterpolate like remapNearest's).  The rounded pixel lies inside the bilinear footprint of the same
// coordinate, so boxes computed with NN = 1 (k_tile_boxes) bound it.
template <int VAR_W, int ROT, int K, int OWN, int INTERIOR, int MPOLY, int MIRROR = 0, int NN = 0, typename TabPtr>
// `rot`: the rotation that applies (ROT != 0): the unit's own record -- the host stores the EFFECTIVE matrix there, the unit's override
// or the chain's composed rotation (plan.hip: fill_unit) -- or c.ray.rot (k_tile_boxes)
// `ext` (K = 4 / 8, BGR, BORDER_CONSTANT: kxk_ext()): a pixel counts as `inside` when its K x K footprint OVERLAPS the source instead of
// lying inside it -- the consumer stages the box with the border colour around the image (stage_load_ext), so the taps beyond the edge
// read what remapBicubic / remapLanczos4's constant-border branch substitutes for them (oracle/vr180_oracle.c:415-449) and the
// pixel never takes the per-pixel patch path.
__device__ __forceinline__ void lane_coords(ctx_cref c, rot_cptr rot, const RowCol& rc, int npx, TabPtr tab,
                                            int tab0, int tabn, LaneCoords& L, int ext = 0)
{
    ray_cref P = c.ray;
    geom_cref g = c.g;
    const double sl = rc.sl, cl = rc.cl, hl = rc.hl;
    const double rx32 = P.rx32, ry32 = P.ry32, cx32 = P.cx32, cy32 = P.cy32;

    double A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, B2 = 0, C0 = 0, C1 = 0, C2 = 0;
    if (ROT == 1) {
        A0 = rot[0] * cl, B0 = rot[2] * cl, C0 = rot[1] * sl;
        A1 = rot[3] * cl, B1 = rot[5] * cl, C1 = rot[4] * sl;
        A2 = rot[6] * cl, B2 = rot[8] * cl, C2 = rot[7] * sl;
    }
    const double* slon = rc.slon;
    const double* qlon = rc.qlon;
    // factors of G (see ray_eval): x32 = (G*kx)*fx_[k] + cx32 ; y32 = (G*ky)*fy_[k] + cy32 (rotation)
    //                                                          y32 = G*ky + cy32          (none)
    const double kx = ROT ? rx32 : rx32 * cl, ky = ROT ? ry32 : ry32 * sl;
    double fx_[kPX], fy_[kPX], tt[kPX], mm[kPX];
    int idx[kPX];
    unsigned in_table = MPOLY ? 0xFu : 0u;
    unsigned gen_bad = 0;  // ROT = 2: pixels whose base variable lies outside the S / Cm tables (RayParams::gen_mode 2)
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        double m;
        if (ROT == 2) {
            // the general modes (lat_x; radial stages in front of the rotation): the rotated ray by gen_vector, the very function the
            // generic kernel's ray_eval calls
            if (!gen_vector(P, rot, sl, cl, hl, slon[k], qlon[k], fx_[k], fy_[k], m))
                gen_bad |= 1u << k;
        } else if (ROT) {
            fx_[k] = fma(A0, slon[k], fma(B0, qlon[k], C0));
            fy_[k] = fma(A1, slon[k], fma(B1, qlon[k], C1));
            m = 1.0 - fma(A2, slon[k], fma(B2, qlon[k], C2));
        } else {
            fx_[k] = slon[k];
            fy_[k] = 1.0;
            m = fma(cl, qlon[k], hl);
        }
        mm[k] = m;
        if (MPOLY) {
            tt[k] = 0.0, idx[k] = 0;
            continue;
        }
        const double u = VAR_W ? fast_sqrt_half(m) : m;
        tt[k] = u * P.inv_step;
        const int ir = table_index(tt[k]);
        in_table |= (unsigned)ir < (unsigned)P.n_int ? 1u << k : 0u;
        idx[k] = min(ir, P.n_int - 1);  // clamped: always a readable entry
    }
    if (ROT == 2)
        in_table &= ~gen_bad;

    // radial table: one entry (that of pixel 1) serves all 4 pixels where it may
    double G[kPX];
    if (MPOLY) {
        // interval of pixel 1 from an fp32 root (an index off by one near an interval boundary is
        // covered by the 0.01 margin the entries were validated with)
        const int ic = VAR_W ? (int)(__builtin_amdgcn_sqrtf((float)(0.5 * mm[1])) * P.inv_step_f) : (int)((float)mm[1] * P.inv_step_f);
        for (int k = 0; k < kPX; k++)
            idx[k] = ic;
        double e[kRadialCoefs];
        typedef double __attribute__((ext_vector_type(2))) d2;
        const uint32_t rel = (uint32_t)min(max(ic - tab0, 0), tabn - 1);
        const d2* p2 = (const d2*)(tab + rel * (uint32_t)kRadialCoefs);
#pragma unroll
        for (int q = 0; q < kRadialCoefs / 2; q++) {
            const d2 v = p2[q];
            e[2 * q] = v.x, e[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const double dk = mm[k] - e[kRadialCoefs - 1];  // m - m_c
            double gk = e[kRadialCoefs - 2];
#pragma unroll
            for (int q = kRadialCoefs - 3; q >= 0; q--)
                gk = fma(gk, dk, e[q]);
            G[k] = gk;
        }
    } else {
        const int ic = idx[1];
        double e[kRadialCoefs];
        {
            typedef double __attribute__((ext_vector_type(2))) d2;
            // (slice-relative and clamped: out-of-table pixels read some entry and are discarded)
            const uint32_t rel = (uint32_t)min(max(ic - tab0, 0), tabn - 1);
            const d2* p2 = (const d2*)(tab + rel * (uint32_t)kRadialCoefs);
#pragma unroll
            for (int q = 0; q < kRadialCoefs / 2; q++) {
                const d2 v = p2[q];
                e[2 * q] = v.x, e[2 * q + 1] = v.y;
            }
        }
        // pixel 1's entry serves a pixel inside the range the entry was validated on -- if it is the entry that was read (v1c_core.hpp:
        // shared_entry_serves; a pixel 1 outside the table -- general mode 2: its base variable outside the S / Cm tables, its m
        // arbitrary -- can point beside the slice, and two good pixels of such a lane once took the clamped neighbour's polynomial about
        // the wrong centre: 0.4 px off, tools/fuzz.py seed 34 case 1974)
        const double zc = (double)ic + 0.5;
        unsigned own = 0;  // pixels that must use their own entry
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const double zk = tt[k] - zc;
            if (OWN) {
                const bool usec = shared_entry_serves(zk, e[kRadialDegree], ic, tab0, tabn);
                own |= (!usec & (bool)((in_table >> k) & 1)) ? 1u << k : 0u;
            }
            double gk = e[kRadialDegree];
#pragma unroll
            for (int q = kRadialDegree - 1; q >= 0; q--)
                gk = fma(gk, zk, e[q]);
            G[k] = gk;
        }
        if (OWN && own) {
#pragma unroll
            for (int k = 0; k < kPX; k++) {
                if (own & (1u << k)) {
                    const double* __restrict__ pc = P.radial + (size_t)idx[k] * kRadialCoefs;  // always from global
                    const double zk = tt[k] - ((double)idx[k] + 0.5);
                    double gk = pc[kRadialDegree];
#pragma unroll
                    for (int q = kRadialDegree - 1; q >= 0; q--)
                        gk = fma(gk, zk, pc[q]);
                    G[k] = gk;
                }
            }
        }
    }

    L.ok = INTERIOR ? 0xFu : 0u, L.inside = INTERIOR ? 0xFu : 0u;
    L.idx_lo = 0x7fffffff, L.idx_hi = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const bool it = (in_table >> k) & 1;
        L.idx_lo = min(L.idx_lo, it ? idx[k] : 0x7fffffff), L.idx_hi = max(L.idx_hi, it ? idx[k] : -0x7fffffff);
    }
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const double x32 = fma(G[k] * kx, fx_[k], cx32);
        const double y32 = ROT ? fma(G[k] * ky, fy_[k], cy32) : fma(G[k], ky, cy32);
        const float fxk = (float)x32, fyk = (float)y32;  // = 32 * float32(x)
        if (INTERIOR) {
            // cvRound by the 1.5 * 2^23 trick (two full-rate instructions instead of v_rndne_f32 +
            // v_cvt_i32_f32): the add rounds to the nearest integer, ties to even, exactly like
            // rint(); valid for |32 x| < 2^22, and interior coordinates are inside the source (< 2^20)
            // INTERIOR == 2: nothing is known about the tile yet -- the coordinates are clamped into
            // the trick's range (NaN -> lower bound), which leaves every coordinate inside a source
            // (< 2^15 px) untouched; the caller derives "interior" from the bounding box of ALL
            // pixels and redoes the tile with INTERIOR = 0 when it is not.
            const float ax = INTERIOR == 2 ? __builtin_amdgcn_fmed3f(fxk, -4194303.0f, 4194303.0f) : fxk;
            const float ay = INTERIOR == 2 ? __builtin_amdgcn_fmed3f(fyk, -4194303.0f, 4194303.0f) : fyk;
            if (NN) {  // 32 * cvRound(x): x = fxk / 32 exactly (a power of two)
                L.sx[k] = (__float_as_int(ax * 0.03125f + 12582912.0f) - 0x4B400000) * 32;
                L.sy[k] = (__float_as_int(ay * 0.03125f + 12582912.0f) - 0x4B400000) * 32;
                continue;
            }
            L.sx[k] = __float_as_int(ax + 12582912.0f) - 0x4B400000;
            L.sy[k] = __float_as_int(ay + 12582912.0f) - 0x4B400000;
            if (MIRROR && !ROT)
                L.sy2[k] = __float_as_int((float)fma(G[k], -ky, cy32) + 12582912.0f) - 0x4B400000;
            continue;
        }
        // flagged intervals carry NaN coefficients; |32 x| < 2^30 keeps the int conversion exact
        const bool good = (bool)((in_table >> k) & 1) & (fabsf(fxk) < 1073741824.0f) & (fabsf(fyk) < 1073741824.0f);
        const bool okk = good & (k < npx);
        L.ok |= okk ? 1u << k : 0u;
        // branch-free: the conversion always sees an in-range float (NaN -> clamped by med3);
        // pixels that are not `good` are never used
        if (NN) {  // saturate_cast<short>(cvRound(x)) * 32
            L.sx[k] = __float2int_rn(__builtin_amdgcn_fmed3f(fxk * 0.03125f, -32768.0f, 32767.0f)) * 32;
            L.sy[k] = __float2int_rn(__builtin_amdgcn_fmed3f(fyk * 0.03125f, -32768.0f, 32767.0f)) * 32;
        } else {
            L.sx[k] = __float2int_rn(__builtin_amdgcn_fmed3f(fxk, -1073741824.0f, 1073741824.0f));
            L.sy[k] = __float2int_rn(__builtin_amdgcn_fmed3f(fyk, -1073741824.0f, 1073741824.0f));
        }
        const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
        // whole footprint inside the source (remapBilinear / remapBicubic / remapLanczos4 inlier
        // test); the bilinear path additionally wants 8 readable bytes per row for its global-memory
        // fallback, hence w - 2 there
        constexpr int off = K / 2 - 1;
        // (ext: top-left tap in [-(K - 1), len - 1], i.e. the footprint shares at least one row and one column with the source)
        const int lo_e = (K != 2 && ext) ? K - 1 : 0;
        const unsigned wb = (K != 2 && ext) ? (unsigned)(g.src_w + K - 1) : (unsigned)max(g.src_w - (K - 1), 0);
        const unsigned hb = (K != 2 && ext) ? (unsigned)(g.src_h + K - 1) : (unsigned)max(g.src_h - (K - 1), 0);
        const bool in = K == 2 ? okk & ((unsigned)ix < (unsigned)max(g.src_w - 2, 0)) & ((unsigned)iy < (unsigned)max(g.src_h - 1, 0))
                               : okk & ((unsigned)(ix - off + lo_e) < wb) & ((unsigned)(iy - off + lo_e) < hb);
        L.inside |= in ? 1u << k : 0u;
    }
}

// ---- workgroup-wide bounding box of the inside pixels (DPP mins + one LDS exchange) ----
template <int K, int NW>
__device__ __forceinline__ TileBox reduce_box(const LaneCoords& L, int* red, int tid)
{
    int xmn = 32767, ymn = 32767, nxmx = 32767, nymx = 32767;  // running mins of x, y, -x, -y
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
        const bool in = (L.inside >> k) & 1;
        xmn = min(xmn, in ? ix : 32767), ymn = min(ymn, in ? iy : 32767);
        nxmx = min(nxmx, in ? -ix : 32767), nymx = min(nymx, in ? -iy : 32767);
    }
    xmn = wave_min_to_lane63(xmn), ymn = wave_min_to_lane63(ymn);
    nxmx = wave_min_to_lane63(nxmx), nymx = wave_min_to_lane63(nymx);
    if ((tid & 63) == 63) {
        int* r = red + (tid >> 6) * 4;
        r[0] = xmn, r[1] = ymn, r[2] = nxmx, r[3] = nymx;
    }
    __syncthreads();
    int m0 = red[0], m1 = red[1], m2 = red[2], m3 = red[3];
#pragma unroll
    for (int w = 1; w < NW; w++)
        m0 = min(m0, red[4 * w]), m1 = min(m1, red[4 * w + 1]), m2 = min(m2, red[4 * w + 2]), m3 = min(m3, red[4 * w + 3]);
    const int bx0r = __builtin_amdgcn_readfirstlane(m0);
    const int by0 = __builtin_amdgcn_readfirstlane(m1);
    const int bx1 = -__builtin_amdgcn_readfirstlane(m2);
    const int by1 = -__builtin_amdgcn_readfirstlane(m3);
    // footprint of pixel (ix, iy): columns ix-off .. ix-off+K-1, rows iy-off .. iy-off+K-1
    constexpr int off = K / 2 - 1;
    TileBox b;
    b.x0 = (bx0r - off) & ~3;  // the box starts on a 4-pixel (12-byte) boundary
    b.y0 = by0 - off;
    b.cpr = bx0r <= bx1 ? (bx1 - off + K - b.x0 + 3) >> 2 : 0;
    b.nrows = by1 - by0 + K;
    b.idx0 = b.nidx = b.interior = 0;
    b.magic = b.cpr > 0 ? (int)(((1u << 20) + (unsigned)b.cpr - 1u) / (unsigned)b.cpr) : 0;
    return b;
}

__device__ __forceinline__ bool box_fits(const TileBox& b, const uint8_t* src, uint32_t spitch, int max_chunks, int box_dwords)
{
    const int lpw = b.cpr * 4 + 4;
    return (b.cpr > 0) & (b.cpr <= kMaxCpr) & (b.nrows * b.cpr <= max_chunks) & (b.nrows * lpw <= box_dwords) &
           (((((uintptr_t)src) | spitch) & 3) == 0);
}

// ---- bounding box of ALL of the tile's pixels (no validity masks), K = 2 ----
// Four minima (x, y, -x, -y of cv2's fixed point, shifted to pixels at the end) in 10 VALU
// instructions per wave: v_permlane32_swap / v_permlane16_swap (gfx950) fold two registers' halves
// into one with a single v_min each -- after them row r of 16 lanes holds quantity r's partial
// minimum -- and four v_min_i32_dpp row_ror steps finish each row.  (The masked reduction above
// costs 48 + 16 selects.)  xmin / xmax / ymin / ymax come back wave-uniform in SGPRs.
struct BoxAll {
    int xmin, xmax, ymin, ymax;  // source pixel of the top-left tap, over all 1024 pixels
};

template <int CTRL>
__device__ __forceinline__ int row_min_step(int v)
{
    return min(v, __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true));
}

template <int NW>
__device__ __forceinline__ BoxAll reduce_box_all(const LaneCoords& L, int* red, int tid)
{
    const int a = min(min(L.sx[0], L.sx[1]), min(L.sx[2], L.sx[3]));
    const int b = min(min(L.sy[0], L.sy[1]), min(L.sy[2], L.sy[3]));
    const int c = -max(max(L.sx[0], L.sx[1]), max(L.sx[2], L.sx[3]));
    const int d = -max(max(L.sy[0], L.sy[1]), max(L.sy[2], L.sy[3]));
    // lanes 0-31: a folded over both halves, lanes 32-63: b folded (likewise c | d)
    const auto ab = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    const auto cd = __builtin_amdgcn_permlane32_swap((unsigned)c, (unsigned)d, false, false);
    const int pab = min((int)ab[0], (int)ab[1]), pcd = min((int)cd[0], (int)cd[1]);
    // rows of 16 lanes: a | c | b | d
    const auto q = __builtin_amdgcn_permlane16_swap((unsigned)pab, (unsigned)pcd, false, false);
    int v = min((int)q[0], (int)q[1]);
    v = row_min_step<0x121>(v);  // row_ror:1
    v = row_min_step<0x122>(v);  // row_ror:2
    v = row_min_step<0x124>(v);  // row_ror:4
    v = row_min_step<0x128>(v);  // row_ror:8: every lane of a row holds the row's minimum
    if ((tid & 15) == 0)
        red[(tid >> 6) * 4 + ((tid >> 4) & 3)] = v;
    __syncthreads();
    int m0 = red[0], m1 = red[1], m2 = red[2], m3 = red[3];
#pragma unroll
    for (int w = 1; w < NW; w++)
        m0 = min(m0, red[4 * w]), m1 = min(m1, red[4 * w + 1]), m2 = min(m2, red[4 * w + 2]), m3 = min(m3, red[4 * w + 3]);
    BoxAll r;
    r.xmin = __builtin_amdgcn_readfirstlane(m0) >> 5;
    r.xmax = (-__builtin_amdgcn_readfirstlane(m1)) >> 5;
    r.ymin = __builtin_amdgcn_readfirstlane(m2) >> 5;
    r.ymax = (-__builtin_amdgcn_readfirstlane(m3)) >> 5;
    return r;
}

struct Staged {
    uint32_t w0[4], w1[4], w2[4];
};

// Where a thread's (up to) four 4-pixel chunks of the box live: computed once per tile -- the
// decomposition chunk -> (row, column) costs half-rate integer multiplies -- and reused by every
// unit the workgroup serves.
struct ChunkMap {
    uint32_t row[4];     // box row of chunk q
    uint32_t xbyte[4];   // byte offset of the chunk inside a source row
    uint32_t lds_dw[4];  // dword index in the LDS box
    unsigned valid;      // bit q: chunk q exists
};

// (`ch0`: first chunk of the round -- boxes of more than 4 NT chunks are staged in two rounds by the pair code: shared_map_tile)
template <int NT, int NQ = 4>
__device__ __forceinline__ void make_chunk_map(const TileBox& b, int tid, ChunkMap& M, int ch0 = 0)
{
    const int nchunks = b.nrows * b.cpr;
    const int lpw = b.cpr * 4 + 4;  // LDS row pitch in dwords (+4: rotate the banks from row to row)
#if V1C_CHUNK_MAP_FP32
    // floor(ch / cpr) for ch < 16 K, cpr <= 64 in fp32: (ch + 0.5) / cpr stays 0.5 / 64 away from every integer and the
    // rounding errors are below 1e-3 -- full-rate instructions instead of quarter-rate v_mul_lo_u32
    const float rcpr = __builtin_amdgcn_rcpf((float)b.cpr);
#else
    const uint32_t magic = (uint32_t)b.magic;  // exact floor(ch / cpr) for ch < 16k, cpr <= 64
#endif
    M.valid = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const uint32_t ch = ch0 + tid + q * NT;
#if V1C_CHUNK_MAP_FP32
        const uint32_t r = (uint32_t)(((float)ch + 0.5f) * rcpr), col = ch - __umul24(r, (uint32_t)b.cpr);
#else
        const uint32_t r = (ch * magic) >> 20, col = ch - r * b.cpr;
#endif
        M.row[q] = b.y0 + r;
        M.xbyte[q] = (uint32_t)(b.x0 + 4 * col) * 3u;
        M.lds_dw[q] = __umul24(r, (uint32_t)lpw) + col * 4;
        M.valid |= ch < (uint32_t)nchunks ? 1u << q : 0u;
    }
}

// ---- issue the box loads (4 source pixels = 12 bytes per chunk) ----
// TAIL: the box may reach past the last byte of the image (decided per tile, wave-uniform)
// ZERO: clear the slots without a chunk (the batch loop's register prefetch schedules better with
// defined values: measured 9 % on C3; the pair path saves the 24 moves)
template <bool TAIL, bool ZERO, int NQ = 4>
__device__ __forceinline__ void stage_load(const ChunkMap& M, const uint8_t* __restrict__ src, uint32_t spitch, uint32_t src_bytes,
                                           Staged& S)
{
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (ZERO)
            S.w0[q] = S.w1[q] = S.w2[q] = 0;
        if (M.valid & (1u << q)) {
            const uint32_t goff = __umul24(M.row[q], spitch) + M.xbyte[q];
            if (!TAIL || goff + 12u <= src_bytes) {
                struct u96 {
                    uint32_t a, b, c;
                };
                typedef u96 __attribute__((aligned(4), may_alias)) u96a4;
                const u96 v = *(const u96a4*)(src + goff);
                S.w0[q] = v.a, S.w1[q] = v.b, S.w2[q] = v.c;
            } else {  // last bytes of the image: never read past the allocation
                uint32_t w[3] = {0, 0, 0};
#pragma unroll 1
                for (int bb = 0; bb < 12; bb++)
                    if (goff + bb < src_bytes)
                        w[bb >> 2] |= (uint32_t)src[goff + bb] << (8 * (bb & 3));
                S.w0[q] = w[0], S.w1[q] = w[1], S.w2[q] = w[2];
            }
        }
    }
}

// ---- the same for a box that leaves the source (bicubic / Lanczos4): the border around the image ----
// remapBicubic / remapLanczos4 give a tap outside the source the border value (the constant-border branch: oracle/vr180_oracle.c:415-449,
// sample_table in v1c_core.hpp).  A box staged with that colour in every cell outside the image lets the K x K gather serve such
// footprints like any other: where the image circle touches the frame (radius = "max", what "auto" finds for a full-frame circle: the
// reference's defaults, remapper.py:333,416) whole tiles next to the poles have every footprint cross the edge, and the per-pixel
// patch path they took was a tail of ~30 waves that doubled the launch (C1L 0.189 against 0.092 ms with the circle 0.5 % inside).
// Chunks are 4 pixels starting at a multiple of 4 columns: they straddle the right edge only (src_w not a multiple of 4).
// (every border mode but TRANSPARENT, whose bicubic / Lanczos4 rule -- skip the pixel when its CENTRE tap is outside, reflect the other
//  taps -- stays with the per-pixel sampler: REPLICATE / REFLECT / WRAP / REFLECT_101 stage, cell by cell, the pixel borderInterpolate
//  maps the cell to -- exactly what sample_table's border branch reads for that tap)
__device__ __forceinline__ int kxk_ext(geom_cref g)
{
    return g.border != V1C_BORDER_TRANSPARENT ? 1 : 0;
}
__device__ __forceinline__ bool box_leaves_source(const TileBox& b, geom_cref g)
{
    return (b.x0 < 0) | (b.y0 < 0) | (b.x0 + 4 * b.cpr > g.src_w) | (b.y0 + b.nrows > g.src_h);
}
template <bool ZERO, int NQ = 4>
__device__ __forceinline__ void stage_load_ext(const ChunkMap& M, const uint8_t* __restrict__ src, uint32_t spitch, geom_cref g, Staged& S)
{
    const uint32_t c0 = g.cval[0], c1 = g.cval[1], c2 = g.cval[2];
    const uint32_t p0 = c0 | (c1 << 8) | (c2 << 16) | (c0 << 24), p1 = c1 | (c2 << 8) | (c0 << 16) | (c1 << 24),
                   p2 = c2 | (c0 << 8) | (c1 << 16) | (c2 << 24);  // four border pixels
    const int wb = g.src_w * 3;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (ZERO)
            S.w0[q] = S.w1[q] = S.w2[q] = 0;
        if (M.valid & (1u << q)) {
            const int row = (int)M.row[q], xb = (int)M.xbyte[q];  // (signed: the box may start left of / above the image)
            const bool row_in = (unsigned)row < (unsigned)g.src_h;
            if (row_in & (xb >= 0) & (xb + 12 <= wb)) {
                struct u96 {
                    uint32_t a, b, c;
                };
                typedef u96 __attribute__((aligned(4), may_alias)) u96a4;
                const u96 v = *(const u96a4*)(src + (__umul24((uint32_t)row, spitch) + (uint32_t)xb));
                S.w0[q] = v.a, S.w1[q] = v.b, S.w2[q] = v.c;
            } else if ((g.border == V1C_BORDER_CONSTANT) & row_in & (xb >= 0) & (xb < wb)) {  // the chunk straddles the right edge: pixel by pixel
                uint32_t w[3] = {p0, p1, p2};
                const uint32_t goff = __umul24((uint32_t)row, spitch) + (uint32_t)xb;
#pragma unroll 1
                for (int bb = 0; bb < 12; bb++)
                    if (xb + bb < wb)
                        w[bb >> 2] = (w[bb >> 2] & ~(0xffu << (8 * (bb & 3)))) | ((uint32_t)src[goff + bb] << (8 * (bb & 3)));
                S.w0[q] = w[0], S.w1[q] = w[1], S.w2[q] = w[2];
            } else if (g.border != V1C_BORDER_CONSTANT) {  // REPLICATE / REFLECT / WRAP / REFLECT_101: the pixel each cell maps to
                const int yi = border_index(row, g.src_h, g.border);
                uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll 1
                for (int px = 0; px < 4; px++) {
                    const int xi = border_index(xb / 3 + px, g.src_w, g.border);
                    const uint8_t* s3 = src + (__umul24((uint32_t)yi, spitch) + (uint32_t)xi * 3u);
                    for (int ch = 0; ch < 3; ch++) {
                        const int bb = 3 * px + ch;
                        w[bb >> 2] |= (uint32_t)s3[ch] << (8 * (bb & 3));
                    }
                }
                S.w0[q] = w[0], S.w1[q] = w[1], S.w2[q] = w[2];
            } else {
                S.w0[q] = p0, S.w1[q] = p1, S.w2[q] = p2;
            }
        }
    }
}

// ---- expand to BGRx and write the box into LDS ----
template <int NQ = 4>
__device__ __forceinline__ void stage_store(const ChunkMap& M, const Staged& S, uint32_t* boxw)
{
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        if (M.valid & (1u << q)) {
            // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3  ->  BGRx x 4
            u128 o;
            o.x = S.w0[q] & 0x00ffffffu;
            o.y = __builtin_amdgcn_perm(S.w1[q], S.w0[q], 0x0c050403u);
            o.z = __builtin_amdgcn_perm(S.w2[q], S.w1[q], 0x0c040302u);
            o.w = S.w2[q] >> 8;
            *(u128*)(boxw + M.lds_dw[q]) = o;
        }
    }
}

// ---- the same for the two eyes of a pair, interleaved per pixel: (A_i, B_i) as one 8-byte cell ----
// Both eyes sample the same box positions (one map per apply() call), so with the cells interleaved
// a tap row of BOTH eyes is one ds_read2_b64 (cells i, i + 1) instead of two ds_read2_b32: the
// scattered gather costs about the same LDS cycles per instruction either way
// (tools/ubench/lds_tap_mapping.hip: 18 vs 2 x 14), i.e. a third fewer for the pair.
// `boxw` then holds nrows x (4 cpr + 4) cells = the two buffers' dwords together.
__device__ __forceinline__ void stage_store_pair(const ChunkMap& M, const Staged& SA, const Staged& SB, uint32_t* boxw)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (M.valid & (1u << q)) {
            const uint32_t a0 = SA.w0[q] & 0x00ffffffu, a1 = __builtin_amdgcn_perm(SA.w1[q], SA.w0[q], 0x0c050403u),
                           a2 = __builtin_amdgcn_perm(SA.w2[q], SA.w1[q], 0x0c040302u), a3 = SA.w2[q] >> 8;
            const uint32_t b0 = SB.w0[q] & 0x00ffffffu, b1 = __builtin_amdgcn_perm(SB.w1[q], SB.w0[q], 0x0c050403u),
                           b2 = __builtin_amdgcn_perm(SB.w2[q], SB.w1[q], 0x0c040302u), b3 = SB.w2[q] >> 8;
            u128* dst = (u128*)(boxw + 2 * M.lds_dw[q]);
            dst[0] = u128{a0, b0, a1, b1};
            dst[1] = u128{a2, b2, a3, b3};
        }
    }
}

// ---- bilinear blend of one pixel from its two tap pairs; SEL_HI = byte index of px1 ----
// out = (sum_ij p_ij * wx_i * wy_j + 512) >> 10 with the four 10-bit products as two u16 pairs,
// scaled by 64 so that the result byte is byte 2 of the accumulator:
//   (64 * sum + 32768) >> 16 == (sum + 512) >> 10.
// Per channel 2 x v_perm_b32 (tap pair -> two zero-extended u16) + 2 x v_dot2_u32_u16; two more
// v_perm_b32 pack the three result bytes.  Identical to the two-step lerp of sample_linear
// (v1c_core.hpp).  The one product that does not fit 16 bits after scaling, 1024 * 64 (both
// fractions zero, the other three weights 0), is stored as 65535: (65535 p + 32768) >> 16 == p.
typedef unsigned short __attribute__((ext_vector_type(2))) ushort2v;

struct BlendW {
    uint32_t wa, wb;  // 64 * (wx0 * wy0, wx1 * wy0) and 64 * (wx0 * wy1, wx1 * wy1) as u16 pairs
};

__device__ __forceinline__ BlendW blend_weights(int sx, int sy)
{
    const uint32_t fq = sx & 31, fr = sy & 31;
    // u16 pair 64 * (wx0, wx1) = (2048 - 64 fq) | (64 fq) << 16 in one multiply-add; times wy0 / wy1 with
    // the packed 16-bit multipliers (src1's low half for both lanes).  The one product that does not
    // fit, 64 * 32 * 32 = 65536 (both fractions zero), saturates to 65535 (clamp) -- see blend3.
    // All 1024 fraction pairs checked against the scalar form: tools/ubench/blend_weights_pk.hip.
    const uint32_t wx64 = __umul24(fq, 0x3FFFC0u) + 2048u;
    const uint32_t frc = 32u - fr;
    BlendW w;
    asm("v_pk_mad_u16 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(w.wa) : "v"(wx64), "v"(frc));
    asm("v_pk_mul_lo_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(w.wb) : "v"(wx64), "v"(fr));
    return w;
}

template <int SEL_HI>
__device__ __forceinline__ uint32_t blend3(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi, const BlendW w)
{
    uint32_t v[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        // bytes (p0c, 0, p1c, 0) of the pixel pair
        constexpr uint32_t base = 0x0c000c00u | ((uint32_t)SEL_HI << 16);
        const uint32_t sel = base + (uint32_t)ch * 0x00010001u;
        const uint32_t pa = __builtin_amdgcn_perm(ahi, alo, sel), pb = __builtin_amdgcn_perm(bhi, blo, sel);
        v[ch] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, pa), __builtin_bit_cast(ushort2v, w.wa), 32768u, false);
        v[ch] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, pb), __builtin_bit_cast(ushort2v, w.wb), v[ch], false);
    }
    // byte 2 of each accumulator (byte 3 is zero: the sums stay below 2^24)
    const uint32_t bg = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u);
    return __builtin_amdgcn_perm(v[2], bg, 0x0c060100u);
}

// ---- K x K taps (bicubic / Lanczos4) from the BGRx box with OpenCV's int16 table ----
// remapBicubic / remapLanczos4 inlier branch: sum over the K x K footprint of S * w, then
// FixedPtCast: saturate((sum + 2^14) >> 15).  `lo` = dword index of the top-left tap.
typedef short __attribute__((ext_vector_type(2))) short2v;

// saturate_cast<uchar>(acc >> 15) of a FixedPtCast sum (acc already holds the + 2^14), for results that are packed into one dword
// by shifts and ors.  The value passes through an empty asm statement: without it the compiler fuses shift + clamp + the packing of
// two channels into v_ashr_pk_u8_i32 (new on gfx950) and treats its 16-bit result as zero-extended, while the instruction leaves the
// upper half of its destination register as it was -- BGRA bicubic / Lanczos4 came out with channels 2 and 3 OR-ed with the bits of
// a stale weight dword (found by tools/fuzz.py in round 4; tests/test_resource_budget.py keeps the instruction out of every object).
__device__ __forceinline__ uint32_t fixpt_u8(int acc)
{
    int v = min(max(acc >> 15, 0), 255);
    asm("" : "+v"(v));
    return (uint32_t)v;
}

// (not inlined: four inlined copies make the scheduler hoist all 4 x K*K tap loads -> 256 VGPRs)
// explicit address spaces: a generic pointer into a noinline function costs a flat-address null
// check (3 VALU) per tap and defeats ds_read2 / global_load selection
typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
typedef const __attribute__((address_space(1))) uint32_t* glb_u32_ptr;

template <int K, typename WPtr>
__device__ __noinline__ uint32_t blend_table(lds_u32_ptr boxw, uint32_t lo, int lpw, WPtr w)
{
    int acc0 = 1 << 14, acc1 = 1 << 14, acc2 = 1 << 14;
#pragma unroll
    for (int r = 0; r < K; r++) {
        uint32_t d[K];
#pragma unroll
        for (int q = 0; q < K; q++)
            d[q] = boxw[lo + r * lpw + q];
        uint32_t wr[K / 2];
#pragma unroll
        for (int q = 0; q < K / 2; q++)
            wr[q] = w[r * (K / 2) + q];
#pragma unroll
        for (int q = 0; q < K / 2; q++) {
            const short2v ww = __builtin_bit_cast(short2v, wr[q]);
            // (channel c of pixel 2q, 0, channel c of pixel 2q+1, 0) = two zero-extended int16
            const uint32_t p0 = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c040c00u);
            const uint32_t p1 = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c050c01u);
            const uint32_t p2 = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c060c02u);
            acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, p0), ww, acc0, false);
            acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, p1), ww, acc1, false);
            acc2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, p2), ww, acc2, false);
        }
    }
    const int o0 = min(max(acc0 >> 15, 0), 255), o1 = min(max(acc1 >> 15, 0), 255), o2 = min(max(acc2 >> 15, 0), 255);
    return (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16);
}

// The same for the two eyes of a pair: both read their taps at the same box offset with the SAME
// weights (one map per apply() call), so the weight row -- 128 B per pixel for Lanczos4, an L2
// read -- is fetched once for both.  Returns (pixel of box A) | (pixel of box B) << 32.
// The boxes are interleaved per pixel ((A_i, B_i) cells, stage_store_pair): a row of K taps of BOTH eyes is
// K / 2 ds_read2_b64 (K = 8: LDS instructions per pixel pair 64 -> 32; the Lanczos4 pair had its
// LDS 67 % busy, 59 % of that bank conflicts).  `lo` = cell index of the top-left tap.
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2v;
typedef const __attribute__((address_space(3))) u32x2v* lds_cell_ptr;

template <int K, typename WPtr>
__device__ __noinline__ uint64_t blend_table_pair(lds_cell_ptr cells, uint32_t lo, int lpw, WPtr w)
{
    int a0 = 1 << 14, a1 = 1 << 14, a2 = 1 << 14, b0 = 1 << 14, b1 = 1 << 14, b2 = 1 << 14;
#pragma unroll
    for (int r = 0; r < K; r++) {
        uint32_t da[K], db[K];
#pragma unroll
        for (int q = 0; q < K; q++) {
            const u32x2v cq = cells[lo + r * lpw + q];
            da[q] = cq.x, db[q] = cq.y;
        }
        uint32_t wr[K / 2];
#pragma unroll
        for (int q = 0; q < K / 2; q++)
            wr[q] = w[r * (K / 2) + q];
#pragma unroll
        for (int q = 0; q < K / 2; q++) {
            const short2v ww = __builtin_bit_cast(short2v, wr[q]);
            a0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(da[2 * q + 1], da[2 * q], 0x0c040c00u)), ww, a0, false);
            a1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(da[2 * q + 1], da[2 * q], 0x0c050c01u)), ww, a1, false);
            a2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(da[2 * q + 1], da[2 * q], 0x0c060c02u)), ww, a2, false);
            b0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(db[2 * q + 1], db[2 * q], 0x0c040c00u)), ww, b0, false);
            b1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(db[2 * q + 1], db[2 * q], 0x0c050c01u)), ww, b1, false);
            b2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(db[2 * q + 1], db[2 * q], 0x0c060c02u)), ww, b2, false);
        }
    }
    const uint32_t pa = (uint32_t)min(max(a0 >> 15, 0), 255) | ((uint32_t)min(max(a1 >> 15, 0), 255) << 8) | ((uint32_t)min(max(a2 >> 15, 0), 255) << 16);
    const uint32_t pb = (uint32_t)min(max(b0 >> 15, 0), 255) | ((uint32_t)min(max(b1 >> 15, 0), 255) << 8) | ((uint32_t)min(max(b2 >> 15, 0), 255) << 16);
    return (uint64_t)pa | ((uint64_t)pb << 32);
}

// Border-aware K x K sampler for the pixels whose footprint leaves the source (same arithmetic as
// sample_table<3, K> in v1c_core.hpp; the row loop rolled, a row's taps together: see inside.  The
// callee's VGPRs count against the kernel -- 117 for K = 8, four waves per SIMD.  With a row's taps in two round trips of four the
// Lanczos4 pair kernels need 88 and run five: C2L +3.3 %, P1L +3 %, C4 +0.6 %, C1L +-0 -- a fifth wave only adds to the LDS queue
// (profiles/r05g_prio/ab_lanczos4_five_waves_per_simd.log).)
template <int K>
// (border mode and the packed BGR border value as plain scalars: a Geom passed by value to a
// non-inlined function had its byte members mis-read -- cval[2] came back as 63)
__device__ __noinline__ uint32_t slow_pixel_table3_t(const uint8_t* src, int64_t pitch, int h, int w, int border, uint32_t cval_bgr,
                                                     const short* itab, float x, float y)
{
    const int cv0 = (int)(cval_bgr & 255u), cv1 = (int)((cval_bgr >> 8) & 255u), cv2 = (int)((cval_bgr >> 16) & 255u);
    const Taps t = quantize(x, y);
    const short* __restrict__ wt = itab + (size_t)(t.fy * 32 + t.fx) * (K * K);
    constexpr int off = K / 2 - 1;
    const int sx = t.ix - off, sy = t.iy - off;
    if (border == V1C_BORDER_TRANSPARENT) {
        // remapBicubic / remapLanczos4: a pixel whose centre tap lies outside the source keeps the destination's bytes (bit 24 of the
        // result, as slow_pixel_linear3_t reports it); the others take their missing taps by BORDER_REFLECT_101
        if ((unsigned)t.ix >= (unsigned)w || (unsigned)t.iy >= (unsigned)h)
            return 1u << 24;
        border = V1C_BORDER_REFLECT_101;
    }
    if (border == V1C_BORDER_CONSTANT && (sx >= w || sx + K <= 0 || sy >= h || sy + K <= 0))  // footprint entirely outside
        return cval_bgr & 0xffffffu;
    // The column indices first (borderInterpolate's loops and switches), then row by row K taps' loads with no control flow between
    // them: one memory round trip per ROW.  With index and load of every tap in one rolled loop each tap waited for its own load -- 64
    // round trips per pixel, 512 per wave of a tile whose every footprint crosses the border: where the image circle touches the frame
    // (radius = "max", what "auto" finds for a full-frame circle: the reference's defaults) a handful of such waves next to the poles ran
    // 170 us after the rest of the launch had finished, and the 2 048² Lanczos4 pair took 0.285 ms instead of 0.087
    // (profiles/r04d_final/kxk_border_tiles.log).
    typedef const __attribute__((address_space(1))) uint8_t* g8;
    typedef const __attribute__((address_space(1))) short* g16;
    int xo[K];  // byte offset of tap j in a row; -1: outside under BORDER_CONSTANT
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int xj = border_index(sx + j, w, border);
        xo[j] = xj < 0 ? -1 : xj * 3;
    }
    const g16 wg = (g16)wt;
    int a0 = 1 << 14, a1 = 1 << 14, a2 = 1 << 14;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        const int yi = border_index(sy + i, h, border);  // -1: outside under BORDER_CONSTANT
        const g8 S = (g8)src + (int64_t)(yi < 0 ? 0 : yi) * pitch;
        uint8_t b0[K], b1[K], b2[K];
        short wv[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const g8 p = S + (xo[j] < 0 ? 0 : xo[j]);
            b0[j] = p[0], b1[j] = p[1], b2[j] = p[2];
            wv[j] = wg[i * K + j];
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            const bool in = (yi >= 0) & (xo[j] >= 0);
            a0 += (in ? (int)b0[j] : cv0) * (int)wv[j];
            a1 += (in ? (int)b1[j] : cv1) * (int)wv[j];
            a2 += (in ? (int)b2[j] : cv2) * (int)wv[j];
        }
    }
    const int o0 = min(max(a0 >> 15, 0), 255), o1 = min(max(a1 >> 15, 0), 255), o2 = min(max(a2 >> 15, 0), 255);
    return (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16);
}

// `aligned`: the lane's 12 bytes start on a dword boundary.  x0 * 3 is a multiple of 12, so this is
// a property of the unit (dst and its pitch), wave-uniform -- see dst_rows_dword_aligned().
// SYS = 1 (the mirror pair kernels): the streaming stores at SYSTEM scope (`sc0 sc1 nt`: written through to memory, nothing
// kept in L2).  With plain `nt` a 128-byte line that two workgroups write half each leaves L2 twice in part (1.10 x the bytes of
// the image: HISTORY.md 4.4c); at system scope a C2 launch writes 99 312 KB for its 98 304 KB of output and runs 2 - 4 % faster
// (profiles/r03d_final/ab_store_policy.log).  The batch, rotation and Lanczos4 launches measured equal or slower with it (C4 +3.8 %)
// and keep `nt`.  No builtin selects that policy without the waits of a volatile access, so the store is written by hand; the
// `s_nop 1` inside the statement is the two wait states gfx940+ needs between a store of more than 8 bytes and a VALU write of its
// data registers -- a hazard the compiler cannot see through an asm statement.
template <int SYS = 0>
__device__ __forceinline__ void store4(uint8_t* drow, const uint32_t (&pix)[kPX], unsigned ok, bool aligned)
{
    if (SYS && ok == 0xFu && aligned) {
        typedef uint32_t __attribute__((ext_vector_type(3))) u32x3;
        const u32x3 v = {pix[0] | (pix[1] << 24), (pix[1] >> 8) | (pix[2] << 16), (pix[2] >> 16) | (pix[3] << 8)};
        asm volatile("global_store_dwordx3 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(drow), "v"(v) : "memory");
    } else if (ok == 0xFu && aligned) {
        uint32_t* d32 = (uint32_t*)drow;
        // non-temporal: the destination is written once and never read by this launch; with the stores marked
        // streaming the L2 / Infinity Cache keep the source halo rows instead (L3-cold bench, r02: C2 -4 %, C1 -5 %,
        // C5 -3.5 %; no difference when the destination was L3-resident from the previous step)
        __builtin_nontemporal_store(pix[0] | (pix[1] << 24), d32 + 0);
        __builtin_nontemporal_store((pix[1] >> 8) | (pix[2] << 16), d32 + 1);
        __builtin_nontemporal_store((pix[2] >> 16) | (pix[3] << 8), d32 + 2);
    } else {
#pragma unroll
        for (int k = 0; k < kPX; k++)
            if (ok & (1u << k)) {
                drow[3 * k + 0] = (uint8_t)pix[k];
                drow[3 * k + 1] = (uint8_t)(pix[k] >> 8);
                drow[3 * k + 2] = (uint8_t)(pix[k] >> 16);
            }
    }
}

struct TileIds {
    int x0, j, xc, jc, npx, flag_tile, flag_stride, box_tile;
    bool active;
};

// tile (tx, ty) of a grid of tiles_x columns; the tile is 64 px wide and `th` = threads/16 rows high
// `g`: anything with dst_w / dst_h -- the plan's geometry or the argument block's copy of the two
template <typename Dst>
__device__ __forceinline__ TileIds tile_ids(const Dst& g, int z, int tid, int tx, int ty, int tiles_x, int th)
{
    TileIds t;
    const int lx = tid % kLanesX, ly = tid / kLanesX;
    t.x0 = (tx * kLanesX + lx) * kPX;
    t.j = ty * th + ly;
    t.active = (t.x0 < g.dst_w) & (t.j < g.dst_h);
    t.xc = min(t.x0, ((g.dst_w + 3) & ~3) - 4), t.jc = min(t.j, g.dst_h - 1);  // clamped for table reads
    t.npx = t.active ? min(kPX, g.dst_w - t.x0) : 0;
    // flag words are indexed like kernels.hip's 256x4 tiles so that MODE_FIXUP finds them
    const int ftx = (g.dst_w + kBlockX * kPX - 1) / (kBlockX * kPX), fty = (g.dst_h + kBlockY - 1) / kBlockY;
    t.flag_tile = (z * fty + t.jc / kBlockY) * ftx + t.xc / (kBlockX * kPX);
    t.flag_stride = fty * ftx;  // flag words per unit
    t.box_tile = ty * tiles_x + tx;
    return t;
}

__device__ __forceinline__ bool box_touches_image_end(const TileBox& b, geom_cref g)
{
    // only the last chunk(s) of the image's last row can reach past the allocation
    return (b.y0 + b.nrows >= g.src_h) & ((b.x0 + 4 * b.cpr) > g.src_w);
}

// ---- the 2x2 taps of a lane's 4 pixels from a staged BGRx box (bilinear) ----
struct Taps2 {
    uint32_t alo[kPX], ahi[kPX], blo[kPX], bhi[kPX];
};

template <bool ALL_IN = false>
__device__ __forceinline__ void read_taps_lds(const LaneCoords& L, const TileBox& b, const uint32_t* boxw, Taps2& T)
{
    const int lpw = b.cpr * 4 + 4;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
        const bool in = ALL_IN || ((L.inside >> k) & 1);
        const uint32_t lo = in ? __umul24(iy - b.y0, lpw) + (uint32_t)(ix - b.x0) : 0u;
        T.alo[k] = boxw[lo], T.ahi[k] = boxw[lo + 1];
        T.blo[k] = boxw[lo + lpw], T.bhi[k] = boxw[lo + lpw + 1];
    }
}

__device__ __forceinline__ void blend_taps(const Taps2& T, const LaneCoords& L, uint32_t (&pix)[kPX])
{
#pragma unroll
    for (int k = 0; k < kPX; k++)
        pix[k] = blend3<4>(T.alo[k], T.ahi[k], T.blo[k], T.bhi[k], blend_weights(L.sx[k], L.sy[k]));
}

// interior tile: every lane is active with 4 valid pixels -- three dword stores when the row
// pointer is dword-aligned (wave-uniform per unit: dst and its pitch), bytes otherwise
__device__ __forceinline__ bool dst_rows_dword_aligned(units_cptr U, int z)
{
    return ((((uintptr_t)U[z].dst) | (uintptr_t)U[z].dst_pitch) & 3u) == 0;
}

// first byte of the lane's 4 pixels: 32-bit offset (the host checks dst_h * dst_pitch < 2^32 and
// dst_pitch < 2^24 before it selects these kernels)
__device__ __forceinline__ uint8_t* dst_ptr(units_cptr U, int z, const TileIds& t)
{
    return U[z].dst + (__umul24((uint32_t)t.j, (uint32_t)U[z].dst_pitch) + (uint32_t)t.x0 * 3u);
}

template <int SYS = 0>
__device__ __forceinline__ void store_interior(units_cptr U, int z, const TileIds& t, const uint32_t (&pix)[kPX])
{
    store4<SYS>(dst_ptr(U, z, t), pix, 0xFu, dst_rows_dword_aligned(U, z));
}

// ---- slow-path patch (pixels with valid coordinates the tiled path did not produce) and store ----
template <int K, int NN = 0>
__device__ __forceinline__ void patch_and_store(ctx_cref c, units_cptr U, int z, const TileIds& t, const LaneCoords& L,
                                                uint32_t (&pix)[kPX], unsigned done, const uint8_t* __restrict__ src)
{
    geom_cref g = c.g;
    const unsigned slow = L.ok & ~done;
    unsigned skip = 0;  // BORDER_TRANSPARENT: pixels the sampler leaves untouched (bilinear: 2 x 2 footprint not inside; K x K: centre tap outside)
    // (Under BORDER_CONSTANT a valid K x K pixel that is not `inside` -- lane_coords' ext: the footprint overlaps the source -- has its whole
    //  footprint outside and is the border value; giving it that value here instead of through the sampler's early return was measured:
    //  C4 +0.7 %, C2C +1.5 % (profiles/r05c_mid/ab_far_pixels_border_value.log) -- the selects cost every tile more than the calls cost the few.)
    if (slow) {
        if (K == 2) {
            // one inlined copy in a rolled loop (a call would pin every live value above the 40
            // caller-saved VGPRs)
#pragma unroll 1
            for (int k = 0; k < kPX; k++) {
                if (slow & (1u << k)) {
                    const int fsx = k == 0 ? L.sx[0] : k == 1 ? L.sx[1] : k == 2 ? L.sx[2] : L.sx[3];
                    const int fsy = k == 0 ? L.sy[0] : k == 1 ? L.sy[1] : k == 2 ? L.sy[2] : L.sy[3];
                    uint32_t r;
                    if constexpr (NN)
                        r = slow_pixel_nearest3_t(src, U[z].src_pitch, g.src_h, g.src_w, g, fsx, fsy);
                    else
                        r = slow_pixel_linear3_t(src, U[z].src_pitch, g.src_h, g.src_w, g, fsx, fsy);
                    skip |= (r >> 24) << k;
#pragma unroll
                    for (int q = 0; q < kPX; q++)
                        pix[q] = q == k ? (r & 0xffffffu) : pix[q];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPX; k++)
                if (slow & (1u << k))
                    // (float)sx / 32 re-quantises to sx while |sx| < 2^24; beyond that the footprint is
                    // outside the source either way (saturated short coordinates)
                {
                    const uint32_t r = slow_pixel_table3_t<K>(src, U[z].src_pitch, g.src_h, g.src_w, g.border,
                                                              (uint32_t)g.cval[0] | ((uint32_t)g.cval[1] << 8) | ((uint32_t)g.cval[2] << 16), c.itab,
                                                              (float)L.sx[k] * 0.03125f, (float)L.sy[k] * 0.03125f);
                    skip |= (r >> 24) << k;  // (BORDER_TRANSPARENT: centre tap outside the source)
                    pix[k] = r & 0xffffffu;
                }
        }
    }
    if (!t.active)
        return;
    store4(dst_ptr(U, z, t), pix, L.ok & ~skip, dst_rows_dword_aligned(U, z));
}

// ---- taps, blend, slow-path patch and store: shared tail of the kernels ----
// `wtab` = OpenCV's int16 weight table for K = 4 / 8 (global memory, or LDS in the persistent kernel)
template <int K, int NN = 0, typename WPtr>
__device__ __forceinline__ void sample_and_store(ctx_cref c, units_cptr U, int z, const TileIds& t, const LaneCoords& L,
                                                 const TileBox& b, bool use_lds, const uint32_t* boxw, WPtr wtab,
                                                 const uint8_t* __restrict__ src, uint32_t spitch, bool all_in = false)
{
    uint32_t pix[kPX];
    unsigned done = 0;  // pixels produced by the tiled path
    if (use_lds) {
        const int lpw = b.cpr * 4 + 4;
        if (K == 2) {
            Taps2 T;
            if (all_in)  // interior tile: no tap predication
                read_taps_lds<true>(L, b, boxw, T);
            else
                read_taps_lds(L, b, boxw, T);
            blend_taps(T, L, pix);
        } else {
            constexpr int off = K / 2 - 1;
#pragma unroll
            for (int k = 0; k < kPX; k++) {
                const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
                const bool in = (L.inside >> k) & 1;
                const uint32_t lo = in ? __umul24(iy - off - b.y0, lpw) + (uint32_t)(ix - off - b.x0) : 0u;
                const uint32_t a = (uint32_t)((L.sy[k] & 31) * 32 + (L.sx[k] & 31));
                pix[k] = blend_table<K>((lds_u32_ptr)boxw, lo, lpw, wtab + a * (K * K / 2));
            }
        }
        done = L.inside;
    } else if (K == 2) {
        uint32_t alo[kPX], ahi[kPX], blo[kPX], bhi[kPX];
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
            const bool in = (L.inside >> k) & 1;
            const uint32_t off = in ? __umul24(iy, spitch) + (uint32_t)(ix * 3) : 0u;
            const u64pair a = load_u64_unaligned(src + off);
            const u64pair bq = load_u64_unaligned(src + off + spitch);
            alo[k] = a.lo, ahi[k] = a.hi, blo[k] = bq.lo, bhi[k] = bq.hi;
        }
#pragma unroll
        for (int k = 0; k < kPX; k++)
            pix[k] = blend3<3>(alo[k], ahi[k], blo[k], bhi[k], blend_weights(L.sx[k], L.sy[k]));
        done = L.inside;
    }

    patch_and_store<K, NN>(c, U, z, t, L, pix, done, src);
}

// ---- K x K taps (bicubic / Lanczos4) of the two eyes of a pair from interleaved cells, against ONE fetch of each pixel's weight rows ----
// `cells`: the (A_i, B_i) cells of the box `b` (stage_store_pair); `xch`: kKxkExchangeBytes of LDS behind them.
template <int K, typename WPtr>
__device__ __forceinline__ void kxk_pair_gather(int tid, const LaneCoords& L, const TileBox& b, uint32_t* cells, uint32_t* xch, WPtr wtab,
                                                uint32_t (&pa)[kPX], uint32_t (&pb)[kPX])
{
    uint32_t* const boxw = cells;
    constexpr int off = K / 2 - 1;
    const int lpw = b.cpr * 4 + 4;
    if (kLanesX == 16 && !V1C_KXK_OWN_LANES) {
        // The gather runs in ANOTHER lane -> pixel mapping than the coordinates and the stores: slot k of lane l
        // samples column 16 k + (l & 15) of the lane's tile row, so that the 16 lanes LDS serves together read
        // ADJACENT cells (with 4 adjacent pixels per lane they read every 4th cell: 4-way bank conflicts at best;
        // the Lanczos4 pair had its LDS 67 % busy, 59 % of that conflicts -- tools/ubench/lanczos_pair_forms.hip:
        // sampler alone 1.29 - 1.93 -> 0.96 - 1.29 ms at C4's size).  Tap origin and weight entry travel as one
        // dword through a wave-private KB of LDS behind the boxes, the two result pixels come back the same way
        // (a wave's LDS operations execute in order: no barrier, only compiler fences).
        const int lane = tid & 63;
        uint32_t* xw = xch + (tid >> 6) * 256;  // [row of the wave][column of the tile]
        u128 own;
        uint32_t pk[kPX];
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
            const bool in = (L.inside >> k) & 1;
            const uint32_t lo = in ? __umul24(iy - off - b.y0, lpw) + (uint32_t)(ix - off - b.x0) : 0u;  // < 2^14 cells
            pk[k] = (lo << 10) | (uint32_t)((L.sy[k] & 31) * 32 + (L.sx[k] & 31));
        }
        own.x = pk[0], own.y = pk[1], own.z = pk[2], own.w = pk[3];
        ((u128*)xw)[lane] = own;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t* col = xw + (lane >> 4) * 64 + (lane & 15);
        uint32_t gk[kPX];
#pragma unroll
        for (int k = 0; k < kPX; k++)
            gk[k] = col[16 * k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (requesting the weight row of slot k + 1 before slot k is blended -- 32 more VGPRs, the blend inlined in a
        // rolled loop -- measured 5 % SLOWER on C4 than these four calls)
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const uint64_t pp = blend_table_pair<K>((lds_cell_ptr)boxw, gk[k] >> 10, lpw, wtab + (gk[k] & 1023u) * (K * K / 2));
            pa[k] = (uint32_t)pp, gk[k] = (uint32_t)(pp >> 32);
        }
#pragma unroll
        for (int k = 0; k < kPX; k++)
            col[16 * k] = pa[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        own = ((const u128*)xw)[lane];
        pa[0] = own.x, pa[1] = own.y, pa[2] = own.z, pa[3] = own.w;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int k = 0; k < kPX; k++)
            col[16 * k] = gk[k];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        own = ((const u128*)xw)[lane];
        pb[0] = own.x, pb[1] = own.y, pb[2] = own.z, pb[3] = own.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
            const bool in = (L.inside >> k) & 1;
            const uint32_t lo = in ? __umul24(iy - off - b.y0, lpw) + (uint32_t)(ix - off - b.x0) : 0u;
            const uint32_t a = (uint32_t)((L.sy[k] & 31) * 32 + (L.sx[k] & 31));
            const uint64_t pp = blend_table_pair<K>((lds_cell_ptr)boxw, lo, lpw, wtab + a * (K * K / 2));
            pa[k] = (uint32_t)pp, pb[k] = (uint32_t)(pp >> 32);
        }
    }
}

// ---- one tile for up to `upb` units that share the map (plan-time boxes) ----
// The units of one launch share the map (the reference computes ONE map per apply() call,
// remapper.py:381-398: both eyes of a pair, all frames of a batch), so the workgroup evaluates the
// tile's coordinates once and then serves the units one after the other.  The LDS box is double
// buffered (`boxw`, `boxw + half_dwords`; the plan sizes it from the largest tile box): the boxes
// of the first two units are requested up front and become visible with the single barrier that
// also publishes the radial-table slice; later units are prefetched into registers one iteration
// ahead and cost one barrier each.
#ifdef V1C_STAMPS
// diagnostic build only: per-phase cycle sums of every wave's lane 0, added to c.xmap[0..7] (as u64)
#define V1C_STAMP(i)                                                                                   \
    do {                                                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter();                                  \
        if ((threadIdx.x & 63) == 0)                                                                   \
            atomicAdd((unsigned long long*)c.xmap + (i), now_ - stamp_);                               \
        stamp_ = __builtin_readcyclecounter();                                                         \
    } while (0)
#else
#define V1C_STAMP(i)
#endif

// PAIR = 1: the launch has at most 2 units per workgroup (apply_lr's two eyes): only the
// straight-line path is compiled, which needs ~30 fewer VGPRs (6 waves per SIMD instead of 4).
// LEAN = 1 (k_ray_lin3_batch_lean): only the lean batch path below is compiled and tiles that are
// not eligible for it (lean_static_ok) exit at once; the host launches the general kernel on the list
// of exactly those tiles (launch_tile_k).
template <int VAR_W, int ROT, int K, int OWN, int PAIR, int NT, int LEAN, int NN = 0, typename WPtr>
__device__ __forceinline__ void shared_map_tile(args_cref a, int n_units, int upb, int zg, int tx, int ty, int tiles_x, uint32_t* boxw,
                                                int half_dwords, double* tabw, WPtr wtab)
{
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    const TileBox* __restrict__ boxes = a.boxes;
    geom_cref g = c.g;
    ray_cref P = c.ray;
    const int tid = threadIdx.x;
#ifdef V1C_STAMPS
    unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
    const int z0 = zg * upb;
    const TileIds t = tile_ids(g, z0, tid, tx, ty, tiles_x, NT / kLanesX);
    const int nu = min(upb, n_units - z0);
    // (PAIR: the first two units' pointers are read here, next to the tile box, not behind it: every dependent scalar load of the
    // prologue is a wait; the batch loop reads each unit's record where it issues its loads -- it has no scalar registers to spare)
    const int z1 = min(z0 + 1, n_units - 1);
    const uint8_t* __restrict__ usrc0 = PAIR ? U[z0].src : nullptr;
    const uint32_t upitch0 = PAIR ? (uint32_t)U[z0].src_pitch : 0u;
    const uint8_t* __restrict__ usrc1 = PAIR ? U[z1].src : nullptr;
    const uint32_t upitch1 = PAIR ? (uint32_t)U[z1].src_pitch : 0u;
    // everything the tile needs from global memory is requested up front: the boxes of the first
    // two units, the radial-table slice and the row / column table entries (one exposed latency)
    if (((V1C_PRIO & 1) && K != 2) || ((V1C_PRIO & 2) && K == 2))  // (A/B: a workgroup's load phase in front of other workgroups' arithmetic)
        __builtin_amdgcn_s_setprio(3);
    const TileBox b = load_tile_box(boxes, t.box_tile);
    const bool tail = box_touches_image_end(b, g);
    // bicubic / Lanczos4 with BORDER_CONSTANT: footprints that cross the edge of the source are served from the box too (lane_coords' `ext`;
    // the plan's boxes were computed the same way: k_tile_boxes)
    const int ext = (K != 2 && !LEAN) ? kxk_ext(g) : 0;
    const bool ext_box = K != 2 && ext && box_leaves_source(b, g);
    // lean batch path (further down): bilinear, more than two units, an interior tile whose table slice
    // and box fit -- the box is the same for all units, only the alignment of a source can differ
    // (the host launches the lean kernel only when every source is dword-aligned and every group has
    // more than two units: launch_tile_k)
    if (LEAN && !lean_static_ok(b.cpr, b.nrows, b.nidx, b.interior, half_dwords))
        return;
    ChunkMap M;
    make_chunk_map<NT>(b, tid, M);

    // The pair code stages boxes of up to 8 NT chunks, in two rounds of 4 per thread (the second one below, behind the first round's LDS
    // stores: no extra registers).  A 64 x 16 tile of an EquirectangularEncoder(is_latitude_y=False) chain has its long side along a
    // meridian: bounding boxes of up to 1 900 chunks under Lanczos4 where the lat_y configurations stay below 1 000 -- a third of
    // such a launch's tiles gathered every tap from global memory before (P3L 1.09 ms).
    // (not in the bilinear pair instantiations of the classic modes: the unrotated ones are compiled for 6 waves per SIMD -- the second
    // round's code cost them 41 - 59 spilled VGPRs -- and the boxes of their 2 x 2 footprints never come near 4 NT chunks)
#ifndef V1C_TWO_ROUNDS
#define V1C_TWO_ROUNDS 1
#endif
    // (nor in any other instantiation of the classic modes: bicubic 78 -> 105 VGPRs, 6 -> 4 waves per SIMD, C2C +18 %; Lanczos4 -- bound by
    // its callee's 117 VGPRs either way -- C1L +3 %, C4 +0.8 % on one box (profiles/r05c_mid/ab_kxk_round4_ext_only_final.log); the general
    // modes (ROT = 2) run at 4 waves anyway and are the ones whose boxes need it)
    constexpr bool kTwoRounds = V1C_TWO_ROUNDS && PAIR && !LEAN && ROT == 2;
    constexpr int kMaxChunks = kTwoRounds ? 8 * NT : 4 * NT;
    auto issue_m = [&](int z, Staged& S, const ChunkMap& Mx) -> bool {  // start the box loads of unit z; false: it must gather from global memory
        const uint8_t* __restrict__ src = !PAIR ? U[z].src : z == z0 ? usrc0 : z == z0 + 1 ? usrc1 : U[z].src;
        const uint32_t spitch = !PAIR ? (uint32_t)U[z].src_pitch : z == z0 ? upitch0 : z == z0 + 1 ? upitch1 : (uint32_t)U[z].src_pitch;
        const uint32_t src_bytes = (uint32_t)(g.src_h - 1) * spitch + (uint32_t)g.src_w * 3u;
        const bool fits = box_fits(b, src, spitch, kMaxChunks, LEAN ? 2 * half_dwords : half_dwords);
        if (fits) {
            if (K != 2 && ext_box)  // (the box leaves the source: border colour around the image)
                stage_load_ext<!PAIR>(Mx, src, spitch, g, S);
            else if (tail)
                stage_load<true, !PAIR>(Mx, src, spitch, src_bytes, S);
            else
                stage_load<false, !PAIR>(Mx, src, spitch, src_bytes, S);
        }
        return fits;
    };
    auto issue = [&](int z, Staged& S) -> bool { return issue_m(z, S, M); };

    // lean kernel: a box larger than one buffer takes both, one unit at a time
    const bool single = LEAN && b.nrows * (b.cpr * 4 + 4) > half_dwords;
    Staged S0, S1;
    const bool fit0 = issue(z0, S0);
    const bool fit1 = nu > 1 && !single ? issue(z0 + 1, S1) : false;
    const bool tab_lds = (b.nidx > 0) & (b.nidx <= kTabSlice);
    typedef double __attribute__((ext_vector_type(2))) d2;
    d2 tv = {0.0, 0.0};
    const bool mpoly = OWN == 0 && (b.interior & 2) != 0;  // slice of the polynomials in m instead
    if (tab_lds && tid < b.nidx * 4)
        tv = ((const d2*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs))[tid];
    RowCol rc;
    load_rowcol<ROT>(P, t.xc, t.jc, rc);
    V1C_STAMP(0);  // setup + issue of all loads
    const bool cells = PAIR && nu == 2 && fit0 && fit1;  // a pair: the two eyes interleaved per pixel
    if (cells) {
        stage_store_pair(M, S0, S1, boxw);
    } else {
        if (fit0)
            stage_store(M, S0, boxw);
        if (fit1)
            stage_store(M, S1, boxw + half_dwords);
    }
    if (kTwoRounds && b.nrows * b.cpr > 4 * NT) {  // the second round of a large box (wave-uniform, rare: see kMaxChunks)
        ChunkMap M2;
        make_chunk_map<NT>(b, tid, M2, 4 * NT);
        if (fit0)
            issue_m(z0, S0, M2);
        if (fit1)
            issue_m(z0 + 1, S1, M2);
        if (cells) {
            stage_store_pair(M2, S0, S1, boxw);
        } else {
            if (fit0)
                stage_store(M2, S0, boxw);
            if (fit1)
                stage_store(M2, S1, boxw + half_dwords);
        }
    }
    if (tab_lds && tid < b.nidx * 4)
        ((d2*)tabw)[tid] = tv;
    V1C_STAMP(1);  // wait for the loads + expand + LDS stores
    __syncthreads();
    if (((V1C_PRIO & 1) && K != 2) || ((V1C_PRIO & 2) && K == 2))
        __builtin_amdgcn_s_setprio(0);
    V1C_STAMP(2);  // barrier
    const bool interior = tab_lds & (b.interior != 0);  // wave-uniform: no validity / inside tests needed
    // Interior tile of a bilinear batch whose every unit can be staged (the common case by far): a
    // lean path with its own coordinate evaluation, which afterwards keeps only the per-pixel tap
    // addresses and blend weights live (not the coordinates, masks and fit flags of the general loop
    // further down).  Compiled into a kernel of its own (k_ray_lin3_batch_lean) so that its register
    // count, not the general loop's, sets the occupancy of the batch workloads.
    // Same buffer rotation as the general loop: unit v in buffer v & 1, one barrier per unit.
    if (LEAN) {
        {
            const uint32_t lpw4 = (uint32_t)(b.cpr * 4 + 4) * 4u;
            uint32_t ta[kPX];
            BlendW W[kPX];
            {
                LaneCoords L;
                if (OWN == 0 && mpoly)
                    lane_coords<VAR_W, ROT, K, 0, 1, 1, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
                else
                    lane_coords<VAR_W, ROT, K, OWN, 1, 0, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
#pragma unroll
                for (int k = 0; k < kPX; k++) {
                    ta[k] = __umul24((L.sy[k] >> 5) - b.y0, lpw4) + (uint32_t)((L.sx[k] >> 5) - b.x0) * 4u;  // byte offset in a box buffer
                    W[k] = blend_weights(L.sx[k], L.sy[k]);
                }
            }
            const lds_u32_ptr lbox = (lds_u32_ptr)boxw;
            auto load_unit = [&](int z) {  // box loads of unit z into S0
                const uint8_t* __restrict__ src = U[z].src;
                const uint32_t spitch = (uint32_t)U[z].src_pitch;
                if (tail)
                    stage_load<true, false>(M, src, spitch, (uint32_t)(g.src_h - 1) * spitch + (uint32_t)g.src_w * 3u, S0);
                else
                    stage_load<false, false>(M, src, spitch, 0u, S0);
            };
            auto sample_unit = [&](int z, uint32_t base) {  // taps from the buffer at byte offset `base`, blend, store
                uint32_t pix[kPX];
#pragma unroll
                for (int k = 0; k < kPX; k++) {
                    const lds_u32_ptr pa = (lds_u32_ptr)((const __attribute__((address_space(3))) char*)lbox + (ta[k] + base));
                    const lds_u32_ptr pb = (lds_u32_ptr)((const __attribute__((address_space(3))) char*)pa + lpw4);
                    pix[k] = blend3<4>(pa[0], pa[1], pb[0], pb[1], W[k]);
                }
                store_interior(U, z, t, pix);
            };
            if (single) {
                // the few tiles whose box needs both buffers (diagonal footprints): one unit at a time,
                // two barriers per unit, the next unit's loads in flight meanwhile
                for (int u = 0; u < nu; u++) {
                    if (u >= 1) {
                        __syncthreads();  // everyone is done with unit u - 1
                        stage_store(M, S0, boxw);
                    }
                    if (u + 1 < nu)
                        load_unit(z0 + u + 1);
                    if (u >= 1)
                        __syncthreads();
                    sample_unit(z0 + u, 0u);
                }
                return;
            }
            for (int u = 0; u < nu; u++) {
                const int z = z0 + u;
                if (u >= 1)
                    __syncthreads();
                if (u + 1 < nu && u + 1 >= 2)
                    stage_store(M, S0, boxw + ((u + 1) & 1) * half_dwords);
                if (u + 2 < nu)
                    load_unit(z + 2);
                sample_unit(z, (uint32_t)(u & 1) * (uint32_t)half_dwords * 4u);
            }
            return;
        }
    }
    if (LEAN)
        return;
    LaneCoords L;
    if (OWN == 0 && interior && mpoly)
        lane_coords<VAR_W, ROT, K, 0, 1, 1, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L, ext);
    else if (interior)
        lane_coords<VAR_W, ROT, K, OWN, 1, 0, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L, ext);
    else if (tab_lds)
        lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L, ext);
    else
        lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z0].rot, rc, t.npx, P.radial, 0, P.n_int, L, ext);
    const bool incomplete = L.ok != (1u << t.npx) - 1;
    V1C_STAMP(3);  // coordinates

    // Units 0 and 1 are in the two buffers.  Unit v >= 2 goes to buffer v & 1: its loads are issued
    // (into S0) while unit v-2 is sampled, its LDS store happens at the top of iteration v-1 behind
    // the barrier that also tells everyone is done with unit v-2, and the barrier at the top of
    // iteration v makes it visible: one barrier per unit, none at all for a pair.
    if (PAIR) {
        // a pair (apply_lr: the two eyes): both boxes are already visible, no further barrier --
        // straight-line code lets the second unit's LDS reads overlap the first unit's blend
        if (incomplete) {
            if (uint32_t* flags = a.tile_flags) {  // (null: the host proved that no fix-up pass is needed)
                flags[t.flag_tile] = 1;
                if (nu == 2)
                    flags[t.flag_tile + t.flag_stride] = 1;
            }
        }
        if (K == 2 && nu == 2 && fit0 && fit1) {
            // taps of both eyes first: the second eye's LDS latency hides behind the first eye's blend
            Taps2 T0, T1;
            uint32_t pix[kPX];
            if (interior) {  // no predication, no slow-path test, unconditional stores
                // cell address = (iy * pitch + ix) * 8 + [box - (y0 * pitch + x0) * 8]: the bracket is a scalar;
                // one ds_read2_b64 per tap row fetches (A_ix, B_ix), (A_ix+1, B_ix+1) -- interior
                // coordinates are non-negative
                typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
                typedef const __attribute__((address_space(3))) u32x2* lds_u64_ptr;
                const uint32_t lpw8 = (uint32_t)(b.cpr * 4 + 4) * 8u;
                const uint32_t base0 = (uint32_t)(uintptr_t)(lds_u32_ptr)boxw - ((uint32_t)b.y0 * lpw8 + (uint32_t)b.x0 * 8u);
#pragma unroll
                for (int k = 0; k < kPX; k++) {
                    const uint32_t rel = __umul24((uint32_t)(L.sy[k] >> 5), lpw8) + (((uint32_t)L.sx[k] >> 2) & ~7u);
                    const lds_u64_ptr ra = (lds_u64_ptr)(uintptr_t)(rel + base0), rb = (lds_u64_ptr)(uintptr_t)(rel + base0 + lpw8);
                    const u32x2 a_lo = ra[0], a_hi = ra[1], b_lo = rb[0], b_hi = rb[1];
                    T0.alo[k] = a_lo.x, T0.ahi[k] = a_hi.x, T0.blo[k] = b_lo.x, T0.bhi[k] = b_hi.x;
                    T1.alo[k] = a_lo.y, T1.ahi[k] = a_hi.y, T1.blo[k] = b_lo.y, T1.bhi[k] = b_hi.y;
                }
                V1C_STAMP(4);  // tap addresses + LDS reads issued
                blend_taps(T0, L, pix);
                store_interior(U, z0, t, pix);
                V1C_STAMP(5);  // blend + store, first eye
                blend_taps(T1, L, pix);
                store_interior(U, z0 + 1, t, pix);
                V1C_STAMP(6);  // blend + store, second eye
                return;
            }
            {  // (the same cells, taps predicated on `inside`)
                const int lpw = b.cpr * 4 + 4;
                const uint2* cellw = (const uint2*)boxw;
#pragma unroll
                for (int k = 0; k < kPX; k++) {
                    const int ix = L.sx[k] >> 5, iy = L.sy[k] >> 5;
                    const bool in = (L.inside >> k) & 1;
                    const uint32_t lo = in ? __umul24(iy - b.y0, lpw) + (uint32_t)(ix - b.x0) : 0u;
                    const uint2 a_lo = cellw[lo], a_hi = cellw[lo + 1], b_lo = cellw[lo + lpw], b_hi = cellw[lo + lpw + 1];
                    T0.alo[k] = a_lo.x, T0.ahi[k] = a_hi.x, T0.blo[k] = b_lo.x, T0.bhi[k] = b_hi.x;
                    T1.alo[k] = a_lo.y, T1.ahi[k] = a_hi.y, T1.blo[k] = b_lo.y, T1.bhi[k] = b_hi.y;
                }
            }
            blend_taps(T0, L, pix);
            patch_and_store<K, NN>(c, U, z0, t, L, pix, L.inside, U[z0].src);
            blend_taps(T1, L, pix);
            patch_and_store<K, NN>(c, U, z0 + 1, t, L, pix, L.inside, U[z0 + 1].src);
            return;
        }
        if constexpr (K != 2) {
        if (nu == 2 && fit0 && fit1) {
            // K x K taps of both eyes against one fetch of the weight row
            uint32_t pa[kPX], pb[kPX];
            kxk_pair_gather<K>(tid, L, b, boxw, boxw + 2 * half_dwords, wtab, pa, pb);
            patch_and_store<K, NN>(c, U, z0, t, L, pa, L.inside, U[z0].src);
            patch_and_store<K, NN>(c, U, z0 + 1, t, L, pb, L.inside, U[z0 + 1].src);
            return;
        }
        }
        sample_and_store<K, NN>(c, U, z0, t, L, b, fit0, boxw, wtab, U[z0].src, (uint32_t)U[z0].src_pitch);
        if (nu == 2)
            sample_and_store<K, NN>(c, U, z0 + 1, t, L, b, fit1, boxw + half_dwords, wtab, U[z0 + 1].src, (uint32_t)U[z0 + 1].src_pitch);
        return;
    }
    bool fit_cur = fit0, fit_nxt = fit1, fit_s = false;
    for (int u = 0; u < nu; u++) {
        const int z = z0 + u;
        if (incomplete)
            if (uint32_t* flags = a.tile_flags)
                flags[t.flag_tile + u * t.flag_stride] = 1;
        if (u >= 1 && nu > 2)
            __syncthreads();
        if (u + 1 < nu && u + 1 >= 2) {
            if (fit_s)
                stage_store(M, S0, boxw + ((u + 1) & 1) * half_dwords);
            fit_nxt = fit_s;
        }
        if (u + 2 < nu)
            fit_s = issue(z + 2, S0);
        sample_and_store<K, NN>(c, U, z, t, L, b, fit_cur, boxw + (u & 1) * half_dwords, wtab, U[z].src, (uint32_t)U[z].src_pitch,
                            interior);
        fit_cur = fit_nxt;
        V1C_STAMP(4 + (u & 1));  // taps + blend + store of one unit
    }
}

// XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
// consecutive ids would scatter neighbouring tiles -- which share source rows -- over all L2s.
// Give each XCD a contiguous run of tiles instead (C2: FETCH_SIZE 128 -> 61 MB per launch).
// `magic` = floor(2^32 / gridDim.x) + 1 from the host: floor(m / gridDim.x) == mulhi(m, magic) for
// every tile index (m * gridDim.x < 2^32).
// `strip_len` > 0 (host: tile_xcd_strips): an XCD's share is not one block of the image but strips of
// strip_len tiles spread over it (strip s of XCD x = global strip 8 s + x): tiles differ in cost (source
// boxes that do not fit LDS gather from global memory, single-buffered boxes, tiles left to the pair
// code) and the expensive ones cluster, so one block per XCD left some XCDs with most of them -- the
// kernel is as slow as its slowest XCD.  `strip_magic` = floor(2^32 / strip_len) + 1.
// `rows`, `row0`: the swizzled part of the grid is rows row0 .. row0 + rows - 1 of workgroups (0: all of gridDim.y); row0 *
// gridDim.x must be a multiple of 8 (the XCD of a workgroup is its linear id modulo 8)
// `gx`, `gy`: the grid (read from the launch's hidden kernel arguments by the wrapper below; the mirror kernels have them preloaded)
__device__ __forceinline__ void xcd_tile_at(unsigned magic, unsigned strip_len, unsigned strip_magic, int& tx, int& ty, unsigned rows, unsigned row0,
                                            const unsigned gx, const unsigned gy)
{
    tx = blockIdx.x, ty = blockIdx.y - row0;
#if V1C_XCD_SWIZZLE
    const unsigned ntile = gx * (rows ? rows : gy), lin = (blockIdx.y - row0) * gx + blockIdx.x;
    const unsigned per = ntile >> 3;  // tiles per XCD; the remainder keeps its natural order
    if (strip_len & 0x80000000u) {
        // block mode (gx a multiple of 8): XCD x serves tile columns [x BW, (x + 1) BW), BW = gx / 8, in blocks of BH =
        // strip_len & 0xffff tile rows, row-major inside a block -- the eight XCDs still work side by side in one band of BH tile rows,
        // but a block's halo (source rows / columns its neighbours fetch too) is its perimeter 2 (BW + BH) instead of the
        // 2 (gx + 2) of a two-row strip
        const unsigned BW = gx >> 3, BH = strip_len & 0xffffu, i = lin >> 3, x = lin & 7u;
        const unsigned blk = BW * BH, kb = i / blk, r = i - kb * blk, rr = r / BW;
        ty = (int)(kb * BH + rr);
        tx = (int)(x * BW + (r - rr * BW));
        return;
    }
    if (lin < per * 8u) {
        unsigned m = (lin & 7u) * per + (lin >> 3);
        if (strip_len) {
            const unsigned i = lin >> 3, sidx = __umulhi(i, strip_magic);  // = i / strip_len (i * strip_len < 2^32)
            // full strips: 8 per round, one per XCD; the last round's strips are the (shorter) remainder of each share
            const unsigned nfull = __umulhi(per, strip_magic), x = lin & 7u;
            m = sidx < nfull ? (sidx * 8u + x) * strip_len + (i - sidx * strip_len)
                             : nfull * 8u * strip_len + x * (per - nfull * strip_len) + (i - nfull * strip_len);
        }
        ty = gx == 1 ? (int)m : (int)__umulhi(m, magic);  // (2^32 / 1 + 1 does not fit the magic)
        tx = (int)(m - (unsigned)ty * gx);
    }
#endif
}

__device__ __forceinline__ void xcd_tile(unsigned magic, unsigned strip_len, unsigned strip_magic, int& tx, int& ty, unsigned rows = 0,
                                         unsigned row0 = 0)
{
    xcd_tile_at(magic, strip_len, strip_magic, tx, ty, rows, row0, gridDim.x, gridDim.y);
}

// ---- one tile of ONE unit that overrides the rotation (per-frame calibration): box reduced in-kernel, table from global memory ----
// `red`: 16 ints of LDS, `boxw`: kBoxBytes + 16 bytes of LDS (BGRx box)
template <int VAR_W, int ROT, int K, int OWN, int PAIR, int NN = 0, typename WPtr>
__device__ __forceinline__ void rot_unit_tile(args_cref a, int z, int btx, int bty, int* red, uint32_t* boxw, WPtr wtab)
{
    constexpr int NT = 256;
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    geom_cref g = c.g;
    ray_cref P = c.ray;
    const int tid = threadIdx.x;
    const TileIds t = tile_ids(g, z, tid, btx, bty, gridDim.x, NT / kLanesX);
    const uint8_t* __restrict__ src = U[z].src;
    const uint32_t spitch = (uint32_t)U[z].src_pitch;
    const uint32_t src_bytes = (uint32_t)(g.src_h - 1) * spitch + (uint32_t)g.src_w * 3u;
    RowCol rc;
    load_rowcol<ROT>(P, t.xc, t.jc, rc);
    LaneCoords L;
    // Fast attempt (bilinear, OWN = 0: the host proved that no pixel of these units leaves the
    // validated part of the radial table): coordinates without any validity logic, bounding box
    // of all 1024 pixels, and if that box lies inside the source the tile is "interior" --
    // unpredicated taps, unconditional stores.  Anything else (tiles cut by the destination's
    // edge, footprints leaving the source) falls through to the general code below.
    // PAIR doubles as "the m-polynomial table is valid for every unit of this launch" here.
    const bool tile_full = ((btx + 1) * kTW <= g.dst_w) & ((bty + 1) * (NT / kLanesX) <= g.dst_h);
    if (K == 2 && OWN == 0 && tile_full) {
        if (PAIR)
            lane_coords<VAR_W, ROT, K, 0, 2, 1, 0, NN>(c, U[z].rot, rc, kPX, P.radial_m, 0, P.n_int, L);
        else
            lane_coords<VAR_W, ROT, K, 0, 2, 0, 0, NN>(c, U[z].rot, rc, kPX, P.radial, 0, P.n_int, L);
        const BoxAll ba = reduce_box_all<NT / 64>(L, red, tid);
        if ((ba.xmin >= 0) & (ba.xmax < g.src_w - 2) & (ba.ymin >= 0) & (ba.ymax < g.src_h - 1)) {
            TileBox fb;
            fb.x0 = ba.xmin & ~3, fb.y0 = ba.ymin;
            fb.cpr = (ba.xmax + 2 - fb.x0 + 3) >> 2, fb.nrows = ba.ymax - ba.ymin + 2;
            fb.idx0 = fb.nidx = 0, fb.interior = 1;
            fb.magic = kChunkMagic.v[min(fb.cpr, kMaxCpr)];  // (cpr > kMaxCpr: the box does not fit, magic unused)
            if (box_fits(fb, src, spitch, 4 * NT, kBoxBytes / 4)) {
                ChunkMap M;
                Staged S;
                // (this kernel is VALU-bound: boxes of at most two chunks per thread -- nearly all --
                // skip the other two slots with a scalar branch)
                if (fb.nrows * fb.cpr <= 2 * NT && !box_touches_image_end(fb, g)) {
                    make_chunk_map<NT, 2>(fb, tid, M);
                    stage_load<false, false, 2>(M, src, spitch, src_bytes, S);
                    stage_store<2>(M, S, boxw);
                } else {
                    make_chunk_map<NT>(fb, tid, M);
                    if (box_touches_image_end(fb, g))
                        stage_load<true, false>(M, src, spitch, src_bytes, S);
                    else
                        stage_load<false, false>(M, src, spitch, src_bytes, S);
                    stage_store(M, S, boxw);
                }
                __syncthreads();
                Taps2 T;
                uint32_t pix[kPX];
                read_taps_lds<true>(L, fb, boxw, T);
                blend_taps(T, L, pix);
                store_interior(U, z, t, pix);
            } else {  // box too large for LDS (strong minification): gather from global memory
                sample_and_store<K, NN>(c, U, z, t, L, fb, false, boxw, wtab, src, spitch);
            }
            return;
        }
        __syncthreads();  // `red` is reused below
    }
    const int ext = K != 2 ? kxk_ext(g) : 0;
    lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z].rot, rc, t.npx, P.radial, 0, P.n_int, L, ext);
    const TileBox b = reduce_box<K, NT / 64>(L, red, tid);
    const bool use_lds = box_fits(b, src, spitch, 4 * NT, kBoxBytes / 4);
    if (use_lds) {
        ChunkMap M;
        make_chunk_map<NT>(b, tid, M);
        Staged S;
        if (K != 2 && ext && box_leaves_source(b, g))
            stage_load_ext<true>(M, src, spitch, g, S);
        else
            stage_load<true, true>(M, src, spitch, src_bytes, S);
        stage_store(M, S, boxw);
    }
    __syncthreads();
    if (L.ok != (1u << t.npx) - 1)
        if (uint32_t* flags = a.tile_flags)
            flags[t.flag_tile] = 1;
    sample_and_store<K, NN>(c, U, z, t, L, b, use_lds, boxw, wtab, src, spitch);
}

// ---- one tile of TWO units that carry the same rotation, without plan-time boxes (bicubic / Lanczos4): v1c_plan_run_auto ----
// radius = "auto" on the device runs the kernels that reduce their boxes themselves; its units share the map (one transformer, one
// radius: remapper.py:474-484), so the two eyes of a pair take ONE evaluation of the coordinates, one box geometry, and -- as in the
// pair path of shared_map_tile -- one fetch of every pixel's weight rows for both (interleaved cells, kxk_pair_gather).
// `red`: 16 ints; `boxw`: 2 kBoxBytes + 16 bytes of cells followed by kKxkExchangeBytes.  A pair whose boxes do not fit 24 KB per eye
// (strong minification: rare) samples every pixel through the border-aware per-pixel sampler.
template <int VAR_W, int K, int OWN, int NN = 0, typename WPtr>
__device__ __forceinline__ void rot_shared_pair_tile(args_cref a, int zA, int zB, int btx, int bty, int* red, uint32_t* boxw, WPtr wtab)
{
    constexpr int NT = 256;
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    geom_cref g = c.g;
    ray_cref P = c.ray;
    const int tid = threadIdx.x;
    const TileIds t = tile_ids(g, zA, tid, btx, bty, gridDim.x, NT / kLanesX);
    RowCol rc;
    load_rowcol<1>(P, t.xc, t.jc, rc);
    const int ext = kxk_ext(g);
    LaneCoords L;
    lane_coords<VAR_W, 1, K, OWN, 0, 0, 0, NN>(c, U[zA].rot, rc, t.npx, P.radial, 0, P.n_int, L, ext);
    const TileBox b = reduce_box<K, NT / 64>(L, red, tid);
    const uint8_t* __restrict__ srcA = U[zA].src;
    const uint8_t* __restrict__ srcB = U[zB].src;
    const uint32_t pitchA = (uint32_t)U[zA].src_pitch, pitchB = (uint32_t)U[zB].src_pitch;
    const bool fits = box_fits(b, srcA, pitchA, 4 * NT, kBoxBytes / 4) & box_fits(b, srcB, pitchB, 4 * NT, kBoxBytes / 4);
    if (!fits) {  // (wave-uniform)
        uint32_t px[kPX] = {0, 0, 0, 0};
        patch_and_store<K, NN>(c, U, zA, t, L, px, 0u, srcA);
        patch_and_store<K, NN>(c, U, zB, t, L, px, 0u, srcB);
        return;
    }
    ChunkMap M;
    make_chunk_map<NT>(b, tid, M);
    Staged SA, SB;
    if (ext && box_leaves_source(b, g)) {
        stage_load_ext<true>(M, srcA, pitchA, g, SA);
        stage_load_ext<true>(M, srcB, pitchB, g, SB);
    } else {
        stage_load<true, true>(M, srcA, pitchA, (uint32_t)(g.src_h - 1) * pitchA + (uint32_t)g.src_w * 3u, SA);
        stage_load<true, true>(M, srcB, pitchB, (uint32_t)(g.src_h - 1) * pitchB + (uint32_t)g.src_w * 3u, SB);
    }
    stage_store_pair(M, SA, SB, boxw);
    __syncthreads();
    if (L.ok != (1u << t.npx) - 1)
        if (uint32_t* flags = a.tile_flags) {
            flags[t.flag_tile] = 1;
            flags[t.flag_tile + (zB - zA) * t.flag_stride] = 1;
        }
    uint32_t pa[kPX], pb[kPX];
    kxk_pair_gather<K>(tid, L, b, boxw, boxw + 2 * (kBoxBytes / 4) + 4, wtab, pa, pb);
    patch_and_store<K, NN>(c, U, zA, t, L, pa, L.inside, srcA);
    patch_and_store<K, NN>(c, U, zB, t, L, pb, L.inside, srcB);
}

// ---- a pair (apply_lr) of an unrotated chain: a tile AND its mirror image about the equator per workgroup ----
// Rows j and mirror_h - j of an unrotated equirectangular chain differ only in the sign of sin(lat): m, the radial
// factor G and the x coordinate are the same numbers, y mirrors about the source centre (lane_coords<..., MIRROR>).
// One workgroup therefore serves tile (tx, ty) of the upper half and the band of 16 rows that mirrors it: one
// prologue, one table slice, one evaluation of the coordinates (the largest block of the pair kernel's VALU work:
// ~130 of 362 instructions per wave) for 2 x 2 x 1024 output pixels.  The LDS box is used twice: the tile's cells
// (both eyes interleaved, as in the pair path of shared_map_tile) are sampled while the loads of the mirrored
// band's box are in flight, then that box replaces them behind a barrier.  Interior tiles only (both boxes fit,
// every pixel valid and inside -- mirror_static_ok, the predicate the host's rest list is built from); the rest
// list (tile rows 0, H/32 and the last one, which the mirrored bands do not cover, plus the tiles and mirror
// bands of ineligible workgroups) rides in grid slice z = 0 through the general pair code.
__host__ __device__ inline bool mirror_box_ok(int x0, int y0, int cpr, int nrows, int half_dwords, int src_h, int src_w)
{
    return cpr > 0 && cpr <= kMaxCpr && nrows * cpr <= 1024 && nrows * (cpr * 4 + 4) <= half_dwords &&
           !((y0 + nrows >= src_h) && (x0 + 4 * cpr > src_w));  // (boxes that reach the image's last bytes stay with the general code)
}

__host__ __device__ inline bool mirror_static_ok(const TileBox& b, const TileBox& q, int half_dwords, int src_h, int src_w)
{
    return b.interior != 0 && q.interior == b.interior && b.nidx > 0 && b.nidx <= kTabSlice && q.idx0 == b.idx0 && q.nidx == b.nidx &&
           mirror_box_ok(b.x0, b.y0, b.cpr, b.nrows, half_dwords, src_h, src_w) &&
           mirror_box_ok(q.x0, q.y0, q.cpr, q.nrows, half_dwords, src_h, src_w);
}

// ... and for k_ray_lin3_pair_mirror_raw: the box as it is in memory, rows of `upr` 16-byte units (LDS-DMA, 16 B per lane);
// a box buffer holds `nwp` wave-passes of 64 units (the plan sizes it: tile_mirror_raw_passes); the last unit of a row
// reads up to 12 bytes past the box (never past the image)
constexpr int kRawMaxWavePasses = 16;  // 16 KB per box and eye
__host__ __device__ inline int raw_units_per_row(int cpr)
{
    return (3 * cpr + 3) >> 2;
}
__host__ __device__ inline bool raw_box_ok(int x0, int y0, int cpr, int nrows, int nwp, int src_h, int src_w)
{
    const int upr = raw_units_per_row(cpr);
    return cpr > 0 && cpr <= kMaxCpr && nrows * upr <= nwp * 64 && !((y0 + nrows >= src_h) && (x0 * 3 + upr * 16 > src_w * 3));
}
// k_ray_lin3_pair_mirror_raw: 1 = all boxes of the workgroup fit their nwp KB buffers; 0 (and 2: they would fit two buffers each -- an
// eye-by-eye form of the workgroup for those was built and cost the kernel 22 VGPRs and a wave per SIMD) = the pair goes to the
// general pair code (rest list).  The band's table entries are a subset of the tile's (its rows mirror the tile's; tile row 0's band has
// one row less): the tile's slice serves both.
__host__ __device__ inline int mirror_raw_fit(const TileBox& b, const TileBox& q, int nwp, int src_h, int src_w)
{
    if (!(b.interior != 0 && q.interior == b.interior && b.nidx > 0 && b.nidx <= kTabSlice && q.nidx > 0 && q.idx0 >= b.idx0 &&
          q.idx0 + q.nidx <= b.idx0 + b.nidx))
        return 0;
    if (raw_box_ok(b.x0, b.y0, b.cpr, b.nrows, nwp, src_h, src_w) && raw_box_ok(q.x0, q.y0, q.cpr, q.nrows, nwp, src_h, src_w))
        return 1;
    if (raw_box_ok(b.x0, b.y0, b.cpr, b.nrows, 2 * nwp, src_h, src_w) && raw_box_ok(q.x0, q.y0, q.cpr, q.nrows, 2 * nwp, src_h, src_w))
        return 2;
    return 0;
}
__host__ __device__ inline bool mirror_raw_static_ok(const TileBox& b, const TileBox& q, int nwp, int src_h, int src_w)
{
    return b.interior != 0 && q.interior == b.interior && b.nidx > 0 && b.nidx <= kTabSlice && q.idx0 == b.idx0 && q.nidx == b.nidx &&
           raw_box_ok(b.x0, b.y0, b.cpr, b.nrows, nwp, src_h, src_w) && raw_box_ok(q.x0, q.y0, q.cpr, q.nrows, nwp, src_h, src_w);
}

// ---- the same workgroup with the boxes brought in by LDS-DMA, as they are in memory ----
// global_load_lds_dwordx4 copies 16 bytes per lane from any dword-aligned address straight into LDS (lane-linear: unit
// u = 256 * pass + tid at byte 16 u), so nothing of a box ever sits in a VGPR: all four boxes of the workgroup (two
// eyes x tile and mirrored band) and the table slice are requested in the prologue and the coordinates are evaluated
// WHILE they are in flight (with register staging that overlap costs the staging registers' occupancy: HISTORY.md 4.4).
// The box stays packed BGR (row pitch upr x 16 bytes); a tap pair (6 bytes at byte 3 ix) is cut out of three dwords
// read at the dword below it (ds_read2_b32 + ds_read_b32: b64 / b96 reads that are not naturally aligned are
// microcoded, 64 cycles) with two v_alignbyte_b32, then blended as the global-memory fallback does (blend3<3>).
// vmcnt counts requests in issue order, so the waits below are exact: every lane of a wave issues each of the wave's passes
// (units past the box are clamped to its last unit and land in the unused tail of the box buffer).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// A box by LDS-DMA with a lane -> unit mapping that is the same for every wave-instruction ("pass"): R = floor(64 / upr) whole
// rows per pass, lane l copies unit l % upr of row l / upr of the pass (lanes >= R * upr idle: exec-masked), so that a pass is a
// scalar source base, a scalar LDS base (M0) and one instruction -- no per-lane address arithmetic per pass (the lane-linear
// form divided every lane's unit index by upr in every pass: ~13 VALU instructions per pass, 65 per wave and tile pair on
// C2).  Pass k starts at row k R; the last one is moved up to end with the box (rows copied twice are the same bytes).
// Wave w runs passes w, w + 4, ...; returns how many (wave-uniform: the caller's vmcnt bookkeeping).
struct RawLanes {
    uint32_t row_l, col16;  // per lane: row of the pass and byte offset in the row
    int R, upr;             // wave-uniform
};

__device__ __forceinline__ RawLanes raw_lanes_upr(int upr, int lane)
{
    RawLanes m;
    m.upr = upr;
    // floor(64 / upr) and floor(l / upr) without a division: (64 + 0.5) / upr and (l + 0.5) / upr stay 0.5 / 48 away from
    // every integer, fp32's error here is < 1e-4
    const float rupr = __builtin_amdgcn_rcpf((float)m.upr);
    m.R = __builtin_amdgcn_readfirstlane((int)(64.5f * rupr));
    m.row_l = (uint32_t)(((float)lane + 0.5f) * rupr);
    m.col16 = ((uint32_t)lane - __umul24(m.row_l, (uint32_t)m.upr)) * 16u;
    return m;
}

__device__ __forceinline__ RawLanes raw_lanes(int cpr, int lane)
{
    return raw_lanes_upr(raw_units_per_row(cpr), lane);
}

// BPP: bytes per source pixel (3: packed BGR; 1 / 4: k_ray_lin_cn)
template <int BPP = 3>
__device__ __forceinline__ int raw_box_dma(const TileBox& b, const RawLanes& m, const uint8_t* __restrict__ src, uint32_t spitch, int lane,
                                            int wave, uint32_t lds_box)
{
    // request address = scalar base (source + first row of the pass + box column, all wave-uniform: scalar ALU) + the lane's 32-bit
    // offset (row in the pass x pitch + column unit; the same for every pass and every box of this geometry): the request's
    // SGPR-base form, no vector arithmetic per pass
    const uint32_t sp = (uint32_t)__builtin_amdgcn_readfirstlane((int)spitch);
    uint32_t voff;  // (v_mul_u32_u24 by hand: with a scalar factor the compiler picks the quarter-rate v_mul_lo_u32)
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(voff) : "v"(m.row_l), "s"(sp));
    voff += m.col16;
    const int rows = min(m.R, b.nrows);
    const int last0 = b.nrows - rows;  // first row of the last pass
    const bool active = lane < rows * m.upr;
    const uint64_t org = (uint64_t)(uintptr_t)src + (uint64_t)((uint32_t)b.y0 * sp + (uint32_t)b.x0 * (uint32_t)BPP);
    const uint32_t lpitch = (uint32_t)m.upr * 16u;
    int n = 0;
    for (int r0 = wave * m.R; r0 < b.nrows; r0 += 4 * m.R) {  // wave-uniform
        const int rs = min(r0, last0);
        const uint64_t base = org + (uint64_t)((uint32_t)rs * sp);
        const uint32_t m0v = lds_box + (uint32_t)rs * lpitch;
        // (by hand: as a pointer expression the compiler sums the two offsets first and addresses through a 64-bit VGPR pair --
        // a quarter-rate v_mul_lo_u32 and two 64-bit adds per pass.  M0 = LDS destination of lane 0; it is put back, the compiler
        // keeps its own value there across statements it does not know to write it.  s_nop 2: with the two moves five wait states
        // in front of the request -- what a VMEM instruction needs behind a VALU write (v_readfirstlane) of an SGPR it reads, a
        // hazard the compiler does not see inside an asm statement; it also covers the wait state M0 needs)
        uint32_t m0_saved;
        if (active)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved)
                         : "v"(voff), "s"(base), "s"(m0v)
                         : "memory");
        n++;
    }
    return n;
}

// s_waitcnt vmcnt(n) + s_barrier for a wave-uniform n; no fence: see the kernels.  The count is an immediate, so n selects one of
// 21 sixteen-byte blocks (s_waitcnt | s_barrier | s_branch end | pad) by a computed jump: 6 scalar instructions and two jumps.
// (As a switch the compiler structurised the 21 cases -- each holds a convergent barrier -- into a cascade of flag tests: ~30
// scalar instructions and ~10 taken branches per wait, as many scalar as vector instructions in the unit loop of the batch kernel.)
// n > 20 waits for vmcnt(20): a smaller count only waits longer.
#define V1C_WAIT_BLOCK(i, extra) "s_waitcnt vmcnt(" #i ")" extra "\n\ts_barrier\n\ts_branch 1f\n\ts_nop 0\n\t"
#define V1C_WAIT_TABLE(extra)                                                                                                      \
    V1C_WAIT_BLOCK(0, extra) V1C_WAIT_BLOCK(1, extra) V1C_WAIT_BLOCK(2, extra) V1C_WAIT_BLOCK(3, extra) V1C_WAIT_BLOCK(4, extra)    \
    V1C_WAIT_BLOCK(5, extra) V1C_WAIT_BLOCK(6, extra) V1C_WAIT_BLOCK(7, extra) V1C_WAIT_BLOCK(8, extra) V1C_WAIT_BLOCK(9, extra)    \
    V1C_WAIT_BLOCK(10, extra) V1C_WAIT_BLOCK(11, extra) V1C_WAIT_BLOCK(12, extra) V1C_WAIT_BLOCK(13, extra)                         \
    V1C_WAIT_BLOCK(14, extra) V1C_WAIT_BLOCK(15, extra) V1C_WAIT_BLOCK(16, extra) V1C_WAIT_BLOCK(17, extra)                         \
    V1C_WAIT_BLOCK(18, extra) V1C_WAIT_BLOCK(19, extra) V1C_WAIT_BLOCK(20, extra)
#define V1C_WAIT_JUMP(extra)                                                                                                       \
    uint32_t off_;                                                                                                                 \
    asm volatile("s_getpc_b64 vcc\n\t"          /* vcc = address of the next instruction */                                     \
                 "s_lshl_b32 %0, %1, 4\n\t"     /* 16 bytes per block ... */                                                    \
                 "s_add_u32 %0, %0, 20\n\t"     /* ... behind these five 4-byte instructions */                                  \
                 "s_add_u32 vcc_lo, vcc_lo, %0\n\t"                                                                              \
                 "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"                                                                              \
                 "s_setpc_b64 vcc\n\t" V1C_WAIT_TABLE(extra) "1:"                                                                \
                 : "=&s"(off_)                                                                                                     \
                 : "s"(__builtin_amdgcn_readfirstlane((int)min((uint32_t)max(n, 0), 20u)))                                                                            \
                 : "vcc", "scc", "memory")

__device__ __forceinline__ void wait_vm_barrier(int n)
{
    V1C_WAIT_JUMP("");
}

// ... for a count known at compile time
template <int N>
__device__ __forceinline__ void wait_vm_barrier_imm()
{
    static_assert(N >= 0 && N <= 2, "add the case");
    if (N == 0)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if (N == 1)
        asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
}

// taps of both eyes of a lane's 4 pixels from the raw boxes at LDS byte addresses `raw` (eye 0) and `raw + eye_off` (eye 1)
__device__ __forceinline__ void gather_pair_raw(const TileBox& b, uint32_t raw, uint32_t eye_off, const int (&sx)[kPX], const int (&sy)[kPX],
                                                uint32_t (&pix0)[kPX], uint32_t (&pix1)[kPX])
{
    const uint32_t pitch = (uint32_t)raw_units_per_row(b.cpr) * 16u;
    const uint32_t base0 = raw - ((uint32_t)b.y0 * pitch + (uint32_t)b.x0 * 3u);
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t ix = (uint32_t)sx[k] >> 5;
        const uint32_t a = __umul24((uint32_t)(sy[k] >> 5), pitch) + (ix * 2u + ix) + base0;  // LDS byte address of the pixel
        const uint32_t d = a & ~3u;
        const lds_u32_ptr r0 = (lds_u32_ptr)(uintptr_t)d, r1 = (lds_u32_ptr)(uintptr_t)(d + pitch);
        const lds_u32_ptr s0 = (lds_u32_ptr)(uintptr_t)(d + eye_off), s1 = (lds_u32_ptr)(uintptr_t)(d + pitch + eye_off);
        const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        const uint32_t c0 = s0[0], c1 = s0[1], c2 = s0[2], e0 = s1[0], e1 = s1[1], e2 = s1[2];
        const BlendW w = blend_weights(sx[k], sy[k]);
        pix0[k] = blend3<3>(__builtin_amdgcn_alignbyte(a1, a0, a), __builtin_amdgcn_alignbyte(a2, a1, a), __builtin_amdgcn_alignbyte(b1, b0, a),
                            __builtin_amdgcn_alignbyte(b2, b1, a), w);
        pix1[k] = blend3<3>(__builtin_amdgcn_alignbyte(c1, c0, a), __builtin_amdgcn_alignbyte(c2, c1, a), __builtin_amdgcn_alignbyte(e1, e0, a),
                            __builtin_amdgcn_alignbyte(e2, e1, a), w);
    }
}

__device__ __forceinline__ void store_pair_row(units_cptr U, const TileIds& t, int j, const uint32_t (&pix0)[kPX], const uint32_t (&pix1)[kPX])
{
    const uint32_t row_off = (uint32_t)t.x0 * 3u;
    store4<1>(U[0].dst + (__umul24((uint32_t)j, (uint32_t)U[0].dst_pitch) + row_off), pix0, 0xFu, dst_rows_dword_aligned(U, 0));
    store4<1>(U[1].dst + (__umul24((uint32_t)j, (uint32_t)U[1].dst_pitch) + row_off), pix1, 0xFu, dst_rows_dword_aligned(U, 1));
}

// taps of a lane's 4 pixels from ONE raw box at LDS byte address `raw`
__device__ __forceinline__ void gather_one_raw(const TileBox& b, uint32_t raw, const int (&sx)[kPX], const int (&sy)[kPX], uint32_t (&pix)[kPX])
{
    const uint32_t pitch = (uint32_t)raw_units_per_row(b.cpr) * 16u;
    const uint32_t base0 = raw - ((uint32_t)b.y0 * pitch + (uint32_t)b.x0 * 3u);
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t ix = (uint32_t)sx[k] >> 5;
        const uint32_t a = __umul24((uint32_t)(sy[k] >> 5), pitch) + (ix * 2u + ix) + base0;
        const uint32_t d = a & ~3u;
        const lds_u32_ptr r0 = (lds_u32_ptr)(uintptr_t)d, r1 = (lds_u32_ptr)(uintptr_t)(d + pitch);
        const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        pix[k] = blend3<3>(__builtin_amdgcn_alignbyte(a1, a0, a), __builtin_amdgcn_alignbyte(a2, a1, a), __builtin_amdgcn_alignbyte(b1, b0, a),
                           __builtin_amdgcn_alignbyte(b2, b1, a), blend_weights(sx[k], sy[k]));
    }
}

// ------------------------------------------------------------------------------------------
// host helpers shared by the launchers
// ------------------------------------------------------------------------------------------
// ---- host side: the argument block of a launch ----
static inline TileArgs tile_args(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags)
{
    TileArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ctx = cdev;
    a.units = lu.dev;
    a.tile_flags = flags;
    a.n_units = lu.n;
    a.col_s = c.ray.col_s, a.col_c = c.ray.col_c, a.col_h = c.ray.col_h;
    a.row_s = c.ray.row_s, a.row_c = c.ray.row_c, a.row_h = c.ray.row_h;
    a.dst_w = c.g.dst_w, a.dst_h = c.g.dst_h;
    if (!lu.dev)  // (lu.n <= kInlineUnits: plan.hip)
        std::memcpy(a.inl, lu.host, sizeof(DevUnit) * (size_t)std::min(lu.n, kInlineUnits));
    return a;
}

static inline bool units_dword_aligned(const LaunchUnits& lu)
{
    for (int k = 0; k < lu.n; k++)
        if (((((uintptr_t)lu.host[k].src) | (uintptr_t)lu.host[k].src_pitch) & 3u) != 0)
            return false;
    return true;
}

static inline int taps_of(int interp)
{
    // (INTER_NEAREST rides the bilinear kernels: lane_coords<..., NN = 1>)
    return (interp == V1C_INTER_LINEAR || interp == V1C_INTER_NEAREST) ? 2 : interp == V1C_INTER_CUBIC ? 4 : interp == V1C_INTER_LANCZOS4 ? 8 : 0;
}

// threads per workgroup (tile = 64 x threads/16) the plan-time boxes are computed for
static inline int tile_threads(const Geom&)
{
    return 256;
}

static inline dim3 tile_grid(const Geom& g, int nt, int nz)
{
    const int th = nt / kLanesX;
    return dim3((g.dst_w + kTW - 1) / kTW, (g.dst_h + th - 1) / th, nz);
}

}  // namespace v1c
