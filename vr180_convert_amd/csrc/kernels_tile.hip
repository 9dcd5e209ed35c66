// kernels_tile.hip -- LDS-tiled kernels for the hot configurations: BGR, every border mode but TRANSPARENT, fused ray
// path, INTER_LINEAR and (with K x K table taps) INTER_CUBIC / INTER_LANCZOS4.  Same arithmetic as
// kernels.hip: the tests compare both with the oracle bit for bit.
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
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

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
    const double* __restrict__ pq = (ROT ? P.col_c : P.col_h) + xc;
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
__device__ __forceinline__ void lane_coords(ctx_cref c, rot_cptr rot, const RowCol& rc, int npx, TabPtr tab,
                                            int tab0, int tabn, LaneCoords& L)
{
    ray_cref P = c.ray;
    geom_cref g = c.g;
    const double sl = rc.sl, cl = rc.cl, hl = rc.hl;
    const double rx32 = P.rx32, ry32 = P.ry32, cx32 = P.cx32, cy32 = P.cy32;

    double A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, B2 = 0, C0 = 0, C1 = 0, C2 = 0;
    if (ROT) {
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
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        double m;
        if (ROT) {
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
        // validity level of the entry (radial_fit.hpp): the polynomial holds on |z| <= 0.5 + level
        const double zlim = 0.5 + (double)(__double2loint(e[kRadialDegree]) & 3);
        const double zc = (double)ic + 0.5;
        unsigned own = 0;  // pixels that must use their own entry
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const double zk = tt[k] - zc;
            if (OWN) {
                const bool usec = fabs(zk) <= zlim;
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
        const bool in = K == 2 ? okk & ((unsigned)ix < (unsigned)(g.src_w - 2)) & ((unsigned)iy < (unsigned)(g.src_h - 1))
                               : okk & ((unsigned)(ix - off) < (unsigned)max(g.src_w - (K - 1), 0)) &
                                     ((unsigned)(iy - off) < (unsigned)max(g.src_h - (K - 1), 0));
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

template <int NT, int NQ = 4>
__device__ __forceinline__ void make_chunk_map(const TileBox& b, int tid, ChunkMap& M)
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
        const uint32_t ch = tid + q * NT;
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

// Border-aware K x K sampler for the rare pixel whose footprint leaves the source (same
// arithmetic as sample_table<3, K> in v1c_core.hpp, loops kept rolled: a small register
// footprint matters more than speed here because the callee's VGPRs count against the kernel).
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
    if (border == V1C_BORDER_CONSTANT && (sx >= w || sx + K <= 0 || sy >= h || sy + K <= 0))  // footprint entirely outside
        return cval_bgr & 0xffffffu;
    int a0 = 1 << 14, a1 = 1 << 14, a2 = 1 << 14;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        const int yi = border_index(sy + i, h, border);  // -1: outside under BORDER_CONSTANT
        const uint8_t* S = src + (int64_t)(yi < 0 ? 0 : yi) * pitch;
#pragma unroll 1
        for (int j = 0; j < K; j++) {
            const int xj = border_index(sx + j, w, border);
            const bool in = (yi >= 0) & (xj >= 0);
            const int wv = wt[i * K + j];
            const uint8_t* p = S + (in ? xj : 0) * 3;
            a0 += (in ? (int)p[0] : cv0) * wv;
            a1 += (in ? (int)p[1] : cv1) * wv;
            a2 += (in ? (int)p[2] : cv2) * wv;
        }
    }
    const int o0 = min(max(a0 >> 15, 0), 255), o1 = min(max(a1 >> 15, 0), 255), o2 = min(max(a2 >> 15, 0), 255);
    return (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16);
}

// `aligned`: the lane's 12 bytes start on a dword boundary.  x0 * 3 is a multiple of 12, so this is
// a property of the unit (dst and its pitch), wave-uniform -- see dst_rows_dword_aligned().
// SYS = 1 (the mirror pair kernels): the streaming stores at SYSTEM scope (`sc0 sc1 nt`: written through to memory, nothing
// kept in L2).  With plain `nt` a 128-byte line that two workgroups write half each leaves L2 twice in part (1.10 x the bytes of
// the image: DESIGN.md 4.4c); at system scope a C2 launch writes 99 312 KB for its 98 304 KB of output and runs 2 - 4 % faster
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

// Plan-time pass: the source box and the radial-table slice of every 64 x (NT/16) tile (chain
// rotation as stored in the plan).
// OWN as in the kernel that will consume the boxes: with OWN = 0 a pixel is evaluated with pixel 1's
// table entry, with OWN = 1 possibly with its own -- both within tolerance, but the box must
// bound the coordinates the consumer will actually compute.
template <int VAR_W, int ROT, int K, int NT, int OWN, int NN = 0>
// `mirror_h` > 0: the boxes of the MIRRORED bands instead -- entry (tx, ty) describes output rows mirror_h - j for
// the rows j of tile (tx, ty) (k_ray_lin3_pair_mirror evaluates a tile and its mirror image from one set of coordinates).
__global__ __launch_bounds__(NT) void k_tile_boxes(TileArgs a_)  // a.boxes: written here; a.mirror_h
{
    args_cref a = kernel_args();
    ctx_cref c = args_ctx(a);
    TileBox* boxes = const_cast<TileBox*>(a.boxes);
    const int mirror_h = a.mirror_h;
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) int red[4 * NW];
    __shared__ int red2[2 * NW];
    const int tid = threadIdx.x;
    const TileIds t = tile_ids(c.g, 0, tid, blockIdx.x, blockIdx.y, gridDim.x, NT / kLanesX);
    RowCol rc;
    load_rowcol<ROT>(c.ray, t.xc, mirror_h > 0 ? min(max(mirror_h - t.j, 0), c.g.dst_h - 1) : t.jc, rc);
    LaneCoords L;
    lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, c.ray.rot, rc, t.npx, c.ray.radial, 0, c.ray.n_int, L);
    TileBox b = reduce_box<K, NW>(L, red, tid);
    int interior = __syncthreads_and((int)(t.active & (t.npx == kPX) & (L.ok == 0xFu) & (L.inside == 0xFu))) ? 1 : 0;
    const int lo = wave_min_to_lane63(L.idx_lo), nhi = wave_min_to_lane63(-L.idx_hi);
    if ((tid & 63) == 63)
        red2[(tid >> 6) * 2] = lo, red2[(tid >> 6) * 2 + 1] = nhi;
    __syncthreads();
    int i0 = red2[0], n1 = red2[1];
    for (int w = 1; w < NW; w++)
        i0 = min(i0, red2[2 * w]), n1 = min(n1, red2[2 * w + 1]);
    const int i1 = -n1;
    // Interior tiles of a w-table whose every reachable interval (one either side for the fp32
    // index) has a valid polynomial in m: the consumer will evaluate THAT (no fp64 square root), so
    // the box and the interior verdict are recomputed from those coordinates (bit 1 of `interior`).
    const bool mpoly = OWN == 0 && c.ray.radial_m != nullptr && interior && i0 <= i1 && i0 - 1 >= c.ray.mp_first_ok &&
                       i1 - i0 + 3 <= kTabSlice;
    if (mpoly) {
        __syncthreads();  // red / red2 are reused
        LaneCoords L2;
        lane_coords<VAR_W, ROT, K, 0, 0, 1, 0, NN>(c, c.ray.rot, rc, t.npx, c.ray.radial_m, 0, c.ray.n_int, L2);
        b = reduce_box<K, NW>(L2, red, tid);
        interior = __syncthreads_and((int)(t.active & (t.npx == kPX) & (L2.ok == 0xFu) & (L2.inside == 0xFu)));
        interior = interior ? 3 : 0;  // (not interior any more: an ordinary tile, evaluated through w)
        if (!interior) {
            __syncthreads();
            b = reduce_box<K, NW>(L, red, tid);
        }
    }
    if (tid == 0) {
        // (m-polynomial tiles keep one more entry either side of the slice for the fp32 index)
        const int j0 = interior == 3 ? i0 - 1 : i0, j1 = interior == 3 ? i1 + 1 : i1;
        b.idx0 = i0 <= i1 ? j0 : 0;
        b.nidx = i0 <= i1 ? j1 - j0 + 1 : 0;
        b.interior = interior;
        b.magic = b.cpr > 0 ? (int)(((1u << 20) + (unsigned)b.cpr - 1u) / (unsigned)b.cpr) : 0;
        boxes[t.box_tile] = b;
    }
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

__device__ __forceinline__ void store_interior(units_cptr U, int z, const TileIds& t, const uint32_t (&pix)[kPX])
{
    store4(dst_ptr(U, z, t), pix, 0xFu, dst_rows_dword_aligned(U, z));
}

// ---- slow-path patch (pixels with valid coordinates the tiled path did not produce) and store ----
template <int K>
__device__ __forceinline__ void patch_and_store(ctx_cref c, units_cptr U, int z, const TileIds& t, const LaneCoords& L,
                                                uint32_t (&pix)[kPX], unsigned done, const uint8_t* __restrict__ src)
{
    geom_cref g = c.g;
    const unsigned slow = L.ok & ~done;
    unsigned skip = 0;  // BORDER_TRANSPARENT (bilinear): pixels whose 2 x 2 footprint leaves the source keep the destination's bytes
    if (slow) {
        if (K == 2) {
            // one inlined copy in a rolled loop (a call would pin every live value above the 40
            // caller-saved VGPRs)
#pragma unroll 1
            for (int k = 0; k < kPX; k++) {
                if (slow & (1u << k)) {
                    const int fsx = k == 0 ? L.sx[0] : k == 1 ? L.sx[1] : k == 2 ? L.sx[2] : L.sx[3];
                    const int fsy = k == 0 ? L.sy[0] : k == 1 ? L.sy[1] : k == 2 ? L.sy[2] : L.sy[3];
                    const uint32_t r = slow_pixel_linear3_t(src, U[z].src_pitch, g.src_h, g.src_w, g, fsx, fsy);
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
                    pix[k] = slow_pixel_table3_t<K>(src, U[z].src_pitch, g.src_h, g.src_w, g.border,
                                                    (uint32_t)g.cval[0] | ((uint32_t)g.cval[1] << 8) | ((uint32_t)g.cval[2] << 16), c.itab,
                                                    (float)L.sx[k] * 0.03125f, (float)L.sy[k] * 0.03125f);
        }
    }
    if (!t.active)
        return;
    store4(dst_ptr(U, z, t), pix, L.ok & ~skip, dst_rows_dword_aligned(U, z));
}

// ---- taps, blend, slow-path patch and store: shared tail of the kernels ----
// `wtab` = OpenCV's int16 weight table for K = 4 / 8 (global memory, or LDS in the persistent kernel)
template <int K, typename WPtr>
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

    patch_and_store<K>(c, U, z, t, L, pix, done, src);
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
    const TileBox b = load_tile_box(boxes, t.box_tile);
    const bool tail = box_touches_image_end(b, g);
    // lean batch path (further down): bilinear, more than two units, an interior tile whose table slice
    // and box fit -- the box is the same for all units, only the alignment of a source can differ
    // (the host launches the lean kernel only when every source is dword-aligned and every group has
    // more than two units: launch_tile_k)
    if (LEAN && !lean_static_ok(b.cpr, b.nrows, b.nidx, b.interior, half_dwords))
        return;
    ChunkMap M;
    make_chunk_map<NT>(b, tid, M);

    auto issue = [&](int z, Staged& S) -> bool {  // start the box loads of unit z; false: it must gather from global memory
        const uint8_t* __restrict__ src = !PAIR ? U[z].src : z == z0 ? usrc0 : z == z0 + 1 ? usrc1 : U[z].src;
        const uint32_t spitch = !PAIR ? (uint32_t)U[z].src_pitch : z == z0 ? upitch0 : z == z0 + 1 ? upitch1 : (uint32_t)U[z].src_pitch;
        const uint32_t src_bytes = (uint32_t)(g.src_h - 1) * spitch + (uint32_t)g.src_w * 3u;
        const bool fits = box_fits(b, src, spitch, 4 * NT, LEAN ? 2 * half_dwords : half_dwords);
        if (fits) {
            if (tail)
                stage_load<true, !PAIR>(M, src, spitch, src_bytes, S);
            else
                stage_load<false, !PAIR>(M, src, spitch, src_bytes, S);
        }
        return fits;
    };

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
    if (tab_lds && tid < b.nidx * 4)
        ((d2*)tabw)[tid] = tv;
    V1C_STAMP(1);  // wait for the loads + expand + LDS stores
    __syncthreads();
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
        lane_coords<VAR_W, ROT, K, 0, 1, 1, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
    else if (interior)
        lane_coords<VAR_W, ROT, K, OWN, 1, 0, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
    else if (tab_lds)
        lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
    else
        lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z0].rot, rc, t.npx, P.radial, 0, P.n_int, L);
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
            patch_and_store<K>(c, U, z0, t, L, pix, L.inside, U[z0].src);
            blend_taps(T1, L, pix);
            patch_and_store<K>(c, U, z0 + 1, t, L, pix, L.inside, U[z0 + 1].src);
            return;
        }
        if constexpr (K != 2) {
        if (nu == 2 && fit0 && fit1) {
            // K x K taps of both eyes against one fetch of the weight row
            constexpr int off = K / 2 - 1;
            const int lpw = b.cpr * 4 + 4;
            uint32_t pa[kPX], pb[kPX];
            if (kLanesX == 16 && !V1C_KXK_OWN_LANES) {
                // The gather runs in ANOTHER lane -> pixel mapping than the coordinates and the stores: slot k of lane l
                // samples column 16 k + (l & 15) of the lane's tile row, so that the 16 lanes LDS serves together read
                // ADJACENT cells (with 4 adjacent pixels per lane they read every 4th cell: 4-way bank conflicts at best;
                // the Lanczos4 pair had its LDS 67 % busy, 59 % of that conflicts -- tools/ubench/lanczos_pair_forms.hip:
                // sampler alone 1.29 - 1.93 -> 0.96 - 1.29 ms at C4's size).  Tap origin and weight entry travel as one
                // dword through a wave-private KB of LDS behind the boxes, the two result pixels come back the same way
                // (a wave's LDS operations execute in order: no barrier, only compiler fences).
                const int lane = tid & 63;
                uint32_t* xw = boxw + 2 * half_dwords + (tid >> 6) * 256;  // [row of the wave][column of the tile]
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
            patch_and_store<K>(c, U, z0, t, L, pa, L.inside, U[z0].src);
            patch_and_store<K>(c, U, z0 + 1, t, L, pb, L.inside, U[z0 + 1].src);
            return;
        }
        }
        sample_and_store<K>(c, U, z0, t, L, b, fit0, boxw, wtab, U[z0].src, (uint32_t)U[z0].src_pitch);
        if (nu == 2)
            sample_and_store<K>(c, U, z0 + 1, t, L, b, fit1, boxw + half_dwords, wtab, U[z0 + 1].src, (uint32_t)U[z0 + 1].src_pitch);
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
        sample_and_store<K>(c, U, z, t, L, b, fit_cur, boxw + (u & 1) * half_dwords, wtab, U[z].src, (uint32_t)U[z].src_pitch,
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
                sample_and_store<K>(c, U, z, t, L, fb, false, boxw, wtab, src, spitch);
            }
            return;
        }
        __syncthreads();  // `red` is reused below
    }
    lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, U[z].rot, rc, t.npx, P.radial, 0, P.n_int, L);
    const TileBox b = reduce_box<K, NT / 64>(L, red, tid);
    const bool use_lds = box_fits(b, src, spitch, 4 * NT, kBoxBytes / 4);
    if (use_lds) {
        ChunkMap M;
        make_chunk_map<NT>(b, tid, M);
        Staged S;
        stage_load<true, true>(M, src, spitch, src_bytes, S);
        stage_store(M, S, boxw);
    }
    __syncthreads();
    if (L.ok != (1u << t.npx) - 1)
        if (uint32_t* flags = a.tile_flags)
            flags[t.flag_tile] = 1;
    sample_and_store<K>(c, U, z, t, L, b, use_lds, boxw, wtab, src, spitch);
}

// BOXES = 1: boxes (+ table slices) precomputed by k_tile_boxes, coordinates shared by `upb` units.
// BOXES = 0: units that override the rotation (per-frame calibration): one unit per workgroup, box
//   reduced in-kernel, table read from global memory.
template <int VAR_W, int ROT, int BOXES, int K, int OWN, int PAIR, int LIST = 0, int NN = 0>
// LIST = 1 (BOXES = 1): blockIdx.x indexes a.rest_list (ty << 16 | tx) instead of the tile grid;
// a.tiles_x = tile columns of the full grid then.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((BOXES && PAIR && K == 2 && !ROT && !OWN) ? V1C_PAIR_WAVES : (!BOXES && K == 2) ? V1C_NOBOX_WAVES : 1, 8))) void k_ray_lin3_tile(TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) int red[16];
    __shared__ __attribute__((aligned(16))) double tabw[BOXES ? kTabSlice * kRadialCoefs : 2];
    args_cref a = kernel_args();
    const glb_u32_ptr wtab = K == 2 ? (glb_u32_ptr) nullptr : (glb_u32_ptr)args_ctx(a).itab;
    if (BOXES) {
        int tx, ty, tiles_x;
        if (LIST) {
            const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[blockIdx.x];
            tx = (int)(v & 0xffffu), ty = (int)(v >> 16);
            tiles_x = a.tiles_x;
        } else {
            xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, tx, ty);
            tiles_x = gridDim.x;
        }
        // two box buffers of half_dwords each, sized by the plan from its largest tile box
        extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];
        shared_map_tile<VAR_W, ROT, K, OWN, PAIR, NT, 0, NN>(a, a.n_units, a.upb, blockIdx.z, tx, ty, tiles_x, dyn_box, a.half_dwords, tabw, wtab);
    } else {
        __shared__ __attribute__((aligned(16))) uint32_t boxw[kBoxBytes / 4 + 4];
        // (natural tile order unless the host passes strips: with one block of tiles per XCD the swizzle measured 6 % slower on C5)
        int btx = blockIdx.x, bty = blockIdx.y;
        if (a.strip_len)
            xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, btx, bty);
        rot_unit_tile<VAR_W, ROT, K, OWN, PAIR, NN>(a, (int)blockIdx.z, btx, bty, red, boxw, wtab);
    }
}

#ifdef V1C_TUNING
// The lean batch path of shared_map_tile as a kernel of its own (bilinear, plan-time boxes, more than
// two units per workgroup; register-staged boxes): the A/B partner (V1C_LEAN_RAW=0) of k_ray_lin3_batch_lean_raw, tuning build only.
template <int VAR_W, int ROT, int OWN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROT ? 4 : (OWN ? 5 : V1C_LEAN_WAVES), 8))) void k_ray_lin3_batch_lean(TileArgs a_)
{
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];
    args_cref a = kernel_args();
    int tx, ty;
    // a.rest_list != null: grid slice z = 0 serves the tiles the lean path leaves out, two units per
    // workgroup through the pair code, and the lean groups are slices 1 .. n_groups: the few
    // latency-bound workgroups are dispatched first and run alongside the lean ones instead of in a
    // launch of their own behind them (C3 0.210 -> 0.202 ms).  LDS is sized for the lean path: a
    // remaining tile whose box needs more gathers from global memory.
    int zl = (int)blockIdx.z;
    if (a.rest_list != nullptr)
        zl -= 1;
    if (zl < 0) {
        const unsigned pairs = (unsigned)(a.n_units + 1) / 2u;
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= (unsigned)a.n_rest * pairs)
            return;
        const unsigned ti = lin / pairs, zg = lin - ti * pairs;
        const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[ti];
        shared_map_tile<VAR_W, ROT, 2, OWN, 1, 256, 0>(a, a.n_units, 2, (int)zg, (int)(v & 0xffffu), (int)(v >> 16), (int)gridDim.x, dyn_box,
                                                       a.half_dwords, tabw, (glb_u32_ptr) nullptr);
        return;
    }
    xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, tx, ty);
    shared_map_tile<VAR_W, ROT, 2, OWN, 0, 256, 1>(a, a.n_units, a.upb, zl, tx, ty, gridDim.x, dyn_box, a.half_dwords, tabw,
                                                   (glb_u32_ptr) nullptr);
}
#endif

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

// taps of both eyes of a lane's 4 pixels from the interleaved cells of box `b` (fixed-point rows `sy`), blend, store
// into output row `j`
__device__ __forceinline__ void sample_pair_cells(units_cptr U, const TileIds& t, int j, const TileBox& b, const uint32_t* boxw,
                                                  const int (&sx)[kPX], const int (&sy)[kPX])
{
    typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
    typedef const __attribute__((address_space(3))) u32x2* lds_u64_ptr;
    const uint32_t lpw8 = (uint32_t)(b.cpr * 4 + 4) * 8u;
    const uint32_t base0 = (uint32_t)(uintptr_t)(lds_u32_ptr)boxw - ((uint32_t)b.y0 * lpw8 + (uint32_t)b.x0 * 8u);
    Taps2 T0, T1;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t rel = __umul24((uint32_t)(sy[k] >> 5), lpw8) + (((uint32_t)sx[k] >> 2) & ~7u);
        const lds_u64_ptr ra = (lds_u64_ptr)(uintptr_t)(rel + base0), rb = (lds_u64_ptr)(uintptr_t)(rel + base0 + lpw8);
        const u32x2 a_lo = ra[0], a_hi = ra[1], b_lo = rb[0], b_hi = rb[1];
        T0.alo[k] = a_lo.x, T0.ahi[k] = a_hi.x, T0.blo[k] = b_lo.x, T0.bhi[k] = b_hi.x;
        T1.alo[k] = a_lo.y, T1.ahi[k] = a_hi.y, T1.blo[k] = b_lo.y, T1.bhi[k] = b_hi.y;
    }
    uint32_t pix[kPX];
    const uint32_t row_off = (uint32_t)t.x0 * 3u;
#pragma unroll
    for (int e = 0; e < 2; e++) {
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const BlendW w = blend_weights(sx[k], sy[k]);
            pix[k] = e == 0 ? blend3<4>(T0.alo[k], T0.ahi[k], T0.blo[k], T0.bhi[k], w) : blend3<4>(T1.alo[k], T1.ahi[k], T1.blo[k], T1.bhi[k], w);
        }
        uint8_t* drow = U[e].dst + (__umul24((uint32_t)j, (uint32_t)U[e].dst_pitch) + row_off);
        store4(drow, pix, 0xFu, dst_rows_dword_aligned(U, e));
    }
}

#ifdef V1C_TUNING  // (the register-staged form: A/B partner, V1C_MIRROR_RAW=0, of the LDS-DMA kernels below)
template <int VAR_W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_ray_lin3_pair_mirror(TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];  // one pair of interleaved boxes: 2 x half_dwords
    args_cref a = kernel_args();
    const int tid = threadIdx.x;
    if (blockIdx.z == 0) {  // the tiles the mirror path leaves out, through the general pair code
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= (unsigned)a.n_rest)
            return;
        const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[lin];
        shared_map_tile<VAR_W, 0, 2, 0, 1, NT, 0>(a, 2, 2, 0, (int)(v & 0xffffu), (int)(v >> 16), (int)gridDim.x, dyn_box, a.half_dwords,
                                                  tabw, (glb_u32_ptr) nullptr);
        return;
    }
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    const TileBox* __restrict__ boxes = a.boxes;
    const TileBox* __restrict__ mboxes = a.mboxes;
    const int half_dwords = a.half_dwords, mirror_h = a.mirror_h;
    geom_cref g = c.g;
    ray_cref P = c.ray;
    int tx, ty;
    xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, tx, ty);
    ty += 1;  // tile row 0 has no mirror image (row 0 <-> row mirror_h)
    const TileIds t = tile_ids(g, 0, tid, tx, ty, (int)gridDim.x, NT / kLanesX);
    const uint8_t* __restrict__ src0 = U[0].src;
    const uint8_t* __restrict__ src1 = U[1].src;
    const uint32_t pitch0 = (uint32_t)U[0].src_pitch, pitch1 = (uint32_t)U[1].src_pitch;
    const TileBox b = load_tile_box(boxes, t.box_tile), q = load_tile_box(mboxes, t.box_tile);
    if (!mirror_static_ok(b, q, half_dwords, g.src_h, g.src_w))
        return;
    ChunkMap M;
    make_chunk_map<NT>(b, tid, M);
    Staged S0, S1;
    stage_load<false, false>(M, src0, pitch0, 0u, S0);
    stage_load<false, false>(M, src1, pitch1, 0u, S1);
    typedef double __attribute__((ext_vector_type(2))) d2;
    d2 tv = {0.0, 0.0};
    const bool mpoly = (b.interior & 2) != 0;
    if (tid < b.nidx * 4)
        tv = ((const d2*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs))[tid];
    RowCol rc;
    load_rowcol<0>(P, t.xc, t.jc, rc);
    stage_store_pair(M, S0, S1, dyn_box);
    if (tid < b.nidx * 4)
        ((d2*)tabw)[tid] = tv;
    __syncthreads();
    // the mirrored band's box: requested now, in flight while the tile itself is evaluated and sampled
    make_chunk_map<NT>(q, tid, M);
    stage_load<false, false>(M, src0, pitch0, 0u, S0);
    stage_load<false, false>(M, src1, pitch1, 0u, S1);
    LaneCoords L;
    if (mpoly)
        lane_coords<VAR_W, 0, 2, 0, 1, 1, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
    else
        lane_coords<VAR_W, 0, 2, 0, 1, 0, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
    sample_pair_cells(U, t, t.j, b, dyn_box, L.sx, L.sy);
    __syncthreads();  // every wave has read its taps of the tile's box
    stage_store_pair(M, S0, S1, dyn_box);
    __syncthreads();
    sample_pair_cells(U, t, mirror_h - t.j, q, dyn_box, L.sx, L.sy2);
}
#endif  // V1C_TUNING

// ---- the same workgroup with the boxes brought in by LDS-DMA, as they are in memory ----
// global_load_lds_dwordx4 copies 16 bytes per lane from any dword-aligned address straight into LDS (lane-linear: unit
// u = 256 * pass + tid at byte 16 u), so nothing of a box ever sits in a VGPR: all four boxes of the workgroup (two
// eyes x tile and mirrored band) and the table slice are requested in the prologue and the coordinates are evaluated
// WHILE they are in flight (with register staging that overlap costs the staging registers' occupancy: DESIGN 4.4).
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

// NE = number of eyes (units) of the launch: 2 = apply_lr's pair; 1 = a single image (apply() of one image, BASELINE config 1):
// the same workgroup with two boxes instead of four
__device__ __forceinline__ void gather_one_raw(const TileBox& b, uint32_t raw, const int (&sx)[kPX], const int (&sy)[kPX], uint32_t (&pix)[kPX]);

template <int VAR_W, int NE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V1C_RAW_WAVES, 8))) void k_ray_lin3_pair_mirror_raw(V1C_MIRROR_HEAD, TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];  // 2 NE raw boxes (or the general code's cell buffers)
    args_cref a = kernel_args<kMirrorHeadBytes>();
    const int tid = threadIdx.x;
    // grid: first `rest_rows` rows of workgroups for the tiles this path leaves out (general pair code: they take longest, so
    // they are dispatched first -- dispatched last they were a tail: C1 0.0188 -> 0.0245 ms), then the rows of tile pairs
    const unsigned tiles_x = gx_rest & 0xffffu, rest_rows = gx_rest >> 16;
    if (blockIdx.y < rest_rows) {
        const unsigned lin = blockIdx.y * tiles_x + blockIdx.x;
        if (lin >= (unsigned)a.n_rest)
            return;
        const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[lin];
        shared_map_tile<VAR_W, 0, 2, 0, 1, NT, 0>(a, NE, 2, 0, (int)(v & 0xffffu), (int)(v >> 16), (int)tiles_x, dyn_box, a.half_dwords,
                                                  tabw, (glb_u32_ptr) nullptr);
        return;
    }
    // from the preloaded head alone (V1C_MIRROR_HEAD): the tile, its row / column values (vector loads) and its pair of boxes (one scalar
    // load) are requested before the argument block has been read at all
    int tx, ty;
    {
        const unsigned slen = (rows_strip >> 16) * tiles_x;  // strips of 0 / 2 tile rows; floor(2^32 / (2 t)) == floor(floor(2^32 / t) / 2)
        xcd_tile_at(tiles_x_magic, slen, ((tiles_x_magic - 1u) >> 1) + 1u, tx, ty, rows_strip & 0xffffu, rest_rows, tiles_x, 0u);
    }
    // tile rows 0 .. TY / 2: row 0 of the image has no mirror image (its band row would be row H: not stored), row H / 2 is its
    // own (tile row TY / 2 and its band rewrite rows their neighbours write too -- with the same bytes)
    const DstSize dsz{(int)(dst_wh & 0xffffu), (int)(dst_wh >> 16)};
    const TileIds t = tile_ids(dsz, 0, tid, tx, ty, (int)tiles_x, NT / kLanesX);
    RowCol rc;
    load_rowcol<0>(rowcol_tables_at(rowcol_tables, dsz.dst_w, dsz.dst_h), t.xc, t.jc, rc);
    const TileBoxPair bq = load_tile_box_pair(pairs, t.box_tile);
    const TileBox &b = bq.b, &q = bq.q;
    ctx_cref c = *(const V1C_CONST KernelCtx*)ctxp;
    const units_cptr U = (units_cptr)a.inl;  // (one or two units: always the block's own records, at a known offset)
    geom_cref g = c.g;
    ray_cref P = c.ray;
    touch_plan_and_units<0, NE>(c, U, 0, NE - 1);
    const int nwp = (int)(kb_mh & 0xffffu);
    if (mirror_raw_fit(b, q, nwp, g.src_h, g.src_w) != 1)
        return;
    const bool mpoly = (b.interior & 2) != 0;
    // (the row / column values are consumed here: the compiler's own wait for them then sits in front of the DMA requests,
    // not -- as vmcnt(0), it does not count LDS-DMA -- in front of the coordinates)
#pragma unroll
    for (int k = 0; k < kPX; k++)
        asm volatile("" ::"v"(rc.slon[k]), "v"(rc.qlon[k]));
    asm volatile("" ::"v"(rc.sl), "v"(rc.cl), "v"(rc.hl));
    const uint32_t lds_tab = (uint32_t)(uintptr_t)(lds_u32_ptr)(const uint32_t*)tabw;
    const uint32_t box_bytes = (uint32_t)nwp * 1024u;
    // (one image: two boxes -- half the LDS of a pair's workgroup, 7 workgroups per CU at 67 VGPRs)
    const uint32_t raw_b = (uint32_t)(uintptr_t)(lds_u32_ptr)dyn_box, raw_q = raw_b + (uint32_t)NE * box_bytes;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    {  // table slice: nidx * 4 units of 16 bytes, one pass (units past the slice: clamped, they land in the unused tail of tabw)
        const uint32_t u = min((uint32_t)tid, (uint32_t)(b.nidx * 4 - 1));
        const uint8_t* gp = (const uint8_t*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs) + u * 16u;
        __builtin_amdgcn_global_load_lds((glb_void_ptr)gp, (lds_void_ptr)(uintptr_t)(lds_tab + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u), 16, 0, 0);
    }
    const RawLanes mb = raw_lanes(b.cpr, lane), mq = raw_lanes(q.cpr, lane);
    const uint32_t row_off = (uint32_t)t.x0 * 3u;
    const int mirror_h = (int)(kb_mh >> 16);
    const int jm = mirror_h - t.j;              // the band's row
    const bool band_row = jm < dsz.dst_h;         // (false for row 0 of the image only)
    LaneCoords L;
    uint32_t p0[kPX], p1[kPX];
    const int nb = raw_box_dma(b, mb, U[0].src, (uint32_t)U[0].src_pitch, lane, wave, raw_b);  // this wave's requests per box of the tile ...
    if (NE == 2)
        raw_box_dma(b, mb, U[NE - 1].src, (uint32_t)U[NE - 1].src_pitch, lane, wave, raw_b + box_bytes);
    const int nq = raw_box_dma(q, mq, U[0].src, (uint32_t)U[0].src_pitch, lane, wave, raw_q);  // ... and of the mirrored band
    if (NE == 2)
        raw_box_dma(q, mq, U[NE - 1].src, (uint32_t)U[NE - 1].src_pitch, lane, wave, raw_q + box_bytes);
    // Barriers without __syncthreads()' fence (it would wait for every load in flight): each wave waits for its own part of
    // what the barrier publishes -- vmcnt counts in issue order -- then joins.
    wait_vm_barrier(NE * (nb + nq));  // table slice landed (this wave's box loads may still be in flight)
    if (mpoly)
        lane_coords<VAR_W, 0, 2, 0, 1, 1, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
    else
        lane_coords<VAR_W, 0, 2, 0, 1, 0, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
    wait_vm_barrier(NE * nq);  // the tile's boxes
    if constexpr (NE == 2) {
        gather_pair_raw(b, raw_b, box_bytes, L.sx, L.sy, p0, p1);
        // the mirrored band's boxes: waited for BEFORE the tile's stores are issued (stores count in vmcnt too)
        wait_vm_barrier_imm<0>();
        store_pair_row(U, t, t.j, p0, p1);
        gather_pair_raw(q, raw_q, box_bytes, L.sx, L.sy2, p0, p1);
        if (band_row)
            store_pair_row(U, t, jm, p0, p1);
    } else {
        gather_one_raw(b, raw_b, L.sx, L.sy, p0);
        wait_vm_barrier_imm<0>();
        store4<1>(U[0].dst + (__umul24((uint32_t)t.j, (uint32_t)U[0].dst_pitch) + row_off), p0, 0xFu, dst_rows_dword_aligned(U, 0));
        gather_one_raw(q, raw_q, L.sx, L.sy2, p1);
        if (band_row)
            store4<1>(U[0].dst + (__umul24((uint32_t)jm, (uint32_t)U[0].dst_pitch) + row_off), p1, 0xFu, dst_rows_dword_aligned(U, 0));
    }
}

// ---- the pair kernel with the eyes one after the other: two box buffers instead of four ----
// k_ray_lin3_pair_mirror_raw holds four boxes (two eyes x tile and band) in LDS at once.  Here the workgroup keeps two buffers (tile
// box, band box) and serves eye 0, then eye 1 with the SAME tap addresses and weights (one map per call: both eyes read the same
// box positions), requesting eye 1's tile box as soon as every wave has sampled eye 0's, and its band box likewise:
//   requests: table slice, b(eye 0), q(eye 0) | coordinates -> tap addresses + weights of tile and band (24 registers)
//   gather b | -> request b(eye 1) | store | gather q | -> request q(eye 1) | store | gather b | store | gather q | store
// With the LDS of the four-box form each buffer holds boxes of twice the size -- the pairs that went to the general pair code for
// their size (2 % of the tiles, 3 - 6 % of a C2 launch) stay here -- or, with the same capacity, a workgroup takes half the LDS
// (C1's 9 KB boxes: 4 -> 6 workgroups per CU); a gather holds one eye's taps (24 registers instead of 48).
#ifndef V1C_SEQ_WAVES
#define V1C_SEQ_WAVES 6
#endif
__device__ __forceinline__ void gather_taps_raw(const uint32_t (&ta)[kPX], const BlendW (&W)[kPX], uint32_t pitch, uint32_t (&pix)[kPX])
{
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t a = ta[k], d = a & ~3u;
        const lds_u32_ptr r0 = (lds_u32_ptr)(uintptr_t)d, r1 = (lds_u32_ptr)(uintptr_t)(d + pitch);
        const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        pix[k] = blend3<3>(__builtin_amdgcn_alignbyte(a1, a0, a), __builtin_amdgcn_alignbyte(a2, a1, a), __builtin_amdgcn_alignbyte(b1, b0, a),
                           __builtin_amdgcn_alignbyte(b2, b1, a), W[k]);
    }
}

// REST = 0: the plan's rest list is empty (C2: every tile pair fits) -- the instantiation without the general pair code, whose
// registers (74 against 68 VGPRs, 94 against 66 SGPRs) and LDS (its cell buffers) otherwise set the occupancy of a launch that never
// runs it: 7 instead of 6 workgroups per CU
template <int VAR_W, int REST = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(REST ? V1C_SEQ_WAVES : V1C_SEQ_WAVES + 1, 8))) void k_ray_lin3_pair_mirror_seq(V1C_MIRROR_HEAD, TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];  // 2 raw boxes of a.kb KB (or the general code's cell buffers)
    args_cref a = kernel_args<kMirrorHeadBytes>();
    const int tid = threadIdx.x;
    const unsigned tiles_x = gx_rest & 0xffffu, rest_rows = REST ? gx_rest >> 16 : 0u;
    if constexpr (REST) {
        if (blockIdx.y < rest_rows) {
            const unsigned lin = blockIdx.y * tiles_x + blockIdx.x;
            if (lin >= (unsigned)a.n_rest)
                return;
            const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[lin];
            shared_map_tile<VAR_W, 0, 2, 0, 1, NT, 0>(a, 2, 2, 0, (int)(v & 0xffffu), (int)(v >> 16), (int)tiles_x, dyn_box,
                                                      a.half_dwords, tabw, (glb_u32_ptr) nullptr);
            return;
        }
    }
    // from the preloaded head alone (V1C_MIRROR_HEAD): the tile, its row / column values (vector loads) and its pair of boxes (one scalar
    // load) are requested before the argument block has been read at all
    int tx, ty;
    {
        const unsigned slen = (rows_strip >> 16) * tiles_x;  // strips of 0 / 2 tile rows; floor(2^32 / (2 t)) == floor(floor(2^32 / t) / 2)
        xcd_tile_at(tiles_x_magic, slen, ((tiles_x_magic - 1u) >> 1) + 1u, tx, ty, rows_strip & 0xffffu, rest_rows, tiles_x, 0u);
    }
    // tile rows 0 .. TY / 2: row 0 of the image has no mirror image (its band row would be row H: not stored), row H / 2 is its
    // own (tile row TY / 2 and its band rewrite rows their neighbours write too -- with the same bytes)
    const DstSize dsz{(int)(dst_wh & 0xffffu), (int)(dst_wh >> 16)};
    const TileIds t = tile_ids(dsz, 0, tid, tx, ty, (int)tiles_x, NT / kLanesX);
    RowCol rc;
    load_rowcol<0>(rowcol_tables_at(rowcol_tables, dsz.dst_w, dsz.dst_h), t.xc, t.jc, rc);
    const TileBoxPair bq = load_tile_box_pair(pairs, t.box_tile);
    const TileBox &b = bq.b, &q = bq.q;
    ctx_cref c = *(const V1C_CONST KernelCtx*)ctxp;
    const units_cptr U = (units_cptr)a.inl;  // (one or two units: always the block's own records, at a known offset)
    geom_cref g = c.g;
    ray_cref P = c.ray;
    touch_plan_and_units<0, 2>(c, U, 0, 1);
    const int cap_kb = (int)(kb_mh & 0xffffu);
    if (mirror_raw_fit(b, q, cap_kb, g.src_h, g.src_w) != 1)
        return;
    const bool mpoly = (b.interior & 2) != 0;
#pragma unroll
    for (int k = 0; k < kPX; k++)
        asm volatile("" ::"v"(rc.slon[k]), "v"(rc.qlon[k]));
    asm volatile("" ::"v"(rc.sl), "v"(rc.cl), "v"(rc.hl));
    const uint32_t lds_tab = (uint32_t)(uintptr_t)(lds_u32_ptr)(const uint32_t*)tabw;
    const uint32_t raw_b = (uint32_t)(uintptr_t)(lds_u32_ptr)dyn_box, raw_q = raw_b + (uint32_t)cap_kb * 1024u;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    {
        const uint32_t u = min((uint32_t)tid, (uint32_t)(b.nidx * 4 - 1));
        const uint8_t* gp = (const uint8_t*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs) + u * 16u;
        __builtin_amdgcn_global_load_lds((glb_void_ptr)gp, (lds_void_ptr)(uintptr_t)(lds_tab + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u), 16, 0, 0);
    }
    const RawLanes mb = raw_lanes(b.cpr, lane), mq = raw_lanes(q.cpr, lane);
    const int nb = raw_box_dma(b, mb, U[0].src, (uint32_t)U[0].src_pitch, lane, wave, raw_b);
    const int nq = raw_box_dma(q, mq, U[0].src, (uint32_t)U[0].src_pitch, lane, wave, raw_q);
    wait_vm_barrier(nb + nq);  // table slice
    const uint32_t pitch_b = (uint32_t)raw_units_per_row(b.cpr) * 16u, pitch_q = (uint32_t)raw_units_per_row(q.cpr) * 16u;
    uint32_t ta_b[kPX], ta_q[kPX];
    BlendW W_b[kPX], W_q[kPX];
    {
        LaneCoords L;
        if (mpoly)
            lane_coords<VAR_W, 0, 2, 0, 1, 1, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
        else
            lane_coords<VAR_W, 0, 2, 0, 1, 0, 1>(c, nullptr, rc, kPX, (const double*)tabw, b.idx0, b.nidx, L);
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const uint32_t ixb = (uint32_t)((L.sx[k] >> 5) - b.x0), ixq = (uint32_t)((L.sx[k] >> 5) - q.x0);
            ta_b[k] = __umul24((uint32_t)((L.sy[k] >> 5) - b.y0), pitch_b) + (ixb * 2u + ixb) + raw_b;
            ta_q[k] = __umul24((uint32_t)((L.sy2[k] >> 5) - q.y0), pitch_q) + (ixq * 2u + ixq) + raw_q;
            W_b[k] = blend_weights(L.sx[k], L.sy[k]);
            W_q[k] = blend_weights(L.sx[k], L.sy2[k]);
        }
    }
    const uint32_t row_off = (uint32_t)t.x0 * 3u;
    const int jm = (int)(kb_mh >> 16) - t.j;
    const bool band_row = jm < dsz.dst_h;
    uint32_t pix[kPX];
    // ---- eye 0 ----
    wait_vm_barrier(nq);  // tile box
    gather_taps_raw(ta_b, W_b, pitch_b, pix);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // band box landed; every wave has sampled the tile box
    raw_box_dma(b, mb, U[1].src, (uint32_t)U[1].src_pitch, lane, wave, raw_b);
    store4<1>(U[0].dst + (__umul24((uint32_t)t.j, (uint32_t)U[0].dst_pitch) + row_off), pix, 0xFu, dst_rows_dword_aligned(U, 0));
    gather_taps_raw(ta_q, W_q, pitch_q, pix);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave has sampled the band box
    raw_box_dma(q, mq, U[1].src, (uint32_t)U[1].src_pitch, lane, wave, raw_q);
    if (band_row)
        store4<1>(U[0].dst + (__umul24((uint32_t)jm, (uint32_t)U[0].dst_pitch) + row_off), pix, 0xFu, dst_rows_dword_aligned(U, 0));
    // ---- eye 1 (behind its tile box: eye 0's tile store and its band's nq requests; the predicated band store is not counted) ----
    wait_vm_barrier(nq + 1);
    gather_taps_raw(ta_b, W_b, pitch_b, pix);
    wait_vm_barrier_imm<0>();
    store4<1>(U[1].dst + (__umul24((uint32_t)t.j, (uint32_t)U[1].dst_pitch) + row_off), pix, 0xFu, dst_rows_dword_aligned(U, 1));
    gather_taps_raw(ta_q, W_q, pitch_q, pix);
    if (band_row)
        store4<1>(U[1].dst + (__umul24((uint32_t)jm, (uint32_t)U[1].dst_pitch) + row_off), pix, 0xFu, dst_rows_dword_aligned(U, 1));
}

// (Round 3 also built k_ray_lin3_pair_mirror_pipe here -- two tile rows per workgroup, the second pair's boxes requested into the
// buffers the first pair had just been sampled from -- bit-exact and 5 % slower than one pair per workgroup at equal occupancy
// (profiles/r03a_mid): "the workgroups do not wait for their boxes".  Removed in round 4 with its A/B switch.)

// ---- batches (units sharing one map) with the boxes by LDS-DMA: k_ray_lin3_batch_lean's loop on raw boxes ----
// V1C_LEAN_RING box buffers of nwp KB in a ring: unit u is sampled from its buffer while the boxes of the next ring - 1 units
// are in flight (requested behind the barrier that tells everyone is done with the unit whose buffer they take): the
// register-staged loop without its 12 staging registers (74 instead of 80 VGPRs, rotated batches 77 instead of 106), expansion
// VALU and ds_write_b128; a tap row is 3 dwords at a lane stride of 12 bytes instead of 2 at 16 (9 instead of 14 LDS cycles
// per wave: tools/ubench/dma_raw_forms.hip, lds_tap_mapping.hip).
__host__ __device__ inline bool lean_raw_static_ok(const TileBox& b, int nwp, int src_h, int src_w)
{
    return b.interior != 0 && b.nidx > 0 && b.nidx <= kTabSlice && raw_box_ok(b.x0, b.y0, b.cpr, b.nrows, nwp, src_h, src_w);
}

template <int VAR_W, int ROT, int OWN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROT ? 4 : (OWN ? 5 : V1C_LEAN_WAVES), 8))) void k_ray_lin3_batch_lean_raw(TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];
    args_cref a = kernel_args();
    touch_args(a);
    const int tid = threadIdx.x;
    int zl = (int)blockIdx.z;
    if (a.rest_list != nullptr)
        zl -= 1;  // slice 0: the tiles this path leaves out, two units per workgroup through the pair code
    if (zl < 0) {
        const unsigned pairs = (unsigned)(a.n_units + 1) / 2u;
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= (unsigned)a.n_rest * pairs)
            return;
        const unsigned ti = lin / pairs, zg = lin - ti * pairs;
        const uint32_t v = ((const V1C_CONST uint32_t*)a.rest_list)[ti];
        shared_map_tile<VAR_W, ROT, 2, OWN, 1, NT, 0>(a, a.n_units, 2, (int)zg, (int)(v & 0xffffu), (int)(v >> 16), (int)gridDim.x, dyn_box,
                                                      a.half_dwords, tabw, (glb_u32_ptr) nullptr);
        return;
    }
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    geom_cref g = c.g;
    ray_cref P = c.ray;
    int tx, ty;
    xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, tx, ty);
    const int upb = a.upb, nwp = a.kb;
    const int z0 = zl * upb;
    const TileIds t = tile_ids(a, z0, tid, tx, ty, (int)gridDim.x, NT / kLanesX);
    const int nu = min(upb, a.n_units - z0);
    RowCol rc;
    load_rowcol<ROT>(a, t.xc, t.jc, rc);
    const TileBox b = load_tile_box(a.boxes, t.box_tile);
    touch_plan_and_units<ROT, 1>(c, U, z0, z0);
    if (!lean_raw_static_ok(b, nwp, g.src_h, g.src_w))
        return;
    const bool mpoly = OWN == 0 && (b.interior & 2) != 0;
#pragma unroll
    for (int k = 0; k < kPX; k++)
        asm volatile("" ::"v"(rc.slon[k]), "v"(rc.qlon[k]));
    asm volatile("" ::"v"(rc.sl), "v"(rc.cl), "v"(rc.hl));
    const uint32_t lds_tab = (uint32_t)(uintptr_t)(lds_u32_ptr)(const uint32_t*)tabw;
    const uint32_t box_bytes = (uint32_t)nwp * 1024u;
    const uint32_t raw0 = (uint32_t)(uintptr_t)(lds_u32_ptr)dyn_box;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const RawLanes ml = raw_lanes(b.cpr, lane);
    {
        const uint32_t u = min((uint32_t)tid, (uint32_t)(b.nidx * 4 - 1));
        const uint8_t* gp = (const uint8_t*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs) + u * 16u;
        __builtin_amdgcn_global_load_lds((glb_void_ptr)gp, (lds_void_ptr)(uintptr_t)(lds_tab + (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63) * 16u), 16, 0, 0);
    }
    // Requests are counted per wave in issue order (vmcnt's order): `issued` so far, `done_at[i]` = the count right behind the
    // box requested into buffer i -- that box has landed once at most issued - done_at[i] requests are outstanding.
    constexpr int R = V1C_LEAN_RING;
    int issued = 1, done_at[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        done_at[i] = 0;
        if (i < nu) {
            issued += raw_box_dma(b, ml, U[z0 + i].src, (uint32_t)U[z0 + i].src_pitch, lane, wave, raw0 + (uint32_t)i * box_bytes);
            done_at[i] = issued;
        }
    }
    wait_vm_barrier(issued - 1);  // table slice landed
    const uint32_t pitch = (uint32_t)raw_units_per_row(b.cpr) * 16u;
    uint32_t ta[kPX];
    BlendW W[kPX];
    {
        LaneCoords L;
        if (OWN == 0 && mpoly)
            lane_coords<VAR_W, ROT, 2, 0, 1, 1>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
        else
            lane_coords<VAR_W, ROT, 2, OWN, 1, 0>(c, U[z0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const uint32_t ix = (uint32_t)((L.sx[k] >> 5) - b.x0);
            ta[k] = __umul24((uint32_t)((L.sy[k] >> 5) - b.y0), pitch) + (ix * 2u + ix) + raw0;  // LDS byte address in buffer 0
            W[k] = blend_weights(L.sx[k], L.sy[k]);
        }
    }
    uint32_t cur = 0;  // byte offset of unit u's buffer (u % R)
    int slot = 0;
    for (int u = 0; u < nu; u++) {
        int mark = done_at[0];
#pragma unroll
        for (int i = 1; i < R; i++)
            mark = slot == i ? done_at[i] : mark;
        wait_vm_barrier(issued - mark);  // unit u's box landed (every wave's part of it: barrier)
        // everyone is done with unit u - 1: its buffer takes the box of unit u - 1 + R
        if (u >= 1 && u - 1 + R < nu) {
            const int z = z0 + u - 1 + R;
            const uint32_t prev = cur == 0 ? (uint32_t)(R - 1) * box_bytes : cur - box_bytes;
            issued += raw_box_dma(b, ml, U[z].src, (uint32_t)U[z].src_pitch, lane, wave, raw0 + prev);
#pragma unroll
            for (int i = 0; i < R; i++)
                done_at[i] = (slot == 0 ? R - 1 : slot - 1) == i ? issued : done_at[i];
        }
        asm volatile("" ::: "memory");  // (the counts rely on the program order request -> taps -> store)
        uint32_t pix[kPX];
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const uint32_t a = ta[k] + cur, d = a & ~3u;
            const lds_u32_ptr r0 = (lds_u32_ptr)(uintptr_t)d, r1 = (lds_u32_ptr)(uintptr_t)(d + pitch);
            const uint32_t a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
            pix[k] = blend3<3>(__builtin_amdgcn_alignbyte(a1, a0, a), __builtin_amdgcn_alignbyte(a2, a1, a), __builtin_amdgcn_alignbyte(b1, b0, a),
                               __builtin_amdgcn_alignbyte(b2, b1, a), W[k]);
        }
        store_interior(U, z0 + u, t, pix);
        issued += 1;  // the unit's store (at least one instruction; more only make the next wait longer than needed)
        cur = cur == (uint32_t)(R - 1) * box_bytes ? 0u : cur + box_bytes;
        slot = slot == R - 1 ? 0 : slot + 1;
    }
}

// ---- units that override the rotation (per-frame calibration, BASELINE config 5): two units per workgroup, boxes by LDS-DMA ----
// k_ray_lin3_tile<..., BOXES = 0> serves such a unit with one workgroup per tile: coordinates -> box (reduction) -> register-staged
// box -> barrier -> taps, a serial chain whose loads nothing of its own overlaps, with 60 VALU instructions of staging per lane
// (chunk map, v_perm expansion to BGRx, ds_write_b128).  Here a workgroup serves the tile for TWO units (the two eyes of a
// frame: units 2 z, 2 z + 1 of the launch) and brings the boxes in by LDS-DMA as they are in memory:
//   coordinates A -> box A -> request A | coordinates B -> box B -> request B | taps + blend + store A | taps + blend + store B
// so A's box flies behind B's coordinates and B's behind A's sampling; the row / column table values and the prologue are
// paid once for the two.  Bilinear, OWN = 0 (the host proved that no ray of these units leaves the validated part of the radial
// table), full tiles whose box lies inside the source and fits a buffer; every other (unit, tile) goes through rot_unit_tile.
// Waits: a box is followed by at least the other unit's requests / by the first unit's store (lower bounds: loads the
// coordinates make in between only make a wait longer).
// (The separable form of the rotated ray -- per-column vectors T_k = R_k0 sin(lon) + R_k2 cos(lon) through LDS, 17 instead of 33 fp64
// operations per lane and unit -- was built in round 3, bit-exact and 18 % slower (profiles/r03b_final/ab_rot_separable.log: a serial
// load -> compute -> barrier prologue and 12 LDS reads per lane and unit); removed in round 4.)
#ifndef V1C_ROTPAIR_WAVES
#define V1C_ROTPAIR_WAVES 5
#endif
constexpr int kRotPairRedInts = 32;

struct RawBox {  // a TileBox's geometry for raw_box_dma / the gather
    int x0, y0, cpr, nrows;
};

// reduce_box_all without __syncthreads()' fence (which waits for every request in flight): LDS writes are waited for, then a bare barrier
template <int NW>
__device__ __forceinline__ BoxAll reduce_box_all_nofence(const LaneCoords& L, int* red, int tid)
{
    const int a = min(min(L.sx[0], L.sx[1]), min(L.sx[2], L.sx[3]));
    const int b = min(min(L.sy[0], L.sy[1]), min(L.sy[2], L.sy[3]));
    const int c = -max(max(L.sx[0], L.sx[1]), max(L.sx[2], L.sx[3]));
    const int d = -max(max(L.sy[0], L.sy[1]), max(L.sy[2], L.sy[3]));
    const auto ab = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    const auto cd = __builtin_amdgcn_permlane32_swap((unsigned)c, (unsigned)d, false, false);
    const int pab = min((int)ab[0], (int)ab[1]), pcd = min((int)cd[0], (int)cd[1]);
    const auto q = __builtin_amdgcn_permlane16_swap((unsigned)pab, (unsigned)pcd, false, false);
    int v = min((int)q[0], (int)q[1]);
    v = row_min_step<0x121>(v);
    v = row_min_step<0x122>(v);
    v = row_min_step<0x124>(v);
    v = row_min_step<0x128>(v);
    if ((tid & 15) == 0)
        red[(tid >> 6) * 4 + ((tid >> 4) & 3)] = v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// NC = 1: the host has bounded every pixel's fixed-point coordinates inside the cvRound trick's range (launch_ray_lin3_tile's
// `coords_bounded`): the speculative coordinates need no clamps (2 v_med3_f32 per pixel of a launch that is bound by vector issue)
template <int VAR_W, int MP, int NC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V1C_ROTPAIR_WAVES, 8))) void k_ray_lin3_rot_pair_raw(TileArgs a_)
{
    constexpr int NT = 256;
    // dynamic LDS: two box buffers of slot_bytes (rot_unit_tile: its BGRx box of kBoxBytes + 16) | kRotPairRedInts ints
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];
    args_cref a = kernel_args();
    touch_args(a);
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    geom_cref g = c.g;
    ray_cref P = c.ray;
    const int tid = threadIdx.x;
    int btx = blockIdx.x, bty = blockIdx.y;
    if (a.strip_len)
        xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, btx, bty);
    const int zA = 2 * (int)blockIdx.z, zB = zA + 1;
    const bool hasB = zB < a.n_units;
    const int slot_bytes = a.kb;
    const uint32_t red_off = (uint32_t)max(2 * slot_bytes, kBoxBytes + 16);
    int* red = (int*)((uint8_t*)dyn_box + red_off);
    const glb_u32_ptr wtab = (glb_u32_ptr) nullptr;  // (bilinear: no weight table)
    const bool tile_full = ((btx + 1) * kTW <= a.dst_w) & ((bty + 1) * (NT / kLanesX) <= a.dst_h);
    if (!tile_full) {
        rot_unit_tile<VAR_W, 1, 2, 0, MP>(a, zA, btx, bty, red, dyn_box, wtab);
        if (hasB) {
            __syncthreads();
            rot_unit_tile<VAR_W, 1, 2, 0, MP>(a, zB, btx, bty, red, dyn_box, wtab);
        }
        return;
    }
    const TileIds t = tile_ids(a, zA, tid, btx, bty, gridDim.x, NT / kLanesX);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32_ptr)dyn_box;
    RowCol rc;
    load_rowcol<1>(a, t.xc, t.jc, rc);  // (in flight during the scalar round below)
    touch_plan_and_units<1, 2>(c, U, zA, hasB ? zB : zA);
    // box of all 1024 pixels of a unit -> is the tile interior, does its raw box fit a buffer
    auto raw_box = [&](const BoxAll& ba, int z, TileBox& b) -> bool {
        b.x0 = ba.xmin & ~3, b.y0 = ba.ymin;
        b.cpr = (ba.xmax + 2 - b.x0 + 3) >> 2, b.nrows = ba.ymax - ba.ymin + 2;
        b.idx0 = b.nidx = 0, b.interior = 1, b.magic = 0;
        const int upr = raw_units_per_row(b.cpr);
        const bool inside = (ba.xmin >= 0) & (ba.xmax < g.src_w - 2) & (ba.ymin >= 0) & (ba.ymax < g.src_h - 1);
        const bool aligned = ((((uintptr_t)U[z].src) | (uintptr_t)U[z].src_pitch) & 3u) == 0;
        return inside & aligned & (b.cpr <= kMaxCpr) & (b.nrows * upr * 16 <= slot_bytes) &
               !((b.y0 + b.nrows >= g.src_h) & (b.x0 * 3 + upr * 16 > g.src_w * 3));
    };
    LaneCoords LA, LB;
    TileBox bA, bB;
    bool fastA, fastB = false;
    int nB = 0;
    lane_coords<VAR_W, 1, 2, 0, (NC ? 1 : 2), MP>(c, U[zA].rot, rc, kPX, MP ? P.radial_m : P.radial, 0, P.n_int, LA);
    fastA = raw_box(reduce_box_all_nofence<NT / 64>(LA, red, tid), zA, bA);
    if (fastA) {
        const RawLanes m = raw_lanes(bA.cpr, lane);
        raw_box_dma(bA, m, U[zA].src, (uint32_t)U[zA].src_pitch, lane, wave, lds0);
    }
    if (hasB) {
        lane_coords<VAR_W, 1, 2, 0, (NC ? 1 : 2), MP>(c, U[zB].rot, rc, kPX, MP ? P.radial_m : P.radial, 0, P.n_int, LB);
        fastB = raw_box(reduce_box_all_nofence<NT / 64>(LB, red + 16, tid), zB, bB);
        if (fastB) {
            const RawLanes m = raw_lanes(bB.cpr, lane);
            nB = raw_box_dma(bB, m, U[zB].src, (uint32_t)U[zB].src_pitch, lane, wave, lds0 + (uint32_t)slot_bytes);
        }
    }
    uint32_t pix[kPX];
    if (fastA) {
        wait_vm_barrier(nB);  // A's box: behind it at least B's requests
        gather_one_raw(bA, lds0, LA.sx, LA.sy, pix);
        store_interior(U, zA, t, pix);
    }
    if (fastB) {
        wait_vm_barrier(fastA ? 1 : 0);  // B's box: behind it at least A's store
        gather_one_raw(bB, lds0 + (uint32_t)slot_bytes, LB.sx, LB.sy, pix);
        store_interior(U, zB, t, pix);
    }
    // the rest (rare: rays leaving the source, boxes beyond a buffer): one unit at a time through the general code
    if (!fastA) {
        __syncthreads();
        rot_unit_tile<VAR_W, 1, 2, 0, MP>(a, zA, btx, bty, red, dyn_box, wtab);
    }
    if (hasB && !fastB) {
        __syncthreads();
        rot_unit_tile<VAR_W, 1, 2, 0, MP>(a, zB, btx, bty, red, dyn_box, wtab);
    }
}


// ---- grayscale and BGRA (cn = 1 / 4), bilinear, plan-time boxes: k_ray_lin_cn ----
// The reference hands cv2.remap whatever array the caller passes (remapper.py:388-398); BGR has the kernels above, the other two channel
// counts cv2 images come in run this one: the same tiles, boxes, radial-table slices and coordinates (lane_coords), the source box as it
// is in memory by LDS-DMA (rows of 16-byte units, as k_ray_lin3_batch_lean_raw), two box buffers -- the next unit's box is in flight while
// this one is sampled -- and one workgroup per tile walking all units of the launch (they share the map).  Pixels the box cannot serve
// (footprint leaving the source: border rules; a box beyond the buffers; an unaligned source) take the border-aware per-pixel sampler
// sample_linear_t<CN> from global memory; table intervals the fit flagged go to the fix-up launch like everywhere else.
template <int CN>
__host__ __device__ inline int cn_units_per_row(int cpr)
{
    return CN == 1 ? (cpr + 3) >> 2 : cpr;  // 4 cpr pixels of CN bytes in 16-byte units
}
template <int CN>
__host__ __device__ inline bool cn_box_ok(int x0, int y0, int cpr, int nrows, int kb, int src_h, int src_w)
{
    const int upr = cn_units_per_row<CN>(cpr);
    // (the last unit of a row may read past the box -- never past the image's last row)
    return cpr > 0 && cpr <= kMaxCpr && nrows > 0 && nrows * upr <= kb * 64 && !((y0 + nrows >= src_h) && (x0 * CN + upr * 16 > src_w * CN));
}

// result byte of one channel: taps (p0, p1) of the upper and the lower row as bytes `LO`, `HI` of the 8 bytes (ahi : alo) / (bhi : blo)
template <uint32_t SEL>
__device__ __forceinline__ uint32_t blend_channel(uint32_t alo, uint32_t ahi, uint32_t blo, uint32_t bhi, const BlendW w)
{
    const uint32_t pa = __builtin_amdgcn_perm(ahi, alo, SEL), pb = __builtin_amdgcn_perm(bhi, blo, SEL);
    uint32_t v = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, pa), __builtin_bit_cast(ushort2v, w.wa), 32768u, false);
    v = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, pb), __builtin_bit_cast(ushort2v, w.wb), v, false);
    return v;  // the channel is byte 2
}

// taps of a lane's 4 pixels from a raw box in LDS (`ta`: byte address of the top-left tap), blend; one dword per pixel (CN = 4) or one
// byte per pixel in the low byte (CN = 1)
template <int CN>
__device__ __forceinline__ void gather_cn(const uint32_t (&ta)[kPX], const BlendW (&W)[kPX], uint32_t pitch, uint32_t (&pix)[kPX])
{
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t a = ta[k], d = a & ~3u;
        const lds_u32_ptr r0 = (lds_u32_ptr)(uintptr_t)d, r1 = (lds_u32_ptr)(uintptr_t)(d + pitch);
        const uint32_t a0 = r0[0], a1 = r0[1], b0 = r1[0], b1 = r1[1];
        if (CN == 1) {
            const uint32_t ra = __builtin_amdgcn_alignbyte(a1, a0, a), rb = __builtin_amdgcn_alignbyte(b1, b0, a);
            pix[k] = blend_channel<0x0c010c00u>(ra, 0u, rb, 0u, W[k]) >> 16;
        } else {
            const uint32_t v0 = blend_channel<0x0c040c00u>(a0, a1, b0, b1, W[k]), v1 = blend_channel<0x0c050c01u>(a0, a1, b0, b1, W[k]);
            const uint32_t v2 = blend_channel<0x0c060c02u>(a0, a1, b0, b1, W[k]), v3 = blend_channel<0x0c070c03u>(a0, a1, b0, b1, W[k]);
            const uint32_t lo = __builtin_amdgcn_perm(v1, v0, 0x0c0c0602u), hi = __builtin_amdgcn_perm(v3, v2, 0x06020c0cu);
            pix[k] = lo | hi;
        }
    }
}

template <int CN>
__device__ __forceinline__ void store_cn(uint8_t* drow, const uint32_t (&pix)[kPX], unsigned ok, bool aligned)
{
    if (ok == 0xFu && aligned) {
        uint32_t* d32 = (uint32_t*)drow;
        if (CN == 1) {
            __builtin_nontemporal_store((pix[0] & 255u) | ((pix[1] & 255u) << 8) | ((pix[2] & 255u) << 16) | (pix[3] << 24), d32);
        } else {
#pragma unroll
            for (int k = 0; k < kPX; k++)
                __builtin_nontemporal_store(pix[k], d32 + k);
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPX; k++)
            if (ok & (1u << k)) {
#pragma unroll
                for (int ch = 0; ch < CN; ch++)
                    drow[CN * k + ch] = (uint8_t)(pix[k] >> (8 * ch));
            }
    }
}

// slow_pixel_table3_t for CN = 1 / 4 channels: the rare pixel whose K x K footprint leaves the source (border rules; BORDER_TRANSPARENT
// never comes here), loops rolled -- a small register footprint, no local arrays (sample_table<CN, K> of v1c_core.hpp keeps its tap
// columns in one: scratch)
template <int CN, int K>
__device__ __noinline__ uint32_t slow_pixel_table_cn(const uint8_t* src, int64_t pitch, int h, int w, int border, uint32_t cval, const short* itab,
                                                     int fsx, int fsy)
{
    const Taps t = taps_from_fixed(fsx, fsy);
    const short* __restrict__ wt = itab + (size_t)(t.fy * 32 + t.fx) * (K * K);
    constexpr int off = K / 2 - 1;
    const int sx = t.ix - off, sy = t.iy - off;
    if (border == V1C_BORDER_CONSTANT && (sx >= w || sx + K <= 0 || sy >= h || sy + K <= 0))  // footprint entirely outside
        return CN == 1 ? (cval & 255u) : cval;
    int acc[CN];
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        acc[ch] = 1 << 14;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        const int yi = border_index(sy + i, h, border);  // -1: outside under BORDER_CONSTANT
        const uint8_t* S = src + (int64_t)(yi < 0 ? 0 : yi) * pitch;
#pragma unroll 1
        for (int j = 0; j < K; j++) {
            const int xj = border_index(sx + j, w, border);
            const bool in = (yi >= 0) & (xj >= 0);
            const int wv = wt[i * K + j];
            const uint8_t* p = S + (in ? xj : 0) * CN;
#pragma unroll
            for (int ch = 0; ch < CN; ch++)
                acc[ch] += (in ? (int)p[ch] : (int)((cval >> (8 * ch)) & 255u)) * wv;
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        out |= (uint32_t)min(max(acc[ch] >> 15, 0), 255) << (8 * ch);
    return out;
}

// ---- K x K taps (bicubic / Lanczos4) of one pixel from a raw box: OpenCV's int16 table entry `w` (K * K / 2 dwords, global memory) ----
// CN = 4: a tap is an aligned dword, K of them per row (ds_read2_b32 pairs), per channel and tap pair one v_perm_b32 + one v_dot2 as in the
// BGR kernels; CN = 1: a row's K bytes are cut out of K / 4 + 1 dwords (v_alignbyte_b32), two taps per v_perm_b32 + v_dot2.
// `a`: LDS byte address of the top-left tap.  Returns the pixel (CN bytes from bit 0).
template <int CN, int K>
__device__ __forceinline__ uint32_t blend_table_cn(uint32_t a, uint32_t pitch, glb_u32_ptr w)
{
    int acc[CN];
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        acc[ch] = 1 << 14;
#pragma unroll
    for (int r = 0; r < K; r++) {
        const uint32_t ar = a + (uint32_t)r * pitch;
        const lds_u32_ptr p = (lds_u32_ptr)(uintptr_t)(ar & ~3u);
        uint32_t wr[K / 2];
#pragma unroll
        for (int q = 0; q < K / 2; q++)
            wr[q] = w[r * (K / 2) + q];
        if constexpr (CN == 4) {
            uint32_t d[K];
#pragma unroll
            for (int q = 0; q < K; q++)
                d[q] = p[q];
#pragma unroll
            for (int q = 0; q < K / 2; q++) {
                const short2v ww = __builtin_bit_cast(short2v, wr[q]);
                acc[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c040c00u)), ww, acc[0], false);
                acc[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c050c01u)), ww, acc[1], false);
                acc[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c060c02u)), ww, acc[2], false);
                acc[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x0c070c03u)), ww, acc[3], false);
            }
        } else {
            uint32_t d[K / 4 + 1], b[K / 4];
#pragma unroll
            for (int q = 0; q < K / 4 + 1; q++)
                d[q] = p[q];
#pragma unroll
            for (int q = 0; q < K / 4; q++)
                b[q] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], ar);  // bytes 4 q .. 4 q + 3 of the row
#pragma unroll
            for (int q = 0; q < K / 2; q++) {
                const short2v ww = __builtin_bit_cast(short2v, wr[q]);
                const uint32_t pr = (q & 1) ? __builtin_amdgcn_perm(0u, b[q / 2], 0x0c030c02u) : __builtin_amdgcn_perm(0u, b[q / 2], 0x0c010c00u);
                acc[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pr), ww, acc[0], false);
            }
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        out |= (uint32_t)min(max(acc[ch] >> 15, 0), 255) << (8 * ch);
    return out;
}

// K = 2: bilinear, or nearest with NN = 1 (lane_coords<..., NN>: fixed point 32 * cvRound(x), fractions zero, for which the blend returns its
// top-left tap exactly; every border mode but TRANSPARENT, whose skip rule differs from the bilinear one); K = 4 / 8: bicubic / Lanczos4
// (blend_table_cn; every border mode but TRANSPARENT).
// BOXES = 1: plan-time boxes, one workgroup per tile walking all units of the launch (they share the map).
// BOXES = 0 (ROT = 1): units that override the rotation -- one unit per workgroup (blockIdx.z), coordinates with the unit's matrix,
// the bounding box of its inside pixels reduced in the kernel (reduce_box), then the same requests, gather and patch path.
template <int VAR_W, int ROT, int CN, int NN = 0, int K = 2, int BOXES = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_ray_lin_cn(TileArgs a_)
{
    __shared__ __attribute__((aligned(16))) double tabw[BOXES ? kTabSlice * kRadialCoefs : 2];
    __shared__ __attribute__((aligned(16))) int red[16];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];  // two box buffers of kb KB (+ 16 bytes: the gather's last dword)
    args_cref a = kernel_args();
    ctx_cref c = args_ctx(a);
    const units_cptr U = args_units(a);
    geom_cref g = c.g;
    ray_cref P = c.ray;
    const int tid = threadIdx.x;
    const int kb = a.kb;
    const int u0 = BOXES ? 0 : (int)blockIdx.z, n_units = BOXES ? a.n_units : u0 + 1;  // the units this workgroup serves
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    constexpr int off = K / 2 - 1;  // the footprint's top-left tap is (ix - off, iy - off)
    int tx = blockIdx.x, ty = blockIdx.y;
    if (BOXES)
        xcd_tile(a.tiles_x_magic, a.strip_len, a.strip_magic, tx, ty);
    const TileIds t = tile_ids(g, u0, tid, tx, ty, (int)gridDim.x, 16);
    const glb_u32_ptr wtab = K == 2 ? (glb_u32_ptr) nullptr : (glb_u32_ptr)c.itab;
    RowCol rc;
    LaneCoords L;
    TileBox b;
    if (BOXES) {
        b = load_tile_box(a.boxes, t.box_tile);
        const bool tab_lds = (b.nidx > 0) & (b.nidx <= kTabSlice);
        const bool mpoly = (b.interior & 2) != 0;
        typedef double __attribute__((ext_vector_type(2))) d2;
        d2 tv = {0.0, 0.0};
        if (tab_lds && tid < b.nidx * 4)
            tv = ((const d2*)(radial_table(P, mpoly) + (size_t)b.idx0 * kRadialCoefs))[tid];
        load_rowcol<ROT>(P, t.xc, t.jc, rc);
        if (tab_lds && tid < b.nidx * 4)
            ((d2*)tabw)[tid] = tv;
        __syncthreads();
    } else {
        load_rowcol<ROT>(P, t.xc, t.jc, rc);
        lane_coords<VAR_W, ROT, K, 0, 0, 0, 0, NN>(c, U[u0].rot, rc, t.npx, P.radial, 0, P.n_int, L);
        b = reduce_box<K, 4>(L, red, tid);  // (contains a barrier; idx0 / nidx / interior = 0: unknown)
    }
    // every source dword-aligned (host) and the box inside the buffers: wave-uniform
    const bool fits = cn_box_ok<CN>(b.x0, b.y0, b.cpr, b.nrows, kb, g.src_h, g.src_w);
    const int upr = cn_units_per_row<CN>(b.cpr);
    const RawLanes m = raw_lanes_upr(max(upr, 1), lane);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_u32_ptr)dyn_box, buf_bytes = (uint32_t)kb * 1024u;
    if (fits)  // the first unit's box flies behind the coordinates
        raw_box_dma<CN>(b, m, U[u0].src, (uint32_t)U[u0].src_pitch, lane, wave, lds0);
    if (BOXES) {
        const bool tab_lds = (b.nidx > 0) & (b.nidx <= kTabSlice);
        const bool mpoly = (b.interior & 2) != 0;
        const bool interior = tab_lds & (b.interior != 0);
        if (interior && mpoly)
            lane_coords<VAR_W, ROT, K, 0, 1, 1, 0, NN>(c, U[0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
        else if (interior)
            lane_coords<VAR_W, ROT, K, 0, 1, 0, 0, NN>(c, U[0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
        else if (tab_lds)
            lane_coords<VAR_W, ROT, K, 0, 0, 0, 0, NN>(c, U[0].rot, rc, t.npx, (const double*)tabw, b.idx0, b.nidx, L);
        else
            lane_coords<VAR_W, ROT, K, 0, 0, 0, 0, NN>(c, U[0].rot, rc, t.npx, P.radial, 0, P.n_int, L);
    }
    const bool incomplete = L.ok != (1u << t.npx) - 1;
    const uint32_t lpitch = (uint32_t)upr * 16u;
    uint32_t ta[kPX];
    BlendW W[kPX];   // K = 2: the blend weights
    uint32_t we[kPX];  // K > 2: the table entry (fy * 32 + fx)
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const bool in = (L.inside >> k) & 1;
        ta[k] = in ? __umul24((uint32_t)((L.sy[k] >> 5) - off - b.y0), lpitch) + (uint32_t)((L.sx[k] >> 5) - off - b.x0) * (uint32_t)CN : 0u;
        if (K == 2)
            W[k] = blend_weights(L.sx[k], L.sy[k]);
        else
            we[k] = (uint32_t)((L.sy[k] & 31) * 32 + (L.sx[k] & 31));
    }
    for (int u = u0; u < n_units; u++) {
        if (incomplete)
            if (uint32_t* flags = a.tile_flags)
                flags[t.flag_tile + (BOXES ? u * t.flag_stride : 0)] = 1;  // (BOXES = 0: tile_ids() counted the unit in)
        uint32_t pix[kPX] = {0u, 0u, 0u, 0u};
        unsigned done = 0;
        if (fits) {
            // unit u's box has landed in every wave's share, and every wave is done reading unit u - 1's buffer
            wait_vm_barrier_imm<0>();
            if (u + 1 < n_units)
                raw_box_dma<CN>(b, m, U[u + 1].src, (uint32_t)U[u + 1].src_pitch, lane, wave, lds0 + (uint32_t)((u + 1) & 1) * buf_bytes);
            const uint32_t base = lds0 + (uint32_t)((u - u0) & 1) * buf_bytes;
            if constexpr (K == 2) {
                uint32_t tb[kPX];
#pragma unroll
                for (int k = 0; k < kPX; k++)
                    tb[k] = ta[k] + base;
                gather_cn<CN>(tb, W, lpitch, pix);
            } else {
#pragma unroll 1
                for (int k = 0; k < kPX; k++) {
                    const uint32_t tk = k == 0 ? ta[0] : k == 1 ? ta[1] : k == 2 ? ta[2] : ta[3];
                    const uint32_t ek = k == 0 ? we[0] : k == 1 ? we[1] : k == 2 ? we[2] : we[3];
                    const uint32_t r = blend_table_cn<CN, K>(tk + base, lpitch, wtab + ek * (K * K / 2));
#pragma unroll
                    for (int q = 0; q < kPX; q++)
                        pix[q] = q == k ? r : pix[q];
                }
            }
            done = L.inside;
        }
        const unsigned slow = L.ok & ~done;
        unsigned skip = 0;  // BORDER_TRANSPARENT: the destination keeps its bytes
        if (slow) {
            const Image im{U[u].src, U[u].src_pitch, g.src_h, g.src_w};
            const Geom gg = geom_copy(g);
#pragma unroll 1
            for (int k = 0; k < kPX; k++) {
                if (slow & (1u << k)) {
                    const int fsx = k == 0 ? L.sx[0] : k == 1 ? L.sx[1] : k == 2 ? L.sx[2] : L.sx[3];
                    const int fsy = k == 0 ? L.sy[0] : k == 1 ? L.sy[1] : k == 2 ? L.sy[2] : L.sy[3];
                    uint32_t r;
                    if constexpr (K == 2) {
                        uint8_t px[4] = {0, 0, 0, 0};
                        const bool st = sample_linear_t<CN>(im, gg, taps_from_fixed(fsx, fsy), px);
                        r = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)px[3] << 24);
                        skip |= (st ? 0u : 1u) << k;
                    } else {
                        r = slow_pixel_table_cn<CN, K>(im.p, im.pitch, im.h, im.w, gg.border,
                                                       (uint32_t)gg.cval[0] | ((uint32_t)gg.cval[1] << 8) | ((uint32_t)gg.cval[2] << 16) | ((uint32_t)gg.cval[3] << 24),
                                                       c.itab, fsx, fsy);
                    }
#pragma unroll
                    for (int q = 0; q < kPX; q++)
                        pix[q] = q == k ? r : pix[q];
                }
            }
        }
        if (t.active) {
            uint8_t* drow = U[u].dst + (__umul24((uint32_t)t.j, (uint32_t)U[u].dst_pitch) + (uint32_t)t.x0 * (uint32_t)CN);
            store_cn<CN>(drow, pix, L.ok & ~skip, dst_rows_dword_aligned(U, u));
        }
    }
}

// ---- host side: the argument block of a launch ----
static TileArgs tile_args(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags)
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

static bool units_dword_aligned(const LaunchUnits& lu)
{
    for (int k = 0; k < lu.n; k++)
        if (((((uintptr_t)lu.host[k].src) | (uintptr_t)lu.host[k].src_pitch) & 3u) != 0)
            return false;
    return true;
}

// Units of a launch longer than the kernel-argument block holds: copied into a slot of the plan's device ring by launches of their
// own (kPutUnits records each, carried in THEIR kernel arguments -- 3.6 KB of the 4 KB a launch may carry): stream-ordered,
// graph-capturable, no staging buffer to keep alive.  A 64-unit launch (BASELINE config 5 per GPU) costs two of them, ~9 us in front
// of a 2.3 ms launch.
constexpr int kPutUnits = 32;
struct PutArgs {
    DevUnit u[kPutUnits];
};
__global__ __launch_bounds__(256) void k_put_units(DevUnit* dst, PutArgs ua_, int n)
{
    typedef const V1C_CONST uint32_t* cu32;
    const cu32 src = (cu32)((const V1C_CONST uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() + 8);  // (behind `dst`)
    const int words = n * (int)(sizeof(DevUnit) / 4);
    for (int i = threadIdx.x; i < words; i += 256)
        ((uint32_t*)dst)[i] = src[i];
}

hipError_t launch_put_units(DevUnit* dst, const DevUnit* host, int n, hipStream_t stream)
{
    static_assert(sizeof(DevUnit) % 4 == 0 && alignof(PutArgs) == 8 && sizeof(PutArgs) + 16 <= 4096,
                  "k_put_units reads its records at byte 8 of the kernel arguments");
    for (int base = 0; base < n; base += kPutUnits) {
        const int m = std::min(kPutUnits, n - base);
        PutArgs ua;
        std::memset(&ua, 0, sizeof(ua));
        std::memcpy(ua.u, host + base, sizeof(DevUnit) * (size_t)m);
        hipLaunchKernelGGL(k_put_units, dim3(1), dim3(256), 0, stream, dst + base, ua, m);
    }
    return hipGetLastError();
}

static int taps_of(int interp)
{
    // (INTER_NEAREST rides the bilinear kernels: lane_coords<..., NN = 1>)
    return (interp == V1C_INTER_LINEAR || interp == V1C_INTER_NEAREST) ? 2 : interp == V1C_INTER_CUBIC ? 4 : interp == V1C_INTER_LANCZOS4 ? 8 : 0;
}

bool tile_kernel_supports(const Geom& g)
{
    // every border mode (the border only matters to pixels whose footprint leaves the source, and those take the generic
    // samplers) -- BORDER_TRANSPARENT for INTER_LINEAR only: its skipped pixels (remapBilinear: the 2 x 2 footprint not fully inside)
    // are what the bilinear patch path leaves unstored; the NEAREST / bicubic / Lanczos4 skip rules differ and stay generic
    return g.cn == 3 && (g.border != V1C_BORDER_TRANSPARENT || g.interp == V1C_INTER_LINEAR) && taps_of(g.interp) != 0 && g.src_w >= 3 &&
           g.src_h >= 2;
}

// threads per workgroup (tile = 64 x threads/16) the plan-time boxes are computed for
int tile_threads(const Geom&)
{
    return 256;
}

static dim3 tile_grid(const Geom& g, int nt, int nz)
{
    const int th = nt / kLanesX;
    return dim3((g.dst_w + kTW - 1) / kTW, (g.dst_h + th - 1) / th, nz);
}

size_t tile_box_bytes(const Geom& g)
{
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    return (size_t)d.x * d.y * sizeof(TileBox);
}

// k_ray_lin_cn: grayscale / BGRA; bilinear with every border mode, nearest / bicubic / Lanczos4 with every border mode but TRANSPARENT
bool cn_kernel_supports(const Geom& g)
{
    return (g.cn == 1 || g.cn == 4) && taps_of(g.interp) != 0 && (g.interp == V1C_INTER_LINEAR || g.border != V1C_BORDER_TRANSPARENT) &&
           g.src_w >= 3 && g.src_h >= 2;
}

// box buffer size (KB) of k_ray_lin_cn for a plan: the smallest that holds 99 % of the tile boxes (at most 16)
int tile_cn_box_kb(const void* host_boxes, const Geom& g)
{
    const TileBox* hb = (const TileBox*)host_boxes;
    const dim3 full = tile_grid(g, 256, 1);
    const size_t n = (size_t)full.x * full.y;
    size_t hist[17] = {0};
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        const TileBox& b = hb[i];
        if (b.cpr <= 0 || b.cpr > kMaxCpr || b.nrows <= 0)
            continue;
        const int upr = g.cn == 1 ? cn_units_per_row<1>(b.cpr) : cn_units_per_row<4>(b.cpr);
        const int kb = (b.nrows * upr + 63) / 64;
        hist[std::min(std::max(kb, 1), 16)]++, total++;
    }
    size_t acc = 0;
    for (int k = 1; k <= 16; k++) {
        acc += hist[k];
        if (acc * 100 >= total * 99)
            return std::max(k, 2);
    }
    return 16;
}

// `boxes` == null: the units override the rotation -- one unit per workgroup, boxes reduced in the kernel (kb: box buffer KB, one buffer)
hipError_t launch_ray_lin_cn(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const void* boxes,
                             int kb, hipStream_t stream)
{
    const bool bx = boxes != nullptr;
    const dim3 block(256, 1, 1), grid = tile_grid(c.g, 256, bx ? 1 : lu.n);
    TileArgs a = tile_args(c, cdev, lu, flags);
    a.boxes = (const TileBox*)boxes;
    a.kb = kb;
    a.tiles_x_magic = (unsigned)(0x100000000ull / grid.x) + 1u;
    // XCD interleave: strips of two tile rows (as the BGR launches)
    const unsigned two_rows = 2u * grid.x;
    a.strip_len = bx && two_rows < ((grid.x * grid.y) >> 3) ? two_rows : 0u;
    a.strip_magic = a.strip_len ? (unsigned)(0x100000000ull / a.strip_len) + 1u : 0u;
    const size_t lds = (size_t)(bx ? 2 : 1) * 1024 * (size_t)kb + 16;
    const int mode = c.g.interp == V1C_INTER_NEAREST ? 1 : c.g.interp == V1C_INTER_CUBIC ? 2 : c.g.interp == V1C_INTER_LANCZOS4 ? 3 : 0;
#define V1C_CN_MODE(VW, R, CN, BX)                                                                   \
    do {                                                                                             \
        switch (mode) {                                                                              \
        case 0: hipLaunchKernelGGL((k_ray_lin_cn<VW, R, CN, 0, 2, BX>), grid, block, lds, stream, a); break; \
        case 1: hipLaunchKernelGGL((k_ray_lin_cn<VW, R, CN, 1, 2, BX>), grid, block, lds, stream, a); break; \
        case 2: hipLaunchKernelGGL((k_ray_lin_cn<VW, R, CN, 0, 4, BX>), grid, block, lds, stream, a); break; \
        default: hipLaunchKernelGGL((k_ray_lin_cn<VW, R, CN, 0, 8, BX>), grid, block, lds, stream, a); break; \
        }                                                                                            \
    } while (0)
#define V1C_CN_LAUNCH(VW, R, CN)            \
    do {                                    \
        if (bx)                             \
            V1C_CN_MODE(VW, R, CN, 1);      \
        else if constexpr (R == 1)          \
            V1C_CN_MODE(VW, 1, CN, 0);      \
    } while (0)
    const int sel = (c.ray.var_is_w ? 4 : 0) | ((use_rot || !bx) ? 2 : 0) | (c.g.cn == 4 ? 1 : 0);
    switch (sel) {
    case 0: V1C_CN_LAUNCH(0, 0, 1); break;
    case 1: V1C_CN_LAUNCH(0, 0, 4); break;
    case 2: V1C_CN_LAUNCH(0, 1, 1); break;
    case 3: V1C_CN_LAUNCH(0, 1, 4); break;
    case 4: V1C_CN_LAUNCH(1, 0, 1); break;
    case 5: V1C_CN_LAUNCH(1, 0, 4); break;
    case 6: V1C_CN_LAUNCH(1, 1, 1); break;
    default: V1C_CN_LAUNCH(1, 1, 4); break;
    }
#undef V1C_CN_LAUNCH
#undef V1C_CN_MODE
    return hipGetLastError();
}


// Rest list of the mirror launch; false when the plan cannot use it (geometry, or more remaining tiles than the
// launch's first grid slice holds).
// Box buffer size (wave-passes of 64 sixteen-byte units) of k_ray_lin3_pair_mirror_raw for a plan: the smallest that holds the
// boxes of 98 % of the tile pairs (the others go to the general code with the rest list); 4 boxes of nwp KB each set the
// workgroups per CU.
// `permille`: the share of the tile pairs the buffers must hold (980: k_ray_lin3_pair_mirror_raw; 998 with `max_kb` 11:
// k_ray_lin3_pair_mirror_seq, whose two buffers leave room for nearly every box)
int tile_mirror_raw_passes(const void* host_boxes, const void* host_mboxes, const Geom& g, int permille, int max_kb)
{
    const TileBox* b = (const TileBox*)host_boxes;
    const TileBox* q = (const TileBox*)host_mboxes;
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    std::vector<int> hist(kRawMaxWavePasses + 2, 0);
    size_t n = 0;
    for (unsigned ty = 0; ty <= d.y / 2; ty++)
        for (unsigned tx = 0; tx < d.x; tx++) {
            const size_t i = (size_t)ty * d.x + tx;
            if (b[i].cpr <= 0 || q[i].cpr <= 0 || b[i].cpr > kMaxCpr || q[i].cpr > kMaxCpr)
                continue;
            const int u = std::max(b[i].nrows * raw_units_per_row(b[i].cpr), q[i].nrows * raw_units_per_row(q[i].cpr));
            hist[std::min((u + 63) / 64, kRawMaxWavePasses + 1)]++, n++;
        }
    size_t acc = 0;
    for (int k = 0; k <= kRawMaxWavePasses; k++) {
        acc += hist[k];
        if (acc * 1000 >= n * (size_t)permille)
            return std::min(std::max(k, 4), max_kb);
    }
    return max_kb;
}

// `full_rows` (raw_nwp > 0): the list for k_ray_lin3_pair_mirror_raw, whose workgroups serve tile rows 0 .. TY / 2 (mirror_raw_fit) and
// `n_eyes` units; otherwise tile rows 1 .. TY / 2 - 1 pair up and rows 0, TY / 2 and TY - 1 are always on the list
bool tile_mirror_rest(const void* host_boxes, const void* host_mboxes, const Geom& g, int half_dwords, int mirror_h,
                      std::vector<uint32_t>& rest, int raw_nwp, bool full_rows, int n_eyes)
{
    const TileBox* b = (const TileBox*)host_boxes;
    const TileBox* q = (const TileBox*)host_mboxes;
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    rest.clear();
    const int th = tile_threads(g) / kLanesX;  // tile height
    if (mirror_h != g.dst_h || g.dst_h % (2 * th) != 0 || g.dst_w % 4 != 0 || d.x > 0xffffu || d.y > 0xffffu)
        return false;
    const unsigned TY = d.y, TYh = TY / 2;
    if (TYh < 3)
        return false;
    std::vector<uint8_t> in_rest((size_t)d.x * d.y, 0);
    auto add = [&](unsigned tx, unsigned ty) { in_rest[(size_t)ty * d.x + tx] = 1; };
    if (full_rows && raw_nwp > 0) {
        for (unsigned ty = 0; ty <= TYh; ty++)
            for (unsigned tx = 0; tx < d.x; tx++) {
                const size_t i = (size_t)ty * d.x + tx;
                const int fit = mirror_raw_fit(b[i], q[i], raw_nwp, g.src_h, g.src_w);
                if (fit != 1) {
                    add(tx, ty);
                    if (ty < TYh) {  // its band: 15 rows of tile row TY - 1 - ty and (ty > 0) the first row of tile row TY - ty
                        add(tx, TY - 1 - ty);
                        if (ty > 0)
                            add(tx, TY - ty);
                    }
                }
            }
    } else {
        for (unsigned tx = 0; tx < d.x; tx++)
            add(tx, 0), add(tx, TYh), add(tx, TY - 1);
        for (unsigned ty = 1; ty < TYh; ty++)
            for (unsigned tx = 0; tx < d.x; tx++) {
                const size_t i = (size_t)ty * d.x + tx;
                if (raw_nwp > 0 ? !mirror_raw_static_ok(b[i], q[i], raw_nwp, g.src_h, g.src_w) : !mirror_static_ok(b[i], q[i], half_dwords, g.src_h, g.src_w))
                    add(tx, ty), add(tx, TY - 1 - ty), add(tx, TY - ty);
            }
    }
    for (unsigned ty = 0; ty < d.y; ty++)
        for (unsigned tx = 0; tx < d.x; tx++)
            if (in_rest[(size_t)ty * d.x + tx])
                rest.push_back(ty << 16 | tx);
    // worth it only while the mirror path serves most of the image, and the list must fit grid slice 0
    return rest.size() <= (size_t)d.x * (TYh - 1) && rest.size() * 4 <= (size_t)d.x * d.y;
}

// lu.n = 2: apply_lr's pair; 1 (raw_nwp > 0): a single image through the one-eye instantiation of the LDS-DMA kernel
// seq_kb > 0 (pairs): k_ray_lin3_pair_mirror_seq, two box buffers of seq_kb KB, the eyes one after the other
hipError_t launch_ray_lin3_pair_mirror(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, const void* boxes,
                                       const void* mboxes, int half_dwords, int mirror_h, const uint32_t* rest_list, int n_rest, int raw_nwp,
                                       hipStream_t stream, int seq_kb)
{
    const int n_eyes = lu.n;
    const dim3 full = tile_grid(c.g, 256, 1);
    const dim3 block(256, 1, 1);
    TileArgs a = tile_args(c, cdev, lu, flags);
    a.boxes = (const TileBox*)boxes, a.mboxes = (const TileBox*)mboxes;
    a.rest_list = rest_list, a.n_rest = n_rest;
    a.half_dwords = half_dwords, a.mirror_h = mirror_h;
    a.tiles_x_magic = (unsigned)(0x100000000ull / full.x) + 1u;
    static const unsigned strip_rows = [] {  // V1C_MIRROR_STRIP_ROWS=<n>: A/B override (0: one block per XCD)
        const char* e = tuning_env("V1C_MIRROR_STRIP_ROWS");
        return e ? (unsigned)std::atoi(e) : 2u;
    }();
    static const unsigned block_rows = [] {  // V1C_MIRROR_BLOCK_ROWS=<n>: XCD blocks of gridDim.x / 8 columns x n tile rows (xcd_tile)
        const char* e = tuning_env("V1C_MIRROR_BLOCK_ROWS");
        return e ? (unsigned)std::atoi(e) : 0u;
    }();
    // the LDS-DMA kernels serve tile rows 0 .. TY / 2 (two more than the register-staged pairing of rows 1 .. TY / 2 - 1)
    [[maybe_unused]] auto strips = [&](unsigned rows) {  // two tile rows per strip (tile_xcd_strips): the tuning build's A/B partners
        const unsigned per = (full.x * rows) >> 3;
        unsigned slen = strip_rows && strip_rows * full.x < per ? strip_rows * full.x : 0u;
        if (block_rows && full.x % 8 == 0 && raw_nwp > 0)
            slen = 0x80000000u | std::min(block_rows, 0xffffu);
        a.strip_len = slen;
        a.strip_magic = (slen && !(slen & 0x80000000u)) ? (unsigned)(0x100000000ull / slen) + 1u : 0u;
    };
    const unsigned raw_rows = full.y / 2 + 1;
    const unsigned rest_rows = ((((unsigned)n_rest + full.x - 1) / full.x) + 7u) & ~7u;  // whole rows, a multiple of 8: the pair rows keep their XCDs
    // The LDS-DMA kernels take the head of their arguments as preloaded scalar parameters (V1C_MIRROR_HEAD): the (tile, band) box pairs
    // (`mboxes`), tiles_x | rest_rows << 16, rows of tile pairs | XCD strip rows << 16 (two tile rows per strip, or one block per XCD), the
    // destination size, the base of the plan's row / column tables (= col_s: one buffer), the context, box KB | mirror row << 16
    const unsigned srows = strip_rows == 2u && 2u * full.x < ((full.x * raw_rows) >> 3) ? 2u : 0u;
    const unsigned rows_strip = raw_rows | srows << 16, dst_wh = (unsigned)c.g.dst_w | (unsigned)c.g.dst_h << 16;
#define V1C_MIRROR_LAUNCH(KERNEL, GRID, LDS, RESTROWS)                                                                                    \
    hipLaunchKernelGGL(KERNEL, GRID, block, LDS, stream, a.mboxes, a.tiles_x_magic, full.x | (unsigned)(RESTROWS) << 16, rows_strip, dst_wh, \
                       a.col_s, a.ctx, (unsigned)a.kb | (unsigned)a.mirror_h << 16, a)
    if (seq_kb > 0 && n_eyes == 2) {  // the eyes one after the other: two buffers of seq_kb KB (rest list made for that size)
        static const bool norest_off = [] {  // V1C_SEQ_NOREST=0 (A/B): the instantiation with the general pair code for every plan
            const char* e = tuning_env("V1C_SEQ_NOREST");
            return e && e[0] == '0';
        }();
        a.kb = seq_kb;
        if (n_rest == 0 && !norest_off) {  // nothing for the general pair code: the instantiation (and the LDS) without it
            const dim3 rgrid(full.x, raw_rows, 1);
            const size_t slds = (size_t)2 * 1024 * (size_t)seq_kb + 16;
            a.rest_rows = 0;
            if (c.ray.var_is_w)
                V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_seq<1, 0>), rgrid, slds, 0u);
            else
                V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_seq<0, 0>), rgrid, slds, 0u);
            return hipGetLastError();
        }
        const dim3 rgrid(full.x, raw_rows + rest_rows, 1);
        const size_t slds = std::max((size_t)half_dwords * 8 + 16, (size_t)2 * 1024 * (size_t)seq_kb);
        a.rest_rows = rest_rows;
        if (c.ray.var_is_w)
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_seq<1>), rgrid, slds, rest_rows);
        else
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_seq<0>), rgrid, slds, rest_rows);
        return hipGetLastError();
    }
    if (raw_nwp > 0 && n_eyes == 1) {  // a single image: the LDS-DMA kernel's one-eye instantiation
        const dim3 rgrid(full.x, raw_rows + rest_rows, 1);
        // two boxes; the general pair code serves one unit from one cell buffer
        const size_t lds = std::max((size_t)half_dwords * 4 + 16, (size_t)2 * 1024 * (size_t)raw_nwp);
        a.kb = raw_nwp, a.rest_rows = rest_rows;
        if (c.ray.var_is_w)
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_raw<1, 1>), rgrid, lds, rest_rows);
        else
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_raw<0, 1>), rgrid, lds, rest_rows);
        return hipGetLastError();
    }
#ifdef V1C_TUNING  // A/B partners of the seq kernel: the four-buffer LDS-DMA pair kernel (V1C_MIRROR_SEQ=0) and the register-staged one (V1C_MIRROR_RAW=0)
    if (n_eyes != 2)
        return hipErrorInvalidValue;
    const size_t lds = std::max((size_t)half_dwords * 8 + 16, (size_t)4 * 1024 * (size_t)std::max(raw_nwp, 0));
    if (raw_nwp > 0) {
        dim3 rgrid(full.x, raw_rows + rest_rows, 1);
        strips(raw_rows);
        a.kb = raw_nwp, a.rest_rows = rest_rows;
        // V1C_MIRROR_SKIP=1: timing experiment, the rest rows are not launched (their tiles stay unwritten); =2: ONLY the rest rows
        static const int skip = [] {
            const char* e = tuning_env("V1C_MIRROR_SKIP");
            return e ? std::atoi(e) : 0;
        }();
        if (skip == 1)
            a.n_rest = 0, a.rest_rows = 0, rgrid.y = raw_rows;
        if (skip == 2)
            rgrid.y = rest_rows;
        const unsigned rr = skip == 1 ? 0u : rest_rows;
        if (c.ray.var_is_w)
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_raw<1, 2>), rgrid, lds, rr);
        else
            V1C_MIRROR_LAUNCH((k_ray_lin3_pair_mirror_raw<0, 2>), rgrid, lds, rr);
        return hipGetLastError();
    }
    const dim3 grid(full.x, full.y / 2 - 1, 2);
    strips(grid.y);
    if (c.ray.var_is_w)
        hipLaunchKernelGGL((k_ray_lin3_pair_mirror<1>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((k_ray_lin3_pair_mirror<0>), grid, block, lds, stream, a);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;  // (plan.hip selects the seq / one-eye forms only)
#endif
}

// LDS dwords one box buffer must hold so that every stageable tile box of `host_boxes` fits
// (capped at kMaxHalfDwords: larger boxes gather from global memory)
constexpr int kMaxHalfDwords = 8192;  // 32 KB per buffer

int tile_half_dwords(const void* host_boxes, size_t n_tiles)
{
    const TileBox* b = (const TileBox*)host_boxes;
    int m = 256;
    for (size_t i = 0; i < n_tiles; i++) {
        if (b[i].cpr <= 0 || b[i].cpr > kMaxCpr || b[i].nrows * b[i].cpr > 1024)
            continue;
        const int need = b[i].nrows * (b[i].cpr * 4 + 4);
        if (need <= kMaxHalfDwords)
            m = std::max(m, need);
    }
    return (m + 3) & ~3;
}

// Box buffer size (dwords) of the lean batch kernel: capped so that 6 workgroups -- what its 80 VGPRs
// allow -- also fit the CU's 160 KB of LDS (2 buffers + 4 KB table slice each); tiles with larger
// boxes go to the general kernel.
int tile_lean_half_dwords(int half_dwords)
{
    static const int cap = [] {  // V1C_LEAN_CAP=<dwords>: A/B override
        const char* e = tuning_env("V1C_LEAN_CAP");
        const int v = e ? std::atoi(e) : 0;
        return v >= 256 ? v : 2816;
    }();
    return std::min(half_dwords, cap);
}

// Strip length (tiles) of the XCD interleave (xcd_tile) for a plan, 0 = one block per XCD: a cost model
// of the tiles (box beyond the LDS buffers: gathers from global memory; box needing both lean
// buffers; not interior) is summed per XCD for 1, 2, 4, 8, 16 strips per XCD; the smallest count
// within 3 % of the best balance wins.  Short launches keep one block (their first workgroups would
// all start in the same corner of the image).
int tile_xcd_strips(const void* host_boxes, const Geom& g, int half_dwords, int lean_half)
{
    static const int forced = [] {  // V1C_XCD_STRIPS=<n>: A/B override (1 = one block per XCD)
        const char* e = tuning_env("V1C_XCD_STRIPS");
        return e ? std::atoi(e) : 0;
    }();
    const TileBox* b = (const TileBox*)host_boxes;
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    const unsigned ntile = d.x * d.y, per = ntile >> 3;
    if (per < 768 && forced <= 0)
        return 0;
    std::vector<float> cost(ntile);
    for (unsigned i = 0; i < ntile; i++) {
        const int need = b[i].cpr > 0 ? b[i].nrows * (b[i].cpr * 4 + 4) : 0;
        const bool stageable = b[i].cpr > 0 && b[i].cpr <= kMaxCpr && b[i].nrows * b[i].cpr <= 1024 && need <= half_dwords;
        cost[i] = b[i].cpr <= 0 ? 0.5f : 1.0f + (stageable ? 0.0f : 4.0f) + (need > lean_half ? 0.5f : 0.0f) + (b[i].interior ? 0.0f : 0.5f) + (float)need / 16384.0f;
    }
    // candidates: one block, then strips of 16, 8, 4, 2 tile rows (forced: V1C_XCD_STRIPS = strips per XCD, rounded to
    // whole tile rows; the last strip of a share may be shorter)
    std::vector<unsigned> cand_len;
    std::vector<double> cand_load;
    auto model = [&](unsigned L) {
        double load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned nfull = per / L;
        for (unsigned x = 0; x < 8; x++)
            for (unsigned i = 0; i < per; i++) {
                const unsigned sidx = i / L;
                const unsigned m = sidx < nfull ? (sidx * 8u + x) * L + (i - sidx * L) : nfull * 8u * L + x * (per - nfull * L) + (i - nfull * L);
                load[x] += cost[m];
            }
        const double mx = *std::max_element(load, load + 8);
        if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
            std::fprintf(stderr, "[v1c] XCD interleave model: strips of %u tiles (%.1f per XCD) -> max load %.0f (mean %.0f)\n", L, (double)per / L, mx,
                         (load[0] + load[1] + load[2] + load[3] + load[4] + load[5] + load[6] + load[7]) / 8);
        return mx;
    };
    if (forced > 0) {
        if (forced == 1)
            return 0;
        const unsigned rows = std::max(1u, (per / (unsigned)forced + d.x / 2) / d.x);
        const unsigned L = rows * d.x;
        return L < per ? (int)L : 0;
    }
    cand_len.push_back(per), cand_load.push_back(model(per));
    for (unsigned rows : {16u, 8u, 4u, 2u})
        if (rows * d.x < per)
            cand_len.push_back(rows * d.x), cand_load.push_back(model(rows * d.x));
    // Measured with source and destination L3-cold (r02, interleaved A/B runs): the finer the interleave the
    // faster -- C2 1 / 2 / 4 / 8 / 16 strips 0.0570 / 0.0564 / 0.0555 / 0.0543 / 0.0532 ms, C4 8 -> 16 strips 1.363 ->
    // 1.336 ms -- all eight XCDs then work on the same band of the image, i.e. on the same open HBM pages and
    // Infinity-Cache sets, instead of on eight bands 1/8 of the image apart; below two tile rows per strip every halo
    // row is shared between two XCDs' L2s and it turns (C2 32 strips 0.0531 vs 0.0527).  (Round 1 tuned this on an
    // L3-resident buffer set, where strips only paid for balancing tile costs.)  So: the finest candidate, unless the
    // model says a coarser one balances the tile costs 3 % better.
    size_t q_best = cand_len.size() - 1;
    for (size_t q = q_best; q-- > 0;)
        if (cand_load[q] < 0.97 * cand_load[q_best])
            q_best = q;
    return cand_len[q_best] < per ? (int)cand_len[q_best] : 0;
}

// tiles the lean batch kernel leaves to the general one, as ty << 16 | tx (row-major order)
// Box buffer size (KB = wave-passes of 64 units) of k_ray_lin3_batch_lean_raw: holds the boxes of 98 % of the interior tiles
int tile_lean_raw_passes(const void* host_boxes, const Geom& g)
{
    const TileBox* b = (const TileBox*)host_boxes;
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    std::vector<int> hist(kRawMaxWavePasses + 2, 0);
    size_t n = 0;
    for (size_t i = 0; i < (size_t)d.x * d.y; i++) {
        if (b[i].interior == 0 || b[i].cpr <= 0 || b[i].cpr > kMaxCpr)
            continue;
        hist[std::min((b[i].nrows * raw_units_per_row(b[i].cpr) + 63) / 64, kRawMaxWavePasses + 1)]++, n++;
    }
    size_t acc = 0;
    for (int k = 0; k <= kRawMaxWavePasses; k++) {
        acc += hist[k];
        if (acc * 100 >= n * 98)
            return std::max(k, 4);
    }
    return kRawMaxWavePasses;
}

// `raw_nwp` > 0: the list for k_ray_lin3_batch_lean_raw with box buffers of raw_nwp KB instead
std::vector<uint32_t> tile_rest_list(const void* host_boxes, const Geom& g, int half_dwords, int raw_nwp)
{
    const TileBox* b = (const TileBox*)host_boxes;
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    std::vector<uint32_t> out;
    if (d.x > 0xffffu || d.y > 0xffffu)
        return out;
    for (unsigned ty = 0; ty < d.y; ty++)
        for (unsigned tx = 0; tx < d.x; tx++) {
            const TileBox& q = b[(size_t)ty * d.x + tx];
            if (raw_nwp > 0 ? !lean_raw_static_ok(q, raw_nwp, g.src_h, g.src_w) : !lean_static_ok(q.cpr, q.nrows, q.nidx, q.interior, half_dwords))
                out.push_back(ty << 16 | tx);
        }
    return out;
}

// plan creation: fill `boxes` (device, tile_box_bytes()) for the plan's own rotation
template <int K, int NT>
static void launch_boxes_k(const KernelCtx& c, const TileArgs& a, bool shared_entry, hipStream_t stream)
{
    const dim3 block(NT, 1, 1), grid = tile_grid(c.g, NT, 1);
    const bool rot = c.ray.has_rot != 0;
    const bool nn = c.g.interp == V1C_INTER_NEAREST;
#define V1C_BOXES(VW, RT)                                                                              \
    do {                                                                                               \
        if constexpr (K == 2) {                                                                        \
            if (nn) {                                                                                  \
                if (shared_entry)                                                                      \
                    hipLaunchKernelGGL((k_tile_boxes<VW, RT, K, NT, 0, 1>), grid, block, 0, stream, a); \
                else                                                                                   \
                    hipLaunchKernelGGL((k_tile_boxes<VW, RT, K, NT, 1, 1>), grid, block, 0, stream, a); \
                break;                                                                                 \
            }                                                                                          \
        }                                                                                              \
        if (shared_entry)                                                                              \
            hipLaunchKernelGGL((k_tile_boxes<VW, RT, K, NT, 0>), grid, block, 0, stream, a);           \
        else                                                                                           \
            hipLaunchKernelGGL((k_tile_boxes<VW, RT, K, NT, 1>), grid, block, 0, stream, a);           \
    } while (0)
    if (c.ray.var_is_w) {
        if (rot)
            V1C_BOXES(1, 1);
        else
            V1C_BOXES(1, 0);
    } else {
        if (rot)
            V1C_BOXES(0, 1);
        else
            V1C_BOXES(0, 0);
    }
#undef V1C_BOXES
}

// `shared_entry`: the value the launches consuming these boxes will pass to launch_ray_lin3_tile
// (`cdev` must hold the plan's tables already: the kernel reads them through it)
hipError_t launch_tile_boxes(const KernelCtx& c, const KernelCtx* cdev, void* boxes, bool shared_entry, hipStream_t stream, int mirror_h)
{
    TileArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ctx = cdev;
    a.boxes = (const TileBox*)boxes;
    a.mirror_h = mirror_h;
    a.col_s = c.ray.col_s, a.col_c = c.ray.col_c, a.col_h = c.ray.col_h;
    a.row_s = c.ray.row_s, a.row_c = c.ray.row_c, a.row_h = c.ray.row_h;
    a.dst_w = c.g.dst_w, a.dst_h = c.g.dst_h;
    switch (taps_of(c.g.interp)) {
    case 2: launch_boxes_k<2, 256>(c, a, shared_entry, stream); break;
    case 4: launch_boxes_k<4, 256>(c, a, shared_entry, stream); break;
    case 8: launch_boxes_k<8, 256>(c, a, shared_entry, stream); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int K>
static void launch_tile_k(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const TileBox* bx,
                          int half_dwords, bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half, int strip_len,
                          int lean_raw_nwp, hipStream_t stream, bool coords_bounded)
{
    const int n_units = lu.n;
    // with precomputed boxes a workgroup serves up to kUnitsPerBlock units that share the map
    static const int upb_max = [] {  // V1C_UPB=<n>: A/B override of the units per workgroup
        const char* e = tuning_env("V1C_UPB");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1 && v <= kUnitsPerBlock ? v : kUnitsPerBlock;
    }();
    // balanced groups (10 units: 5 + 5, not 8 + 2): no short tail group, and every group of a batch
    // has more than two units, which the lean batch kernel wants
    const int n_groups = (n_units + upb_max - 1) / upb_max;
    int upb = bx ? (n_units + n_groups - 1) / n_groups : 1;
    // the template's PAIR slot: with boxes "at most 2 units per workgroup", without (one unit per
    // workgroup anyway) "the m-polynomial table serves every pixel of every unit" (bilinear, OWN = 0)
    // (decided below for launches with boxes)
    bool pair = bx ? upb <= 2 : (K == 2 && shared_entry && mpoly_all && c.ray.radial_m != nullptr);
    const dim3 block(256, 1, 1);
    dim3 grid = tile_grid(c.g, 256, (n_units + upb - 1) / upb);
    static const size_t lds_pad = [] {  // V1C_LDS_PAD=<bytes>: occupancy experiments (fewer workgroups per CU)
        const char* e = tuning_env("V1C_LDS_PAD");
        return e ? (size_t)std::atoi(e) : (size_t)0;
    }();
    // batches (more than two units per workgroup) of a bilinear plan: the interior tiles go to the lean
    // kernel, everything else stays with the general one (same grid; each skips the other's tiles)
    static const bool lean_off = [] {
        const char* e = tuning_env("V1C_DISABLE_LEAN");
        return e && e[0] == '1';
    }();
    // (every group with more than two units, every source dword-aligned, tile coordinates fit 16 bits)
    static const bool lean_pair = [] {  // V1C_LEAN_PAIR=1: A/B switch, pairs through the lean kernel too
        const char* e = tuning_env("V1C_LEAN_PAIR");
        return e && e[0] == '1';
    }();
    const bool nn = c.g.interp == V1C_INTER_NEAREST;  // through k_ray_lin3_tile<..., NN = 1> only
    bool lean = bx && K == 2 && !nn && !lean_off && rest_list != nullptr &&
                (lean_pair ? (upb >= 2 && n_units % upb != 1) : (upb > 2 && (n_units % upb == 0 || n_units % upb > 2)));
#ifndef V1C_TUNING
    lean = lean && lean_raw_nwp > 0;  // (the register-staged lean kernel is an A/B partner: tuning build only)
#endif
    lean = lean && units_dword_aligned(lu);
    // Whatever does not take the lean batch kernel (NEAREST, bicubic / Lanczos4, unaligned sources, one or two units left over) is served
    // TWO units per workgroup by the pair code: round 1 measured the general loop -- up to 8 units per workgroup, 100 - 124 VGPRs -- no
    // faster pair by pair (C3 277 against 267 us), and its 24 instantiations were the last kernels that spilled scalar registers into
    // vector lanes (9 - 12 each).  They exist in the tuning build only, behind V1C_UPB=<n>.
    static const bool upb_forced = tuning_env("V1C_UPB") != nullptr;
    if (bx && !lean && !upb_forced && upb > 2) {
        upb = 2;
        pair = true;
        grid = tile_grid(c.g, 256, (n_units + 1) / 2);
    }
    // two box buffers (+ the K x K pair path's exchange buffer: 1 KB per wave, see shared_map_tile)
    const size_t lds = bx ? (size_t)half_dwords * 8 + 16 + (K != 2 && pair ? kKxkExchangeBytes : 0) + lds_pad : 0;
    // (An LDS-DMA form of the plain pair kernel -- k_ray_lin3_pair_mirror_raw without the mirror image -- was built and removed:
    // bit-identical, but 0.0535 against 0.0511 ms on an unrotated 4080^2 pair and 0.0733 against 0.0684 ms on a rotated
    // 4096^2 pair (94 VGPRs): with one tile per workgroup the interleaved cells' single ds_read2_b64 per tap row wins.)
    // the few remaining tiles are served two units per workgroup (the pair instantiation): a workgroup
    // looping over 8 units would be one long serial chain with nothing to overlap it
    const dim3 rest_grid((unsigned)std::max(n_rest, 1), 1, (unsigned)((n_units + 1) / 2));
    const size_t lean_lds = (size_t)lean_half * 8 + 16;
    // the remaining tiles ride in the lean launch as one more grid slice when they fit one (else, or
    // with V1C_DISABLE_MERGE=1, in a launch of their own)
    static const bool merge_off = [] {
        const char* e = tuning_env("V1C_DISABLE_MERGE");
        return e && e[0] == '1';
    }();
    const bool merged = !merge_off && n_rest > 0 && (size_t)n_rest * ((n_units + 1) / 2) <= (size_t)grid.x * grid.y;
    const dim3 merged_grid(grid.x, grid.y, grid.z + 1);
    static const unsigned nobox_strip_rows = [] {  // V1C_NOBOX_STRIP_ROWS=<n>: XCD strips of n tile rows for launches without boxes
        const char* e = tuning_env("V1C_NOBOX_STRIP_ROWS");
        return e ? (unsigned)std::atoi(e) : 0u;
    }();
    TileArgs a = tile_args(c, cdev, lu, flags);
    a.boxes = bx;
    a.upb = upb, a.half_dwords = half_dwords;
    a.tiles_x = (int)grid.x;
    a.tiles_x_magic = (unsigned)(0x100000000ull / grid.x) + 1u;
    a.strip_len = bx ? (strip_len > 0 && (unsigned)strip_len < ((grid.x * grid.y) >> 3) ? (unsigned)strip_len : 0u)
                     : (nobox_strip_rows * grid.x < ((grid.x * grid.y) >> 3) ? nobox_strip_rows * grid.x : 0u);
    a.strip_magic = a.strip_len ? (unsigned)(0x100000000ull / a.strip_len) + 1u : 0u;
    // units that override the rotation, bilinear, OWN = 0: two units per workgroup with their boxes by LDS-DMA (V1C_ROT_PAIR=0: A/B
    // switch, the one-unit-per-workgroup kernel)
    static const bool rot_pair_off = [] {
        const char* e = tuning_env("V1C_ROT_PAIR");
        return e && e[0] == '0';
    }();
    static const int rot_pair_slot = [] {  // V1C_ROT_PAIR_SLOT=<bytes>: box buffer size (A/B)
        const char* e = tuning_env("V1C_ROT_PAIR_SLOT");
        const int v = e ? std::atoi(e) : 0;
        return v >= 1024 ? (v & ~15) : 12288;
    }();
    if constexpr (K == 2) {
        if (!bx && shared_entry && !rot_pair_off && !nn) {
            const dim3 pgrid(grid.x, grid.y, (unsigned)((n_units + 1) / 2));
            const size_t plds = (size_t)std::max(2 * rot_pair_slot, kBoxBytes + 16) + kRotPairRedInts * sizeof(int);
            const bool mp = mpoly_all && c.ray.radial_m != nullptr;
            a.kb = rot_pair_slot;
#define V1C_ROTPAIR(VW, MPV)                                                                            \
    do {                                                                                                \
        if (coords_bounded)                                                                             \
            hipLaunchKernelGGL((k_ray_lin3_rot_pair_raw<VW, MPV, 1>), pgrid, block, plds, stream, a);   \
        else                                                                                            \
            hipLaunchKernelGGL((k_ray_lin3_rot_pair_raw<VW, MPV, 0>), pgrid, block, plds, stream, a);   \
    } while (0)
            if (c.ray.var_is_w) {
                if (mp)
                    V1C_ROTPAIR(1, 1);
                else
                    V1C_ROTPAIR(1, 0);
            } else {
                if (mp)
                    V1C_ROTPAIR(0, 1);
                else
                    V1C_ROTPAIR(0, 0);
            }
#undef V1C_ROTPAIR
            return;
        }
    }
    // Only combinations a plan can select are instantiated: the lean batch kernel and the tile-list form exist for
    // bilinear plans with boxes; launches without boxes (units that override the rotation) always rotate.
#ifdef V1C_TUNING
#define V1C_LEAN_STAGED(VW, RT, OW) hipLaunchKernelGGL((k_ray_lin3_batch_lean<VW, RT, OW>), merged ? merged_grid : grid, block, lean_lds, stream, la)
#else
#define V1C_LEAN_STAGED(VW, RT, OW) (void)0
#endif
#define V1C_TILE_P(VW, RT, BX, OW, PR)                                                                                                \
    do {                                                                                                                              \
        if constexpr (K == 2 && BX == 1) {                                                                                            \
            if (lean) {                                                                                                               \
                /* (running the remaining tiles on a side stream, forked and joined with events so that their */                      \
                /* latency-bound kernel overlaps the lean one, measured 4 % slower on C3 than back to back) */                        \
                TileArgs la = a;                                                                                                      \
                la.half_dwords = lean_half, la.kb = lean_raw_nwp;                                                                     \
                la.rest_list = merged ? rest_list : (const uint32_t*)nullptr, la.n_rest = n_rest;                                     \
                if (lean_raw_nwp > 0)                                                                                                 \
                    hipLaunchKernelGGL((k_ray_lin3_batch_lean_raw<VW, RT, OW>), merged ? merged_grid : grid, block,                   \
                                       std::max(lean_lds, (size_t)V1C_LEAN_RING * 1024 * (size_t)lean_raw_nwp), stream, la);          \
                else                                                                                                                  \
                    V1C_LEAN_STAGED(VW, RT, OW);                                                                                      \
                if (n_rest > 0 && !merged) {                                                                                          \
                    TileArgs ra = a;                                                                                                  \
                    ra.upb = 2, ra.rest_list = rest_list, ra.strip_len = ra.strip_magic = 0u;                                         \
                    hipLaunchKernelGGL((k_ray_lin3_tile<VW, RT, 1, K, OW, 1, 1>), rest_grid, block, lds, stream, ra);                 \
                }                                                                                                                     \
                break;                                                                                                                \
            }                                                                                                                         \
        }                                                                                                                             \
        if constexpr (K == 2) {                                                                                                       \
            if (nn) {                                                                                                                 \
                hipLaunchKernelGGL((k_ray_lin3_tile<VW, RT, BX, K, OW, PR, 0, 1>), grid, block, lds, stream, a);                      \
                break;                                                                                                                \
            }                                                                                                                         \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_ray_lin3_tile<VW, RT, BX, K, OW, PR>), grid, block, lds, stream, a);                                    \
    } while (0)
#ifdef V1C_TUNING
    constexpr bool kBatchLoop = true;
#else
    constexpr bool kBatchLoop = false;  // (launches with boxes always run the pair instantiation: see `upb` above)
#endif
#define V1C_TILE_O(VW, RT, BX, OW)                       \
    do {                                                 \
        if (pair || (BX && lean))  /* (a lean launch: decided inside V1C_TILE_P) */ \
            V1C_TILE_P(VW, RT, BX, OW, (BX || !OW) ? 1 : 0); \
        else if constexpr (BX == 0 || kBatchLoop)        \
            V1C_TILE_P(VW, RT, BX, OW, 0);               \
    } while (0)
#define V1C_TILE(VW, RT)                \
    do {                                \
        if (bx && shared_entry)         \
            V1C_TILE_O(VW, RT, 1, 0);   \
        else if (bx)                    \
            V1C_TILE_O(VW, RT, 1, 1);   \
        else if (shared_entry)          \
            V1C_TILE_O(VW, 1, 0, 0);    \
        else                            \
            V1C_TILE_O(VW, 1, 0, 1);    \
    } while (0)
    if (!bx)
        use_rot = true;  // (units override the rotation: plan.hip passes any_rot || has_rot)
    if (c.ray.var_is_w) {
        if (use_rot)
            V1C_TILE(1, 1);
        else
            V1C_TILE(1, 0);
    } else {
        if (use_rot)
            V1C_TILE(0, 1);
        else
            V1C_TILE(0, 0);
    }
#undef V1C_TILE
#undef V1C_TILE_O
#undef V1C_TILE_P
#undef V1C_LEAN_STAGED
}

// `boxes` may be null (the units override the rotation): then boxes are reduced in-kernel.
// (A persistent variant keeping OpenCV's 128 KB Lanczos4 weight table in LDS was tried: with one
// 512-thread workgroup per CU it cannot hide LDS latency and its 128-byte weight rows land on 8
// banks -- 6x slower than reading the weights through L2.  See DESIGN.md 4.5.)
// `shared_entry`: no lane needs more than pixel 1's table entry (proved by the caller).
// `mpoly_all` (boxes == null only): the m-polynomial table is valid on every interval these units reach.
// `rest_list` / `n_rest` / `lean_half` (boxes != null; list may be null): device copy of tile_rest_list() and the box
// buffer size (dwords) it was made for.
// `flags`: the plan's tile-flag words when a fix-up pass follows this launch, else null.
hipError_t launch_ray_lin3_tile(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const void* boxes,
                                int half_dwords, bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half,
                                int strip_len, int lean_raw_nwp, hipStream_t stream, bool coords_bounded)
{
    const TileBox* bx = (const TileBox*)boxes;
    switch (taps_of(c.g.interp)) {
    case 2: launch_tile_k<2>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded); break;
    case 4: launch_tile_k<4>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded); break;
    case 8: launch_tile_k<8>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace v1c
