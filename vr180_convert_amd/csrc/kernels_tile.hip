// kernels_tile.hip -- the general LDS-tiled kernels: the plan-time box pass (k_tile_boxes), one tile for the units that share a map
// (k_ray_lin3_tile: pairs, bicubic / Lanczos4, NEAREST, rest tiles), batches with LDS-DMA boxes (k_ray_lin3_batch_lean_raw) and units
// with a rotation of their own (k_ray_lin3_rot_pair_raw), with their launchers and the host-side sizing of the plan (box buffers, XCD
// strips, rest lists).  Building blocks: tile_device.hpp.  Same arithmetic as kernels.hip: the tests compare both with the oracle bit
// for bit.
#include "tile_device.hpp"

namespace v1c {

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
    // bicubic / Lanczos4 boxes of BGR plans with BORDER_CONSTANT cover the footprints that cross the edge of the source too (the tile kernel
    // stages the border colour around the image: stage_load_ext); the cn kernels' raw LDS-DMA boxes cannot, their plans keep the strict boxes
    const int ext = (K != 2 && c.g.cn == 3) ? kxk_ext(c.g) : 0;
    LaneCoords L;
    lane_coords<VAR_W, ROT, K, OWN, 0, 0, 0, NN>(c, c.ray.rot, rc, t.npx, c.ray.radial, 0, c.ray.n_int, L, ext);
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
        lane_coords<VAR_W, ROT, K, 0, 0, 1, 0, NN>(c, c.ray.rot, rc, t.npx, c.ray.radial_m, 0, c.ray.n_int, L2, ext);
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

// Bicubic / Lanczos4 without plan-time boxes for units that all carry ONE rotation (radius = "auto" on the device, v1c_plan_run_auto):
// a workgroup serves the tile for units 2 z and 2 z + 1 with one evaluation of the coordinates, one reduced box and one fetch of each
// pixel's weight rows (rot_shared_pair_tile).  Launched for an even number of units.
template <int VAR_W, int K, int OWN>
__global__ __launch_bounds__(256) void k_ray_kxk_auto_pair(TileArgs a_)
{
    __shared__ __attribute__((aligned(16))) int red[16];
    __shared__ __attribute__((aligned(16))) uint32_t boxw[2 * (kBoxBytes / 4) + 4 + kKxkExchangeBytes / 4];
    args_cref a = kernel_args();
    const glb_u32_ptr wtab = (glb_u32_ptr)args_ctx(a).itab;
    rot_shared_pair_tile<VAR_W, K, OWN>(a, 2 * (int)blockIdx.z, 2 * (int)blockIdx.z + 1, (int)blockIdx.x, (int)blockIdx.y, red, boxw, wtab);
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

// (kernels_mirror.hip: round 3 also built k_ray_lin3_pair_mirror_pipe -- two tile rows per workgroup, the second pair's boxes requested into the
// buffers the first pair had just been sampled from -- bit-exact and 5 % slower than one pair per workgroup at equal occupancy
// (profiles/r03a_mid, HISTORY.md 4.4c): "the workgroups do not wait for their boxes".  Removed in round 4 with its A/B switch.)

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

#ifndef V1C_LEAN_SYS_STORE
// the batch kernel's stores at system scope like the mirror pair kernels' (store4<SYS = 1>: written through to memory, the two halves of a
// 128-byte line two tiles share no longer leave L2 twice): a 16-unit C3 launch writes 398.4 MB for its 398.1 MB of output instead of
// 422.9 MB, at 0.1780 against 0.1784 ms (profiles/r05c_mid/ab_c3_system_scope_stores.log; round 3 had measured the time only: "+-0")
#define V1C_LEAN_SYS_STORE 1
#endif
template <int VAR_W, int ROT, int OWN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ROT ? 4 : (OWN ? 5 : V1C_LEAN_WAVES), 8))) void k_ray_lin3_batch_lean_raw(TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];
    if (V1C_PRIO & 8)
        __builtin_amdgcn_s_setprio(3);
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
    if (V1C_PRIO & 8)
        __builtin_amdgcn_s_setprio(0);
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
        store_interior<V1C_LEAN_SYS_STORE>(U, z0 + u, t, pix);
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
#ifndef V1C_ROTPAIR_XCOLS
#define V1C_ROTPAIR_XCOLS 0  // 1 (A/B builds): grid columns padded to a multiple of 8, so that XCD x serves the tile COLUMNS x, x + 8, ...
                             // (vertically adjacent tiles -- which share their halo rows -- on one L2) without any index arithmetic
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
#if V1C_ROTPAIR_XCOLS
    if (btx * kTW >= a.dst_w)  // (the grid's columns are padded to a multiple of 8: see the launcher)
        return;
#endif
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
    const BoxAll baA = reduce_box_all_nofence<NT / 64>(LA, red, tid);
    fastA = raw_box(baA, zA, bA);
    if (fastA) {
        const RawLanes m = raw_lanes(bA.cpr, lane);
        raw_box_dma(bA, m, U[zA].src, (uint32_t)U[zA].src_pitch, lane, wave, lds0);
    }
    if (hasB) {
        if (a.same_rot) {
            // every unit of the launch carries the same rotation (radius = "auto" on the device: v1c_plan_run_auto): the two units of
            // the workgroup share the map -- one evaluation of the coordinates, one box geometry (wave-uniform branch)
            LB = LA;
            fastB = raw_box(baA, zB, bB);
        } else {
            lane_coords<VAR_W, 1, 2, 0, (NC ? 1 : 2), MP>(c, U[zB].rot, rc, kPX, MP ? P.radial_m : P.radial, 0, P.n_int, LB);
            fastB = raw_box(reduce_box_all_nofence<NT / 64>(LB, red + 16, tid), zB, bB);
        }
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


bool tile_kernel_supports(const Geom& g)
{
    // every border mode (the border only matters to pixels whose footprint leaves the source, and those take the per-pixel samplers
    // of the patch path) -- BORDER_TRANSPARENT with the skip rule of the interpolation: remapBilinear leaves a pixel untouched when its
    // 2 x 2 footprint is not fully inside, remapBicubic / remapLanczos4 when its centre tap is outside, remapNearest when the pixel
    // itself is (the NN = 1 kernels gather through the bilinear footprint, their patch path samples by remapNearest's rules)
    return g.cn == 3 && taps_of(g.interp) != 0 && g.src_w >= 3 && g.src_h >= 2;
}

size_t tile_box_bytes(const Geom& g)
{
    const dim3 d = tile_grid(g, tile_threads(g), 1);
    return (size_t)d.x * d.y * sizeof(TileBox);
}

// LDS dwords one box buffer must hold so that every stageable tile box of `host_boxes` fits
// (capped at kMaxHalfDwords: larger boxes gather from global memory)
constexpr int kMaxHalfDwords = 8192;  // 32 KB per buffer

// `max_chunks`: boxes of more chunks are left out (they gather from global memory): 1024 = one staging round; the pair code stages a
// second round (shared_map_tile) -- worth its LDS for bicubic / Lanczos4, whose global-memory path samples tap by tap (a lat_x Lanczos4
// pair: 1.09 -> 0.12 ms), not for bilinear launches, where the larger buffers cost every tile occupancy (the same pair bilinear:
// 0.050 -> 0.060 ms with them)
int tile_half_dwords(const void* host_boxes, size_t n_tiles, int max_chunks, bool occupancy_classes)
{
    const TileBox* b = (const TileBox*)host_boxes;
    int m = 256;
    size_t stageable = 0;
    for (size_t i = 0; i < n_tiles; i++) {
        if (b[i].cpr <= 0 || b[i].cpr > kMaxCpr || b[i].nrows * b[i].cpr > max_chunks)
            continue;
        const int need = b[i].nrows * (b[i].cpr * 4 + 4);
        if (need <= kMaxHalfDwords)
            m = std::max(m, need), stageable++;
    }
    m = (m + 3) & ~3;
    // Bilinear / nearest pairs: the pair code's 74 VGPRs allow six workgroups per CU, its LDS (two buffers of `half` dwords + 4 KB of
    // table) decides how many there are -- and the launch wants them more than it wants the largest boxes staged: a rotated 4096^2 pair
    // whose largest box needs 3 744 dwords (four workgroups) ran 0.0659 ms, 0.0617 with the buffers capped at 2 688 (six; the 3 % of
    // the tiles beyond them gather from global memory).  The smallest class that still stages 98 % of the tiles; chains whose boxes are
    // large throughout (lat_x, planar + rotation: capped they lose 6 - 10 %) keep the maximum.  profiles/r05g_prio/ab_half_cap.log
    if (occupancy_classes && stageable > 0) {
        static const int classes[] = {2880, 3568};  // six / five workgroups per CU: (160 KB / n - 4 160 B static) / 8
        for (int cap : classes) {
            if (cap >= m)
                break;
            size_t fit = 0;
            for (size_t i = 0; i < n_tiles; i++) {
                if (b[i].cpr <= 0 || b[i].cpr > kMaxCpr || b[i].nrows * b[i].cpr > max_chunks)
                    continue;
                fit += b[i].nrows * (b[i].cpr * 4 + 4) <= cap;
            }
            if (fit * 100 >= stageable * 98)
                return cap;
        }
    }
    return m;
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
        const bool stageable = b[i].cpr > 0 && b[i].cpr <= kMaxCpr && b[i].nrows * b[i].cpr <= 2048 && need <= half_dwords;
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
    // ... and then as much more as the CU holds the same number of workgroups with (two buffers + 4 KB of table each; six is what the
    // kernel's registers allow): the 2 % beyond the quantile ride the general pair code, a few of them less is free -- C3 with 10 instead
    // of 9 KB: 0.1765 -> 0.1745 ms, with 12 (five workgroups) 0.1814 (profiles/r05g_prio/ab_lean_raw_kb.log)
    auto per_cu = [](int k) { return std::min(163840 / (2 * 1024 * k + 4224), 6); };
    int largest = 0;
    for (int k = 0; k <= kRawMaxWavePasses; k++)
        if (hist[k])
            largest = k;
    size_t acc = 0;
    for (int k = 0; k <= kRawMaxWavePasses; k++) {
        acc += hist[k];
        if (acc * 100 >= n * 98) {
            int kk = std::max(k, 4);
            while (kk < largest && kk < kRawMaxWavePasses && per_cu(kk + 1) == per_cu(kk))
                kk++;
            return kk;
        }
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
    const bool gen = c.ray.gen_mode != 0;  // the general modes (lat_x, radial stages in front of the rotation): ROT = 2
    if (c.ray.var_is_w) {
        if (gen)
            V1C_BOXES(1, 2);
        else if (rot)
            V1C_BOXES(1, 1);
        else
            V1C_BOXES(1, 0);
    } else {
        if (gen)
            V1C_BOXES(0, 2);
        else if (rot)
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
                          int lean_raw_nwp, hipStream_t stream, bool coords_bounded, int* kind, bool same_rot)
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
    const bool gen = c.ray.gen_mode != 0;  // the general modes run the pair instantiation of k_ray_lin3_tile<..., ROT = 2> (plan-time boxes only)
    lean = lean && !gen;
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
    size_t lds = bx ? (size_t)half_dwords * 8 + 16 + (K != 2 && pair ? kKxkExchangeBytes : 0) + lds_pad : 0;
    // Lanczos4 pairs: at most three workgroups per CU.  Their gather keeps the LDS pipe busier than anything else in the engine (8 x 8 cells
    // of two eyes per pixel); a fourth resident workgroup -- which boxes below ~36 KB would admit -- only lengthens its queue: C2L 0.3050 ->
    // 0.2965 ms with the launch padded to three, C1L and C4 (boxes of 41 - 44 KB: three anyway) slower by 7 - 11 % at two, bicubic faster with
    // every workgroup it can get (profiles/r05g_prio/ab_lds_pad*.log).
#ifndef V1C_K8_MIN_LDS
#define V1C_K8_MIN_LDS 40960
#endif
    if (bx && K == 8 && pair)
        lds = std::max(lds, (size_t)V1C_K8_MIN_LDS);
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
    a.same_rot = (!bx && same_rot) ? 1 : 0;
    if constexpr (K != 2) {
        if (a.same_rot && n_units % 2 == 0 && c.g.interp != V1C_INTER_NEAREST) {
            const dim3 pgrid(grid.x, grid.y, (unsigned)(n_units / 2));
            if (c.ray.var_is_w) {
                if (shared_entry)
                    hipLaunchKernelGGL((k_ray_kxk_auto_pair<1, K, 0>), pgrid, block, 0, stream, a);
                else
                    hipLaunchKernelGGL((k_ray_kxk_auto_pair<1, K, 1>), pgrid, block, 0, stream, a);
            } else {
                if (shared_entry)
                    hipLaunchKernelGGL((k_ray_kxk_auto_pair<0, K, 0>), pgrid, block, 0, stream, a);
                else
                    hipLaunchKernelGGL((k_ray_kxk_auto_pair<0, K, 1>), pgrid, block, 0, stream, a);
            }
            return;
        }
    }
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
            if (kind)
                *kind = V1C_LAUNCH_ROT_PAIR;
            const dim3 pgrid(V1C_ROTPAIR_XCOLS ? ((grid.x + 7u) & ~7u) : grid.x, grid.y, (unsigned)((n_units + 1) / 2));
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
        if constexpr (K == 2 && BX == 1 && RT != 2) {                                                                                 \
            if (lean) {                                                                                                               \
                /* (running the remaining tiles on a side stream, forked and joined with events so that their */                      \
                /* latency-bound kernel overlaps the lean one, measured 4 % slower on C3 than back to back) */                        \
                if (kind)                                                                                                             \
                    *kind = V1C_LAUNCH_BATCH;                                                                                         \
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
    // (the general modes have no form without plan-time boxes: plan.hip sends units that override their rotation to the interpreter)
#define V1C_TILE_GEN(VW)              \
    do {                              \
        if (shared_entry)             \
            V1C_TILE_O(VW, 2, 1, 0);  \
        else                          \
            V1C_TILE_O(VW, 2, 1, 1);  \
    } while (0)
    if (c.ray.var_is_w) {
        if (gen && bx)
            V1C_TILE_GEN(1);
        else if (use_rot)
            V1C_TILE(1, 1);
        else
            V1C_TILE(1, 0);
    } else {
        if (gen && bx)
            V1C_TILE_GEN(0);
        else if (use_rot)
            V1C_TILE(0, 1);
        else
            V1C_TILE(0, 0);
    }
#undef V1C_TILE_GEN
#undef V1C_TILE
#undef V1C_TILE_O
#undef V1C_TILE_P
#undef V1C_LEAN_STAGED
}

// `boxes` may be null (the units override the rotation): then boxes are reduced in-kernel.
// (A persistent variant keeping OpenCV's 128 KB Lanczos4 weight table in LDS was tried: with one
// 512-thread workgroup per CU it cannot hide LDS latency and its 128-byte weight rows land on 8
// banks -- 6x slower than reading the weights through L2.  See HISTORY.md 4.5.)
// `shared_entry`: no lane needs more than pixel 1's table entry (proved by the caller).
// `mpoly_all` (boxes == null only): the m-polynomial table is valid on every interval these units reach.
// `rest_list` / `n_rest` / `lean_half` (boxes != null; list may be null): device copy of tile_rest_list() and the box
// buffer size (dwords) it was made for.
// `flags`: the plan's tile-flag words when a fix-up pass follows this launch, else null.
hipError_t launch_ray_lin3_tile(const KernelCtx& c, const KernelCtx* cdev, const LaunchUnits& lu, uint32_t* flags, bool use_rot, const void* boxes,
                                int half_dwords, bool shared_entry, bool mpoly_all, const uint32_t* rest_list, int n_rest, int lean_half,
                                int strip_len, int lean_raw_nwp, hipStream_t stream, bool coords_bounded, int* kind, bool same_rot)
{
    const TileBox* bx = (const TileBox*)boxes;
    if (kind)
        *kind = V1C_LAUNCH_TILE;  // (the general kernel, unless launch_tile_k picks the batch / rotation-pair kernel)
    switch (taps_of(c.g.interp)) {
    case 2: launch_tile_k<2>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded, kind, same_rot); break;
    case 4: launch_tile_k<4>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded, kind, same_rot); break;
    case 8: launch_tile_k<8>(c, cdev, lu, flags, use_rot, bx, half_dwords, shared_entry, mpoly_all, rest_list, n_rest, lean_half, strip_len, lean_raw_nwp, stream, coords_bounded, kind, same_rot); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace v1c
