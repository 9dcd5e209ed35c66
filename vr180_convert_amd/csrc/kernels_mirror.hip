// kernels_mirror.hip -- pairs (apply_lr) and single images of an unrotated chain: one workgroup serves a tile AND the band that
// mirrors it about the equator from one set of coordinates, the boxes brought in by LDS-DMA (k_ray_lin3_pair_mirror_seq: pairs, the
// eyes one after the other through two buffers; k_ray_lin3_pair_mirror_raw: single images; their A/B partners in the tuning build).
// Building blocks: tile_device.hpp.
#include "tile_device.hpp"

namespace v1c {

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

// NE = number of eyes (units) of the launch: 2 = apply_lr's pair; 1 = a single image (apply() of one image, BASELINE config 1):
// the same workgroup with two boxes instead of four
template <int VAR_W, int NE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(V1C_RAW_WAVES, 8))) void k_ray_lin3_pair_mirror_raw(V1C_MIRROR_HEAD, TileArgs a_)
{
    constexpr int NT = 256;
    __shared__ __attribute__((aligned(16))) double tabw[kTabSlice * kRadialCoefs];
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_box[];  // 2 NE raw boxes (or the general code's cell buffers)
    if (V1C_PRIO & 4)
        __builtin_amdgcn_s_setprio(3);
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
    if (V1C_PRIO & 4)
        __builtin_amdgcn_s_setprio(0);
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
    if (V1C_PRIO & 4)
        __builtin_amdgcn_s_setprio(3);
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
    if (V1C_PRIO & 4)
        __builtin_amdgcn_s_setprio(0);
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

}  // namespace v1c
