// kernels_cn.hip -- grayscale and BGRA (cn = 1 / 4) through the tile machinery (k_ray_lin_cn: bilinear, nearest, bicubic, Lanczos4;
// plan-time boxes or, for units with a rotation of their own, boxes reduced in the kernel), and the upload of long unit lists into the
// plan's device ring (k_put_units).  Building blocks: tile_device.hpp.
#include "tile_device.hpp"

namespace v1c {

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

// slow_pixel_table3_t for CN = 1 / 4 channels: the pixels whose K x K footprint leaves the source (border rules; BORDER_TRANSPARENT:
// bit 32 of the result when the centre tap is outside); the row loop rolled, a row's taps in registers (sample_table<CN, K> of v1c_core.hpp
// keeps its tap columns in a local array: scratch)
template <int CN, int K>
// Returns the pixel in the low dword; bit 32: leave the destination untouched.  (By value: a reference parameter of a non-inlined
// function is a stack slot -- scratch memory for the whole kernel.)
__device__ __noinline__ uint64_t slow_pixel_table_cn(const uint8_t* src, int64_t pitch, int h, int w, int border, uint32_t cval, const short* itab,
                                                     int fsx, int fsy)
{
    const Taps t = taps_from_fixed(fsx, fsy);
    const short* __restrict__ wt = itab + (size_t)(t.fy * 32 + t.fx) * (K * K);
    constexpr int off = K / 2 - 1;
    const int sx = t.ix - off, sy = t.iy - off;
    if (border == V1C_BORDER_TRANSPARENT) {  // centre tap outside the source: the destination keeps its bytes; else REFLECT_101
        if ((unsigned)t.ix >= (unsigned)w || (unsigned)t.iy >= (unsigned)h)
            return 1ull << 32;
        border = V1C_BORDER_REFLECT_101;
    }
    if (border == V1C_BORDER_CONSTANT && (sx >= w || sx + K <= 0 || sy >= h || sy + K <= 0))  // footprint entirely outside
        return CN == 1 ? (cval & 255u) : cval;
    int acc[CN];
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        acc[ch] = 1 << 14;
    // (column indices first, then a row's K taps requested together: one memory round trip per row instead of one per tap --
    //  slow_pixel_table3_t, tile_device.hpp)
    typedef const __attribute__((address_space(1))) uint8_t* g8;
    typedef const __attribute__((address_space(1))) short* g16;
    int xo[K];  // byte offset of tap j in a row; -1: outside under BORDER_CONSTANT
#pragma unroll
    for (int j = 0; j < K; j++) {
        const int xj = border_index(sx + j, w, border);
        xo[j] = xj < 0 ? -1 : xj * CN;
    }
    const g16 wg = (g16)wt;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
        const int yi = border_index(sy + i, h, border);  // -1: outside under BORDER_CONSTANT
        const g8 S = (g8)src + (int64_t)(yi < 0 ? 0 : yi) * pitch;
        uint32_t d[K];  // a tap's CN bytes (BGRA: one dword load, at whatever alignment the source has)
        short wv[K];
        typedef uint32_t __attribute__((aligned(1), may_alias)) u32_u;
#pragma unroll
        for (int j = 0; j < K; j++) {
            const g8 p = S + (xo[j] < 0 ? 0 : xo[j]);
            if constexpr (CN == 4)
                d[j] = *(const __attribute__((address_space(1))) u32_u*)p;
            else
                d[j] = p[0];
            wv[j] = wg[i * K + j];
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint32_t v = ((yi >= 0) & (xo[j] >= 0)) ? d[j] : cval;
#pragma unroll
            for (int ch = 0; ch < CN; ch++)
                acc[ch] += (int)((v >> (8 * ch)) & 255u) * (int)wv[j];
        }
    }
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        out |= fixpt_u8(acc[ch]) << (8 * ch);
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
        out |= fixpt_u8(acc[ch]) << (8 * ch);
    return out;
}

// The same for TWO units that share the map (both eyes of a pair, consecutive frames of a batch): their taps sit at the same offset of
// their box buffers and take the SAME weights, so the weight rows -- 32 / 128 B per pixel from L2, what the gray / BGRA K x K launches
// were bound by (gray Lanczos4 pairs ran slower than BGR ones with a third of the multiplies) -- are fetched once for both.
template <int CN, int K>
__device__ __forceinline__ void blend_table_cn_pair(uint32_t a0, uint32_t a1, uint32_t pitch, glb_u32_ptr w, uint32_t& out0, uint32_t& out1)
{
    int acc0[CN], acc1[CN];
#pragma unroll
    for (int ch = 0; ch < CN; ch++)
        acc0[ch] = acc1[ch] = 1 << 14;
#pragma unroll
    for (int r = 0; r < K; r++) {
        const uint32_t ar0 = a0 + (uint32_t)r * pitch, ar1 = a1 + (uint32_t)r * pitch;
        const lds_u32_ptr p0 = (lds_u32_ptr)(uintptr_t)(ar0 & ~3u), p1 = (lds_u32_ptr)(uintptr_t)(ar1 & ~3u);
        uint32_t wr[K / 2];
#pragma unroll
        for (int q = 0; q < K / 2; q++)
            wr[q] = w[r * (K / 2) + q];
        if constexpr (CN == 4) {
            uint32_t d0[K], d1[K];
#pragma unroll
            for (int q = 0; q < K; q++)
                d0[q] = p0[q], d1[q] = p1[q];
#pragma unroll
            for (int q = 0; q < K / 2; q++) {
                const short2v ww = __builtin_bit_cast(short2v, wr[q]);
                acc0[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d0[2 * q + 1], d0[2 * q], 0x0c040c00u)), ww, acc0[0], false);
                acc0[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d0[2 * q + 1], d0[2 * q], 0x0c050c01u)), ww, acc0[1], false);
                acc0[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d0[2 * q + 1], d0[2 * q], 0x0c060c02u)), ww, acc0[2], false);
                acc0[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d0[2 * q + 1], d0[2 * q], 0x0c070c03u)), ww, acc0[3], false);
                acc1[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d1[2 * q + 1], d1[2 * q], 0x0c040c00u)), ww, acc1[0], false);
                acc1[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d1[2 * q + 1], d1[2 * q], 0x0c050c01u)), ww, acc1[1], false);
                acc1[2] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d1[2 * q + 1], d1[2 * q], 0x0c060c02u)), ww, acc1[2], false);
                acc1[3] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(d1[2 * q + 1], d1[2 * q], 0x0c070c03u)), ww, acc1[3], false);
            }
        } else {
            uint32_t d0[K / 4 + 1], d1[K / 4 + 1], b0[K / 4], b1[K / 4];
#pragma unroll
            for (int q = 0; q < K / 4 + 1; q++)
                d0[q] = p0[q], d1[q] = p1[q];
#pragma unroll
            for (int q = 0; q < K / 4; q++) {
                b0[q] = __builtin_amdgcn_alignbyte(d0[q + 1], d0[q], ar0);  // bytes 4 q .. 4 q + 3 of the row
                b1[q] = __builtin_amdgcn_alignbyte(d1[q + 1], d1[q], ar1);
            }
#pragma unroll
            for (int q = 0; q < K / 2; q++) {
                const short2v ww = __builtin_bit_cast(short2v, wr[q]);
                const uint32_t sel = (q & 1) ? 0x0c030c02u : 0x0c010c00u;
                acc0[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, b0[q / 2], sel)), ww, acc0[0], false);
                acc1[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, b1[q / 2], sel)), ww, acc1[0], false);
            }
        }
    }
    out0 = out1 = 0;
#pragma unroll
    for (int ch = 0; ch < CN; ch++) {
        out0 |= fixpt_u8(acc0[ch]) << (8 * ch);
        out1 |= fixpt_u8(acc1[ch]) << (8 * ch);
    }
}

// K = 2: bilinear, or nearest with NN = 1 (lane_coords<..., NN>: fixed point 32 * cvRound(x), fractions zero, for which the blend returns its
// top-left tap exactly); K = 4 / 8: bicubic / Lanczos4 (blend_table_cn).  Every border mode, BORDER_TRANSPARENT included: the pixels a
// box cannot serve take the border-aware per-pixel samplers (slow_pixel_table_cn, sample_linear_t / sample_nearest), which apply the
// skip rule of the interpolation and report it to the store mask (skip bits).
// BOXES = 1: plan-time boxes, one workgroup per tile walking all units of the launch (they share the map) -- one at a time with the
// next unit's box in flight (K = 2), two at a time against one fetch of the weight rows (K = 4 / 8: blend_table_cn_pair).
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
    constexpr bool kPairs = K > 2 && BOXES;  // K x K taps: the units in pairs, both boxes resident (see the unit loop)
    if (fits) {  // the first unit's box flies behind the coordinates (K x K: the first two units' boxes)
        raw_box_dma<CN>(b, m, U[u0].src, (uint32_t)U[u0].src_pitch, lane, wave, lds0);
        if (kPairs && u0 + 1 < n_units)
            raw_box_dma<CN>(b, m, U[u0 + 1].src, (uint32_t)U[u0 + 1].src_pitch, lane, wave, lds0 + buf_bytes);
    }
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
    if constexpr (kPairs) {
        // pixels the boxes did not serve (footprints leaving the source, a box beyond the buffers), then the store of unit u
        // (a lambda of this block only: shared with the loop below it cost the bilinear instantiations 14 VGPRs and a wave per SIMD)
        auto patch_and_store_cn = [&](int u, uint32_t (&pix)[kPX], unsigned done) {
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
                            const uint64_t rr = slow_pixel_table_cn<CN, K>(im.p, im.pitch, im.h, im.w, gg.border,
                                                                           (uint32_t)gg.cval[0] | ((uint32_t)gg.cval[1] << 8) | ((uint32_t)gg.cval[2] << 16) | ((uint32_t)gg.cval[3] << 24),
                                                                           c.itab, fsx, fsy);
                            r = (uint32_t)rr;
                            skip |= (uint32_t)(rr >> 32) << k;
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
        };
        // K x K taps: units u, u + 1 from the two buffers against ONE fetch of each pixel's weight rows.  Per pair: (everyone is done
        // with the previous pair's buffers -> ) requests of both boxes -> all landed -> gather.  No box is in flight during a gather;
        // the other workgroups of the CU cover that latency (64 / 16 taps per pixel keep a workgroup busy for long).
        for (int u = u0; u < n_units; u += 2) {
            const bool two = u + 1 < n_units;
            if (incomplete)
                if (uint32_t* flags = a.tile_flags) {
                    flags[t.flag_tile + u * t.flag_stride] = 1;
                    if (two)
                        flags[t.flag_tile + (u + 1) * t.flag_stride] = 1;
                }
            uint32_t p0[kPX] = {0u, 0u, 0u, 0u}, p1[kPX] = {0u, 0u, 0u, 0u};
            unsigned done = 0;
            if (fits) {
                if (u > u0) {
                    wait_vm_barrier_imm<0>();  // every wave has its taps of the previous pair (their values went into its stores)
                    raw_box_dma<CN>(b, m, U[u].src, (uint32_t)U[u].src_pitch, lane, wave, lds0);
                    if (two)
                        raw_box_dma<CN>(b, m, U[u + 1].src, (uint32_t)U[u + 1].src_pitch, lane, wave, lds0 + buf_bytes);
                }
                wait_vm_barrier_imm<0>();
#pragma unroll 1
                for (int k = 0; k < kPX; k++) {
                    const uint32_t tk = k == 0 ? ta[0] : k == 1 ? ta[1] : k == 2 ? ta[2] : ta[3];
                    const uint32_t ek = k == 0 ? we[0] : k == 1 ? we[1] : k == 2 ? we[2] : we[3];
                    uint32_t r0, r1 = 0u;
                    if (two)
                        blend_table_cn_pair<CN, K>(tk + lds0, tk + lds0 + buf_bytes, lpitch, wtab + ek * (K * K / 2), r0, r1);
                    else
                        r0 = blend_table_cn<CN, K>(tk + lds0, lpitch, wtab + ek * (K * K / 2));
#pragma unroll
                    for (int q = 0; q < kPX; q++)
                        p0[q] = q == k ? r0 : p0[q], p1[q] = q == k ? r1 : p1[q];
                }
                done = L.inside;
            }
            patch_and_store_cn(u, p0, done);
            if (two)
                patch_and_store_cn(u + 1, p1, done);
        }
        return;
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
                        // (NN: remapNearest's border rules -- TRANSPARENT skips by the pixel itself, not by the bilinear footprint)
                        const bool st = NN ? sample_nearest<CN>(im, gg, (float)(fsx >> 5), (float)(fsy >> 5), px)
                                           : sample_linear_t<CN>(im, gg, taps_from_fixed(fsx, fsy), px);
                        r = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)px[3] << 24);
                        skip |= (st ? 0u : 1u) << k;
                    } else {
                        const uint64_t rr = slow_pixel_table_cn<CN, K>(im.p, im.pitch, im.h, im.w, gg.border,
                                                                       (uint32_t)gg.cval[0] | ((uint32_t)gg.cval[1] << 8) | ((uint32_t)gg.cval[2] << 16) | ((uint32_t)gg.cval[3] << 24),
                                                                       c.itab, fsx, fsy);
                        r = (uint32_t)rr;
                        skip |= (uint32_t)(rr >> 32) << k;
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

// Units of a launch longer than the kernel-argument block holds: copied into a slot of the plan's device ring by launches of their
// own (kPutUnits records each, carried in THEIR kernel arguments -- 3.6 KB of the 4 KB a launch may carry): stream-ordered,
// graph-capturable, no staging buffer to keep alive.  A 64-unit launch (BASELINE config 5 per GPU) costs two of them, ~9 us in front
// of a 2.3 ms launch.
constexpr int kPutUnits = 32;
struct PutArgs {
    DevUnit u[kPutUnits];
};
// the kernel-argument segment of k_put_units as the ABI lays it out (natural alignment, declaration order): where the records start
struct PutKernargs {
    DevUnit* dst;
    PutArgs ua;
    int n;
};
__global__ __launch_bounds__(256) void k_put_units(DevUnit* dst, PutArgs ua_, int n)
{
    typedef const V1C_CONST uint32_t* cu32;
    const cu32 src = (cu32)((const V1C_CONST uint8_t*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(PutKernargs, ua));  // (behind `dst`)
    const int words = n * (int)(sizeof(DevUnit) / 4);
    for (int i = threadIdx.x; i < words; i += 256)
        ((uint32_t*)dst)[i] = src[i];
}

hipError_t launch_put_units(DevUnit* dst, const DevUnit* host, int n, hipStream_t stream)
{
    static_assert(sizeof(DevUnit) % 4 == 0 && alignof(PutArgs) == 8 && sizeof(PutArgs) + 16 <= 4096 && offsetof(PutKernargs, ua) == 8,
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

// k_ray_lin_cn: grayscale / BGRA; every interpolation with every border mode (no border test here on purpose: BORDER_TRANSPARENT's skip
// rules live in the kernel's patch path)
bool cn_kernel_supports(const Geom& g)
{
    return (g.cn == 1 || g.cn == 4) && taps_of(g.interp) != 0 &&
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


}  // namespace v1c
