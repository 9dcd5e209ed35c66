// kernels_tile.hip -- LDS-tiled kernel for the hot configuration (BGR, INTER_LINEAR,
// BORDER_CONSTANT, fused ray path).  Same arithmetic as kernels.hip / kernels_fast.hip.
//
// Why: rocprof counters on the gather-from-global version show the vector L1 (TCP) as the limit:
// it retires about one 64-byte access per clock, and a wave's 64 unaligned 8-byte gathers cost
// ~100 accesses per load instruction (no coalescing), a dwordx4 table read 16.  Here
//   * a workgroup owns a 64 x 16 output tile (lane = 4 px of one row, wave = 64 px x 4 rows);
//   * the bounding box of the tile's source taps is reduced across the workgroup (packed int16
//     min/max), then copied from HBM to LDS with 16-byte, row-contiguous, dword-aligned loads
//     (a 1 KiB wave load = 16 L1 accesses for ~340 source pixels);
//   * the 2x2 cells are read back from LDS with unaligned ds_read_b64 (12-byte lane stride is
//     coprime with the 64 banks);
//   * each lane reads ONE 64-byte radial-table entry for its 4 pixels: intervals whose polynomial
//     was validated at plan time on the 3x wider range carry a flag in the LSB of c7.
// Tiles whose box does not fit the LDS budget (strong rotation / minification) gather from global
// memory like kernels_fast.hip; pixels with taps outside the source go through the generic
// border-aware sampler; pixels outside the radial table's domain are left to the fix-up launch.
#include "kernels.hpp"

namespace v1c {

constexpr int kTW = 64, kTH = 16;          // output tile (px)
constexpr int kBoxBytes = 16 * 1024;       // LDS budget for the source box
constexpr int kMaxCpr = 64;                // 16-byte chunks per box row (magic division bound)

typedef short __attribute__((ext_vector_type(2))) s16x2;

__device__ __forceinline__ int pk_min(int a, int b)
{
    return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}

__device__ __forceinline__ int wave_pk_min(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1)
        v = pk_min(v, __shfl_xor(v, m));
    return v;
}

__device__ __noinline__ uint32_t slow_pixel_linear3_t(const uint8_t* src, int64_t pitch, int h, int w, Geom g, float x, float y)
{
    uint8_t px[3] = {0, 0, 0};
    const Image im{src, pitch, h, w};
    sample_linear<3>(im, g, x, y, px);
    return (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
}

struct u128 {
    uint32_t x, y, z, w;
};

template <int VAR_W, int ROT>
__global__ __launch_bounds__(256) void k_ray_lin3_tile(KernelCtx c, UnitArgs ua)
{
    __shared__ int red[8];
    __shared__ __attribute__((aligned(16))) uint8_t box[kBoxBytes + 16];

    const RayParams& P = c.ray;
    const Geom& g = c.g;
    const int z = blockIdx.z;
    const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
    const int x0 = (blockIdx.x * 16 + lx) * kPX;
    const int j = blockIdx.y * kTH + ly;
    const bool active = x0 < g.dst_w && j < g.dst_h;
    const int xc = min(x0, ((g.dst_w + 3) & ~3) - 4), jc = min(j, g.dst_h - 1);  // clamped for table reads
    // flag words are indexed like kernels.hip's 256x4 tiles so that MODE_FIXUP finds them
    const int tiles_x = (g.dst_w + kBlockX * kPX - 1) / (kBlockX * kPX), tiles_y = (g.dst_h + kBlockY - 1) / kBlockY;
    const int tile = (z * tiles_y + jc / kBlockY) * tiles_x + xc / (kBlockX * kPX);

    const uint8_t* __restrict__ src = ua.u[z].src;
    const uint32_t spitch = (uint32_t)ua.u[z].src_pitch;
    const double sl = P.row_s[jc], cl = P.row_c[jc], hl = P.row_h[jc];
    const double rx32 = 32.0 * P.rx, ry32 = 32.0 * P.ry, cx32 = 32.0 * P.cx, cy32 = 32.0 * P.cy;

    double A0 = 0, A1 = 0, A2 = 0, B0 = 0, B1 = 0, B2 = 0, C0 = 0, C1 = 0, C2 = 0;
    if (ROT) {
        double R[9];
#pragma unroll
        for (int q = 0; q < 9; q++)
            R[q] = ua.u[z].has_rot ? ua.u[z].rot[q] : P.rot[q];
        A0 = R[0] * cl, B0 = R[2] * cl, C0 = R[1] * sl;
        A1 = R[3] * cl, B1 = R[5] * cl, C1 = R[4] * sl;
        A2 = R[6] * cl, B2 = R[8] * cl, C2 = R[7] * sl;
    }

    // ---- column tables -> table variable of the 4 pixels ----
    double slon[kPX], qlon[kPX];
    {
        const double* __restrict__ ps = P.col_s + xc;
        const double* __restrict__ pq = (ROT ? P.col_c : P.col_h) + xc;
#pragma unroll
        for (int k = 0; k < kPX; k++)
            slon[k] = ps[k], qlon[k] = pq[k];
    }
    double vx[kPX], vy[kPX], tt[kPX];
    int idx[kPX];
    unsigned in_table = 0;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        double m;
        if (ROT) {
            vx[k] = fma(A0, slon[k], fma(B0, qlon[k], C0));
            vy[k] = fma(A1, slon[k], fma(B1, qlon[k], C1));
            m = 1.0 - fma(A2, slon[k], fma(B2, qlon[k], C2));
        } else {
            vx[k] = cl * slon[k];
            vy[k] = sl;
            m = fma(cl, qlon[k], hl);
        }
        const double u = VAR_W ? fast_sqrt_half(m) : m;
        tt[k] = u * P.inv_step;
        const bool in = tt[k] >= 0.0 && tt[k] < (double)P.n_int;
        in_table |= in ? 1u << k : 0u;
        idx[k] = in ? (int)tt[k] : 0;
    }

    // ---- radial table: one entry (that of pixel 1) serves all 4 pixels where it may ----
    double G[kPX];
    {
        const int ic = idx[1];
        double e[kRadialCoefs];
        {
            typedef double __attribute__((ext_vector_type(2))) d2;
            const d2* p2 = (const d2*)(P.radial + (size_t)ic * kRadialCoefs);
#pragma unroll
            for (int q = 0; q < kRadialCoefs / 2; q++) {
                const d2 v = p2[q];
                e[2 * q] = v.x, e[2 * q + 1] = v.y;
            }
        }
        const bool ext = (__double2loint(e[kRadialDegree]) & 1) != 0;  // validated on |z| <= 1.5
        unsigned own = 0;  // pixels that must use their own entry
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const double zk = tt[k] - ((double)ic + 0.5);
            const bool usec = idx[k] == ic || (ext && fabs(zk) <= 1.5);
            own |= (!usec && ((in_table >> k) & 1)) ? 1u << k : 0u;
            double gk = e[kRadialDegree];
#pragma unroll
            for (int q = kRadialDegree - 1; q >= 0; q--)
                gk = fma(gk, zk, e[q]);
            G[k] = gk;
        }
        if (own) {
#pragma unroll
            for (int k = 0; k < kPX; k++) {
                if (own & (1u << k)) {
                    const double* __restrict__ pc = P.radial + (size_t)idx[k] * kRadialCoefs;
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

    const int npx = active ? min(kPX, g.dst_w - x0) : 0;
    int sx[kPX], sy[kPX];
    float fxs[kPX], fys[kPX];
    unsigned ok = 0;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const double x32 = fma(G[k] * rx32, vx[k], cx32), y32 = fma(G[k] * ry32, vy[k], cy32);
        fxs[k] = (float)x32, fys[k] = (float)y32;  // = 32 * float32(x)
        const bool good = ((in_table >> k) & 1) && fabs(x32) < 1073741824.0 && fabs(y32) < 1073741824.0;
        ok |= (good && k < npx) ? 1u << k : 0u;
        sx[k] = good ? __float2int_rn(fxs[k]) : 0;
        sy[k] = good ? __float2int_rn(fys[k]) : 0;
    }
    if (ok != (1u << npx) - 1)
        c.tile_flags[tile] = 1;

    // ---- bounding box of the fully-inside 2x2 cells ----
    unsigned inside = 0;
    int xmn = 32767, ymn = 32767, nxmx = 32767, nymx = 32767;  // running mins of x, y, -x, -y
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const int ix = sx[k] >> 5, iy = sy[k] >> 5;
        const bool in = ((ok >> k) & 1) && (unsigned)ix < (unsigned)(g.src_w - 2) && (unsigned)iy < (unsigned)(g.src_h - 1);
        inside |= in ? 1u << k : 0u;
        xmn = in ? min(xmn, ix) : xmn, ymn = in ? min(ymn, iy) : ymn;
        nxmx = in ? min(nxmx, -ix) : nxmx, nymx = in ? min(nymx, -iy) : nymx;
    }
    int pa = (xmn & 0xffff) | (ymn << 16), pb = (nxmx & 0xffff) | (nymx << 16);
    pa = wave_pk_min(pa), pb = wave_pk_min(pb);
    if ((tid & 63) == 0)
        red[(tid >> 6) * 2] = pa, red[(tid >> 6) * 2 + 1] = pb;
    __syncthreads();
    pa = pk_min(pk_min(red[0], red[2]), pk_min(red[4], red[6]));
    pb = pk_min(pk_min(red[1], red[3]), pk_min(red[5], red[7]));
    pa = __builtin_amdgcn_readfirstlane(pa), pb = __builtin_amdgcn_readfirstlane(pb);
    const int bx0 = (short)(pa & 0xffff), by0 = pa >> 16;
    const int bx1 = -(int)(short)(pb & 0xffff), by1 = -(pb >> 16);

    // box rows hold source bytes [a0, a0 + cpr*16) of rows by0 .. by1+1
    const int a0 = (bx0 * 3) & ~3;
    const int cpr = (bx1 * 3 + 8 - a0 + 15) >> 4;
    const int nrows = by1 - by0 + 2;
    const int lp = cpr * 16;
    const bool any_inside = bx0 <= bx1;
    const bool use_lds = any_inside && cpr <= kMaxCpr && nrows * lp <= kBoxBytes && ((((uintptr_t)src) | spitch) & 3) == 0;

    uint32_t alo[kPX], ahi[kPX], blo[kPX], bhi[kPX];
    if (use_lds) {
        const int nchunks = nrows * cpr;  // <= 1024
        const uint32_t magic = (65536u + cpr - 1) / cpr;
        const uint32_t src_bytes = (uint32_t)(g.src_h - 1) * spitch + (uint32_t)g.src_w * 3u;
        u128 v[4];
        uint32_t lds_off[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ch = tid + q * 256;
            const uint32_t r = ((uint32_t)ch * magic) >> 16, col = ch - r * cpr;
            const uint32_t goff = (uint32_t)(by0 + r) * spitch + (uint32_t)a0 + col * 16u;
            lds_off[q] = r * lp + col * 16;
            if (ch < nchunks) {
                if (goff + 16u <= src_bytes) {
                    typedef u128 __attribute__((aligned(4), may_alias)) u128a4;
                    v[q] = *(const u128a4*)(src + goff);
                } else {  // last bytes of the image: never read past the allocation
                    uint32_t w[4] = {0, 0, 0, 0};
                    for (int b = 0; b < 16; b++)
                        if (goff + b < src_bytes)
                            w[b >> 2] |= (uint32_t)src[goff + b] << (8 * (b & 3));
                    v[q] = u128{w[0], w[1], w[2], w[3]};
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (tid + q * 256 < nchunks)
                *(u128*)(box + lds_off[q]) = v[q];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int ix = sx[k] >> 5, iy = sy[k] >> 5;
            const bool in = (inside >> k) & 1;
            const uint32_t lo = in ? (uint32_t)((iy - by0) * lp + ix * 3 - a0) : 0u;
            const u64pair a = load_u64_unaligned(box + lo);
            const u64pair b = load_u64_unaligned(box + lo + lp);
            alo[k] = a.lo, ahi[k] = a.hi, blo[k] = b.lo, bhi[k] = b.hi;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int ix = sx[k] >> 5, iy = sy[k] >> 5;
            const bool in = (inside >> k) & 1;
            const uint32_t off = in ? __umul24(iy, spitch) + (uint32_t)(ix * 3) : 0u;
            const u64pair a = load_u64_unaligned(src + off);
            const u64pair b = load_u64_unaligned(src + off + spitch);
            alo[k] = a.lo, ahi[k] = a.hi, blo[k] = b.lo, bhi[k] = b.hi;
        }
    }

    uint32_t pix[kPX];
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        const uint32_t fq = sx[k] & 31, fr = sy[k] & 31;
        const uint32_t wxp = (32u - fq) | (fq << 8);
        const uint32_t wy0 = 32u - fr, wy1 = fr;
        uint32_t o = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const uint32_t sel = 0x0c0c0000u | ((uint32_t)(ch + 3) << 8) | (uint32_t)ch;
            const uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(ahi[k], alo[k], sel), wxp, 0u, false);
            const uint32_t h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(bhi[k], blo[k], sel), wxp, 0u, false);
            const uint32_t v = __umul24(h0, wy0) + __umul24(h1, wy1) + 512u;
            o |= (v >> 10) << (8 * ch);
        }
        pix[k] = o;
    }

    const unsigned slow = ok & ~inside;
    if (slow) {
#pragma unroll
        for (int k = 0; k < kPX; k++)
            if (slow & (1u << k))
                pix[k] = slow_pixel_linear3_t(src, ua.u[z].src_pitch, g.src_h, g.src_w, g, fxs[k] * 0.03125f, fys[k] * 0.03125f);
    }

    if (!active)
        return;
    uint8_t* drow = ua.u[z].dst + (int64_t)j * ua.u[z].dst_pitch + (int64_t)x0 * 3;
    if (ok == 0xFu && (((uintptr_t)drow) & 3) == 0) {
        uint32_t* d32 = (uint32_t*)drow;
        d32[0] = pix[0] | (pix[1] << 24);
        d32[1] = (pix[1] >> 8) | (pix[2] << 16);
        d32[2] = (pix[2] >> 16) | (pix[3] << 8);
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

hipError_t launch_ray_lin3_tile(const KernelCtx& c, const UnitArgs& ua, int n_units, bool use_rot, hipStream_t stream)
{
    const dim3 block(256, 1, 1);
    const dim3 grid((c.g.dst_w + kTW - 1) / kTW, (c.g.dst_h + kTH - 1) / kTH, n_units);
    if (c.ray.var_is_w) {
        if (use_rot)
            hipLaunchKernelGGL((k_ray_lin3_tile<1, 1>), grid, block, 0, stream, c, ua);
        else
            hipLaunchKernelGGL((k_ray_lin3_tile<1, 0>), grid, block, 0, stream, c, ua);
    } else {
        if (use_rot)
            hipLaunchKernelGGL((k_ray_lin3_tile<0, 1>), grid, block, 0, stream, c, ua);
        else
            hipLaunchKernelGGL((k_ray_lin3_tile<0, 0>), grid, block, 0, stream, c, ua);
    }
    return hipGetLastError();
}

}  // namespace v1c
