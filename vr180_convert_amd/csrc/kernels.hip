// kernels.hip -- gfx950 kernels of the fused remap engine.
//
// One launch covers up to kMaxUnitsPerLaunch independent units (eyes / frames) on grid.z.  A
// workgroup is 64 x 4 threads = 4 waves; every lane produces kPX horizontally adjacent output
// pixels of one row, so a wave writes 64*kPX*3 contiguous bytes with dword stores.
//
// Coordinate producers (template parameter MODE):
//   MODE_LITERAL  fp64 interpreter of the lowered chain (any lowerable chain)
//   MODE_RAY      separable tables + radial table.  Pixels outside the table's validated domain
//                 are NOT written; their tile is flagged and
//   MODE_FIXUP    (a second, normally empty launch) re-evaluates exactly those pixels with the
//                 interpreter.  Keeping the interpreter out of MODE_RAY keeps its register
//                 footprint small.
//   MODE_LUT      caller-supplied float32 maps (v1c_remap_lut)
// All of them feed the same cv2.remap-exact sampler (v1c_core.hpp).
#include "kernels.hpp"

namespace v1c {

// Per-unit data pulled out of the kernel-argument block into registers.  (Taking the address of
// anything inside UnitArgs would make the compiler spill the whole 2 KB argument block to scratch.)
struct UnitView {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_pitch, dst_pitch;
    double rot[9];
    bool has_rot;       // the unit overrides the chain's first rotation
    bool use_rot;       // a rotation applies in ray mode (unit's or the chain's composed one)
};

__device__ __forceinline__ UnitView load_unit(const UnitArgs& ua, const RayParams& rp, int z)
{
    UnitView v;
    v.src = ua.u[z].src;
    v.dst = ua.u[z].dst;
    v.src_pitch = ua.u[z].src_pitch;
    v.dst_pitch = ua.u[z].dst_pitch;
    v.has_rot = ua.u[z].has_rot != 0;
    v.use_rot = v.has_rot || rp.has_rot;
#pragma unroll
    for (int q = 0; q < 9; q++)
        v.rot[q] = v.has_rot ? ua.u[z].rot[q] : rp.rot[q];
    return v;
}

template <int MODE>
struct Coords;

template <>
struct Coords<MODE_LITERAL> {
    __device__ static unsigned eval(const KernelCtx& c, const UnitView& u, int x0, int j, float* fx, float* fy)
    {
        const double* rot = u.has_rot ? u.rot : nullptr;
        for (int k = 0; k < kPX; k++) {
            double x, y;
            eval_chain_literal(c.chain, rot, x0 + k, j, x, y);
            fx[k] = (float)x;  // astype(np.float32), remapper.py:58
            fy[k] = (float)y;
        }
        return (1u << kPX) - 1;
    }
};

template <>
struct Coords<MODE_RAY> {
    __device__ static unsigned eval(const KernelCtx& c, const UnitView& u, int x0, int j, float* fx, float* fy)
    {
        const RayParams& rp = c.ray;
        // the row index is wave-uniform (blockDim.x == 64): tell the compiler, so the row table
        // reads become scalar loads
        const int ju = __builtin_amdgcn_readfirstlane(j);
        const double sl = rp.row_s[ju], cl = rp.row_c[ju], hl = rp.row_h[ju];
        unsigned ok = 0;
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int i = min(x0 + k, c.g.dst_w - 1);
            double x, y;
            if (ray_eval(rp, u.use_rot, u.rot, sl, cl, hl, rp.col_s[i], rp.col_c[i], rp.col_h[i], x, y)) {
                fx[k] = (float)x;
                fy[k] = (float)y;
                ok |= 1u << k;
            }
        }
        return ok;
    }
};

template <>
struct Coords<MODE_FIXUP> {
    __device__ static unsigned eval(const KernelCtx& c, const UnitView& u, int x0, int j, float* fx, float* fy)
    {
        float rx[kPX], ry[kPX];
        const unsigned ok = Coords<MODE_RAY>::eval(c, u, x0, j, rx, ry);
        const double* rot = u.has_rot ? u.rot : nullptr;
        unsigned todo = ~ok & ((1u << kPX) - 1);
        for (int k = 0; k < kPX; k++) {
            if (todo & (1u << k)) {
                double x, y;
                eval_chain_literal(c.chain, rot, min(x0 + k, c.g.dst_w - 1), j, x, y);
                fx[k] = (float)x;
                fy[k] = (float)y;
            }
        }
        return todo;
    }
};

template <>
struct Coords<MODE_LUT> {
    __device__ static unsigned eval(const KernelCtx& c, const UnitView&, int x0, int j, float* fx, float* fy)
    {
        const float* xr = (const float*)((const char*)c.xmap + (int64_t)j * c.map_pitch);
        const float* yr = (const float*)((const char*)c.ymap + (int64_t)j * c.map_pitch);
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            const int i = min(x0 + k, c.g.dst_w - 1);
            fx[k] = xr[i];
            fy[k] = yr[i];
        }
        return (1u << kPX) - 1;
    }
};

template <int CN, int INTERP, int MODE>
__global__ __launch_bounds__(kBlockX* kBlockY) void k_remap(KernelCtx c, UnitArgs ua)
{
    const UnitView u = load_unit(ua, c.ray, blockIdx.z);
    const Geom& g = c.g;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (MODE == MODE_FIXUP) {
        // uniform early exit for tiles the ray pass completed; self-cleaning flag
        const uint32_t flagged = c.tile_flags[tile];
        if (!flagged)
            return;
        __syncthreads();
        if (threadIdx.x == 0 && threadIdx.y == 0)
            c.tile_flags[tile] = 0;
    }
    const int x0 = (blockIdx.x * kBlockX + threadIdx.x) * kPX;
    const int j = blockIdx.y * kBlockY + threadIdx.y;
    if (x0 >= g.dst_w || j >= g.dst_h)
        return;

    float fx[kPX], fy[kPX];
    const unsigned npx_mask = (1u << min(kPX, g.dst_w - x0)) - 1;
    unsigned valid = Coords<MODE>::eval(c, u, x0, j, fx, fy) & npx_mask;
    if (MODE == MODE_RAY && valid != npx_mask)
        c.tile_flags[tile] = 1;

    const Image src{u.src, u.src_pitch, g.src_h, g.src_w};
    uint8_t* drow = u.dst + (int64_t)j * u.dst_pitch + (int64_t)x0 * CN;

    uint32_t words[kPX * CN / 4];
#pragma unroll
    for (int q = 0; q < kPX * CN / 4; q++)
        words[q] = 0;
    unsigned written = 0;
#pragma unroll
    for (int k = 0; k < kPX; k++) {
        uint8_t px[CN];
#pragma unroll
        for (int ch = 0; ch < CN; ch++)
            px[ch] = 0;
        if ((valid & (1u << k)) && sample<CN, INTERP>(src, g, c.itab, fx[k], fy[k], px))
            written |= 1u << k;
#pragma unroll
        for (int ch = 0; ch < CN; ch++) {
            const int b = k * CN + ch;
            words[b / 4] |= (uint32_t)px[ch] << (8 * (b % 4));
        }
    }
    if (written == (1u << kPX) - 1 && (((uintptr_t)drow) & 3) == 0) {
#pragma unroll
        for (int q = 0; q < kPX * CN / 4; q++)
            ((uint32_t*)drow)[q] = words[q];
    } else {
        // ragged right edge, BORDER_TRANSPARENT holes, fix-up pixels, unaligned destination
#pragma unroll
        for (int k = 0; k < kPX; k++) {
            if (written & (1u << k)) {
#pragma unroll
                for (int ch = 0; ch < CN; ch++) {
                    const int b = k * CN + ch;
                    drow[b] = (uint8_t)(words[b / 4] >> (8 * (b % 4)));
                }
            }
        }
    }
}

// coordinate maps only (v1c_plan_get_map): float32 like remapper.py:58.  The ray variant falls
// back to the interpreter in place (performance is irrelevant here).
template <int MODE>
__global__ __launch_bounds__(kBlockX* kBlockY) void k_get_map(KernelCtx c, UnitArgs ua, float* xmap, float* ymap, int64_t pitch)
{
    const UnitView u = load_unit(ua, c.ray, 0);
    const int x0 = (blockIdx.x * kBlockX + threadIdx.x) * kPX;
    const int j = blockIdx.y * kBlockY + threadIdx.y;
    if (x0 >= c.g.dst_w || j >= c.g.dst_h)
        return;
    float fx[kPX], fy[kPX];
    Coords<MODE>::eval(c, u, x0, j, fx, fy);
    if (MODE == MODE_RAY) {
        float lx[kPX], ly[kPX];
        const unsigned todo = Coords<MODE_FIXUP>::eval(c, u, x0, j, lx, ly);
        for (int k = 0; k < kPX; k++)
            if (todo & (1u << k))
                fx[k] = lx[k], fy[k] = ly[k];
    }
    float* xr = (float*)((char*)xmap + (int64_t)j * pitch);
    float* yr = (float*)((char*)ymap + (int64_t)j * pitch);
    for (int k = 0; k < kPX && x0 + k < c.g.dst_w; k++) {
        xr[x0 + k] = fx[k];
        yr[x0 + k] = fy[k];
    }
}

// ------------------------------------------------------------------------------------------
// host-callable launchers (used by plan.hip)
// ------------------------------------------------------------------------------------------
static dim3 grid_for(const Geom& g, int n_units)
{
    return dim3((g.dst_w + kBlockX * kPX - 1) / (kBlockX * kPX), (g.dst_h + kBlockY - 1) / kBlockY, n_units);
}

int tiles_per_unit(const Geom& g)
{
    const dim3 d = grid_for(g, 1);
    return (int)(d.x * d.y);
}

template <int CN, int MODE>
static hipError_t launch_interp(const KernelCtx& c, const UnitArgs& ua, int n_units, hipStream_t stream)
{
    const dim3 block(kBlockX, kBlockY, 1);
    const dim3 grid = grid_for(c.g, n_units);
    switch (c.g.interp) {
    case V1C_INTER_NEAREST:
        hipLaunchKernelGGL((k_remap<CN, V1C_INTER_NEAREST, MODE>), grid, block, 0, stream, c, ua);
        break;
    case V1C_INTER_LINEAR:
        hipLaunchKernelGGL((k_remap<CN, V1C_INTER_LINEAR, MODE>), grid, block, 0, stream, c, ua);
        break;
    case V1C_INTER_CUBIC:
        hipLaunchKernelGGL((k_remap<CN, V1C_INTER_CUBIC, MODE>), grid, block, 0, stream, c, ua);
        break;
    case V1C_INTER_LANCZOS4:
        hipLaunchKernelGGL((k_remap<CN, V1C_INTER_LANCZOS4, MODE>), grid, block, 0, stream, c, ua);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_cn(const KernelCtx& c, const UnitArgs& ua, int n_units, hipStream_t stream)
{
    switch (c.g.cn) {
    case 1: return launch_interp<1, MODE>(c, ua, n_units, stream);
    case 3: return launch_interp<3, MODE>(c, ua, n_units, stream);
    case 4: return launch_interp<4, MODE>(c, ua, n_units, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_remap(int mode, const KernelCtx& c, const UnitArgs& ua, int n_units, hipStream_t stream)
{
    switch (mode) {
    case MODE_LITERAL: return launch_cn<MODE_LITERAL>(c, ua, n_units, stream);
    case MODE_RAY: return launch_cn<MODE_RAY>(c, ua, n_units, stream);
    case MODE_FIXUP: return launch_cn<MODE_FIXUP>(c, ua, n_units, stream);
    case MODE_LUT: return launch_cn<MODE_LUT>(c, ua, n_units, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_get_map(int mode, const KernelCtx& c, const UnitArgs& u, float* xmap, float* ymap, int64_t pitch,
                          hipStream_t stream)
{
    const dim3 block(kBlockX, kBlockY, 1);
    const dim3 grid = grid_for(c.g, 1);
    if (mode == MODE_RAY)
        hipLaunchKernelGGL((k_get_map<MODE_RAY>), grid, block, 0, stream, c, u, xmap, ymap, pitch);
    else
        hipLaunchKernelGGL((k_get_map<MODE_LITERAL>), grid, block, 0, stream, c, u, xmap, ymap, pitch);
    return hipGetLastError();
}

// ---- anaglyph merge (remapper.py:485-497) ----
// One thread per pixel: 3 + 3 bytes in, 3 doubles (24 B, 8-byte aligned) out.  The arithmetic is
// the reference's NumPy float64 expression, operation by operation; contraction stays off so that
// mean * colour + mean * colour is two roundings of the products and one of the sum.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void k_anaglyph(const uint8_t* __restrict__ left, int64_t lp, const uint8_t* __restrict__ right,
                                                  int64_t rp, int h, int w, double* __restrict__ out, int64_t op)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h)
        return;
    const uint8_t* l = left + (int64_t)y * lp + (int64_t)x * 3;
    const uint8_t* r = right + (int64_t)y * rp + (int64_t)x * 3;
    // np.mean(img, axis=-1): float64 sum of the three uint8 channels (exact) divided by 3
    const double ml = (double)((int)l[0] + (int)l[1] + (int)l[2]) / 3.0;
    const double mr = (double)((int)r[0] + (int)r[1] + (int)r[2]) / 3.0;
    double* o = (double*)((char*)out + (int64_t)y * op) + (int64_t)x * 3;
    const double cl[3] = {0.0, 128.0, 255.0}, cr[3] = {255.0, 128.0, 0.0};  // colors[0], colors[1] (:489)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double a = ml * cl[c], b = mr * cr[c];
        o[c] = (a + b) / 255.0;  // combine /= 255 (:497)
    }
}
#pragma clang fp contract(fast)

hipError_t launch_anaglyph(const uint8_t* left, int64_t left_pitch, const uint8_t* right, int64_t right_pitch, int h, int w,
                           double* out, int64_t out_pitch, hipStream_t stream)
{
    const dim3 block(256, 1, 1), grid((w + 255) / 256, h, 1);
    hipLaunchKernelGGL(k_anaglyph, grid, block, 0, stream, left, left_pitch, right, right_pitch, h, w, out, out_pitch);
    return hipGetLastError();
}

// ---- get_radius(), transformer.py:108-140 ----
// One workgroup walks the centre row (w > h) or the centre column: black[i] = mean over the channels < threshold (:133, the float64 mean
// of uint8 values: sum / cn), d[i] = black[i + 1] - black[i] (:134), radius = (last i with d == -1  -  first i with d == +1) / 2 (:137-139;
// the sign quirk of the reference -- a disc on black gives a NEGATIVE value -- is kept); no +1 or no -1 anywhere: the reference indexes an
// empty array (IndexError) -> out[1] = 1, out[0] = NaN.
// black[i] of one pixel of a line: transformer.py:133 (the float64 mean of the uint8 channels against the threshold)
__device__ __forceinline__ int radius_black(const uint8_t* __restrict__ line, int64_t step, int i, int cn, int threshold)
{
    const uint8_t* p = line + (int64_t)i * step;
    int s = 0;
    for (int k = 0; k < cn; k++)
        s += p[k];
    return ((double)s / (double)cn) < (double)threshold ? 1 : 0;
}

// One chunk of 256 differences d[i] = black[i + 1] - black[i], i in [i0, i0 + 256), by one workgroup of 256 threads: every pixel is read
// once (the scan is bound by how many cache-line misses ONE compute unit keeps in flight: a centre column is a line per pixel) and
// passed to its neighbour through `bl` (257 bytes of LDS).  lo / hi: the lane's own first +1 / last -1 (0x7fffffff / -1: none).
__device__ __forceinline__ void radius_chunk(const uint8_t* __restrict__ line, int64_t step, int n, int cn, int threshold, int i0, uint8_t* bl,
                                             int& lo, int& hi)
{
    const int i = i0 + (int)threadIdx.x;
    bl[threadIdx.x] = i < n ? radius_black(line, step, i, cn, threshold) : 0;
    if (threadIdx.x == 0)
        bl[256] = i0 + 256 < n ? radius_black(line, step, i0 + 256, cn, threshold) : 0;
    __syncthreads();
    if (i + 1 < n) {
        const int d = (int)bl[threadIdx.x + 1] - (int)bl[threadIdx.x];
        if (d == 1)
            lo = min(lo, i);
        if (d == -1)
            hi = max(hi, i);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_get_radius(const uint8_t* __restrict__ line, int64_t step, int n, int cn, int threshold,
                                                    double* __restrict__ out)
{
    __shared__ int first_rise, last_fall;
    __shared__ uint8_t bl[257];
    if (threadIdx.x == 0)
        first_rise = 0x7fffffff, last_fall = -1;
    __syncthreads();
    int lo = 0x7fffffff, hi = -1;
    for (int i0 = 0; i0 + 1 < n; i0 += 256)
        radius_chunk(line, step, n, cn, threshold, i0, bl, lo, hi);
    if (lo != 0x7fffffff)
        atomicMin(&first_rise, lo);
    if (hi >= 0)
        atomicMax(&last_fall, hi);
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool ok = first_rise != 0x7fffffff && last_fall >= 0;
        out[0] = ok ? (double)(last_fall - first_rise) / 2.0 : NAN;
        out[1] = ok ? 0.0 : 1.0;
    }
}

hipError_t launch_get_radius(const uint8_t* img, int h, int w, int64_t pitch, int cn, int threshold, double* out, hipStream_t stream)
{
    const bool use_row = w > h;  // transformer.py:126-129
    const uint8_t* line = use_row ? img + (int64_t)(h / 2) * pitch : img + (int64_t)(w / 2) * cn;
    hipLaunchKernelGGL(k_get_radius, dim3(1), dim3(256), 0, stream, line, use_row ? (int64_t)cn : pitch, use_row ? w : h, cn, threshold, out);
    return hipGetLastError();
}

// ---- a radius that never leaves the device (v1c_plan_run_auto) ----
// get_radius_smart("auto") = max over the images of get_radius (remapper.py:83-84); get_map gives the Denormalize stage scale = (radius,
// radius) (remapper.py:55).  One thread takes the maximum of the n estimates v1c_get_radius_async left on the device (rad[2 k] = radius,
// rad[2 k + 1] = 0 / 1 "no black border": the reference raises IndexError there -- the device path cannot: the whole map is sent far
// outside the source instead (scale 0, centre at -40 000 px: every pixel the border colour under BORDER_CONSTANT, inside the range the
// kernels' coordinate proofs hold for)), clamps it to the magnitude the plan's proofs were taken for and writes the numbers of the
// plan-resident context the kernels read: stream-ordered, graph-capturable, no host round trip.
__device__ __forceinline__ void patch_ctx_radius(KernelCtx* ctx, double r, bool bad, double r_limit, double cx32, double cy32)
{
    r = fmin(fmax(r, -r_limit), r_limit);
    if (bad || !(r == r)) {
        r = 0.0;
        cx32 = cy32 = -1280000.0;
    }
    ctx->ray.rx = r, ctx->ray.ry = r;
    ctx->ray.rx32 = 32.0 * r, ctx->ray.ry32 = 32.0 * r;
    ctx->ray.cx32 = cx32, ctx->ray.cy32 = cy32;
    ctx->ray.cx = cx32 * 0.03125, ctx->ray.cy = cy32 * 0.03125;
}

__global__ void k_patch_radius(KernelCtx* ctx, const double* __restrict__ rad, int n, double r_limit, double cx32, double cy32)
{
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    double r = rad[0];
    bool bad = rad[1] != 0.0;
    for (int k = 1; k < n; k++) {
        r = rad[2 * k] > r ? rad[2 * k] : r;  // Python's max(): keeps the first of equal values, NaN never wins
        bad |= rad[2 * k + 1] != 0.0;
    }
    patch_ctx_radius(ctx, r, bad, r_limit, cx32, cy32);
}

// Both steps in ONE launch for the images of the call itself (v1c_plan_run_auto_images).  The first form of this -- an estimate launch
// of one workgroup per image, then the patch launch -- spent 16 - 20 us per image in the scan: one compute unit has only so many misses
// in flight.  Here up to kAutoBlocks workgroups take 256-pixel chunks of the lines in turn; their results meet in `scratch` (plan-resident:
// [k] first +1 of image k, [16 + k] last -1, [32] a ticket counter) and the workgroup that draws the last ticket takes the maximum,
// patches the context and leaves the scratch words as it found them (0x7fffffff / -1 / 0) for the next launch.
__global__ __launch_bounds__(256) void k_auto_radius(KernelCtx* ctx, int* __restrict__ scratch, AutoLines im, double r_limit, double cx32,
                                                     double cy32)
{
    __shared__ int first_rise[kInlineUnits], last_fall[kInlineUnits];
    __shared__ uint8_t bl[257];
    __shared__ int is_last;
    if (threadIdx.x < kInlineUnits)
        first_rise[threadIdx.x] = 0x7fffffff, last_fall[threadIdx.x] = -1;
    __syncthreads();
    const int per_line = (im.n - 1 + 255) / 256, total = per_line * im.count;
    for (int q = blockIdx.x; q < total; q += gridDim.x) {
        const int k = q / per_line;
        int lo = 0x7fffffff, hi = -1;
        radius_chunk(im.line[k], im.step[k], im.n, im.cn, im.threshold, (q - k * per_line) * 256, bl, lo, hi);
        if (lo != 0x7fffffff)
            atomicMin(&first_rise[k], lo);
        if (hi >= 0)
            atomicMax(&last_fall[k], hi);
    }
    __syncthreads();
    if ((int)threadIdx.x < im.count) {
        if (first_rise[threadIdx.x] != 0x7fffffff)
            atomicMin(&scratch[threadIdx.x], first_rise[threadIdx.x]);
        if (last_fall[threadIdx.x] >= 0)
            atomicMax(&scratch[kInlineUnits + threadIdx.x], last_fall[threadIdx.x]);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = atomicAdd(&scratch[2 * kInlineUnits], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!is_last || threadIdx.x != 0)
        return;
    __threadfence();
    double r = 0.0;
    bool bad = false;
    for (int k = 0; k < im.count; k++) {
        const int fr = __hip_atomic_exchange(&scratch[k], 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lf = __hip_atomic_exchange(&scratch[kInlineUnits + k], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double rk = (double)(lf - fr) / 2.0;
        r = (k == 0 || rk > r) ? rk : r;  // Python's max(): keeps the first of equal values
        bad |= fr == 0x7fffffff || lf < 0;
    }
    __hip_atomic_store(&scratch[2 * kInlineUnits], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    patch_ctx_radius(ctx, r, bad, r_limit, cx32, cy32);
}

hipError_t launch_auto_radius(KernelCtx* ctx_dev, int* scratch_dev, const AutoLines& im, double r_limit, double cx32, double cy32, hipStream_t stream)
{
    const int total = ((im.n - 1 + 255) / 256) * im.count;
    hipLaunchKernelGGL(k_auto_radius, dim3(std::max(1, std::min(total, kAutoBlocks))), dim3(256), 0, stream, ctx_dev, scratch_dev, im, r_limit, cx32,
                       cy32);
    return hipGetLastError();
}

hipError_t launch_patch_radius(KernelCtx* ctx_dev, const double* rad_dev, int n, double r_limit, double cx32, double cy32, hipStream_t stream)
{
    hipLaunchKernelGGL(k_patch_radius, dim3(1), dim3(64), 0, stream, ctx_dev, rad_dev, n, r_limit, cx32, cy32);
    return hipGetLastError();
}

}  // namespace v1c
