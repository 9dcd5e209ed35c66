// v1c_core.hpp -- per-pixel building blocks of the fused remap kernels (gfx950).
//
// Everything here is __host__ __device__ so that tests/host_emul can run the very same code on
// the CPU of a GPU-less build container; the product only ever launches it from kernels.hip.
//
// Two halves:
//   * coordinate producers: eval_chain_literal() (fp64 interpreter that follows the reference's
//     MultiTransformer stage by stage, transformer.py:93-98) and the fused "ray" evaluator
//     (separable row/column tables + a piecewise-polynomial radial table);
//   * the sampler: cv2.remap's 5-bit fixed-point NEAREST / LINEAR / CUBIC / LANCZOS4 gather with
//     OpenCV's border modes (call site remapper.py:388-398; semantics SURVEY.md Appendix A).
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/vr180_remap.h"

#define V1C_HD __host__ __device__ inline
#define V1C_HDF __host__ __device__ __forceinline__

namespace v1c {

constexpr int kMaxUnitsPerLaunch = 16;  // units carried in kernel arguments (grid.z)
constexpr int kRadialDegree = 7;        // degree of each radial-table piece
constexpr int kRadialCoefs = kRadialDegree + 1;

struct Image {
    const uint8_t* p;
    int64_t pitch;
    int h, w;
};

struct Geom {
    int src_h, src_w, dst_h, dst_w;
    int cn, interp, border;
    uint8_t cval[4];
};

// One unit of work as the kernels see it (kernel-argument resident).
struct DevUnit {
    const uint8_t* src;
    uint8_t* dst;
    int64_t src_pitch, dst_pitch;
    double rot[9];
    int has_rot, pad;
};

struct UnitArgs {
    DevUnit u[kMaxUnitsPerLaunch];
};

// Parameters of the fused ray path (device pointers are plan-owned).
struct RayParams {
    // per-column: sin(lon_i), cos(lon_i), 1-cos(lon_i) ; per-row: sin(lat_j), cos(lat_j), 1-cos(lat_j)
    const double *col_s, *col_c, *col_h;
    const double *row_s, *row_c, *row_h;
    const double* radial;  // [n_int][kRadialCoefs], monomial coefficients in z in [-0.5, 0.5]
    double inv_step;       // intervals per unit of the table variable
    int n_int;
    int var_is_w;          // 0: table variable m = 1 - v_z ; 1: w = sqrt(m / 2)
    int has_rot;           // chain carries (composed) rotation
    int pad;
    double rot[9];         // composed rotation of the chain (overridden per unit when DevUnit.has_rot)
    double rx, ry, cx, cy;  // Denormalize: x = X*rx + cx, y = Y*ry + cy   (transformer.py:202-203)
    double rx32, ry32, cx32, cy32;  // the same times 32 (exact): kernels produce cv2's 32*x directly
    double n_int_f;                 // (double)n_int
    // Second table (radial_fit.hpp: fit_mpoly_table): per interval a degree-6 polynomial in
    // delta = m - m_c (7 coefficients + m_c), so that tiles away from the image centre need no fp64
    // square root (w-tables) and no fp64 index arithmetic.  Null when absent.  mp_first_ok: every interval >= it (up to the plan's
    // reach) is valid at the level the lanes need; inv_step_f: (float)inv_step for the fp32 index.
    const double* radial_m;
    int mp_first_ok;
    float inv_step_f;
    // General modes (radial_fit.hpp: RayAnalysis; a rotation always applies, identity if the chain has none):
    //   1: EquirectangularEncoder(is_latitude_y=False) -- v = (col_s, col_c row_s, col_c row_c), transformer.py:557-566;
    //   2: radial stages in FRONT of the rotation -- the point enters 3-D through v = (sin t d_x, sin t d_y, cos t), t = F_pre(theta)
    //      (equidistant_to_3d, transformer.py:502-507): v = (S b_x, S b_y, 1 - Cm) with the base point b = (row_c col_s, row_s), the base
    //      variable m0 = row_h + row_c col_h (planar chains: b = (xn, yn), m0 = xn^2 + yn^2 through the substituted tables) and
    //      S = sin(F_pre) / |b|, Cm = 1 - cos(F_pre) tabulated in m0 (or w0 = sqrt(m0 / 2)): pre_s / pre_c, degree 7 per interval,
    //      every pixel its own entry.
    int gen_mode;
    int pre_var_is_w;
    const double* pre_s;
    const double* pre_c;
    double pre_inv_step;
    int pre_n_int, pad3;
};

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
V1C_HDF int cv_round(float v)
{
    // cvRound(float) as SSE cvtss2si: round-half-even; NaN / out of range -> INT_MIN
    // (SURVEY.md Appendix A item 2).  fabsf(NaN) < x is false.
#if defined(__HIP_DEVICE_COMPILE__)
    return (fabsf(v) < 2147483648.0f) ? __float2int_rn(v) : INT_MIN;
#else
    return (fabsf(v) < 2147483648.0f) ? (int)nearbyintf(v) : INT_MIN;
#endif
}

V1C_HDF int clamp_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

V1C_HDF int clip_i(int x, int a, int b)
{
    return x >= a ? (x < b ? x : b - 1) : a;
}

// cv::borderInterpolate; BORDER_CONSTANT -> -1
V1C_HD int border_index(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    switch (border) {
    case V1C_BORDER_REPLICATE:
        return p < 0 ? 0 : len - 1;
    case V1C_BORDER_REFLECT:
    case V1C_BORDER_REFLECT_101: {
        const int delta = border == V1C_BORDER_REFLECT_101;
        if (len == 1)
            return 0;
        do {
            if (p < 0)
                p = -p - 1 + delta;
            else
                p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    case V1C_BORDER_WRAP:
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    default:
        return -1;
    }
}

// ------------------------------------------------------------------------------------------
// literal fp64 chain interpreter (generic path; also the slow path of the fused kernels)
// ------------------------------------------------------------------------------------------
#pragma clang fp contract(off)  // the reference rounds every multiply and add separately

V1C_HD void lit_from_3d(const double v[3], double& x, double& y)
{  // equidistant_from_3d, transformer.py:526-530
    const double theta = acos(v[2]);
    const double phi = atan2(v[0], v[1]);
    x = theta * sin(phi);
    y = theta * cos(phi);
}

V1C_HD void lit_to_3d(double x, double y, double v[3])
{  // equidistant_to_3d, transformer.py:502-507
    const double phi = atan2(x, y);
    const double theta = sqrt(x * x + y * y);
    const double st = sin(theta);
    v[0] = st * sin(phi);
    v[1] = st * cos(phi);
    v[2] = cos(theta);
}

V1C_HD double lit_radial(const v1c_op& op, double t)
{
    const double half_pi = 1.5707963267948966;  // np.pi / 2
    switch (op.iparam) {
    case V1C_RAD_ENC_RECTILINEAR:   return atan(t);
    case V1C_RAD_ENC_STEREOGRAPHIC: return 2 * atan(t);
    case V1C_RAD_ENC_EQUIDISTANT:   return t * half_pi;
    case V1C_RAD_ENC_EQUISOLID:     return 2 * asin(t / 1.4142135623730951);
    case V1C_RAD_ENC_ORTHOGRAPHIC:  return asin(t);
    case V1C_RAD_DEC_RECTILINEAR:   return tan(t);
    case V1C_RAD_DEC_STEREOGRAPHIC: return 2 * tan(t / 2);
    case V1C_RAD_DEC_EQUIDISTANT:   return t / half_pi;
    case V1C_RAD_DEC_EQUISOLID:     return 1.4142135623730951 * sin(t / 2);
    case V1C_RAD_DEC_ORTHOGRAPHIC:  return sin(t);
    case V1C_RAD_POLYNOMIAL: {
        double y = 0.0;
        for (int k = op.nparam - 1; k >= 0; k--)
            y = y * t + op.p[k];
        return y;
    }
    case V1C_RAD_RECTDEC_FWD: return tan(t) * op.p[0];
    case V1C_RAD_RECTDEC_INV: return atan(t / op.p[0]);
    default: return NAN;
    }
}

// Evaluate the whole lowered chain at output pixel (i, j).  `rot` (may be null) replaces the
// matrix of the FIRST rotate stage (per-unit calibration, v1c_unit.rot).
V1C_HD void eval_chain_literal(const v1c_chain* ch, const double* rot, int i, int j, double& ox, double& oy)
{
    double x = (double)i, y = (double)j;
    const double half_pi = 1.5707963267948966;
    bool rot_used = false;
    const int n = ch->n_ops;
    for (int k = 0; k < n; k++) {
        const v1c_op& op = ch->ops[k];
        switch (op.opcode) {
        case V1C_OP_NORMALIZE:
            x = (x - op.p[0]) / op.p[2] * 2;
            y = (y - op.p[1]) / op.p[2] * 2;
            break;
        case V1C_OP_DENORMALIZE:
            x = x * op.p[0] + op.p[2];
            y = y * op.p[1] + op.p[3];
            break;
        case V1C_OP_DENORMALIZE_INV:
            x = (x - op.p[2]) / op.p[0];
            y = (y - op.p[3]) / op.p[1];
            break;
        case V1C_OP_ZOOM:
            x = x / op.p[0];
            y = y / op.p[0];
            break;
        case V1C_OP_ZOOM_INV:
            x = x * op.p[0];
            y = y * op.p[0];
            break;
        case V1C_OP_EQUIRECT_ENC: {
            double v[3];
            if (op.iparam) {
                const double lat = y * half_pi, lon = x * half_pi;
                const double cl = cos(lat);
                v[0] = cl * sin(lon);
                v[1] = sin(lat);
                v[2] = cl * cos(lon);
            } else {
                const double lat = x * half_pi, lon = y * half_pi;
                const double cl = cos(lat);
                v[0] = sin(lat);
                v[1] = cl * sin(lon);
                v[2] = cl * cos(lon);
            }
            lit_from_3d(v, x, y);
            break;
        }
        case V1C_OP_EQUIRECT_DEC: {
            double v[3];
            lit_to_3d(x, y, v);
            if (op.iparam) {
                x = atan2(v[0], v[2]) / half_pi;
                y = asin(v[1]) / half_pi;
            } else {
                x = asin(v[0]) / half_pi;
                y = atan2(v[1], v[2]) / half_pi;
            }
            break;
        }
        case V1C_OP_RADIAL: {
            double theta = sqrt(x * x + y * y);
            const double roll = atan2(y, x);
            theta = lit_radial(op, theta);
            x = theta * cos(roll);
            y = theta * sin(roll);
            break;
        }
        case V1C_OP_ROTATE: {
            const double* m = (rot && !rot_used) ? rot : op.p;
            rot_used = true;
            double v[3], r[3];
            lit_to_3d(x, y, v);
            for (int q = 0; q < 3; q++)
                r[q] = m[3 * q] * v[0] + m[3 * q + 1] * v[1] + m[3 * q + 2] * v[2];
            lit_from_3d(r, x, y);
            break;
        }
        default:
            x = y = NAN;
        }
    }
    ox = x;
    oy = y;
}

#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------
// fused ray path: EquirectangularEncoder -> [rotation] -> radial composite -> Denormalize
//   v   = (cl*sl_on, sl, cl*cl_on)            per-row / per-column tables (separable)
//   v'  = R v                                  optional
//   m   = 1 - v'_z                             (= 2 sin^2(theta/2), theta = angle off the axis)
//   G   = F(theta) / sin(theta)                piecewise polynomial in m (F odd) or w = sqrt(m/2)
//   x   = G*rx*v'_x + cx ; y = G*ry*v'_y + cy  then float32 (remapper.py:58)
// Returns false when the pixel falls outside the table's validated domain: the caller then uses
// eval_chain_literal for that pixel.
// ------------------------------------------------------------------------------------------
// (int)t with saturation: negative / tiny-negative -> 0.., huge -> INT_MAX, so that one unsigned
// compare against n_int decides "inside the table".  t is never NaN (finite tables in, finite out).
V1C_HDF int table_index(double t)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return max((int)t, 0);  // v_cvt_i32_f64 saturates
#else
    return t < 0.0 ? 0 : (t >= 2147483647.0 ? 2147483647 : (int)t);
#endif
}

V1C_HDF double fast_sqrt_half(double m)
{
    // sqrt(m/2) to ~1 ulp: v_rsq_f64 seed, one coupled Newton step, one residual correction
    // (inputs below 1e-280 -- the exact image centre -- are lifted to it: w ~ 1e-140 instead of
    // 0 changes G(w) by nothing representable, and keeps v_rsq_f64 away from 0 and denormals)
    const double a = fmax(0.5 * m, 1e-280);
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;
    double h = 0.5 * y;
    const double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, a), h, g);
    return g;
#else
    return sqrt(a);
#endif
}

// The tile kernels' rule for letting ONE radial-table entry serve several pixels of a lane (tile_device.hpp: lane_coords reads the entry
// of the lane's pixel 1 and evaluates the others with it where this says so; the host model of tests/host_emul runs the same
// function).  `zk` = the pixel's table coordinate relative to the centre of entry `ic`; `c7` = that entry's highest coefficient, whose
// two low mantissa bits hold the range the entry was validated on (|z| <= 0.5 + level: radial_fit.hpp); the entry was read as element
// ic - tab0 -- CLAMPED into [0, tabn) -- of the slice [tab0, tab0 + tabn) a workgroup keeps in LDS: when ic lies outside the slice the
// coefficients are another entry's and nothing may share them (a pixel 1 that is outside the table can point anywhere).
V1C_HDF bool shared_entry_serves(double zk, double c7, int ic, int tab0, int tabn)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int level = __double2loint(c7) & 3;
#else
    uint64_t bits;
    memcpy(&bits, &c7, 8);
    const int level = (int)(bits & 3u);
#endif
    return (fabs(zk) <= 0.5 + (double)level) & ((unsigned)(ic - tab0) < (unsigned)tabn);
}

// The rotated ray of the general modes (RayParams::gen_mode): fx, fy = its x / y components, m = 1 - its z component.  `q` = the
// column's third table value: cos of the column angle in mode 1 (build_ray_host_tables stores it in col_h for base 2), 1 - cos(lon) /
// xn^2 in mode 2.  False: the base variable lies outside the S / Cm tables (the caller leaves the pixel to the interpreter).
// ONE definition for the generic kernel (ray_eval) and the tile kernels (lane_coords): both must agree bit for bit.
// (`RP`: RayParams, or the constant-address-space view of it the tile kernels read the plan's context through)
template <typename RP, typename RotPtr>
V1C_HDF bool gen_vector(RP& P, RotPtr rot, double sl, double cl, double hl, double slon, double q, double& fx, double& fy,
                        double& m)
{
    // (written so that everything that depends on the row only -- B_k, T_k's factors -- is common to the 4 pixels of a lane: the tile
    //  kernels unroll the pixel loop and the compiler evaluates those once)
    // (one set of instantiations for both general modes: a lat_x-only build of the bilinear pair kernel needs 93 instead of 110 VGPRs --
    //  5 instead of 4 waves per SIMD -- which did not seem worth 48 more kernels)
    if (P.gen_mode == 1) {
        // R (s_c, c_c s_r, c_c c_r) = R_k0 s_c + (R_k1 s_r + R_k2 c_r) c_c
        const double B0 = fma(rot[1], sl, rot[2] * cl), B1 = fma(rot[4], sl, rot[5] * cl), B2 = fma(rot[7], sl, rot[8] * cl);
        fx = fma(rot[0], slon, B0 * q);
        fy = fma(rot[3], slon, B1 * q);
        m = 1.0 - fma(rot[6], slon, B2 * q);
        return true;
    }
    const double m0 = fma(cl, q, hl);
    const double u = P.pre_var_is_w ? fast_sqrt_half(m0) : m0;
    const double t = u * P.pre_inv_step;
    const int idx = table_index(t);
    const bool in = (unsigned)idx < (unsigned)P.pre_n_int;
    const int ic = in ? idx : 0;
    const double z = t - ((double)ic + 0.5);
    typedef double __attribute__((ext_vector_type(2))) d2v;  // (entries are 64-byte aligned: 16-byte loads)
    const d2v* cs = (const d2v*)(P.pre_s + (size_t)ic * kRadialCoefs);
    const d2v* cc = (const d2v*)(P.pre_c + (size_t)ic * kRadialCoefs);
    double es[kRadialCoefs], ec[kRadialCoefs];
#pragma unroll
    for (int k = 0; k < kRadialCoefs / 2; k++) {
        const d2v a = cs[k], b = cc[k];
        es[2 * k] = a.x, es[2 * k + 1] = a.y, ec[2 * k] = b.x, ec[2 * k + 1] = b.y;
    }
    double S = es[kRadialDegree], Cm = ec[kRadialDegree];
#pragma unroll
    for (int k = kRadialDegree - 1; k >= 0; k--)
        S = fma(S, z, es[k]), Cm = fma(Cm, z, ec[k]);
    // R (S b_x, S b_y, 1 - Cm) with the base point b = (cl slon, sl): S T_k + R_k2 (1 - Cm), T_k = (R_k0 cl) slon + R_k1 sl
    const double vz = 1.0 - Cm;
    const double T0 = fma(rot[0] * cl, slon, rot[1] * sl), T1 = fma(rot[3] * cl, slon, rot[4] * sl), T2 = fma(rot[6] * cl, slon, rot[7] * sl);
    fx = fma(S, T0, rot[2] * vz);
    fy = fma(S, T1, rot[5] * vz);
    // 1 - (R v)_z without the cancellation of 1 - (... + r22 (1 - Cm)): (1 - r22) + r22 Cm - S T_2
    m = fma(-S, T2, fma(rot[8], Cm, 1.0 - rot[8]));
    return in;  // (flagged intervals carry NaN coefficients: the coordinates come out NaN and fail the caller's range test)
}

V1C_HDF bool ray_eval(const RayParams& P, bool use_rot, const double (&rot)[9], double sl, double cl, double hl,
                      double slon, double clon, double hlon, double& ox, double& oy)
{
    // NOTE: tile_device.hpp (lane_coords) evaluates these very expressions (same operations, same order); the
    // fix-up pass relies on both agreeing bit for bit on which pixels are inside the table.
    double m, x32, y32;
    double sx_, sy_;  // the factors G multiplies: x32 = (G*kx)*sx_ + cx32, y32 = (G*ky)*sy_ + cy32
    double kx, ky;
    if (P.gen_mode) {
        if (!gen_vector(P, rot, sl, cl, hl, slon, hlon, sx_, sy_, m))
            return false;
        kx = P.rx32, ky = P.ry32;
        use_rot = true;
    } else if (use_rot) {
        // R*v with v = (cl*slon, sl, cl*clon), grouped as (R_k0*cl)*slon + (R_k2*cl)*clon + R_k1*sl so
        // that the row-constant factors can be hoisted out of the pixel loop
        sx_ = fma(rot[0] * cl, slon, fma(rot[2] * cl, clon, rot[1] * sl));
        sy_ = fma(rot[3] * cl, slon, fma(rot[5] * cl, clon, rot[4] * sl));
        m = 1.0 - fma(rot[6] * cl, slon, fma(rot[8] * cl, clon, rot[7] * sl));
        kx = P.rx32, ky = P.ry32;
    } else {
        // v = (cl*slon, sl, .): the row factors cl / sl are folded into the scale
        sx_ = slon, sy_ = 1.0;
        kx = P.rx32 * cl, ky = P.ry32 * sl;
        m = fma(cl, hlon, hl);  // 1 - cl*clon without cancellation
    }
    const double u = P.var_is_w ? fast_sqrt_half(m) : m;
    const double t = u * P.inv_step;
    // a rotated ray can give m = -1e-17 (t = -0.0...): the saturating conversion clamps it to entry 0
    const int idx = table_index(t);
    if ((unsigned)idx >= (unsigned)P.n_int)
        return false;
    const double z = t - ((double)idx + 0.5);
    const double* c = P.radial + (size_t)idx * kRadialCoefs;
    double g = c[kRadialDegree];
#pragma unroll
    for (int k = kRadialDegree - 1; k >= 0; k--)
        g = fma(g, z, c[k]);
    // computed scaled by 32 (exact): float32(32*x) == 32*float32(x) is cv2's fixed-point input
    x32 = fma(g * kx, sx_, P.cx32);
    y32 = use_rot ? fma(g * ky, sy_, P.cy32) : fma(g, ky, P.cy32);
    // flagged intervals carry NaN coefficients; |float32(32 x)| < 2^30 keeps the int conversion exact
    if (!(fabsf((float)x32) < 1073741824.0f && fabsf((float)y32) < 1073741824.0f))
        return false;
    ox = x32 * 0.03125;
    oy = y32 * 0.03125;
    return true;
}

// ------------------------------------------------------------------------------------------
// sampler
// ------------------------------------------------------------------------------------------
struct Taps {
    int ix, iy;  // integer source position (top-left of the 2x2 cell / nearest pixel)
    int fx, fy;  // 1/32 fractions
};

// from cv2's fixed-point coordinates sx = cvRound(x*32), sy = cvRound(y*32)
V1C_HDF Taps taps_from_fixed(int sx, int sy)
{
    Taps t;
    t.ix = clamp_short(sx >> 5);
    t.iy = clamp_short(sy >> 5);
    t.fx = sx & 31;
    t.fy = sy & 31;
    return t;
}

V1C_HDF Taps quantize(float x, float y)
{
    // RemapInvoker, planar float maps: sx = cvRound(x*32); ix = sat<short>(sx >> 5); fx = sx & 31
    const int sx = cv_round(x * 32.0f), sy = cv_round(y * 32.0f);
    Taps t;
    t.ix = clamp_short(sx >> 5);
    t.iy = clamp_short(sy >> 5);
    t.fx = sx & 31;
    t.fy = sy & 31;
    return t;
}

V1C_HDF uint32_t load_u32_unaligned(const uint8_t* p)
{
    typedef uint32_t __attribute__((aligned(1), may_alias)) u32_u;
    return *(const u32_u*)p;
}

struct u64pair {
    uint32_t lo, hi;
};

V1C_HDF u64pair load_u64_unaligned(const uint8_t* p)
{
    typedef uint64_t __attribute__((aligned(1), may_alias)) u64_u;
    const uint64_t v = *(const u64_u*)p;
    u64pair r;
    r.lo = (uint32_t)v;
    r.hi = (uint32_t)(v >> 32);
    return r;
}

// NEAREST: remapNearest.  Returns false when the pixel must be left untouched (TRANSPARENT).
template <int CN>
V1C_HD bool sample_nearest(const Image& s, const Geom& g, float x, float y, uint8_t* out)
{
    int sx = clamp_short(cv_round(x)), sy = clamp_short(cv_round(y));
    const uint8_t* S;
    if ((unsigned)sx < (unsigned)s.w && (unsigned)sy < (unsigned)s.h) {
        S = s.p + (int64_t)sy * s.pitch + (int64_t)sx * CN;
    } else if (g.border == V1C_BORDER_TRANSPARENT) {
        return false;
    } else if (g.border == V1C_BORDER_CONSTANT) {
#pragma unroll
        for (int k = 0; k < CN; k++)
            out[k] = g.cval[k];
        return true;
    } else {
        sx = border_index(sx, s.w, g.border);
        sy = border_index(sy, s.h, g.border);
        S = s.p + (int64_t)sy * s.pitch + (int64_t)sx * CN;
    }
#pragma unroll
    for (int k = 0; k < CN; k++)
        out[k] = S[k];
    return true;
}

// LINEAR: remapBilinear with the fixed-point table BilinearTab_i.  The table entry for fractions
// (fx, fy) is 32*(32-fx | fx)*(32-fy | fy) -- except entry (0,0), which OpenCV's saturate_cast
// + fix-up turns into {32767, 0, 0, 1}; for uint8 pixels both give (sum + 2^14) >> 15 == p00, so
// the two-step lerp below ((h0*(32-fy) + h1*fy + 512) >> 10) is bit-identical.
template <int CN>
V1C_HD bool sample_linear_t(const Image& s, const Geom& g, const Taps t, uint8_t* out)
{
    const int wx1 = t.fx, wx0 = 32 - t.fx, wy1 = t.fy, wy0 = 32 - t.fy;
    // (s.w - 2 clamped at 0: a source ONE pixel wide made the unsigned bound 2^32 - 1 and every pixel "inside" -- reads far outside
    //  the image, a GPU memory fault; found by tools/fuzz.py in round 4)
    if (CN == 3 && (unsigned)t.ix < (unsigned)(s.w > 2 ? s.w - 2 : 0) && (unsigned)t.iy < (unsigned)(s.h - 1)) {
        // whole 2x2 cell inside and 8 readable bytes per row: two unaligned 8-byte loads
        const uint8_t* p0 = s.p + (int64_t)t.iy * s.pitch + t.ix * 3;
        const u64pair a = load_u64_unaligned(p0);
        const u64pair b = load_u64_unaligned(p0 + s.pitch);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int p00 = (a.lo >> (8 * k)) & 255;
            const int p01 = k == 0 ? (a.lo >> 24) : ((a.hi >> (8 * (k - 1))) & 255);
            const int p10 = (b.lo >> (8 * k)) & 255;
            const int p11 = k == 0 ? (b.lo >> 24) : ((b.hi >> (8 * (k - 1))) & 255);
            const int h0 = p00 * wx0 + p01 * wx1;
            const int h1 = p10 * wx0 + p11 * wx1;
            out[k] = (uint8_t)((h0 * wy0 + h1 * wy1 + 512) >> 10);
        }
        return true;
    }
    const int W = s.w, H = s.h;
    int x0, x1, y0, y1;
    if ((unsigned)t.ix < (unsigned)(W - 1) && (unsigned)t.iy < (unsigned)(H - 1)) {
        x0 = t.ix, x1 = t.ix + 1, y0 = t.iy, y1 = t.iy + 1;
    } else {
        if (g.border == V1C_BORDER_CONSTANT && (t.ix >= W || t.ix + 1 < 0 || t.iy >= H || t.iy + 1 < 0)) {
#pragma unroll
            for (int k = 0; k < CN; k++)
                out[k] = g.cval[k];
            return true;
        }
        if (g.border == V1C_BORDER_TRANSPARENT)
            return false;  // any pixel whose 2x2 cell is not fully inside is skipped
        x0 = border_index(t.ix, W, g.border);
        x1 = border_index(t.ix + 1, W, g.border);
        y0 = border_index(t.iy, H, g.border);
        y1 = border_index(t.iy + 1, H, g.border);
    }
    const uint8_t* r0 = s.p + (int64_t)(y0 < 0 ? 0 : y0) * s.pitch;
    const uint8_t* r1 = s.p + (int64_t)(y1 < 0 ? 0 : y1) * s.pitch;
#pragma unroll
    for (int k = 0; k < CN; k++) {
        const int cv = g.cval[k];
        const int p00 = (x0 >= 0 && y0 >= 0) ? r0[x0 * CN + k] : cv;
        const int p01 = (x1 >= 0 && y0 >= 0) ? r0[x1 * CN + k] : cv;
        const int p10 = (x0 >= 0 && y1 >= 0) ? r1[x0 * CN + k] : cv;
        const int p11 = (x1 >= 0 && y1 >= 0) ? r1[x1 * CN + k] : cv;
        const int h0 = p00 * wx0 + p01 * wx1;
        const int h1 = p10 * wx0 + p11 * wx1;
        out[k] = (uint8_t)((h0 * wy0 + h1 * wy1 + 512) >> 10);
    }
    return true;
}

template <int CN>
V1C_HD bool sample_linear(const Image& s, const Geom& g, float x, float y, uint8_t* out)
{
    return sample_linear_t<CN>(s, g, quantize(x, y), out);
}

// CUBIC (K = 4) / LANCZOS4 (K = 8): remapBicubic / remapLanczos4 with the int16 table built by
// initInterTab2D (host: build_itab in plan.hip).  itab layout [fy*32+fx][K][K].
template <int CN, int K>
V1C_HD bool sample_table(const Image& s, const Geom& g, const short* __restrict__ itab, float x, float y, uint8_t* out)
{
    const Taps t = quantize(x, y);
    const short* __restrict__ w = itab + (size_t)(t.fy * 32 + t.fx) * (K * K);
    const int off = K / 2 - 1;
    const int sx = t.ix - off, sy = t.iy - off;
    const int W = s.w, H = s.h;
    int acc[CN];
#pragma unroll
    for (int k = 0; k < CN; k++)
        acc[k] = 0;
    if ((unsigned)sx < (unsigned)(W - (K - 1) > 0 ? W - (K - 1) : 0) && (unsigned)sy < (unsigned)(H - (K - 1) > 0 ? H - (K - 1) : 0)) {
        const uint8_t* S = s.p + (int64_t)sy * s.pitch + (int64_t)sx * CN;
        for (int i = 0; i < K; i++, S += s.pitch) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int wv = w[i * K + j];
#pragma unroll
                for (int k = 0; k < CN; k++)
                    acc[k] += S[j * CN + k] * wv;
            }
        }
    } else {
        int border = g.border;
        if (border == V1C_BORDER_TRANSPARENT) {
            if ((unsigned)(sx + off) >= (unsigned)W || (unsigned)(sy + off) >= (unsigned)H)
                return false;
            border = V1C_BORDER_REFLECT_101;
        }
        if (border == V1C_BORDER_CONSTANT && (sx >= W || sx + K <= 0 || sy >= H || sy + K <= 0)) {
#pragma unroll
            for (int k = 0; k < CN; k++)
                out[k] = g.cval[k];
            return true;
        }
        int xs[K];
#pragma unroll
        for (int j = 0; j < K; j++)
            xs[j] = border_index(sx + j, W, border);
        for (int i = 0; i < K; i++) {
            const int yi = border_index(sy + i, H, border);
            const uint8_t* S = s.p + (int64_t)(yi < 0 ? 0 : yi) * s.pitch;
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int wv = w[i * K + j];
                const bool in = yi >= 0 && xs[j] >= 0;
#pragma unroll
                for (int k = 0; k < CN; k++)
                    acc[k] += (in ? (int)S[xs[j] * CN + k] : (int)g.cval[k]) * wv;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CN; k++) {
        const int v = (acc[k] + (1 << 14)) >> 15;  // FixedPtCast<int, uchar, 15>
        out[k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    return true;
}

template <int CN, int INTERP>
V1C_HDF bool sample(const Image& s, const Geom& g, const short* itab, float x, float y, uint8_t* out)
{
    if (INTERP == V1C_INTER_NEAREST)
        return sample_nearest<CN>(s, g, x, y, out);
    if (INTERP == V1C_INTER_LINEAR)
        return sample_linear<CN>(s, g, x, y, out);
    if (INTERP == V1C_INTER_CUBIC)
        return sample_table<CN, 4>(s, g, itab, x, y, out);
    return sample_table<CN, 8>(s, g, itab, x, y, out);
}

}  // namespace v1c
