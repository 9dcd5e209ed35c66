// radial_fit.hpp -- plan-time analysis of a lowered chain and fit of the radial table (host only).
//
// The fused "ray" kernel needs, per output pixel, G = F(theta) / sin(theta), where theta is the
// angle between the (rotated) viewing ray and the optical axis and F is the composite of every
// radial stage between the EquirectangularEncoder and the DenormalizeTransformer
// (PolynomialScaler.transform_polar transformer.py:448-451, FisheyeEncoder.inverse_transform_polar
// :379-397, ZoomTransformer :468-473 ...).  G is tabulated as a piecewise degree-7 polynomial in
//     m = 1 - cos(theta)        when that is smooth down to m = 0 (F odd in theta), else in
//     w = sqrt(m / 2) = sin(theta / 2).
// The fit is done in long double (x87, 64-bit mantissa) against the reference's stage semantics
// (including PolarRollTransformer's re-derivation of |theta| and roll, transformer.py:271-272) and
// every interval is validated; intervals that miss the tolerance are flagged (NaN coefficients)
// and pixels landing there are evaluated by the literal fp64 interpreter instead.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/vr180_remap.h"
#include "v1c_core.hpp"

namespace v1c {

// Chain shapes the fused path serves (everything else: the fp64 interpreter), with B = the stage behind Normalize:
//   base 0: B = EquirectangularEncoder(is_latitude_y=True)  (transformer.py:546-555)  -- BASELINE's configurations
//   base 1: B = nothing: the normalised plane point itself ("planar": FisheyeEncoder(...) * ... chains; 7 of the
//           reference's 10 test chains, tests/test_remapper.py:42-91)
//   base 2: B = EquirectangularEncoder(is_latitude_y=False) (transformer.py:557-566)
// followed by   (Radial|Zoom)* [`pre`]   Rotate*   (Radial|Zoom)* [`radial`]   Denormalize.
// Without a rotation everything radial is one composite (`radial`; `pre` stays empty).  With a rotation behind radial stages (`pre`
// not empty: base 0 / 1 only) the ray enters 3-D through v = (sin t d_x, sin t d_y, cos t), t = F_pre(theta) (transformer.py:502-507):
// RayParams::gen_mode 2; base 2 runs gen_mode 1.
struct RayAnalysis {
    bool ok = false;         // chain has a shape the fused kernels handle
    bool has_rot = false;    // a rotation applies in the kernels (gen modes: always, identity if the chain has none)
    int base = 0;            // 0 equirect lat_y, 1 planar, 2 equirect lat_x
    int gen_mode = 0;        // RayParams::gen_mode: 0 classic, 1 lat_x, 2 pre-rotation radial stages
    double rot[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // composition of consecutive rotate stages
    double norm_cx = 0, norm_cy = 0, norm_s = 1;  // Normalize
    bool has_rows = false;                        // Normalize carries the row range of the whole grid (a row band: chain.py)
    double row_lo = 0, row_hi = 0;
    double rx = 1, ry = 1, cx = 0, cy = 0;        // Denormalize
    std::vector<v1c_op> pre;                      // radial / zoom stages in front of the rotation (gen_mode 2)
    std::vector<v1c_op> radial;                   // radial / zoom stages behind it (all of them when nothing rotates)
};

// what a table tabulates, as a function of the table variable u (m, or w = sqrt(m / 2)); F = composite of the stages given
enum {
    FN_RAY_G = 0,     // F(theta) / sin(theta),        m = 1 - cos(theta): the ray path's radial factor
    FN_PLANAR_G = 1,  // F(t) / t,                     m = t^2 = xn^2 + yn^2: planar chains without a rotation
    FN_PLANAR_S = 2,  // sin(F(t)) / t                 } the ray a planar point enters 3-D with:
    FN_PLANAR_CM = 3, // 1 - cos(F(t))                 } v = (S xn, S yn, 1 - Cm)
    FN_RAY_S = 4,     // sin(F(theta)) / sin(theta)    } the same behind an EquirectangularEncoder: v = (S v_x, S v_y, 1 - Cm)
    FN_RAY_CM = 5,    // 1 - cos(F(theta))
};

struct RadialTable {
    int fn = FN_RAY_G;
    double m_max = 0;       // the table covers m in [0, m_max]
    int var_is_w = 0;
    int n_int = 0;
    double inv_step = 0;
    double u_max = 0;
    int first_invalid = 0;  // index of the first flagged interval (== n_int if none)
    int first_below_level[3] = {0, 0, 0};  // index of the first interval whose validity level is < 0 / 1 / 2
    int n_invalid = 0;
    int n_extended = 0;     // intervals whose polynomial is valid beyond their own range (level >= 1)
    std::vector<double> coef;  // n_int * kRadialCoefs
};

constexpr double kTableMMax = 1.9375;  // theta up to ~159.6 deg
constexpr int kTableIntervals = 1024;

inline bool is_radial_op(const v1c_op& op)
{
    return op.opcode == V1C_OP_RADIAL || op.opcode == V1C_OP_ZOOM || op.opcode == V1C_OP_ZOOM_INV;
}

// Does the chain look like Normalize, [EquirectEnc], (Radial|Zoom)*, Rotate*, (Radial|Zoom)*, Denormalize ?  (RayAnalysis)
inline RayAnalysis analyze_chain(const v1c_chain& ch)
{
    RayAnalysis a;
    const int n = ch.n_ops;
    if (n < 2 || ch.ops[0].opcode != V1C_OP_NORMALIZE || ch.ops[n - 1].opcode != V1C_OP_DENORMALIZE)
        return a;
    a.norm_cx = ch.ops[0].p[0];
    a.norm_cy = ch.ops[0].p[1];
    a.norm_s = ch.ops[0].p[2];
    if (ch.ops[0].nparam >= 5 && ch.ops[0].p[4] > ch.ops[0].p[3])
        a.has_rows = true, a.row_lo = ch.ops[0].p[3], a.row_hi = ch.ops[0].p[4];
    const v1c_op& dn = ch.ops[n - 1];
    a.rx = dn.p[0], a.ry = dn.p[1], a.cx = dn.p[2], a.cy = dn.p[3];
    int k = 1;
    if (n >= 3 && ch.ops[1].opcode == V1C_OP_EQUIRECT_ENC) {
        a.base = ch.ops[1].iparam == 1 ? 0 : 2;
        k = 2;
    } else {
        a.base = 1;
    }
    for (; k < n - 1 && is_radial_op(ch.ops[k]); k++)
        a.pre.push_back(ch.ops[k]);
    for (; k < n - 1 && ch.ops[k].opcode == V1C_OP_ROTATE; k++) {
        // v' = M_k (M_{k-1} ... v): left-multiply
        const double* m = ch.ops[k].p;
        double r[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                r[3 * i + j] = m[3 * i] * a.rot[j] + m[3 * i + 1] * a.rot[3 + j] + m[3 * i + 2] * a.rot[6 + j];
        for (int i = 0; i < 9; i++)
            a.rot[i] = r[i];
        a.has_rot = true;
    }
    for (; k < n - 1; k++) {
        const v1c_op& op = ch.ops[k];
        if (is_radial_op(op))
            a.radial.push_back(op);
        else
            return a;  // a second rotation behind radial stages, decoders ...: literal path
    }
    if (!a.has_rot) {  // nothing rotates: one radial composite
        a.radial.insert(a.radial.begin(), a.pre.begin(), a.pre.end());
        a.pre.clear();
    }
    if (a.base == 2) {
        if (!a.pre.empty())
            return a;  // (lat_x, radial stages, rotation: not served)
        a.gen_mode = 1, a.has_rot = true;  // identity unless the chain rotates
    } else if (a.has_rot && (a.base == 1 || !a.pre.empty())) {
        a.gen_mode = 2;  // the ray is entered through the S / Cm tables (for a planar base also with F_pre = identity)
    }
    a.ok = true;
    return a;
}

inline long double radial_ld(const v1c_op& op, long double t)
{
    const long double half_pi = (long double)1.5707963267948966;  // the reference's float64 np.pi/2
    const long double sqrt2 = (long double)1.4142135623730951;    // float64 np.sqrt(2)
    switch (op.iparam) {
    case V1C_RAD_ENC_RECTILINEAR:   return atanl(t);
    case V1C_RAD_ENC_STEREOGRAPHIC: return 2 * atanl(t);
    case V1C_RAD_ENC_EQUIDISTANT:   return t * half_pi;
    case V1C_RAD_ENC_EQUISOLID:     return 2 * asinl(t / sqrt2);
    case V1C_RAD_ENC_ORTHOGRAPHIC:  return asinl(t);
    case V1C_RAD_DEC_RECTILINEAR:   return tanl(t);
    case V1C_RAD_DEC_STEREOGRAPHIC: return 2 * tanl(t / 2);
    case V1C_RAD_DEC_EQUIDISTANT:   return t / half_pi;
    case V1C_RAD_DEC_EQUISOLID:     return sqrt2 * sinl(t / 2);
    case V1C_RAD_DEC_ORTHOGRAPHIC:  return sinl(t);
    case V1C_RAD_POLYNOMIAL: {
        long double y = 0;
        for (int k = op.nparam - 1; k >= 0; k--)
            y = y * t + (long double)op.p[k];
        return y;
    }
    case V1C_RAD_RECTDEC_FWD: return tanl(t) * (long double)op.p[0];
    case V1C_RAD_RECTDEC_INV: return atanl(t / (long double)op.p[0]);
    default: return NAN;
    }
}

// signed radial coordinate after all stages for a point at angle theta > 0 off the axis
inline bool composite_F(const std::vector<v1c_op>& st, long double theta, long double& out)
{
    long double t = theta;
    for (const v1c_op& op : st) {
        if (op.opcode == V1C_OP_ZOOM) {
            t = t / (long double)op.p[0];
        } else if (op.opcode == V1C_OP_ZOOM_INV) {
            t = t * (long double)op.p[0];
        } else {
            // PolarRollTransformer.transform: theta_in = sqrt(x^2+y^2) = |t|; a negative t flips
            // the direction (roll + pi), so the stage output carries the sign of its input.
            const long double s = t < 0 ? -1.0L : 1.0L;
            t = s * radial_ld(op, fabsl(t));
        }
        if (!std::isfinite((double)t))
            return false;
    }
    out = t;
    return true;
}

// value of table function `fn` at table variable u (m when !var_is_w, else w = sqrt(m / 2))
inline bool table_fn(const std::vector<v1c_op>& st, int fn, int var_is_w, long double u, long double& g)
{
    if (!(u > 0))
        return false;
    long double t, den;  // argument of the composite and the denominator
    if (fn == FN_RAY_G || fn == FN_RAY_S || fn == FN_RAY_CM) {
        const long double w = var_is_w ? u : sqrtl(u / 2);
        if (!(w < 1))
            return false;
        t = 2 * asinl(w);                             // theta
        den = 2 * w * sqrtl((1 - w) * (1 + w));       // sin(theta)
    } else {
        t = var_is_w ? u * (long double)1.41421356237309504880168872420969808L : sqrtl(u);  // m = t^2, w = t / sqrt(2)
        den = t;
    }
    long double f;
    if (!composite_F(st, t, f))
        return false;
    switch (fn) {
    case FN_RAY_G:
    case FN_PLANAR_G: g = f / den; break;
    case FN_RAY_S:
    case FN_PLANAR_S: g = sinl(f) / den; break;
    default: {
        const long double sh = sinl(f / 2);
        g = 2 * sh * sh;
    }
    }
    return std::isfinite((double)g);
}
inline bool G_of_u(const std::vector<v1c_op>& st, int var_is_w, long double u, long double& g)
{
    return table_fn(st, FN_RAY_G, var_is_w, u, g);
}

// solve V c = y, V[i][k] = z_i^k, (n x n), long double, partial pivoting
inline bool solve_vandermonde(int n, const long double* z, const long double* y, long double* c)
{
    long double A[16][17];
    for (int i = 0; i < n; i++) {
        long double p = 1;
        for (int k = 0; k < n; k++, p *= z[i])
            A[i][k] = p;
        A[i][n] = y[i];
    }
    for (int col = 0; col < n; col++) {
        int piv = col;
        for (int r = col + 1; r < n; r++)
            if (fabsl(A[r][col]) > fabsl(A[piv][col]))
                piv = r;
        if (A[piv][col] == 0)
            return false;
        if (piv != col)
            for (int k = 0; k <= n; k++)
                std::swap(A[piv][k], A[col][k]);
        for (int r = col + 1; r < n; r++) {
            const long double f = A[r][col] / A[col][col];
            for (int k = col; k <= n; k++)
                A[r][k] -= f * A[col][k];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        long double s = A[i][n];
        for (int k = i + 1; k < n; k++)
            s -= A[i][k] * c[k];
        c[i] = s / A[i][i];
    }
    return true;
}

// The intervals of a table are fitted independently (long-double Vandermonde solve + validation of the double Horner: ~40 us
// each, 1024 of them, up to three tables per plan = the 50 ms a first call used to cost): intervals dealt to up to 16 host
// threads.  The result does not depend on the thread count (no interval reads another's).
template <typename F>
inline void fit_parallel_for(int n, F&& body)
{
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<unsigned>(std::min<unsigned>(hw ? hw : 1u, 16u), (unsigned)std::max(n / 32, 1));
    if (nt <= 1) {
        for (int i = 0; i < n; i++)
            body(i);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    auto run = [&](int t) {
        for (int i = t; i < n; i += nt)  // (interleaved: the intervals that need three attempts cluster)
            body(i);
    };
    int started = 1;  // share 0 runs on this thread
    try {
        for (int t = 1; t < nt; t++) {
            pool.emplace_back(run, t);
            started++;
        }
    } catch (const std::system_error&) {
        // no more threads to be had (resource limits): the shares that did not start run here -- no exception crosses the C ABI
    }
    run(0);
    for (int t = started; t < nt; t++)
        run(t);
    for (auto& th : pool)
        th.join();
}

// Fit one table.  Every interval i (table variable u in [i, i+1) * step, z = u / step - (i + 0.5))
// gets the polynomial of the WIDEST range that validates:
//   level 2: |z| <= 2.5   level 1: |z| <= 1.5   level 0: |z| <= 0.5 (its own interval only)
// (fitting on the wide range keeps the noise of the high coefficients harmless there: nodes on
// [-0.5, 0.5] leave ~1e-16 of long-double rounding in c7, which 1.5^7 would amplify past the
// tolerance).  The level is stored in the two low mantissa bits of c7; the DOUBLE Horner the
// kernels run is validated with those bits already in place.  tile_device.hpp uses pixel 1's
// entry for all 4 pixels of a lane wherever the level allows.
inline RadialTable fit_radial_table(const std::vector<v1c_op>& st, int var_is_w, int n_int = kTableIntervals, int fn = FN_RAY_G,
                                    double m_max = kTableMMax)
{
    RadialTable T;
    T.fn = fn, T.m_max = m_max;
    T.var_is_w = var_is_w;
    T.n_int = n_int;
    T.u_max = var_is_w ? std::sqrt(m_max / 2) : m_max;
    // a step that is exactly representable keeps t = u * inv_step monotone and cheap
    T.inv_step = std::ldexp(std::floor(std::ldexp(T.n_int / T.u_max, 20)), -20);
    const long double step = 1.0L / (long double)T.inv_step;
    T.coef.assign((size_t)T.n_int * kRadialCoefs, NAN);
    T.first_invalid = T.n_int;
    T.first_below_level[0] = T.first_below_level[1] = T.first_below_level[2] = T.n_int;
    const int n = kRadialCoefs;
    long double cheb[16];
    for (int k = 0; k < n; k++)
        cheb[k] = cosl(M_PIl * (2 * k + 1) / (2.0L * n));  // Chebyshev nodes on [-1, 1]
    const double tol = 1.5e-15;
    std::vector<signed char> levels(T.n_int, -1);
    fit_parallel_for(T.n_int, [&](int i) {
        const long double a = i * step;
        double cd[16];
        int level = -1;
        for (int lv = 2; lv >= 0 && level < 0; lv--) {
            // (the table variable is never negative: nothing to cover left of u = 0)
            const long double zlo = std::max(-(0.5L + lv), -(i + 0.5L)), zhi = 0.5L + lv;
            long double zn[16], y[16], c[16];
            bool good = true;
            for (int k = 0; k < n && good; k++) {
                zn[k] = 0.5L * (zlo + zhi) + 0.5L * (zhi - zlo) * cheb[k];
                good = table_fn(st, fn, var_is_w, a + step * (zn[k] + 0.5L), y[k]);
            }
            if (!good || !solve_vandermonde(n, zn, y, c))
                continue;
            for (int k = 0; k < n; k++)
                cd[k] = (double)c[k];
            uint64_t bits;
            std::memcpy(&bits, &cd[n - 1], 8);
            bits = (bits & ~3ull) | (uint64_t)lv;
            std::memcpy(&cd[n - 1], &bits, 8);
            const int ntest = (4 * n + 1) * (2 * lv + 1);
            long double gmax = 0;
            std::vector<long double> gt(ntest), zt(ntest);
            for (int q = 0; q < ntest && good; q++) {
                zt[q] = zlo + (zhi - zlo) * (q + 0.5L) / ntest;
                good = table_fn(st, fn, var_is_w, a + step * (zt[q] + 0.5L), gt[q]);
                gmax = fmaxl(gmax, fabsl(gt[q]));
            }
            for (int q = 0; q < ntest && good; q++) {
                const double z = (double)zt[q];
                double g = cd[n - 1];
                for (int k = n - 2; k >= 0; k--)
                    g = std::fma(g, z, cd[k]);
                good = fabsl((long double)g - gt[q]) <= tol * gmax;
            }
            if (good)
                level = lv;
        }
        levels[i] = (signed char)level;
        if (level >= 0)
            for (int k = 0; k < n; k++)
                T.coef[(size_t)i * n + k] = cd[k];
    });
    for (int i = 0; i < T.n_int; i++) {
        const int level = levels[i];
        for (int lv = 0; lv < 3; lv++)
            if (level < lv && T.first_below_level[lv] == T.n_int)
                T.first_below_level[lv] = i;
        if (level >= 0)
            T.n_extended += level >= 1;
        else
            T.n_invalid++;
    }
    T.first_invalid = T.first_below_level[0];
    return T;
}

// ---- polynomials in m on the w-uniform intervals (tables whose variable is w) ----
// Interval i of a w-table (w in [i, i+1) / inv_step) also gets a degree-6 polynomial of G in
// delta = m - m_c, m = 2 w^2, m_c = 2 w_c^2: entry = {c0..c6, m_c}.  Away from m = 0 (interval >= ~100
// of 1024) this is as accurate as the polynomial in w and lets the kernel skip the fp64 square
// root: the interval index only needs an fp32 root.  Levels as in fit_radial_table (range
// |z| <= 0.5 + level in w-interval units, plus 0.01 for the fp32 index), stored in the low
// mantissa bits of c6; intervals that do not validate are NaN.
constexpr int kMPolyDegree = 6;

struct MPolyTable {
    std::vector<double> coef;               // n_int * kRadialCoefs
    std::vector<signed char> level;         // -1: invalid
};

inline MPolyTable fit_mpoly_table(const std::vector<v1c_op>& st, const RadialTable& T)
{
    MPolyTable M;
    const int n_int = T.n_int, n = kMPolyDegree + 1;
    M.coef.assign((size_t)n_int * kRadialCoefs, NAN);
    M.level.assign(n_int, -1);
    const long double step = 1.0L / (long double)T.inv_step;
    long double cheb[16];
    for (int k = 0; k < n; k++)
        cheb[k] = cosl(M_PIl * (2 * k + 1) / (2.0L * n));
    const double tol = 1.5e-15;
    // table variable u (w or m) -> m
    auto m_of_u = [&](long double u) { return T.var_is_w ? 2 * u * u : u; };
    fit_parallel_for(n_int, [&](int i) {
        const long double uc = (i + 0.5L) * step;
        const double mc = (double)m_of_u(uc);  // the double the kernel subtracts
        for (int lv = 2; lv >= 0 && M.level[i] < 0; lv--) {
            const long double zr = 0.5L + lv + 0.01L;
            // (nothing to cover left of u = 0: the table variable is never negative)
            const long double wl = fmaxl((i + 0.5L - zr) * step, T.var_is_w ? 1e-6L * step : 0.0L), wh = (i + 0.5L + zr) * step;
            if ((T.fn == FN_RAY_G || T.fn == FN_RAY_S || T.fn == FN_RAY_CM) && (T.var_is_w ? !(wh < 1) : !(wh < 2)))
                continue;
            const long double dl = m_of_u(wl) - mc, dh = m_of_u(wh) - mc, scale = fmaxl(fabsl(dl), fabsl(dh));
            long double dn[16], y[16], c[16];
            bool good = true;
            for (int k = 0; k < n && good; k++) {
                const long double d = 0.5L * (dl + dh) + 0.5L * (dh - dl) * cheb[k];
                dn[k] = d / scale;
                good = table_fn(st, T.fn, 0, (long double)mc + d, y[k]);
            }
            if (!good || !solve_vandermonde(n, dn, y, c))
                continue;
            double cd[16];
            long double pw = 1;
            for (int k = 0; k < n; k++, pw *= scale)
                cd[k] = (double)(c[k] / pw);
            uint64_t bits;
            std::memcpy(&bits, &cd[n - 1], 8);
            bits = (bits & ~3ull) | (uint64_t)lv;
            std::memcpy(&cd[n - 1], &bits, 8);
            const int ntest = (4 * n + 1) * (2 * lv + 1);
            long double gmax = 0;
            std::vector<long double> gt(ntest);
            std::vector<double> mt(ntest);
            for (int q = 0; q < ntest && good; q++) {
                const long double u = wl + (wh - wl) * (q + 0.5L) / ntest;
                mt[q] = (double)m_of_u(u);  // the kernel holds m as a double
                good = table_fn(st, T.fn, 0, (long double)mt[q], gt[q]);
                gmax = fmaxl(gmax, fabsl(gt[q]));
            }
            for (int q = 0; q < ntest && good; q++) {
                const double d = mt[q] - mc;
                double g = cd[n - 1];
                for (int k = n - 2; k >= 0; k--)
                    g = std::fma(g, d, cd[k]);
                good = fabsl((long double)g - gt[q]) <= tol * gmax;
            }
            if (good) {
                M.level[i] = (signed char)lv;
                for (int k = 0; k < n; k++)
                    M.coef[(size_t)i * kRadialCoefs + k] = cd[k];
                M.coef[(size_t)i * kRadialCoefs + kRadialCoefs - 1] = mc;
            }
        }
    });
    return M;
}

// smallest index such that every interval from it up to the one holding `m_reach` (+2) has a level
// >= `lv`; n_int if there is none.
inline int mpoly_first_ok(const MPolyTable& M, const RadialTable& T, double m_reach, int lv)
{
    const double u_reach = (T.var_is_w ? std::sqrt(m_reach / 2) : m_reach) * (1 + 1e-9);
    const int last = std::min(T.n_int - 1, (int)(u_reach * T.inv_step) + 2);
    int first = T.n_int;
    for (int i = last; i >= 0 && M.level[i] >= lv; i--)
        first = i;
    return first;
}

// How far the table variable moves between horizontally adjacent output pixels, at most: `m` for tables in m, `w` for tables in w.
// Ray tables: adjacent rays are at most `ray_step` radians apart (rotations preserve angles), |d m| <= |d v| <= angle and
// |d w| = |d sin(theta / 2)| <= angle / 2.  Planar tables (m = xn^2 + yn^2, w = t / sqrt(2)): |d m| <= 2 t_max dx + dx^2, |d w| = dx / sqrt(2).
struct TableStep {
    double m = 0, w = 0;
    double of(const RadialTable& T) const { return T.var_is_w ? w : m; }
};
inline TableStep ray_table_step(double ray_step)
{
    return TableStep{ray_step, 0.5 * ray_step};
}
inline TableStep planar_table_step(double dx, double x_max)
{
    return TableStep{2 * x_max * dx + dx * dx, dx * 0.70710678118654757};
}

// level a lane needs so that pixel 1's entry covers its 4 pixels (see ray_entry_is_shared); 0 if none
inline int shared_entry_level(const RadialTable& T, const TableStep& st)
{
    const double delta = T.inv_step * st.of(T);
    for (int lv = 1; lv <= 2; lv++)
        if (delta <= 0.499 * lv)
            return lv;
    return 0;
}

// flagged intervals of a table up to m = m_front
inline int table_bad_upto(const RadialTable& T, double m_front)
{
    const double u_front = T.var_is_w ? std::sqrt(m_front / 2) : m_front;
    const int last = std::min(T.n_int, (int)(u_front * T.inv_step) + 1);
    int bad = 0;
    for (int i = 0; i < last; i++)
        bad += std::isnan(T.coef[(size_t)i * kRadialCoefs]);
    return bad;
}

// Choose the table variable: m when that fits the front hemisphere (theta <= 90 deg; planar tables: the whole table) without a
// flagged interval, otherwise whichever of m / w flags fewer intervals there.
// (`m_front` <= 0: the front hemisphere for ray tables, the whole table for planar ones)
inline RadialTable build_radial_table(const std::vector<v1c_op>& st, int n_int = kTableIntervals, int fn = FN_RAY_G, double m_max = kTableMMax,
                                      int force_var = -1, double m_front = 0)
{
    if (!(m_front > 0))
        m_front = (fn == FN_RAY_G || fn == FN_RAY_S || fn == FN_RAY_CM) ? 1.0 : m_max;
    if (force_var >= 0)
        return fit_radial_table(st, force_var, n_int, fn, m_max);
    RadialTable M = fit_radial_table(st, 0, n_int, fn, m_max);
    if (table_bad_upto(M, m_front) == 0)
        return M;
    RadialTable W = fit_radial_table(st, 1, n_int, fn, m_max);
    return table_bad_upto(W, m_front) < table_bad_upto(M, m_front) ? W : M;
}

// Separable tables of the fused path (host copies) and the largest value of the BASE variable any pixel reaches: m = 1 - v_z of the
// unrotated ray (bases 0 and 2), xn^2 + yn^2 (base 1).
//
// What the six tables hold, by base (the kernels' formulas are written for base 0 and serve the others through these substitutions):
//   base 0 (equirect lat_y)   col: sin(lon), cos(lon), 1 - cos(lon)       row: sin(lat), cos(lat), 1 - cos(lat)
//        v = (row_c col_s, row_s, row_c col_c),  m = row_h + row_c col_h
//   base 1 (planar)           col: xn, xn^2, xn^2                          row: yn, 1, yn^2
//        the same expressions give (xn, yn, .) and m = yn^2 + xn^2 (= the reference's x**2 + y**2, transformer.py:271), so an
//        unrotated planar chain runs the unrotated ray kernels unchanged: x = G(m) r_x xn + c_x with G = F(t) / t tabulated in m = t^2
//   base 2 (equirect lat_x)   col: sin(lat), cos(lat), cos(lat) [sic]      row: sin(lon), cos(lon), 1 - cos(lon)
//        v = (col_s, col_c row_s, col_c row_c)  (transformer.py:557-566): RayParams::gen_mode 1 (gen_vector, v1c_core.hpp), which
//        reads the column's cosine where the other modes read 1 - cos (one loader for both general modes)
struct RayHostTables {
    std::vector<double> col_s, col_c, col_h, row_s, row_c, row_h;
    double m_reach = 0;
    double t_max = 0, x_max = 0;    // base 1: largest sqrt(xn^2 + yn^2), largest |xn|
    bool front_hemisphere = false;  // every unrotated ray has v_z >= 0 (|lon|, |lat| <= 90 deg); base 1: true (unused)
};

#pragma clang fp contract(off)  // follow the reference's rounding order for lat / lon
inline RayHostTables build_ray_host_tables(const RayAnalysis& a, int dst_w, int dst_h)
{
    RayHostTables t;
    const double half_pi = 1.5707963267948966;
    // padded to a multiple of 4 columns (last entry replicated): the kernels read 4 at a time
    const int wpad = (dst_w + 3) & ~3;
    t.col_s.resize(wpad), t.col_c.resize(wpad), t.col_h.resize(wpad);
    t.row_s.resize(dst_h), t.row_c.resize(dst_h), t.row_h.resize(dst_h);
    double h_max = 0, h_min = 1e300;
    for (int i = 0; i < dst_w; i++) {
        // NormalizeTransformer (transformer.py:162) then lon = x * (pi/2) (:547; lat for is_latitude_y=False, :558)
        const double xn = ((double)i - a.norm_cx) / a.norm_s * 2;
        if (a.base == 1) {
            t.col_s[i] = xn, t.col_c[i] = t.col_h[i] = xn * xn;
        } else {
            const double lon = xn * half_pi;
            const double sh = std::sin(lon * 0.5);
            t.col_s[i] = std::sin(lon), t.col_c[i] = std::cos(lon), t.col_h[i] = a.base == 2 ? t.col_c[i] : 2 * sh * sh;
        }
        h_max = std::max(h_max, t.col_h[i]), h_min = std::min(h_min, t.col_h[i]);
    }
    for (int i = dst_w; i < wpad; i++)
        t.col_s[i] = t.col_s[dst_w - 1], t.col_c[i] = t.col_c[dst_w - 1], t.col_h[i] = t.col_h[dst_w - 1];
    for (int j = 0; j < dst_h; j++) {
        const double yn = ((double)j - a.norm_cy) / a.norm_s * 2;  // :163
        if (a.base == 1) {
            t.row_s[j] = yn, t.row_c[j] = 1.0, t.row_h[j] = yn * yn;
            t.m_reach = std::max(t.m_reach, t.row_h[j] + h_max);
            continue;
        }
        const double lat = yn * half_pi;                            // :546 (lon for is_latitude_y=False, :559)
        const double sh = std::sin(lat * 0.5);
        t.row_s[j] = std::sin(lat), t.row_c[j] = std::cos(lat), t.row_h[j] = 2 * sh * sh;
        if (a.base == 0)
            t.m_reach = std::max(t.m_reach, t.row_h[j] + std::max(t.row_c[j] * h_max, t.row_c[j] * h_min));
        else  // m = 1 - col_c row_c  (col_h holds col_c)
            t.m_reach = std::max(t.m_reach, 1.0 - std::min(t.row_c[j] * h_max, t.row_c[j] * h_min));
    }
    if (a.base == 1 && a.has_rows) {
        // a row band of a larger grid: the reach of the WHOLE grid (its first and last row: yn^2 is largest at one of them), so that
        // every band fits the table the unsplit plan fits
        const double y0 = (a.row_lo - a.norm_cy) / a.norm_s * 2, y1 = ((a.row_hi - 1) - a.norm_cy) / a.norm_s * 2;
        t.m_reach = std::max(t.m_reach, std::max(y0 * y0, y1 * y1) + h_max);
    }
    t.t_max = a.base == 1 ? std::sqrt(t.m_reach) : 0.0;
    t.x_max = a.base == 1 ? std::sqrt(h_max) : 0.0;
    t.front_hemisphere = true;
    if (a.base != 1) {
        for (int i = 0; i < dst_w; i++)
            t.front_hemisphere &= t.col_c[i] >= -1e-12;
        for (int j = 0; j < dst_h; j++)
            t.front_hemisphere &= t.row_c[j] >= -1e-12;
    }
    return t;
}
#pragma clang fp contract(fast)

// true when the table should be used at all: at most a quarter of the intervals up to m_front are flagged (ray tables: the front
// hemisphere, m_front = 1).  Planar tables: up to three quarters -- FisheyeEncoder("orthographic") is NaN beyond t = 1
// (transformer.py:372), i.e. on half of a square output's table, by the reference's own arithmetic; those pixels take the fix-up pass
inline bool ray_table_usable(const RadialTable& T, double m_front = 1.0, int max_bad_quarters = 1)
{
    const int front = std::min(T.n_int, T.var_is_w ? (int)(std::sqrt(m_front / 2) * T.inv_step) : (int)(m_front * T.inv_step));
    int bad = 0;
    for (int i = 0; i < front; i++)
        bad += std::isnan(T.coef[(size_t)i * kRadialCoefs]);
    return bad * 4 < front * max_bad_quarters;
}

// Largest m = 1 - (R v)_z over unit vectors v of the front hemisphere (v_z >= 0): the minimum of
// r2 . v there is -sqrt(r20^2 + r21^2) when r22 >= 0 (attained on the rim), -1 otherwise.
inline double rotated_reach(const double* rot)
{
    return rot[8] >= 0 ? 1.0 + std::sqrt(rot[6] * rot[6] + rot[7] * rot[7]) : 2.0;
}
// ... over the cone of half-angle theta_max about the axis: the angle between r2 (a unit row of R) and the axis is alpha, the largest
// angle between r2 and a v of the cone min(pi, alpha + theta_max)  (theta_max = pi / 2: rotated_reach)
inline double rotated_reach_cone(const double* rot, double theta_max)
{
    const double alpha = std::acos(std::min(1.0, std::max(-1.0, rot[8])));
    const double worst = std::min(3.14159265358979323846, alpha + theta_max * (1 + 1e-9) + 1e-12);
    return 1.0 - std::cos(worst);
}

// true when one table entry may serve the 4 horizontally adjacent pixels of a lane
// (tile_device.hpp, OWN = 0): with pixel 1's entry centred at zc, |t_1 - zc| <= 0.5 and
// |t_k - t_1| <= 2 * delta, so every pixel stays inside the validated range |z| <= 0.5 + level
// when delta <= level / 2 -- provided every entry a pixel can select has that level.
// delta = table units per output pixel (TableStep).
inline bool ray_entry_is_shared(const RadialTable& T, double m_reach, const TableStep& st)
{
    const double delta = T.inv_step * st.of(T);
    const double u = (T.var_is_w ? std::sqrt(m_reach / 2) : m_reach) * (1 + 1e-9);
    for (int lv = 1; lv <= 2; lv++)
        if (delta <= 0.499 * lv && u * T.inv_step + 1.0 < (double)T.first_below_level[lv])
            return true;
    return false;
}

// table resolution for an output whose adjacent pixels are `st` apart in the table variable: the finest of 1024 / 512 / 256
// intervals that keeps a lane's 4 pixels within one level-2 entry.  (Coarser tables for small outputs -- the reference's own tests
// remap to 256 x 256, tests/test_remapper.py:73, whose adjacent rays are further apart than a 256-interval table allows -- were
// tried in round 5: a degree-7 piece of a 128-interval table validates at level 2 on the first 49 intervals only, of 64 a single one,
// so no lane could share an entry there either; small outputs keep the 256-interval table and the kernels with a per-pixel entry.)
constexpr int kTableIntervalsMin = 256;
inline int table_intervals_for(const TableStep& st, double m_max = kTableMMax)
{
    // delta <= 0.998 with inv_step <= n / u_max
    for (int n = kTableIntervals; n > kTableIntervalsMin; n /= 2)
        if ((n / m_max) * st.m <= 0.99 && (n / std::sqrt(m_max / 2)) * st.w <= 0.99)
            return n;
    return kTableIntervalsMin;
}

inline bool ray_reach_is_safe(const RadialTable& T, double m_reach);

// ---- everything a plan derives from a chain and an output size on the host (plan.hip; tests/host_emul runs the same function) ----
struct TableSpec {
    const std::vector<v1c_op>* stages;
    int fn, n_int;
    double m_max;
    int force_var;  // -1: choose m / w; 0 / 1: that variable (the S and Cm tables of a plan share one index)
    double m_front; // the variable is chosen by the flagged intervals up to here (0: build_radial_table's default)
};
struct RayPlanHost {
    RayAnalysis a;
    bool usable = false;       // false: the interpreter serves the chain
    RayHostTables ht;
    RadialTable table;         // the main table: G of the stages behind the rotation (all radial stages when nothing rotates)
    TableStep step;            // ... and how far its variable moves between adjacent pixels
    double ray_step = 0;       // largest angle between horizontally adjacent output rays (ray tables)
    double reach_norot = 0;    // main-table m an unrotated plan reaches (classic and planar plans)
    double reach_rot = 0;      // ... the plan's own rotation reaches (has_rot); 2 = anything
    bool has_pre = false;      // gen_mode 2
    RadialTable pre_s, pre_c;  // FN_*_S / FN_*_CM of a.pre, one variable and step for both
    bool pre_safe = true;      // no pixel's base variable lands in a flagged interval of them
};

// `fit`: RadialTable(const TableSpec&) -- plan.hip passes its per-process cache of fits, the tests fit directly
template <typename Fit>
inline RayPlanHost build_ray_plan_host(const v1c_chain& ch, int dst_w, int dst_h, Fit&& fit)
{
    RayPlanHost H;
    H.a = analyze_chain(ch);
    const RayAnalysis& a = H.a;
    if (!a.ok)
        return H;
    H.ht = build_ray_host_tables(a, dst_w, dst_h);
    const RayHostTables& ht = H.ht;
    const double dx = 2.0 / std::fabs(a.norm_s);                       // adjacent pixels in normalised units
    const double base_ray_step = 1.5707963267948966 * dx;              // bases 0 / 2: lon (lat) = x * pi / 2
    H.has_pre = a.gen_mode == 2;
    if (a.base == 1 && !a.has_rot) {
        // planar, nothing rotates: the main table is G = F(t) / t in m = t^2 up to the corners of the output (5 % beyond them: the
        // proofs about the table's end -- ray_entry_is_shared, the fp32 index of the m-polynomials -- want a few intervals of margin)
        const double m_max = std::max(ht.m_reach * 1.05, 1e-6);
        H.step = planar_table_step(dx, ht.x_max);
        H.table = fit(TableSpec{&a.radial, FN_PLANAR_G, table_intervals_for(H.step, m_max), m_max, -1, ht.m_reach});
        H.reach_norot = ht.m_reach;
        H.usable = ray_table_usable(H.table, ht.m_reach, 3);
        return H;
    }
    double theta_max = 1.5707963267948966;  // largest angle off the axis an unrotated ray has where it meets the rotation
    H.ray_step = base_ray_step;
    if (H.has_pre) {
        // S and Cm of the stages in front of the rotation, in the base variable (planar: m = t^2 up to the corners; equirect: m = 1 - v_z)
        const bool planar = a.base == 1;
        const double m_max = planar ? std::max(ht.m_reach * 1.05, 1e-6) : kTableMMax;
        const int fs = planar ? FN_PLANAR_S : FN_RAY_S, fc = planar ? FN_PLANAR_CM : FN_RAY_CM;
        const double m_front = planar ? ht.m_reach : 1.0;
        H.pre_s = fit(TableSpec{&a.pre, fs, kTableIntervals, m_max, -1, m_front});
        H.pre_c = fit(TableSpec{&a.pre, fc, kTableIntervals, m_max, H.pre_s.var_is_w, m_front});
        if (!ray_table_usable(H.pre_s, m_front, planar ? 3 : 1) || !ray_table_usable(H.pre_c, m_front, planar ? 3 : 1))
            return H;
        const double base_reach = ht.m_reach;
        H.pre_safe = ray_reach_is_safe(H.pre_s, base_reach) && ray_reach_is_safe(H.pre_c, base_reach);
        // how fast the ray turns per unit of the base point, and how far off the axis it gets: sampled (F_pre is smooth where it is
        // finite; 2 % on top)
        const double t_hi = planar ? ht.t_max : std::acos(std::max(-1.0, 1.0 - std::min(base_reach, 2.0)));
        const int ns = 4096;
        long double lip = 1, th = 0, f_prev = 0;
        bool any = false;
        for (int q = 1; q <= ns; q++) {
            const long double t = (long double)t_hi * q / ns;
            long double f;
            if (!composite_F(a.pre, t, f))
                continue;
            const long double den = planar ? t : sinl(t);
            if (den > 0)
                lip = fmaxl(lip, fabsl(sinl(f)) / den);
            if (any)
                lip = fmaxl(lip, fabsl(f - f_prev) * ns / (long double)t_hi);
            th = fmaxl(th, fabsl(f));
            f_prev = f, any = true;
        }
        theta_max = (double)th;
        H.ray_step = (double)lip * 1.02 * (planar ? dx : base_ray_step);
    }
    H.step = ray_table_step(H.ray_step);
    H.table = fit(TableSpec{&a.radial, FN_RAY_G, table_intervals_for(H.step), kTableMMax, -1, 0.0});
    H.usable = ray_table_usable(H.table);
    H.reach_norot = ht.m_reach;
    H.reach_rot = H.has_pre ? rotated_reach_cone(a.rot, theta_max) : (ht.front_hemisphere ? rotated_reach(a.rot) : 2.0);
    return H;
}

// Upper bounds of |G|: entry i of the result bounds every table entry 0 .. i over its whole validated range |z| <= 2.5 (sum of the
// coefficient magnitudes); infinity from the first flagged entry on.  With |x32 - cx32| <= |G| |rx32| it bounds the fixed-point
// coordinates of EVERY pixel of a launch whose rays reach entry i at most, inside the source or not.
inline std::vector<double> radial_table_g_bounds(const RadialTable& T)
{
    std::vector<double> out((size_t)T.n_int);
    double bound = 0;
    for (int i = 0; i < T.n_int; i++) {
        const double* c = &T.coef[(size_t)i * kRadialCoefs];
        double b = 0, zp = 1;
        for (int k = 0; k < kRadialCoefs; k++, zp *= 2.5)
            b += std::fabs(c[k]) * zp;
        bound = (b == b) ? std::max(bound, b) : INFINITY;
        out[(size_t)i] = bound;
    }
    return out;
}
inline double radial_table_g_bound(const RadialTable& T, const std::vector<double>& bounds, double m_reach)
{
    const double u = (T.var_is_w ? std::sqrt(m_reach / 2) : m_reach) * (1 + 1e-9);
    const int last = std::min(T.n_int - 1, (int)(u * T.inv_step) + 3);
    return last >= 0 && (size_t)last < bounds.size() ? bounds[(size_t)last] : INFINITY;
}

// true when no pixel of an unrotated chain can land in a flagged interval
inline bool ray_reach_is_safe(const RadialTable& T, double m_reach)
{
    const double u = (T.var_is_w ? std::sqrt(m_reach / 2) : m_reach) * (1 + 1e-9);
    return u * T.inv_step < (double)T.first_invalid;
}

}  // namespace v1c
