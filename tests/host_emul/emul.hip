// emul.hip -- TEST HARNESS: runs the product's __host__ __device__ per-pixel code on the CPU.
//
// The build container has no GPU.  This file includes the very headers the gfx950 kernels are
// made of (vr180_convert_amd/csrc/v1c_core.hpp, radial_fit.hpp) and drives them with plain loops,
// so that the interpreter, the fused ray path (tables + radial fit + fallback rule) and the
// cv2.remap-exact sampler can be checked against the oracle before any GPU minute is spent.
// It is NOT part of the product: nothing in vr180_convert_amd/ loads it.
// Build: hipcc --cuda-host-only -O2 -shared -fPIC (see tests/host_emul/build.py).
#include <cstring>
#include <vector>

#include "../../vr180_convert_amd/csrc/radial_fit.hpp"
#include "../../vr180_convert_amd/csrc/v1c_core.hpp"

using namespace v1c;

template <int CN>
static void remap_cn(const Image& s, const Geom& g, const short* itab, const float* xm, const float* ym, uint8_t* dst,
                     int64_t dst_pitch)
{
    for (int j = 0; j < g.dst_h; j++)
        for (int i = 0; i < g.dst_w; i++) {
            uint8_t px[4] = {0, 0, 0, 0};
            const float x = xm[(size_t)j * g.dst_w + i], y = ym[(size_t)j * g.dst_w + i];
            bool wr;
            switch (g.interp) {
            case V1C_INTER_NEAREST: wr = sample<CN, V1C_INTER_NEAREST>(s, g, itab, x, y, px); break;
            case V1C_INTER_LINEAR: wr = sample<CN, V1C_INTER_LINEAR>(s, g, itab, x, y, px); break;
            case V1C_INTER_CUBIC: wr = sample<CN, V1C_INTER_CUBIC>(s, g, itab, x, y, px); break;
            default: wr = sample<CN, V1C_INTER_LANCZOS4>(s, g, itab, x, y, px); break;
            }
            if (wr)
                for (int c = 0; c < CN; c++)
                    dst[j * dst_pitch + (int64_t)i * CN + c] = px[c];
        }
}

extern "C" {

// mode 0: literal interpreter; mode 1: ray path with per-pixel literal fix-up (as the kernels do).
// stats[0] = 1 if the ray path was usable, stats[1] = pixels that needed the fix-up,
// stats[2] = table variable (0 m / 1 w), stats[3] = flagged intervals, stats[4] = "no fix-up
// launch needed" verdict of the plan (unrotated reach analysis)
int emul_get_map(const v1c_chain* ch, const double* rot_or_null, int w, int h, int mode, float* xmap, float* ymap,
                 long long* stats)
{
    if (stats)
        std::memset(stats, 0, 5 * sizeof(long long));
    if (mode == 0) {
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                double x, y;
                eval_chain_literal(ch, rot_or_null, i, j, x, y);
                xmap[(size_t)j * w + i] = (float)x;
                ymap[(size_t)j * w + i] = (float)y;
            }
        return 0;
    }
    // as plan.hip: the plan's host-side derivation (radial_fit.hpp), fits uncached
    RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    const RayAnalysis& a = H.a;
    if (!a.ok)
        return 1;
    const RadialTable& T = H.table;
    if (stats) {
        stats[2] = T.var_is_w;
        stats[3] = T.n_invalid;
    }
    if (!H.usable)
        return 2;
    const RayHostTables& ht = H.ht;
    RayParams P{};
    P.col_s = ht.col_s.data(), P.col_c = ht.col_c.data(), P.col_h = ht.col_h.data();
    P.row_s = ht.row_s.data(), P.row_c = ht.row_c.data(), P.row_h = ht.row_h.data();
    P.radial = T.coef.data();
    P.inv_step = T.inv_step, P.n_int = T.n_int, P.var_is_w = T.var_is_w, P.has_rot = a.has_rot;
    for (int q = 0; q < 9; q++)
        P.rot[q] = a.rot[q];
    P.rx = a.rx, P.ry = a.ry, P.cx = a.cx, P.cy = a.cy;
    P.rx32 = 32.0 * a.rx, P.ry32 = 32.0 * a.ry, P.cx32 = 32.0 * a.cx, P.cy32 = 32.0 * a.cy;
    P.n_int_f = (double)P.n_int;
    P.gen_mode = a.gen_mode;
    if (H.has_pre) {
        P.pre_s = H.pre_s.coef.data(), P.pre_c = H.pre_c.coef.data();
        P.pre_var_is_w = H.pre_s.var_is_w, P.pre_inv_step = H.pre_s.inv_step, P.pre_n_int = H.pre_s.n_int;
    }
    double R[9];
    const bool use_rot = rot_or_null || a.has_rot;
    for (int q = 0; q < 9; q++)
        R[q] = rot_or_null ? rot_or_null[q] : a.rot[q];
    long long nfix = 0;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            double x, y;
            if (!ray_eval(P, use_rot, R, ht.row_s[j], ht.row_c[j], ht.row_h[j], ht.col_s[i], ht.col_c[i], ht.col_h[i], x, y)) {
                eval_chain_literal(ch, rot_or_null, i, j, x, y);
                nfix++;
            }
            xmap[(size_t)j * w + i] = (float)x;
            ymap[(size_t)j * w + i] = (float)y;
        }
    if (stats) {
        stats[0] = 1;
        stats[1] = nfix;
        stats[4] = !a.has_rot && !rot_or_null && ray_reach_is_safe(T, H.reach_norot);
    }
    return 0;
}

// The lane of 4 adjacent pixels that starts at column i4 of row j, as tile_device.hpp's lane_coords evaluates it with OWN = 1: pixel 1's
// table entry for every pixel within the entry's validated range, the pixel's own entry otherwise.  out[8 k + 0..7] for pixel k: table
// coordinate t, own index, z relative to pixel 1's entry, level of pixel 1's entry, G by the shared rule, G by the own entry, x, y (own).
int emul_lane_probe(const v1c_chain* ch, int w, int h, int j, int i4, double* out)
{
    RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    if (!H.a.ok || !H.usable)
        return 1;
    const RayAnalysis& a = H.a;
    const RadialTable& T = H.table;
    const RayHostTables& ht = H.ht;
    RayParams P{};
    P.radial = T.coef.data();
    P.inv_step = T.inv_step, P.n_int = T.n_int, P.var_is_w = T.var_is_w;
    P.gen_mode = a.gen_mode;
    if (H.has_pre) {
        P.pre_s = H.pre_s.coef.data(), P.pre_c = H.pre_c.coef.data();
        P.pre_var_is_w = H.pre_s.var_is_w, P.pre_inv_step = H.pre_s.inv_step, P.pre_n_int = H.pre_s.n_int;
    }
    double tt[4], mm[4], fx[4], fy[4];
    int idx[4];
    for (int k = 0; k < 4; k++) {
        const int i = i4 + k;
        double m;
        if (a.gen_mode) {
            gen_vector(P, a.rot, ht.row_s[j], ht.row_c[j], ht.row_h[j], ht.col_s[i], ht.col_h[i], fx[k], fy[k], m);
        } else {
            fx[k] = fy[k] = 0;
            m = fma(ht.row_c[j], ht.col_h[i], ht.row_h[j]);
        }
        mm[k] = m;
        const double u = T.var_is_w ? fast_sqrt_half(m) : m;
        tt[k] = u * T.inv_step;
        idx[k] = std::min(table_index(tt[k]), T.n_int - 1);
    }
    const int ic = idx[1];
    const double* e = T.coef.data() + (size_t)ic * kRadialCoefs;
    uint64_t bits;
    std::memcpy(&bits, &e[kRadialDegree], 8);
    const int level = (int)(bits & 3);
    for (int k = 0; k < 4; k++) {
        const double zk = tt[k] - ((double)ic + 0.5);
        const bool usec = std::fabs(zk) <= 0.5 + level;
        double gs = e[kRadialDegree];
        for (int q = kRadialDegree - 1; q >= 0; q--)
            gs = fma(gs, zk, e[q]);
        const double* pc = T.coef.data() + (size_t)idx[k] * kRadialCoefs;
        const double zo = tt[k] - ((double)idx[k] + 0.5);
        double go = pc[kRadialDegree];
        for (int q = kRadialDegree - 1; q >= 0; q--)
            go = fma(go, zo, pc[q]);
        out[8 * k + 0] = tt[k], out[8 * k + 1] = idx[k], out[8 * k + 2] = zk, out[8 * k + 3] = level;
        out[8 * k + 4] = usec ? gs : go, out[8 * k + 5] = go;
        out[8 * k + 6] = fma(go * 32.0 * a.rx, fx[k], 32.0 * a.cx) / 32.0, out[8 * k + 7] = fma(go * 32.0 * a.ry, fy[k], 32.0 * a.cy) / 32.0;
    }
    return 0;
}

// Host model of how the tile kernels evaluate the radial table in one 64 x 16 tile (tile_device.hpp: lane_coords with OWN = 1, fed by
// the slice k_tile_boxes sizes): the tile's slice = the entries of its IN-TABLE pixels when there are at most 64 of them (else the whole
// table); every lane of 4 adjacent pixels reads the entry of its pixel 1 as slice element ic - tab0 CLAMPED into the slice, and
// shared_entry_serves (v1c_core.hpp: the kernels' own rule) decides which of the lane's pixels take that entry instead of their own.
// out[0] = largest |G_lane - G_own| / max(|G_own|, 1) over the tile's in-table pixels (both are within 1.5e-15 of the function when the
// rule is right), out[1] = in-table pixels, out[2] = pixels served by a shared entry, out[3] = lanes whose pixel 1 points outside the
// slice, out[4] = slice entries (0: whole table), out[5] = in-table pixels with a finite own entry that the rule does NOT let share (what
// the OWN = 0 kernels -- selected when the plan proves "one entry per lane": ray_entry_is_shared -- would evaluate wrongly).
// `ignore_read` = 1: the rule as it was before round 5's fix (the index test dropped).
// (`mp`, `mp_first`, `mp_lv`: the m-polynomial twin of a w-table as plan.hip sets it up, or null.  out[6] = pixels of m-polynomial tiles
//  checked, out[7] = the largest |G_mpoly - G_own| / max(|G_own|, 1) over them and over BOTH interval indices an fp32 root within 0.01 of
//  pixel 1's exact table coordinate can give, out[8] = such candidate indices outside the tile's slice [i0 - 1, i1 + 1] or below the level.)
static int lane_model_of_tile(const RayPlanHost& H, int w, int h, int tx, int ty, int ignore_read, double* out, const MPolyTable* mp = nullptr,
                              int mp_first = 0, int mp_lv = 0, const double* rot_override = nullptr)
{
    for (int q = 0; q < 9; q++)
        out[q] = 0;
    if (!H.a.ok || !H.usable)
        return 1;
    const RayAnalysis& a = H.a;
    const RadialTable& T = H.table;
    const RayHostTables& ht = H.ht;
    RayParams P{};
    P.radial = T.coef.data();
    P.inv_step = T.inv_step, P.n_int = T.n_int, P.var_is_w = T.var_is_w;
    P.gen_mode = a.gen_mode;
    if (H.has_pre) {
        P.pre_s = H.pre_s.coef.data(), P.pre_c = H.pre_c.coef.data();
        P.pre_var_is_w = H.pre_s.var_is_w, P.pre_inv_step = H.pre_s.inv_step, P.pre_n_int = H.pre_s.n_int;
    }
    constexpr int TW = 64, TH = 16;
    double tt[TH][TW], mm[TH][TW];
    int idx[TH][TW];
    bool in_table[TH][TW];
    int lo = 0x7fffffff, hi = -1;
    bool all_in = true;
    for (int r = 0; r < TH; r++)
        for (int c = 0; c < TW; c++) {
            const int j = std::min(ty * TH + r, h - 1), i = std::min(tx * TW + c, w - 1);
            double m, fx, fy;
            bool ok = true;
            if (a.gen_mode) {
                ok = gen_vector(P, a.rot, ht.row_s[j], ht.row_c[j], ht.row_h[j], ht.col_s[i], ht.col_h[i], fx, fy, m);
            } else if (a.has_rot || rot_override) {
                const double* R = rot_override ? rot_override : a.rot;
                m = 1.0 - fma(R[6] * ht.row_c[j], ht.col_s[i], fma(R[8] * ht.row_c[j], ht.col_c[i], R[7] * ht.row_s[j]));
            } else {
                m = fma(ht.row_c[j], ht.col_h[i], ht.row_h[j]);
            }
            const double u = T.var_is_w ? fast_sqrt_half(m) : m;
            mm[r][c] = m;
            tt[r][c] = u * T.inv_step;
            const int ir = table_index(tt[r][c]);
            in_table[r][c] = ok && (unsigned)ir < (unsigned)T.n_int;
            all_in = all_in && in_table[r][c];
            idx[r][c] = std::min(ir, T.n_int - 1);
            if (in_table[r][c])
                lo = std::min(lo, idx[r][c]), hi = std::max(hi, idx[r][c]);
        }
    int tab0 = 0, tabn = T.n_int;
    if (lo <= hi && hi - lo + 1 <= 64)
        tab0 = lo, tabn = hi - lo + 1, out[4] = tabn;
    for (int r = 0; r < TH; r++)
        for (int c4 = 0; c4 < TW; c4 += 4) {
            const int ic = idx[r][c4 + 1];
            const int rel = std::min(std::max(ic - tab0, 0), tabn - 1);
            const double* e = T.coef.data() + (size_t)(tab0 + rel) * kRadialCoefs;
            out[3] += (unsigned)(ic - tab0) >= (unsigned)tabn;
            for (int k = 0; k < 4; k++) {
                if (!in_table[r][c4 + k])
                    continue;
                const double zk = tt[r][c4 + k] - ((double)ic + 0.5);
                const bool usec = ignore_read ? shared_entry_serves(zk, e[kRadialDegree], 0, 0, 1) : shared_entry_serves(zk, e[kRadialDegree], ic, tab0, tabn);
                const double* pc = T.coef.data() + (size_t)idx[r][c4 + k] * kRadialCoefs;
                const double zo = tt[r][c4 + k] - ((double)idx[r][c4 + k] + 0.5);
                double go = pc[kRadialDegree], gs = e[kRadialDegree];
                for (int q = kRadialDegree - 1; q >= 0; q--)
                    go = fma(go, zo, pc[q]), gs = fma(gs, zk, e[q]);
                out[1] += 1;
                if (!std::isfinite(go))
                    continue;  // (a flagged interval: NaN coefficients, the pixel goes to the fix-up pass either way)
                if (usec) {
                    out[2] += 1;
                    const double d = std::fabs(gs - go) / std::max(std::fabs(go), 1.0);
                    out[0] = std::max(out[0], std::isfinite(gs) ? d : 1e300);
                } else {
                    out[5] += 1;
                }
            }
        }
    // m-polynomial tiles (k_tile_boxes: every pixel in the table, its intervals from mp_first + 1 on, the slice with one entry either side
    // within 64): lane_coords<..., MPOLY> takes pixel 1's interval from an fp32 root -- off by one near a boundary -- and evaluates all
    // 4 pixels with that entry's polynomial in m - m_c
    const bool sliced = mp_first >= 0;  // (a launch without boxes reads the whole table: emul_unit_rotation_check)
    if (mp && all_in && (!sliced || (lo - 1 >= mp_first && hi - lo + 3 <= 64)) && (tx + 1) * TW <= w && (ty + 1) * TH <= h) {
        for (int r = 0; r < TH; r++)
            for (int c4 = 0; c4 < TW; c4 += 4) {
                const double t1 = tt[r][c4 + 1];
                const int cand[2] = {std::max((int)std::floor(t1 - 0.01), 0), (int)std::floor(t1 + 0.01)};  // (an fp32 root is never negative)
                for (int q = 0; q < 2; q++) {
                    const int ic = cand[q];
                    if (q == 1 && ic == cand[0])
                        continue;
                    if ((sliced && (ic < lo - 1 || ic > hi + 1)) || ic < 0 || ic >= T.n_int || mp->level[ic] < mp_lv) {
                        out[8] += 1;
                        continue;
                    }
                    const double* e = mp->coef.data() + (size_t)ic * kRadialCoefs;
                    for (int k = 0; k < 4; k++) {
                        const double dk = mm[r][c4 + k] - e[kRadialCoefs - 1];
                        double gk = e[kRadialCoefs - 2];
                        for (int d = kRadialCoefs - 3; d >= 0; d--)
                            gk = fma(gk, dk, e[d]);
                        const double* pc = T.coef.data() + (size_t)idx[r][c4 + k] * kRadialCoefs;
                        const double zo = tt[r][c4 + k] - ((double)idx[r][c4 + k] + 0.5);
                        double go = pc[kRadialDegree];
                        for (int d = kRadialDegree - 1; d >= 0; d--)
                            go = fma(go, zo, pc[d]);
                        out[6] += 1;
                        const double dd = std::fabs(gk - go) / std::max(std::fabs(go), 1.0);
                        out[7] = std::max(out[7], std::isfinite(gk) && std::isfinite(go) ? dd : 1e300);
                    }
                }
            }
    }
    return 0;
}

int emul_tile_lane_model(const v1c_chain* ch, int w, int h, int tx, int ty, int ignore_read, double* out)
{
    const RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    double t[9];
    const int rc = lane_model_of_tile(H, w, h, tx, ty, ignore_read, t);
    for (int q = 0; q < 6; q++)
        out[q] = t[q];
    return rc;
}

// ... over every tile of the output (one plan): out[0] = the largest error, out[1 .. 3], [5] summed, out[4] = tiles with a slice,
// out[6 .. 8] = the m-polynomial model (pixels checked, largest error, candidate indices the tile's slice / levels do not cover)
int emul_lane_model_all(const v1c_chain* ch, int w, int h, int ignore_read, double* out)
{
    const RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    // the m-polynomial twin, set up as plan.hip does (plans that prove one entry per lane for a w- or m-table, level = shared_entry_level)
    const MPolyTable* mp = nullptr;
    MPolyTable M;
    int mp_first = 0, mp_lv = 0;
    if (H.a.ok && H.usable) {
        const double reach = H.a.has_rot ? H.reach_rot : H.reach_norot;
        const bool safe = H.a.has_rot ? (H.reach_rot < 2.0 && H.pre_safe && ray_reach_is_safe(H.table, reach)) : ray_reach_is_safe(H.table, reach);
        if (safe && ray_entry_is_shared(H.table, reach, H.step)) {
            mp_lv = shared_entry_level(H.table, H.step);
            M = fit_mpoly_table(H.a.radial, H.table);
            mp_first = mp_lv > 0 ? mpoly_first_ok(M, H.table, reach, mp_lv) : H.table.n_int;
            if (mp_first < H.table.n_int / 2)
                mp = &M;
        }
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, t[9];
    for (int ty = 0; ty < (h + 15) / 16; ty++)
        for (int tx = 0; tx < (w + 63) / 64; tx++) {
            if (lane_model_of_tile(H, w, h, tx, ty, ignore_read, t, mp, mp_first, mp_lv) != 0)
                return 1;
            acc[0] = std::max(acc[0], t[0]);
            acc[1] += t[1], acc[2] += t[2], acc[3] += t[3], acc[4] += t[4] > 0, acc[5] += t[5];
            acc[6] += t[6], acc[7] = std::max(acc[7], t[7]), acc[8] += t[8];
        }
    for (int q = 0; q < 9; q++)
        out[q] = acc[q];
    return 0;
}

// A unit that overrides the rotation of a classic chain (per-frame calibration, cli.py:308-319; BASELINE config 5; v1c_plan_run_auto): what
// plan.hip's decide_launch CLAIMS for it from closed forms -- the rotated reach stays in valid table intervals ("covered": no fix-up
// pass, no flag words), one entry serves a lane (the kernels without a per-pixel test), every |32 x|, |32 y| < 2^21 (the rounding trick
// without clamps) -- against what the pixels DO.  out[0] = front hemisphere, [1] covered, [2] shared, [3] coordinates bounded (claims);
// [4] = pixels that decline the ray path, [5] = in-table pixels the sharing rule refuses, [6] = largest |32 x|, |32 y|, [7] = largest
// shared-entry error (facts); [8 .. 10] = the m-polynomial model where the launch may use that table (pixel checks, largest error,
// candidate entries below the level).  (The claims restate decide_launch: plan.hip is host code of the device library.)
int emul_unit_rotation_check(const v1c_chain* ch, const double* rot, int w, int h, double* out)
{
    for (int q = 0; q < 11; q++)
        out[q] = 0;
    const RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    if (!H.a.ok || !H.usable || H.a.gen_mode != 0 || H.a.base != 0)
        return 1;
    const RadialTable& T = H.table;
    const double reach = rotated_reach(rot);
    const bool front = H.ht.front_hemisphere;
    const bool covered = front && ray_reach_is_safe(T, reach);
    const bool shared = covered && ray_entry_is_shared(T, reach, H.step);
    const double gb = covered ? radial_table_g_bound(T, radial_table_g_bounds(T), reach) : INFINITY;
    const bool bounded = shared && gb * std::fabs(32.0 * H.a.rx) + std::fabs(32.0 * H.a.cx) < 2097152.0 &&
                         gb * std::fabs(32.0 * H.a.ry) + std::fabs(32.0 * H.a.cy) < 2097152.0;
    out[0] = front, out[1] = covered, out[2] = shared, out[3] = bounded;
    // the m-polynomial twin for such a launch ("mpoly_all": k_ray_lin3_rot_pair_raw<..., MP = 1> reads it for every tile): the plan has
    // one when ITS rotation proves one entry per lane, the unit may use it when every interval up to its reach (+ 2) has the level
    const MPolyTable* mp = nullptr;
    MPolyTable M;
    int mp_lv = 0;
    {
        const double preach = H.a.has_rot ? H.reach_rot : H.reach_norot;
        const bool psafe = H.a.has_rot ? (H.reach_rot < 2.0 && H.pre_safe && ray_reach_is_safe(T, preach)) : ray_reach_is_safe(T, preach);
        if (psafe && ray_entry_is_shared(T, preach, H.step) && shared) {
            mp_lv = shared_entry_level(T, H.step);
            M = fit_mpoly_table(H.a.radial, T);
            const int first = mp_lv > 0 ? mpoly_first_ok(M, T, preach, mp_lv) : T.n_int;
            int upto = -1;
            while (upto + 1 < T.n_int && M.level[upto + 1] >= mp_lv)
                upto++;
            const double u_reach = (T.var_is_w ? std::sqrt(reach / 2) : reach) * (1 + 1e-9);
            if (first < T.n_int / 2 && std::min(T.n_int - 1, (int)(u_reach * T.inv_step) + 2) <= upto)
                mp = &M;
        }
    }
    std::vector<float> xm((size_t)w * h), ym((size_t)w * h);
    long long st[5];
    if (emul_get_map(ch, rot, w, h, 1, xm.data(), ym.data(), st) != 0)
        return 2;
    out[4] = (double)st[1];
    double t[9];
    for (int ty = 0; ty < (h + 15) / 16; ty++)
        for (int tx = 0; tx < (w + 63) / 64; tx++) {
            // (mp_first = -1: no tile is excluded for its intervals -- the launch has no boxes, the claim is about the whole reach)
            if (lane_model_of_tile(H, w, h, tx, ty, 0, t, mp, -1, mp_lv, rot) != 0)
                return 3;
            out[5] += t[5], out[7] = std::max(out[7], t[0]);
            if (mp)
                out[8] += t[6], out[9] = std::max(out[9], t[7]), out[10] += t[8];
        }
    if (covered)  // (pixels that declined carry the interpreter's coordinates: anything)
        for (size_t i = 0; i < xm.size(); i++)
            out[6] = std::max(out[6], (double)std::max(std::fabs(32.0f * xm[i]), std::fabs(32.0f * ym[i])));
    return 0;
}

// What the plan derives from a chain and an output size (radial_fit.hpp: build_ray_plan_host), for tests that assert which path a
// chain takes: out[0] = analysis ok, [1] = usable, [2] = base, [3] = gen_mode, [4] = main table fn, [5] = its variable (0 m / 1 w),
// [6] = intervals, [7] = first interval below level 1, [8] = below level 2, [9] = first flagged, [10] = one entry per lane provable
// for the plan's own rotation / none (ray_entry_is_shared at its reach), [11] = no fix-up pass needed.
int emul_plan_info(const v1c_chain* ch, int w, int h, long long* out)
{
    std::memset(out, 0, 12 * sizeof(long long));
    RayPlanHost H = build_ray_plan_host(*ch, w, h, [](const TableSpec& sp) {
        return build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front);
    });
    out[0] = H.a.ok, out[2] = H.a.base, out[3] = H.a.gen_mode;
    if (!H.a.ok)
        return 0;
    const RadialTable& T = H.table;
    out[1] = H.usable, out[4] = T.fn, out[5] = T.var_is_w, out[6] = T.n_int;
    out[7] = T.first_below_level[1], out[8] = T.first_below_level[2], out[9] = T.first_invalid;
    const double reach = H.a.has_rot ? H.reach_rot : H.reach_norot;
    const bool safe = H.a.has_rot ? (H.reach_rot < 2.0 && H.pre_safe && ray_reach_is_safe(T, reach)) : ray_reach_is_safe(T, reach);  // as plan.hip
    out[10] = safe && ray_entry_is_shared(T, reach, H.step);
    out[11] = safe;
    return 0;
}

// `itab` = the fixed-point table of the PRODUCT (v1c_build_itab of libvr180remap.so), passed in by
// the test; null for NEAREST / LINEAR.
int emul_remap(const uint8_t* src, int src_h, int src_w, int64_t src_pitch, int cn, uint8_t* dst, int dst_h, int dst_w,
               int64_t dst_pitch, const float* xm, const float* ym, int interp, int border, const uint8_t* cval,
               const short* itab)
{
    Geom g{};
    g.src_h = src_h, g.src_w = src_w, g.dst_h = dst_h, g.dst_w = dst_w, g.cn = cn;
    g.interp = interp == V1C_INTER_AREA ? V1C_INTER_LINEAR : interp;
    g.border = border;
    for (int k = 0; k < 4; k++)
        g.cval[k] = cval[k];
    Image s{src, src_pitch, src_h, src_w};
    if (cn == 1)
        remap_cn<1>(s, g, itab, xm, ym, dst, dst_pitch);
    else if (cn == 3)
        remap_cn<3>(s, g, itab, xm, ym, dst, dst_pitch);
    else if (cn == 4)
        remap_cn<4>(s, g, itab, xm, ym, dst, dst_pitch);
    else
        return 1;
    return 0;
}
}
