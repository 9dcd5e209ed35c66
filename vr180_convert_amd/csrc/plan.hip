// plan.hip -- host side of the C ABI declared in include/vr180_remap.h.
//
// A plan owns everything the kernels read besides the images: the device copy of the lowered
// chain, the separable row / column tables and the radial table of the fused ray path, OpenCV's
// fixed-point interpolation table for CUBIC / LANCZOS4, and the tile-flag words that connect the
// ray pass to its fix-up pass.  v1c_plan_run only fills kernel arguments and launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vr180_remap.h"
#include "kernels.hpp"
#include "radial_fit.hpp"

using namespace v1c;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(V1C_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// ------------------------------------------------------------------------------------------
// OpenCV's fixed-point interpolation table (initInterTab2D, imgwarp.cpp)
// ------------------------------------------------------------------------------------------
static int cv_round_host(float v)
{
    return (std::fabs(v) < 2147483648.0f) ? (int)std::nearbyintf(v) : INT_MIN;
}

static short sat_short_host(int v)
{
    return (short)std::min(32767, std::max(-32768, v));
}

static void coeffs_1d(int interp, float x, float* c)
{
    if (interp == V1C_INTER_CUBIC) {
        const float A = -0.75f;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
        return;
    }
    // Lanczos4
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                   {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) {
        for (int i = 0; i < 8; i++)
            c[i] = 0;
        c[3] = 1;
        return;
    }
    float sum = 0;
    const double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    for (int i = 0; i < 8; i++) {
        const double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++)
        c[i] *= sum;
}

#pragma clang fp contract(off)
static std::vector<short> build_itab(int interp)
{
    const int K = interp == V1C_INTER_CUBIC ? 4 : 8;
    std::vector<float> t1(32 * K);
    for (int i = 0; i < 32; i++)
        coeffs_1d(interp, i * (1.f / 32), &t1[i * K]);
    std::vector<short> tab((size_t)1024 * K * K);
    for (int fy = 0; fy < 32; fy++)
        for (int fx = 0; fx < 32; fx++) {
            short* e = &tab[(size_t)(fy * 32 + fx) * K * K];
            int isum = 0;
            for (int k1 = 0; k1 < K; k1++)
                for (int k2 = 0; k2 < K; k2++) {
                    const float v = t1[fy * K + k1] * t1[fx * K + k2];
                    e[k1 * K + k2] = sat_short_host(cv_round_host(v * 32768));
                    isum += e[k1 * K + k2];
                }
            if (isum != 32768) {
                // the surplus goes to one of the taps (K/2 .. K/2+1)^2: the largest if the sum is
                // short, the smallest if it is over; strict comparisons, k1-major scan
                const int diff = isum - 32768, h = K / 2;
                int Mk = h * K + h, mk = h * K + h;
                for (int k1 = h; k1 < h + 2; k1++)
                    for (int k2 = h; k2 < h + 2; k2++) {
                        const int q = k1 * K + k2;
                        if (e[q] < e[mk])
                            mk = q;
                        else if (e[q] > e[Mk])
                            Mk = q;
                    }
                if (diff < 0)
                    e[Mk] = (short)(e[Mk] - diff);
                else
                    e[mk] = (short)(e[mk] - diff);
            }
        }
    return tab;
}
#pragma clang fp contract(fast)

extern "C" int v1c_build_itab(int interp, int16_t* out)
{
    if (!out || (interp != V1C_INTER_CUBIC && interp != V1C_INTER_LANCZOS4))
        return fail(V1C_E_INVALID, "v1c_build_itab: interp must be CUBIC or LANCZOS4, out non-NULL");
    const std::vector<short> tab = build_itab(interp);
    std::memcpy(out, tab.data(), tab.size() * sizeof(short));
    return V1C_OK;
}

// ------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------
struct v1c_plan {
    int device = 0;
    int mode = MODE_LITERAL;
    v1c_chain chain{};
    RayAnalysis ana;
    RadialTable table;
    bool chain_has_rot = false;   // any V1C_OP_ROTATE stage (per-unit override allowed)
    int n_rot_stages = 0;
    bool ray_no_rot_safe = false; // no rotation: reachable m stays below the first flagged interval
    bool ray_plan_rot_safe = false;  // the chain's own rotation keeps every ray inside the validated table
    bool front_hemisphere = false;   // unrotated rays all have v_z >= 0
    KernelCtx ctx{};                 // host copy (what the launchers' decisions read) ...
    KernelCtx* ctx_dev = nullptr;    // ... and the plan-resident device copy the tile kernels read (kernels.hpp: TileArgs)
    // Units of launches longer than the kernel-argument block holds (kInlineUnits) travel through this ring of device buffers of
    // kRingUnits records each.  A slot is rewritten by launch_put_units in stream order; a launch on ANOTHER stream than the slot's
    // last user first waits for that user's event.  Launches recorded into a graph take one of the capture slots instead, each
    // handed out once: a graph owns what it replays.
    static constexpr int kRingSlots = 4, kCaptureSlots = 4, kRingUnits = 256;
    DevUnit* ring = nullptr;         // (kRingSlots + kCaptureSlots) x kRingUnits records
    hipEvent_t ring_ev[kRingSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t ring_last[kRingSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool ring_used[kRingSlots] = {false, false, false, false};
    int ring_next = 0, capture_next = 0;
    std::mutex ring_mu;
    int tiles = 0;
    void* tile_boxes = nullptr;   // per-tile source boxes of the tiled kernel (plan rotation)
    int half_dwords = 256;        // LDS dwords per box buffer of the shared-map tile kernel
    const uint32_t* rest_list = nullptr;  // tiles the lean batch kernel leaves to the general one (device)
    int n_rest = 0;
    int lean_half = 256;          // box buffer dwords of the lean batch kernel (<= half_dwords)
    int lean_raw_nwp = 0;         // > 0: batches through k_ray_lin3_batch_lean_raw, box buffers of so many KB
    int strip_len = 0;            // XCD interleave: tiles per strip (0: one block per XCD), tile_xcd_strips()
    // apply_lr pairs of unrotated chains: boxes of the bands that mirror the tiles about the equator and the tiles
    // the mirror launch leaves to the pair kernel (k_ray_lin3_pair_mirror); mirror_boxes == nullptr: not used
    void* mirror_boxes = nullptr;
    const void* mirror_pairs = nullptr;  // (tile box, band box) interleaved, 64 bytes per tile: what the LDS-DMA mirror kernels read (one scalar load)
    const uint32_t* mirror_rest = nullptr;
    int n_mirror_rest = 0;
    int cn_kb = 0;                           // > 0: grayscale / BGRA bilinear through k_ray_lin_cn with box buffers of so many KB
    int mirror_seq_kb = 0;                   // > 0: pairs through k_ray_lin3_pair_mirror_seq (two buffers of so many KB; mirror_rest is made for it)
    const uint32_t* mirror_rest1 = nullptr;  // ... of single-image launches (one eye: its two boxes have a pair's four buffers)
    int n_mirror_rest1 = -1;                 // -1: no single-image mirror launch
    int mirror_raw_nwp = 0;  // > 0: the mirror launch brings its boxes in by LDS-DMA (k_ray_lin3_pair_mirror_raw), buffers of so many KB
    int mirror_h = 0;
    bool disable_fast = false;    // V1C_DISABLE_FAST=1: always use the generic kernels (A/B testing)
    bool disable_coords_bounded = false;  // V1C_DISABLE_COORDS_BOUNDED=1: k_ray_lin3_rot_pair_raw with its clamps (A/B testing)
    bool disable_shared_entry = false;  // V1C_DISABLE_SHARED_ENTRY=1: keep the per-pixel table fallback compiled in
    bool disable_mpoly = false;         // V1C_DISABLE_MPOLY=1: no m-polynomial table (every tile takes the square root)
    bool plan_shared_entry = false;     // one table entry serves a lane's 4 pixels (ray_entry_is_shared)
    double ray_step = 0;                // largest angle between horizontally adjacent output rays
    int mp_valid_upto = -1;             // m-polynomial table (when uploaded): intervals 0 .. this are all valid at the
                                        // level a lane needs (shared_entry_level)
    int gen_mode = 0;                   // RayParams::gen_mode of the plan (general modes: no per-unit rotations, no cn / mirror kernels)
    TableStep step;                     // how far the main table's variable moves between adjacent output pixels
    double m_reach_norot = 0;           // largest m an unrotated ray reaches
    std::vector<double> g_bounds;       // radial_table_g_bounds(table): |G| over the entries 0 .. i
    // The tile-flag words are per plan: a ray pass sets them, the fix-up pass behind it consumes and clears
    // them.  Launch sequences that use them are serialised ACROSS streams (host: flags_mu; device: the next
    // sequence on another stream waits for flags_ev, recorded behind the previous fix-up pass), so one plan
    // may be run from any number of threads / streams.  Plans that need no fix-up pass never touch them.
    std::mutex flags_mu;
    hipEvent_t flags_ev = nullptr;
    hipStream_t flags_stream = nullptr;
    bool flags_pending = false;
    std::atomic<int> last_launch{-1};   // V1C_LAUNCH_* of the most recent launch group of v1c_plan_run (tests: v1c_plan_last_launch)
    // v1c_plan_run_auto: a second device copy of the context whose Denormalize scale a small kernel rewrites from a device-resident radius
    // in front of every such launch; launches on different streams are ordered by an event, like the flag words
    KernelCtx* ctx_dyn = nullptr;
    int* auto_scratch = nullptr;        // behind it: the words the workgroups of k_auto_radius meet in (kernels.hpp: auto_scratch_init)
    std::mutex dyn_mu;
    hipEvent_t dyn_ev = nullptr;
    hipStream_t dyn_stream = nullptr;
    bool dyn_pending = false;
    std::vector<void*> allocs;
};

template <typename T>
static int upload(v1c_plan* p, const std::vector<T>& h, const T** out)
{
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, std::max<size_t>(h.size() * sizeof(T), 16)));
    p->allocs.push_back(d);
    if (!h.empty())
        HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return V1C_OK;
}

static int validate_chain(const v1c_chain* ch)
{
    if (!ch)
        return fail(V1C_E_INVALID, "chain is NULL");
    if (ch->n_ops < 1 || ch->n_ops > V1C_MAX_OPS)
        return fail(V1C_E_INVALID, "chain.n_ops out of range");
    for (int i = 0; i < ch->n_ops; i++) {
        const v1c_op& op = ch->ops[i];
        if (op.opcode < V1C_OP_NORMALIZE || op.opcode > V1C_OP_ROTATE)
            return fail(V1C_E_INVALID, "unknown opcode in chain");
        if (op.nparam < 0 || op.nparam > V1C_MAX_PARAMS)
            return fail(V1C_E_INVALID, "op.nparam out of range");
        if (op.opcode == V1C_OP_RADIAL && (op.iparam < V1C_RAD_ENC_RECTILINEAR || op.iparam > V1C_RAD_RECTDEC_INV))
            return fail(V1C_E_INVALID, "unknown radial kind in chain");
    }
    return V1C_OK;
}

static int validate_geom(int src_h, int src_w, int dst_h, int dst_w, int cn, int interp, int border)
{
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0)
        return fail(V1C_E_INVALID, "image sizes must be positive");
    // cv2.remap works on short coordinates
    if (src_h >= 32768 || src_w >= 32768 || dst_h >= 32768 || dst_w >= 32768)
        return fail(V1C_E_INVALID, "image sizes must be < 32768 (cv2.remap limit)");
    if (cn != 1 && cn != 3 && cn != 4)
        return fail(V1C_E_INVALID, "cn must be 1, 3 or 4");
    if (interp < V1C_INTER_NEAREST || interp > V1C_INTER_LANCZOS4)
        return fail(V1C_E_INVALID, "unknown interpolation");
    if (border < V1C_BORDER_CONSTANT || border > V1C_BORDER_TRANSPARENT)
        return fail(V1C_E_INVALID, "unknown border mode");
    return V1C_OK;
}

extern "C" int v1c_abi_version(void)
{
    return V1C_ABI_VERSION;
}

extern "C" int v1c_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return fail(V1C_E_NODEVICE, "hipGetDeviceCount failed");
    return n;
}

extern "C" const char* v1c_last_error(void)
{
    return g_err.c_str();
}

extern "C" int v1c_plan_destroy(v1c_plan* p)
{
    if (!p)
        return V1C_OK;
    DeviceGuard dg(p->device);
    for (void* d : p->allocs)
        (void)hipFree(d);
    if (p->flags_ev)
        (void)hipEventDestroy(p->flags_ev);
    if (p->dyn_ev)
        (void)hipEventDestroy(p->dyn_ev);
    for (hipEvent_t e : p->ring_ev)
        if (e)
            (void)hipEventDestroy(e);
    delete p;
    return V1C_OK;
}

static int plan_common(v1c_plan* p, int device, int src_h, int src_w, int dst_h, int dst_w, int cn, int interp,
                       int border_mode, const uint8_t border_val[4])
{
    p->device = device;
    Geom& g = p->ctx.g;
    g.src_h = src_h, g.src_w = src_w, g.dst_h = dst_h, g.dst_w = dst_w;
    g.cn = cn;
    g.interp = interp == V1C_INTER_AREA ? V1C_INTER_LINEAR : interp;  // cv2.remap: AREA -> LINEAR
    g.border = border_mode;
    for (int k = 0; k < 4; k++)
        g.cval[k] = border_val ? border_val[k] : 0;
    p->tiles = tiles_per_unit(g);
    if (g.interp == V1C_INTER_CUBIC || g.interp == V1C_INTER_LANCZOS4) {
        const std::vector<short> tab = build_itab(g.interp);
        int rc = upload(p, tab, &p->ctx.itab);
        if (rc)
            return rc;
    }
    return V1C_OK;
}

// The fitted radial tables depend on the chain's radial stages and the interval count only -- not on radius, image sizes, rotation or
// pixel format -- and fitting them (long double evaluation + validation of up to 1 024 intervals, twice for a w-table chain, plus the
// m-polynomial fit) is most of a plan's creation time.  A caller whose radius changes from image to image (radius="auto", the
// reference's default, remapper.py:62-90) creates a plan per image: the fits are kept, the 16 most recently used.
namespace {
struct FitKey {
    std::string ops;  // the radial stages, byte for byte
    int n_int, fn, force_var;
    double m_max, m_front;
    bool operator==(const FitKey& o) const
    {
        return n_int == o.n_int && fn == o.fn && force_var == o.force_var && m_max == o.m_max && m_front == o.m_front && ops == o.ops;
    }
};
struct FitEntry {
    FitKey key;
    std::shared_ptr<const RadialTable> table;
    std::shared_ptr<const MPolyTable> mpoly;  // (fitted on demand)
};
std::mutex g_fit_mu;
std::list<FitEntry> g_fits;  // front = most recently used

FitKey fit_key(const std::vector<v1c_op>& radial, int n_int, int fn, double m_max, int force_var, double m_front = 0)
{
    FitKey k;
    k.n_int = n_int, k.fn = fn, k.m_max = m_max, k.force_var = force_var, k.m_front = m_front;
    if (!radial.empty())
        k.ops.assign((const char*)radial.data(), radial.size() * sizeof(v1c_op));
    return k;
}

// the entry of `key`, moved to the front (created by `make` when missing); g_fit_mu held by the caller
template <typename Make>
FitEntry& fit_entry(const FitKey& key, Make make)
{
    for (auto it = g_fits.begin(); it != g_fits.end(); ++it)
        if (it->key == key) {
            g_fits.splice(g_fits.begin(), g_fits, it);
            return g_fits.front();
        }
    g_fits.push_front(FitEntry{key, make(), nullptr});
    while (g_fits.size() > 24)
        g_fits.pop_back();
    return g_fits.front();
}
}  // namespace

static RadialTable cached_radial_table(const TableSpec& sp)
{
    std::lock_guard<std::mutex> lk(g_fit_mu);
    return *fit_entry(fit_key(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front), [&] {
                return std::make_shared<const RadialTable>(build_radial_table(*sp.stages, sp.n_int, sp.fn, sp.m_max, sp.force_var, sp.m_front));
            }).table;
}

// (the key of the table the polynomials belong to: the spec it was fitted from, variable chosen)
static MPolyTable cached_mpoly_table(const std::vector<v1c_op>& radial, const RadialTable& table)
{
    std::lock_guard<std::mutex> lk(g_fit_mu);
    FitEntry& e = fit_entry(fit_key(radial, table.n_int, table.fn, table.m_max, -2 - table.var_is_w), [&] { return std::make_shared<const RadialTable>(table); });
    if (!e.mpoly)
        e.mpoly = std::make_shared<const MPolyTable>(fit_mpoly_table(radial, table));
    return *e.mpoly;
}

extern "C" int v1c_plan_create(v1c_plan** out, int device, const v1c_chain* chain, int src_h, int src_w, int dst_h,
                               int dst_w, int cn, int interp, int border_mode, const uint8_t border_val[4])
{
    if (!out)
        return fail(V1C_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate_chain(chain);
    if (rc)
        return rc;
    rc = validate_geom(src_h, src_w, dst_h, dst_w, cn, interp, border_mode);
    if (rc)
        return rc;
    DeviceGuard dg(device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");

    v1c_plan* p = new v1c_plan();
    p->chain = *chain;
    {
        const char* e = tuning_env("V1C_DISABLE_FAST");
        p->disable_fast = e && e[0] == '1';
        const char* ecb = tuning_env("V1C_DISABLE_COORDS_BOUNDED");
        p->disable_coords_bounded = ecb && ecb[0] == '1';
        e = tuning_env("V1C_DISABLE_SHARED_ENTRY");
        p->disable_shared_entry = e && e[0] == '1';
        e = tuning_env("V1C_DISABLE_MPOLY");
        p->disable_mpoly = e && e[0] == '1';
    }
    rc = plan_common(p, device, src_h, src_w, dst_h, dst_w, cn, interp, border_mode, border_val);
    if (rc) {
        v1c_plan_destroy(p);
        return rc;
    }
    for (int i = 0; i < chain->n_ops; i++)
        p->n_rot_stages += chain->ops[i].opcode == V1C_OP_ROTATE;
    p->chain_has_rot = p->n_rot_stages > 0;

    // device copy of the chain for the interpreter
    {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, sizeof(v1c_chain));
        if (e == hipSuccess) {
            p->allocs.push_back(d);
            e = hipMemcpy(d, chain, sizeof(v1c_chain), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            v1c_plan_destroy(p);
            return fail(V1C_E_HIP, std::string("chain upload: ") + hipGetErrorString(e));
        }
        p->ctx.chain = (const v1c_chain*)d;
    }

    // (V1C_DEBUG=1, tuning build: where the time of a plan's creation goes)
    const bool dbg_t = [] { const char* d = tuning_env("V1C_DEBUG"); return d && d[0] == '1'; }();
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now_ms();
    double t_fit = t_start, t_up = t_start, t_boxes = t_start;
    RayPlanHost H = build_ray_plan_host(*chain, dst_w, dst_h, [](const TableSpec& sp) { return cached_radial_table(sp); });
    t_fit = now_ms();
    p->ana = H.a;
    if (p->ana.ok) {
        p->ray_step = H.ray_step;
        p->step = H.step;
        p->table = H.table;
        p->gen_mode = H.a.gen_mode;
        if (H.usable) {
            p->mode = MODE_RAY;
            const RayAnalysis& a = p->ana;
            const RayHostTables& ht = H.ht;
            RayParams& r = p->ctx.ray;
            // the six row / column tables in ONE buffer, in the order col_s | col_c | col_h (wpad entries each) | row_s | row_c | row_h
            // (dst_h each): the mirror kernels get its base among their preloaded arguments and derive the others (tile_device.hpp:
            // rowcol_tables_at)
            {
                std::vector<double> all;
                all.reserve(3 * ht.col_s.size() + 3 * ht.row_s.size());
                for (const std::vector<double>* v : {&ht.col_s, &ht.col_c, &ht.col_h, &ht.row_s, &ht.row_c, &ht.row_h})
                    all.insert(all.end(), v->begin(), v->end());
                const double* base = nullptr;
                if ((rc = upload(p, all, &base)) || (rc = upload(p, p->table.coef, &r.radial))) {
                    v1c_plan_destroy(p);
                    return rc;
                }
                const size_t wpad = ht.col_s.size(), hh = ht.row_s.size();
                r.col_s = base, r.col_c = base + wpad, r.col_h = base + 2 * wpad;
                r.row_s = base + 3 * wpad, r.row_c = base + 3 * wpad + hh, r.row_h = base + 3 * wpad + 2 * hh;
            }
            r.gen_mode = a.gen_mode;
            r.pre_s = r.pre_c = nullptr, r.pre_var_is_w = 0, r.pre_inv_step = 0, r.pre_n_int = 0, r.pad3 = 0;
            if (H.has_pre) {
                if ((rc = upload(p, H.pre_s.coef, &r.pre_s)) || (rc = upload(p, H.pre_c.coef, &r.pre_c))) {
                    v1c_plan_destroy(p);
                    return rc;
                }
                r.pre_var_is_w = H.pre_s.var_is_w, r.pre_inv_step = H.pre_s.inv_step, r.pre_n_int = H.pre_s.n_int;
            }
            r.inv_step = p->table.inv_step;
            r.n_int = p->table.n_int;
            r.var_is_w = p->table.var_is_w;
            r.has_rot = a.has_rot;
            for (int q = 0; q < 9; q++)
                r.rot[q] = a.rot[q];
            r.rx = a.rx, r.ry = a.ry, r.cx = a.cx, r.cy = a.cy;
            r.rx32 = 32.0 * a.rx, r.ry32 = 32.0 * a.ry, r.cx32 = 32.0 * a.cx, r.cy32 = 32.0 * a.cy;
            r.n_int_f = (double)r.n_int;
            p->m_reach_norot = H.reach_norot;
            p->g_bounds = radial_table_g_bounds(p->table);
            p->ray_no_rot_safe = !a.has_rot && ray_reach_is_safe(p->table, H.reach_norot);
            p->front_hemisphere = ht.front_hemisphere;
            // (general modes: the S / Cm tables must cover every pixel too; their rotated reach is the cone's, radial_fit.hpp)
            p->ray_plan_rot_safe = a.has_rot && H.reach_rot < 2.0 && H.pre_safe && ray_reach_is_safe(p->table, H.reach_rot);
            p->plan_shared_entry = (p->ray_no_rot_safe && ray_entry_is_shared(p->table, H.reach_norot, p->step)) ||
                                   (p->ray_plan_rot_safe && ray_entry_is_shared(p->table, H.reach_rot, p->step));
            // polynomials in m on the same intervals: tiles whose intervals all qualify need no fp64
            // square root (w-tables) and no fp64 index arithmetic (tile_device.hpp, lane_coords<..., MPOLY>)
            r.radial_m = nullptr, r.mp_first_ok = r.n_int, r.inv_step_f = (float)r.inv_step;
            if (p->plan_shared_entry && !p->disable_shared_entry && !p->disable_mpoly) {
                const double reach = a.has_rot ? H.reach_rot : H.reach_norot;
                const int lv = shared_entry_level(p->table, p->step);
                const MPolyTable mp = cached_mpoly_table(p->ana.radial, p->table);
                const int first = lv > 0 ? mpoly_first_ok(mp, p->table, reach, lv) : r.n_int;
                if (first < r.n_int / 2) {  // worth a second table: most of the image qualifies
                    if ((rc = upload(p, mp.coef, &r.radial_m))) {
                        v1c_plan_destroy(p);
                        return rc;
                    }
                    r.mp_first_ok = first;
                    p->mp_valid_upto = -1;
                    while (p->mp_valid_upto + 1 < r.n_int && mp.level[p->mp_valid_upto + 1] >= lv)
                        p->mp_valid_upto++;
                }
            }
            if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                std::fprintf(stderr, "[v1c] m-polynomial table: %s, first usable interval %d of %d\n", r.radial_m ? "yes" : "no",
                             r.mp_first_ok, r.n_int);
            if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                std::fprintf(stderr,
                             "[v1c] ray plan: base=%d gen_mode=%d fn=%d var=%s n_int=%d first_invalid=%d first_below_level1=%d first_below_level2=%d m_reach=%.6f "
                             "reach_rot=%.6f step=%.3e no_rot_safe=%d plan_rot_safe=%d shared_entry=%d pre=%d pre_safe=%d\n",
                             a.base, a.gen_mode, p->table.fn, r.var_is_w ? "w" : "m", r.n_int, p->table.first_invalid, p->table.first_below_level[1], p->table.first_below_level[2],
                             ht.m_reach, H.reach_rot, p->step.of(p->table), (int)p->ray_no_rot_safe, (int)p->ray_plan_rot_safe, (int)p->plan_shared_entry, (int)H.has_pre, (int)H.pre_safe);
            // tile flags for kMaxUnitsPerLaunch units
            void* d = nullptr;
            const size_t nflag = (size_t)p->tiles * kMaxUnitsPerLaunch * sizeof(uint32_t);
            hipError_t e = hipMalloc(&d, nflag);
            if (e == hipSuccess) {
                p->allocs.push_back(d);
                e = hipMemset(d, 0, nflag);
            }
            if (e != hipSuccess) {
                v1c_plan_destroy(p);
                return fail(V1C_E_HIP, std::string("tile flags: ") + hipGetErrorString(e));
            }
            p->ctx.tile_flags = (uint32_t*)d;
            e = hipEventCreateWithFlags(&p->flags_ev, hipEventDisableTiming);
            if (e != hipSuccess) {
                v1c_plan_destroy(p);
                return fail(V1C_E_HIP, std::string("tile flags event: ") + hipGetErrorString(e));
            }
#ifdef V1C_STAMPS
            {
                void* sb = nullptr;
                if (hipMalloc(&sb, 64) == hipSuccess) {
                    p->allocs.push_back(sb);
                    (void)hipMemset(sb, 0, 64);
                    p->ctx.xmap = (const float*)sb;
                }
            }
#endif
            // the device copy of the context (complete from here on: geometry, tables, weight table) and the unit ring
            {
                void* dctx = nullptr;
                e = hipMalloc(&dctx, sizeof(KernelCtx));
                if (e == hipSuccess) {
                    p->allocs.push_back(dctx);
                    e = hipMemcpy(dctx, &p->ctx, sizeof(KernelCtx), hipMemcpyHostToDevice);
                }
                void* ring = nullptr;
                if (e == hipSuccess)
                    e = hipMalloc(&ring, sizeof(DevUnit) * (size_t)(v1c_plan::kRingSlots + v1c_plan::kCaptureSlots) * v1c_plan::kRingUnits);
                if (e == hipSuccess)
                    p->allocs.push_back(ring);
                for (int k = 0; k < v1c_plan::kRingSlots && e == hipSuccess; k++)
                    e = hipEventCreateWithFlags(&p->ring_ev[k], hipEventDisableTiming);
                if (e != hipSuccess) {
                    v1c_plan_destroy(p);
                    return fail(V1C_E_HIP, std::string("plan context upload: ") + hipGetErrorString(e));
                }
                p->ctx_dev = (KernelCtx*)dctx;
                p->ring = (DevUnit*)ring;
                // ... and the second copy v1c_plan_run_auto rewrites the Denormalize scale of (created here so that the call itself
                // neither allocates nor synchronises: it may be the first thing a stream capture records)
                void* ddyn = nullptr;
                const size_t ctx_bytes = (sizeof(KernelCtx) + 15) & ~(size_t)15;
                e = hipMalloc(&ddyn, ctx_bytes + kAutoScratchInts * sizeof(int));
                if (e == hipSuccess) {
                    p->allocs.push_back(ddyn);
                    e = hipMemcpy(ddyn, &p->ctx, sizeof(KernelCtx), hipMemcpyHostToDevice);
                }
                if (e == hipSuccess) {
                    int init[kAutoScratchInts];
                    auto_scratch_init(init);
                    e = hipMemcpy((char*)ddyn + ctx_bytes, init, sizeof(init), hipMemcpyHostToDevice);
                }
                if (e == hipSuccess)
                    e = hipEventCreateWithFlags(&p->dyn_ev, hipEventDisableTiming);
                if (e != hipSuccess) {
                    v1c_plan_destroy(p);
                    return fail(V1C_E_HIP, std::string("plan context (dynamic radius): ") + hipGetErrorString(e));
                }
                p->ctx_dyn = (KernelCtx*)ddyn;
                p->auto_scratch = (int*)((char*)ddyn + ctx_bytes);
            }
            t_up = now_ms();
            // source boxes of the tiled kernel, computed once (BGR, constant border, linear/cubic/lanczos4)
            const Geom& g = p->ctx.g;
            if (cn_kernel_supports(g) && p->plan_shared_entry && !p->disable_shared_entry && p->gen_mode == 0) {
                // grayscale / BGRA: the same boxes, consumed by k_ray_lin_cn
                void* bx = nullptr;
                e = hipMalloc(&bx, tile_box_bytes(g));
                if (e == hipSuccess) {
                    p->allocs.push_back(bx);
                    e = launch_tile_boxes(p->ctx, p->ctx_dev, bx, true, nullptr);
                }
                std::vector<char> hb(tile_box_bytes(g));
                if (e == hipSuccess)
                    e = hipMemcpy(hb.data(), bx, hb.size(), hipMemcpyDeviceToHost);
                if (e != hipSuccess) {
                    v1c_plan_destroy(p);
                    return fail(V1C_E_HIP, std::string("tile boxes (cn): ") + hipGetErrorString(e));
                }
                p->tile_boxes = bx;
                p->cn_kb = tile_cn_box_kb(hb.data(), g);
                if (const char* kbsw = tuning_env("V1C_CN_KB"))  // A/B: box buffer size; 0 = the generic kernel
                    p->cn_kb = std::min(std::max(std::atoi(kbsw), 0), 16);
                if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                    std::fprintf(stderr, "[v1c] cn = %d tile kernel: box buffers of %d KB\n", g.cn, p->cn_kb);
            }
            if (tile_kernel_supports(g)) {
                void* bx = nullptr;
                e = hipMalloc(&bx, tile_box_bytes(g));
                if (e == hipSuccess) {
                    p->allocs.push_back(bx);
                    e = launch_tile_boxes(p->ctx, p->ctx_dev, bx, p->plan_shared_entry && !p->disable_shared_entry, nullptr);
                }
                if (e != hipSuccess) {
                    v1c_plan_destroy(p);
                    return fail(V1C_E_HIP, std::string("tile boxes: ") + hipGetErrorString(e));
                }
                p->tile_boxes = bx;
                // size the kernel's two LDS box buffers from the largest tile box of this plan
                std::vector<char> hb(tile_box_bytes(g));
                e = hipMemcpy(hb.data(), bx, hb.size(), hipMemcpyDeviceToHost);
                if (e != hipSuccess) {
                    v1c_plan_destroy(p);
                    return fail(V1C_E_HIP, std::string("tile boxes readback: ") + hipGetErrorString(e));
                }
                p->half_dwords = tile_half_dwords(hb.data(), hb.size() / 32, (p->gen_mode != 0 && (g.interp == V1C_INTER_LANCZOS4 || g.interp == V1C_INTER_CUBIC)) ? 2048 : 1024,
                                                  g.interp == V1C_INTER_LINEAR || g.interp == V1C_INTER_NEAREST);
                if (const char* e = tuning_env("V1C_HALF_CAP"); e && std::atoi(e) >= 256)  // A/B: cap the LDS box buffers
                    p->half_dwords = std::min(p->half_dwords, std::atoi(e));
                {
                    p->lean_half = tile_lean_half_dwords(p->half_dwords);
                    p->strip_len = tile_xcd_strips(hb.data(), g, p->half_dwords, p->lean_half);
                    if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                        std::fprintf(stderr, "[v1c] XCD interleave: strips of %d tiles\n", p->strip_len);
                    {  // V1C_LEAN_RAW=0: the register-staged lean kernel, <n> > 1: n KB per box buffer
                        const char* sw = tuning_env("V1C_LEAN_RAW");
                        const int v = sw ? std::atoi(sw) : 1;
                        p->lean_raw_nwp = v == 1 ? tile_lean_raw_passes(hb.data(), g) : v > 1 ? std::min(v, 16) : 0;
                    }
                    const std::vector<uint32_t> rest = tile_rest_list(hb.data(), g, p->lean_half, p->lean_raw_nwp);
                    const dim3 full((unsigned)((g.dst_w + 63) / 64), (unsigned)((g.dst_h + 15) / 16));
                    if (full.x <= 0xffffu && full.y <= 0xffffu) {
                        if ((rc = upload(p, rest, &p->rest_list))) {
                            v1c_plan_destroy(p);
                            return rc;
                        }
                        p->n_rest = (int)rest.size();
                    }
                    if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                        std::fprintf(stderr, "[v1c] lean batch kernel: %d of %zu tiles left to the general kernel (raw box KB %d)\n", p->n_rest, hb.size() / 32, p->lean_raw_nwp);
                }
                t_boxes = now_ms();
                // bilinear pairs of an unrotated chain whose rows mirror about an integer row (the default Normalize
                // centre H / 2): boxes of the mirrored bands + the list of tiles that launch leaves to the pair kernel
                {
                    const char* off = tuning_env("V1C_DISABLE_MIRROR");
                    const double two_cy = 2.0 * p->ana.norm_cy;
                    if (!(off && off[0] == '1') && g.interp == V1C_INTER_LINEAR && !p->ana.has_rot && p->plan_shared_entry &&
                        !p->disable_shared_entry && two_cy == (double)g.dst_h) {
                        void* mbx = nullptr;
                        e = hipMalloc(&mbx, tile_box_bytes(g));
                        if (e == hipSuccess) {
                            p->allocs.push_back(mbx);
                            e = launch_tile_boxes(p->ctx, p->ctx_dev, mbx, true, nullptr, g.dst_h);
                        }
                        std::vector<char> hm(tile_box_bytes(g));
                        if (e == hipSuccess)
                            e = hipMemcpy(hm.data(), mbx, hm.size(), hipMemcpyDeviceToHost);
                        if (e != hipSuccess) {
                            v1c_plan_destroy(p);
                            return fail(V1C_E_HIP, std::string("mirror boxes: ") + hipGetErrorString(e));
                        }
                        std::vector<uint32_t> mrest;
                        // boxes by LDS-DMA (k_ray_lin3_pair_mirror_raw); V1C_MIRROR_RAW=0: the register-staged form, <n> > 1: n KB per box
                        const char* rawsw = tuning_env("V1C_MIRROR_RAW");
                        const int rawv = rawsw ? std::atoi(rawsw) : 1;
                        p->mirror_raw_nwp = rawv == 1 ? tile_mirror_raw_passes(hb.data(), hm.data(), g) : rawv > 1 ? std::min(rawv, 16) : 0;
                        // pairs: the eyes one after the other through two buffers of twice the size (k_ray_lin3_pair_mirror_seq: 99.8 % of
                        // the tile pairs fit, C2 -0.4 ... -2.6 %, C1 -1.6 ... -3 % against the four-buffer kernel on three boxes,
                        // HISTORY.md 4.4c); V1C_MIRROR_SEQ=0 (A/B): k_ray_lin3_pair_mirror_raw for pairs too; V1C_MIRROR_SEQ_KB=<n>: buffer size
                        const char* seqsw = tuning_env("V1C_MIRROR_SEQ");
                        const char* seqkb = tuning_env("V1C_MIRROR_SEQ_KB");
                        if (p->mirror_raw_nwp > 0 && !(seqsw && seqsw[0] == '0'))
                            p->mirror_seq_kb = seqkb ? std::min(std::max(std::atoi(seqkb), 2), 16) : tile_mirror_raw_passes(hb.data(), hm.data(), g, 998, 11);
                        if (tile_mirror_rest(hb.data(), hm.data(), g, p->half_dwords, g.dst_h, mrest,
                                             p->mirror_seq_kb > 0 ? p->mirror_seq_kb : p->mirror_raw_nwp, true, 2)) {
                            if ((rc = upload(p, mrest, &p->mirror_rest))) {
                                v1c_plan_destroy(p);
                                return rc;
                            }
                            p->mirror_boxes = mbx, p->n_mirror_rest = (int)mrest.size(), p->mirror_h = g.dst_h;
                            {
                                std::vector<char> pairs(2 * hb.size());
                                for (size_t i = 0; i < hb.size() / 32; i++) {
                                    std::memcpy(&pairs[64 * i], &hb[32 * i], 32);
                                    std::memcpy(&pairs[64 * i + 32], &hm[32 * i], 32);
                                }
                                const char* dp = nullptr;
                                if ((rc = upload(p, pairs, &dp))) {
                                    v1c_plan_destroy(p);
                                    return rc;
                                }
                                p->mirror_pairs = dp;
                            }
                            std::vector<uint32_t> mrest1;
                            if (p->mirror_raw_nwp > 0 && tile_mirror_rest(hb.data(), hm.data(), g, p->half_dwords, g.dst_h, mrest1, p->mirror_raw_nwp, true, 1)) {
                                if ((rc = upload(p, mrest1, &p->mirror_rest1))) {
                                    v1c_plan_destroy(p);
                                    return rc;
                                }
                                p->n_mirror_rest1 = (int)mrest1.size();
                            }
                            if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                                std::fprintf(stderr, "[v1c] mirror launch: seq buffers %d KB\n", p->mirror_seq_kb);
                        }
                        if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1')
                            std::fprintf(stderr, "[v1c] mirror pair launch: %s, %zu of %zu tiles left to the pair kernel (raw wave-passes %d)\n",
                                         p->mirror_boxes ? "yes" : "no", mrest.size(), hb.size() / 32, p->mirror_raw_nwp);
                    }
                }
                if (const char* dbg = tuning_env("V1C_DEBUG"); dbg && dbg[0] == '1') {
                    // histogram of the LDS dwords each tile box needs
                    const int* bi = (const int*)hb.data();
                    const size_t nt = hb.size() / 32;
                    int hist[12] = {0};
                    for (size_t i = 0; i < nt; i++) {
                        const int cpr = bi[i * 8 + 2], nrows = bi[i * 8 + 3];
                        const int need = cpr > 0 ? nrows * (cpr * 4 + 4) : 0;  // = tile_device.hpp LDS row pitch
                        int k = 0;
                        while (k < 11 && need > (512 << k) / 2)
                            k++;
                        hist[k]++;
                    }
                    std::fprintf(stderr, "[v1c] tile boxes: %zu tiles, half_dwords=%d; need<=256:%d <=512:%d <=1k:%d <=2k:%d <=4k:%d <=8k:%d <=16k:%d more:%d\n",
                                 nt, p->half_dwords, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7] + hist[8] + hist[9] + hist[10] + hist[11]);
                }
            }
        }
    }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        v1c_plan_destroy(p);
        return fail(V1C_E_HIP, std::string("plan upload: ") + hipGetErrorString(e));
    }
    if (dbg_t)
        std::fprintf(stderr, "[v1c] plan_create: analysis + fits %.3f ms, tables / context / ring %.3f, tile boxes + sizing + rest list %.3f, mirror boxes + lists + sync %.3f\n",
                     t_fit - t_start, t_up - t_fit, t_boxes - t_up, now_ms() - t_boxes);
    *out = p;
    return V1C_OK;
}

#ifdef V1C_STAMPS
// diagnostic build only (tools/README.md): per-phase cycle sums written by the tile kernel
extern "C" int v1c_debug_read_stamps(v1c_plan* p, unsigned long long* out8)
{
    if (!p || !p->ctx.xmap)
        return V1C_E_INVALID;
    DeviceGuard dg(p->device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out8, p->ctx.xmap, 64, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset((void*)p->ctx.xmap, 0, 64));
    return V1C_OK;
}
#endif

extern "C" int v1c_plan_path(const v1c_plan* p)
{
    if (!p)
        return fail(V1C_E_INVALID, "plan is NULL");
    return p->mode == MODE_RAY ? (p->ana.base == 1 ? 2 : 1) : 0;
}

static int fill_unit(const v1c_plan* p, const v1c_unit& in, DevUnit& out)
{
    if (!in.src || !in.dst)
        return fail(V1C_E_INVALID, "unit src/dst is NULL");
    const Geom& g = p->ctx.g;
    if (in.src_pitch < (int64_t)g.src_w * g.cn || in.dst_pitch < (int64_t)g.dst_w * g.cn)
        return fail(V1C_E_INVALID, "unit pitch smaller than a row");
    if (in.has_rot && !p->chain_has_rot)
        return fail(V1C_E_INVALID, "unit carries a rotation but the plan's chain has no rotate stage");
    out.src = in.src, out.dst = in.dst;
    out.src_pitch = in.src_pitch, out.dst_pitch = in.dst_pitch;
    out.has_rot = in.has_rot ? 1 : 0;
    out.pad = 0;
    // `rot` always holds the EFFECTIVE rotation of the fused ray path -- the unit's override or the chain's composed one --: the tile
    // kernels read it unconditionally (the generic kernels and the interpreter still look at has_rot)
    const bool plan_rot = p->mode == MODE_RAY && p->ana.has_rot;
    for (int q = 0; q < 9; q++)
        out.rot[q] = in.has_rot ? in.rot[q] : (plan_rot ? p->ana.rot[q] : 0.0);
    return V1C_OK;
}

// the by-value argument block of the generic kernels (kernels.hip): up to kMaxUnitsPerLaunch records
static UnitArgs unit_args(const DevUnit* du, int n)
{
    UnitArgs ua;
    std::memset(&ua, 0, sizeof(ua));
    std::memcpy(ua.u, du, sizeof(DevUnit) * (size_t)n);
    return ua;
}

// What a launch of units [0, n) of `du` may use, decided from the plan and the units themselves
struct LaunchPlan {
    bool any_rot = false;        // some unit overrides the rotation: no plan-time boxes
    bool fast = false;           // the hand-tuned tile kernels (32-bit source / destination offsets)
    bool need_fixup = false;     // a fix-up pass must follow (pixels may land in flagged table intervals)
    bool shared_entry = false;   // one table entry per lane, no per-pixel fallback in the kernel
    bool mpoly_all = false;      // the m-polynomial table serves every interval the units reach
    bool coords_bounded = false; // every |32 x|, |32 y| < 2^21
};

// `r32_limit` > 0: bound the coordinates for |32 radius| up to it instead of the plan's own Denormalize scale (v1c_plan_run_auto)
static LaunchPlan decide_launch(const v1c_plan* p, const DevUnit* du, int n, double r32_limit = 0)
{
    LaunchPlan d;
    for (int k = 0; k < n; k++)
        d.any_rot |= du[k].has_rot != 0;
    const Geom& g = p->ctx.g;
    d.fast = p->tile_boxes != nullptr && !p->disable_fast;
    for (int k = 0; k < n && d.fast; k++)
        d.fast = (uint64_t)g.src_h * (uint64_t)du[k].src_pitch < 0xFFFFFF00ull && du[k].src_pitch < (1 << 24) &&
                 (uint64_t)g.dst_h * (uint64_t)du[k].dst_pitch < 0xFFFFFF00ull && du[k].dst_pitch < (1 << 24);
    // the fix-up pass is skipped when no pixel can land in a flagged table interval: proven at
    // plan time for the chain's own rotation, per unit for overriding rotations; likewise
    // `shared_entry` (one table entry per lane, no per-pixel fallback in the kernel)
    d.need_fixup = !(p->ray_no_rot_safe || p->ray_plan_rot_safe);
    d.shared_entry = p->plan_shared_entry;
    // the m-polynomial table (no fp64 index arithmetic) serves a launch of overriding rotations
    // when every interval up to each unit's reach is valid at the level the lanes need
    d.mpoly_all = d.any_rot && p->ctx.ray.radial_m != nullptr;
    // every pixel's |32 x|, |32 y| provably below 2^21 (half the cvRound trick's range): |x32 - cx32| <= |G| |rx32| with G bounded
    // over every table entry in reach -- only ever used together with shared_entry && !need_fixup
    d.coords_bounded = d.any_rot && p->front_hemisphere;
    if (d.any_rot) {
        d.need_fixup = !p->front_hemisphere;
        d.shared_entry = p->front_hemisphere;
        for (int k = 0; k < n && !d.need_fixup; k++) {
            const double* r = du[k].has_rot ? du[k].rot : p->ana.rot;
            const bool rotated = du[k].has_rot || p->ana.has_rot;
            const bool covered = rotated ? ray_reach_is_safe(p->table, rotated_reach(r)) : p->ray_no_rot_safe;
            d.need_fixup = !covered;
            if (d.coords_bounded) {
                const RayParams& rp = p->ctx.ray;
                const double gb = covered ? radial_table_g_bound(p->table, p->g_bounds, rotated ? rotated_reach(r) : p->m_reach_norot) : INFINITY;
                const double ax = r32_limit > 0 ? r32_limit : std::fabs(rp.rx32), ay = r32_limit > 0 ? r32_limit : std::fabs(rp.ry32);
                d.coords_bounded = gb * ax + std::fabs(rp.cx32) < 2097152.0 && gb * ay + std::fabs(rp.cy32) < 2097152.0;
            }
            d.shared_entry = d.shared_entry && covered &&
                             (rotated ? ray_entry_is_shared(p->table, rotated_reach(r), p->step) : p->plan_shared_entry);
            if (d.mpoly_all) {  // (same interval bound as mpoly_first_ok: reach + 2 for the fp32 index)
                const double mr = rotated ? rotated_reach(r) : p->m_reach_norot;
                const double u_reach = (p->table.var_is_w ? std::sqrt(mr / 2) : mr) * (1 + 1e-9);
                d.mpoly_all = covered && std::min(p->table.n_int - 1, (int)(u_reach * p->table.inv_step) + 2) <= p->mp_valid_upto;
            }
        }
        d.shared_entry = d.shared_entry && !d.need_fixup;
        d.mpoly_all = d.mpoly_all && d.shared_entry;
        d.coords_bounded = d.coords_bounded && d.shared_entry;
    }
    return d;
}

// device copy of the units of a launch longer than the kernel arguments hold: a ring slot (class comment of v1c_plan).
// The caller holds ring_mu from here until ring_done() has recorded the slot's event behind the remap launch that reads it: the slot's
// bookkeeping (ring_used / ring_last / ring_ev) is only complete then, and a second thread that picked the same slot in between would
// see the stale state, skip its hipStreamWaitEvent and overwrite records a launch in flight is still reading (advisor finding of round
// 4).  Everything between the two calls is asynchronous launches: the critical section is a few microseconds.
static int ring_put(v1c_plan* p, hipStream_t st, const DevUnit* du, int n, const DevUnit** out, int* slot_out)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess)
        cap = hipStreamCaptureStatusNone;
    int slot;
    if (cap != hipStreamCaptureStatusNone) {
        if (p->capture_next >= v1c_plan::kCaptureSlots)
            return fail(V1C_E_UNSUPPORTED, "too many graph-captured launches of more than 16 units for one plan (v1c_plan_release_captures "
                                           "hands the slots out again once their graphs are destroyed)");
        slot = v1c_plan::kRingSlots + p->capture_next++;
        *slot_out = -1;
    } else {
        slot = p->ring_next;
        p->ring_next = (p->ring_next + 1) % v1c_plan::kRingSlots;
        if (p->ring_used[slot] && p->ring_last[slot] != st)
            HIP_TRY(hipStreamWaitEvent(st, p->ring_ev[slot], 0));
        *slot_out = slot;
    }
    DevUnit* dst = p->ring + (size_t)slot * v1c_plan::kRingUnits;
    HIP_TRY(launch_put_units(dst, du, n, st));
    *out = dst;
    return V1C_OK;
}

static int ring_done(v1c_plan* p, hipStream_t st, int slot)
{
    if (slot < 0)
        return V1C_OK;
    HIP_TRY(hipEventRecord(p->ring_ev[slot], st));
    p->ring_last[slot] = st, p->ring_used[slot] = true;
    return V1C_OK;
}

// Graph-captured launches of more than 16 units take one of the plan's kCaptureSlots capture-owned unit buffers each (a graph owns what
// it replays).  Once every graph that recorded such a launch of this plan has been destroyed the caller may hand the slots out again.
extern "C" int v1c_plan_release_captures(v1c_plan* p)
{
    if (!p)
        return fail(V1C_E_INVALID, "plan is NULL");
    std::lock_guard<std::mutex> lk(p->ring_mu);
    p->capture_next = 0;
    return V1C_OK;
}

extern "C" int v1c_plan_run(v1c_plan* p, void* stream, const v1c_unit* units, int n_units)
{
    if (!p || !units || n_units <= 0)
        return fail(V1C_E_INVALID, "v1c_plan_run: bad arguments");
    DeviceGuard dg(p->device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    DevUnit du_small[kMaxUnitsPerLaunch];
    std::vector<DevUnit> du_big;
    DevUnit* du = du_small;
    if (n_units > kMaxUnitsPerLaunch) {
        du_big.resize((size_t)n_units);
        du = du_big.data();
    }
    for (int k = 0; k < n_units; k++) {
        int rc = fill_unit(p, units[k], du[k]);
        if (rc)
            return rc;
    }
    for (int base = 0; base < n_units;) {
        if (p->mode != MODE_RAY) {
            const int n = std::min(kMaxUnitsPerLaunch, n_units - base);
            HIP_TRY(launch_remap(MODE_LITERAL, p->ctx, unit_args(du + base, n), n, st));
            p->last_launch.store(V1C_LAUNCH_GENERIC, std::memory_order_relaxed);
            base += n;
            continue;
        }
        // A launch of the tile kernels takes any number of units (beyond kInlineUnits through the ring) -- unless a fix-up pass
        // must follow (its flag words cover kMaxUnitsPerLaunch unit slots) or the generic kernels serve it (by-value arguments)
        int n = std::min(v1c_plan::kRingUnits, n_units - base);
        LaunchPlan d = decide_launch(p, du + base, n);
        // (the general modes -- lat_x, radial stages in front of the rotation -- have their reach, boxes and proofs for the plan's own rotation only)
        const bool literal = d.any_rot && (p->n_rot_stages > 1 || p->gen_mode != 0);
        if (n > kMaxUnitsPerLaunch && (literal || !d.fast || d.need_fixup || p->ctx.g.cn != 3)) {
            n = kMaxUnitsPerLaunch;
            d = decide_launch(p, du + base, n);
        }
        const DevUnit* u = du + base;
        base += n;
        // units overriding one of SEVERAL rotate stages take the interpreter (rare); all other
        // cases run the ray pass and, unless provably unnecessary, the fix-up pass
        if (literal) {
            HIP_TRY(launch_remap(MODE_LITERAL, p->ctx, unit_args(u, n), n, st));
            p->last_launch.store(V1C_LAUNCH_GENERIC, std::memory_order_relaxed);
            continue;
        }
        std::unique_lock<std::mutex> flags_lk(p->flags_mu, std::defer_lock);
        bool capturing = false;  // (a recorded launch neither waits for nor records the plan's events: v1c_plan_run_auto's comment)
        if (d.need_fixup) {
            flags_lk.lock();
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cap) != hipSuccess)
                cap = hipStreamCaptureStatusNone;
            capturing = cap != hipStreamCaptureStatusNone;
            if (!capturing && p->flags_pending && p->flags_stream != st)
                HIP_TRY(hipStreamWaitEvent(st, p->flags_ev, 0));
        }
        uint32_t* flags = d.need_fixup ? p->ctx.tile_flags : nullptr;
        LaunchUnits lu{u, nullptr, n};
        int slot = -1;
        std::unique_lock<std::mutex> ring_lk(p->ring_mu, std::defer_lock);  // slot selection ... the slot's event (ring_put's comment)
        if (d.fast && n > kInlineUnits) {
            ring_lk.lock();
            int rc = ring_put(p, st, u, n, &lu.dev, &slot);
            if (rc)
                return rc;
        }
        const bool shared = d.shared_entry && !p->disable_shared_entry;
        bool aligned = true;  // every source and its pitch dword-aligned (LDS-DMA kernels)
        for (int k = 0; k < n && aligned; k++)
            aligned = ((((uintptr_t)u[k].src) | (uintptr_t)u[k].src_pitch) & 3u) == 0;
        // a pair (apply_lr) of an unrotated chain: the tile + mirror-image launch
        // (a single image -- apply() of one image, BASELINE config 1 -- takes the LDS-DMA form's one-eye instantiation)
        const bool mirror = d.fast && (n == 2 || (n == 1 && p->mirror_raw_nwp > 0 && p->n_mirror_rest1 >= 0)) && !d.any_rot &&
                            p->mirror_boxes != nullptr && shared && aligned;
        int kind = V1C_LAUNCH_GENERIC;
        if (mirror) {
            kind = V1C_LAUNCH_MIRROR;
            HIP_TRY(launch_ray_lin3_pair_mirror(p->ctx, p->ctx_dev, lu, flags, p->tile_boxes, p->mirror_raw_nwp > 0 ? p->mirror_pairs : p->mirror_boxes,
                                                p->half_dwords, p->mirror_h,
                                                n == 1 ? p->mirror_rest1 : p->mirror_rest, n == 1 ? p->n_mirror_rest1 : p->n_mirror_rest,
                                                p->mirror_raw_nwp, st, p->mirror_seq_kb));
        } else if (d.fast && p->ctx.g.cn != 3) {
            // grayscale / BGRA: k_ray_lin_cn (one table entry per lane; plan-time boxes for the plan's own rotation, boxes reduced in the
            // kernel -- one unit per workgroup, a 12 KB buffer -- for units that override it)
            if (p->cn_kb > 0 && !d.any_rot && shared && aligned) {
                kind = V1C_LAUNCH_CN;
                HIP_TRY(launch_ray_lin_cn(p->ctx, p->ctx_dev, lu, flags, p->ana.has_rot, p->tile_boxes, p->cn_kb, st));
            } else if (d.any_rot && shared && aligned && cn_kernel_supports(p->ctx.g)) {
                kind = V1C_LAUNCH_CN_ROT;
                HIP_TRY(launch_ray_lin_cn(p->ctx, p->ctx_dev, lu, flags, true, nullptr, 12, st));
            } else {
                HIP_TRY(launch_remap(MODE_RAY, p->ctx, unit_args(u, n), n, st));
            }
        } else if (d.fast) {
            // precomputed tile boxes describe the plan's own rotation only
            kind = V1C_LAUNCH_TILE;
            HIP_TRY(launch_ray_lin3_tile(p->ctx, p->ctx_dev, lu, flags, d.any_rot || p->ana.has_rot, d.any_rot ? nullptr : p->tile_boxes,
                                         p->half_dwords, shared, d.mpoly_all && !p->disable_mpoly, d.any_rot ? nullptr : p->rest_list, p->n_rest,
                                         p->lean_half, p->strip_len, p->lean_raw_nwp, st, d.coords_bounded && !p->disable_coords_bounded, &kind));
        } else {
            HIP_TRY(launch_remap(MODE_RAY, p->ctx, unit_args(u, n), n, st));
        }
        if (slot >= 0) {
            int rc = ring_done(p, st, slot);
            if (rc)
                return rc;
        }
        if (ring_lk.owns_lock())
            ring_lk.unlock();
        p->last_launch.store(kind | (d.need_fixup ? V1C_LAUNCH_FIXUP : 0), std::memory_order_relaxed);
        if (d.need_fixup) {
            HIP_TRY(launch_remap(MODE_FIXUP, p->ctx, unit_args(u, n), n, st));
            if (!capturing) {
                HIP_TRY(hipEventRecord(p->flags_ev, st));
                p->flags_stream = st, p->flags_pending = true;
            }
        }
    }
    return V1C_OK;
}

// The same launch with the radius taken from device memory: remapper.py:379-386 with radius="auto" -- get_radius_smart's max over the
// images of get_radius (:83-84) becomes the Denormalize scale (radius, radius) of get_map (:55) -- without the value ever visiting the
// host.  `rad_dev`: n_rad (radius, status) pairs as v1c_get_radius_async writes them.  Nothing of the plan that depends on the radius is
// used: the launch runs the kernels that reduce their source boxes themselves (the per-unit-rotation family, identity where nothing
// rotates), reads the scale from a second plan-resident context a one-thread kernel has just rewritten, and is only taken when the
// plan proves -- for ANY radius up to r_limit in magnitude -- that no fix-up pass is needed.  Launch-only: graph-capturable.
// (`rad_dev` null: the estimates are taken by the launch itself from the units' own source images with `threshold`)
static int run_auto(v1c_plan* p, void* stream, const v1c_unit* units, int n_units, const double* rad_dev, int n_rad, int threshold)
{
    if (!p || !units || n_units <= 0 || (rad_dev && n_rad <= 0))
        return fail(V1C_E_INVALID, "v1c_plan_run_auto: bad arguments");
    if (n_units > kInlineUnits)
        return fail(V1C_E_UNSUPPORTED, "v1c_plan_run_auto: at most 16 units per call");
    if (p->mode != MODE_RAY || p->gen_mode != 0 || p->ana.base != 0 || p->n_rot_stages > 1 || p->disable_fast)
        return fail(V1C_E_UNSUPPORTED, "v1c_plan_run_auto: chains of the form EquirectangularEncoder() * [one rotation] * radial stages only");
    DeviceGuard dg(p->device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    hipStream_t st = (hipStream_t)stream;
    const Geom& g = p->ctx.g;
    DevUnit du[kInlineUnits];
    bool aligned = true;
    for (int k = 0; k < n_units; k++) {
        int rc = fill_unit(p, units[k], du[k]);
        if (rc)
            return rc;
        if (!du[k].has_rot && !p->ana.has_rot)
            for (int q = 0; q < 9; q++)
                du[k].rot[q] = (q % 4 == 0) ? 1.0 : 0.0;
        du[k].has_rot = 1;  // (every unit carries its effective rotation: the kernels without plan-time boxes read it)
        aligned = aligned && ((((uintptr_t)du[k].src) | (uintptr_t)du[k].src_pitch) & 3u) == 0;
    }
    bool same_rot = true;  // every unit the same rotation (no overrides, or identical ones): pairs of units share their coordinates
    for (int k = 1; k < n_units && same_rot; k++)
        same_rot = std::memcmp(du[k].rot, du[0].rot, sizeof(du[0].rot)) == 0;
    // any radius a caller can mean: the image circle a few times the source's size at most (the patch kernel clamps to it)
    const double r_limit = 4.0 * (double)std::max(g.src_h, g.src_w);
    const LaunchPlan d = decide_launch(p, du, n_units, 32.0 * r_limit);
    if (!d.fast || d.need_fixup)
        return fail(V1C_E_UNSUPPORTED, "v1c_plan_run_auto: this chain needs a fix-up pass (or 64-bit offsets): use the radius on the host");
    const bool shared = d.shared_entry && !p->disable_shared_entry;
    if (g.cn != 3 && !(shared && aligned && cn_kernel_supports(g)))
        return fail(V1C_E_UNSUPPORTED, "v1c_plan_run_auto: grayscale / BGRA need dword-aligned sources and one table entry per lane");
    std::lock_guard<std::mutex> lk(p->dyn_mu);
    if (!p->ctx_dyn)
        return fail(V1C_E_UNSUPPORTED, "v1c_plan_run_auto: the plan has no tile kernels");
    // (a launch that is being recorded into a graph neither waits for nor records the plan's event: an event recorded outside a capture
    //  cannot be waited for inside one, nor the other way round -- whoever replays the graph orders it against other users of the plan)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess)
        cap = hipStreamCaptureStatusNone;
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && p->dyn_pending && p->dyn_stream != st)
        HIP_TRY(hipStreamWaitEvent(st, p->dyn_ev, 0));
    if (rad_dev) {
        HIP_TRY(launch_patch_radius(p->ctx_dyn, rad_dev, n_rad, r_limit, p->ctx.ray.cx32, p->ctx.ray.cy32, st));
    } else {
        AutoLines im;
        std::memset(&im, 0, sizeof(im));
        const bool use_row = g.src_w > g.src_h;  // transformer.py:126-129
        for (int k = 0; k < n_units; k++)
            im.line[k] = use_row ? du[k].src + (int64_t)(g.src_h / 2) * du[k].src_pitch : du[k].src + (int64_t)(g.src_w / 2) * g.cn;
        im.n = use_row ? g.src_w : g.src_h, im.cn = g.cn, im.threshold = threshold, im.count = n_units;
        for (int k = 0; k < n_units; k++)
            im.step[k] = use_row ? (int64_t)g.cn : du[k].src_pitch;
        HIP_TRY(launch_auto_radius(p->ctx_dyn, p->auto_scratch, im, r_limit, p->ctx.ray.cx32, p->ctx.ray.cy32, st));
    }
    LaunchUnits lu{du, nullptr, n_units};
    int kind = V1C_LAUNCH_GENERIC;
    if (g.cn != 3) {
        kind = V1C_LAUNCH_CN_ROT;
        HIP_TRY(launch_ray_lin_cn(p->ctx, p->ctx_dyn, lu, nullptr, true, nullptr, 12, st));
    } else {
        kind = V1C_LAUNCH_TILE;
        HIP_TRY(launch_ray_lin3_tile(p->ctx, p->ctx_dyn, lu, nullptr, true, nullptr, p->half_dwords, shared, d.mpoly_all && !p->disable_mpoly, nullptr, 0,
                                     p->lean_half, p->strip_len, p->lean_raw_nwp, st, d.coords_bounded && !p->disable_coords_bounded, &kind, same_rot));
    }
    if (!capturing) {
        HIP_TRY(hipEventRecord(p->dyn_ev, st));
        p->dyn_stream = st, p->dyn_pending = true;
    }
    p->last_launch.store(kind, std::memory_order_relaxed);
    return V1C_OK;
}

extern "C" int v1c_plan_run_auto(v1c_plan* p, void* stream, const v1c_unit* units, int n_units, const double* rad_dev, int n_rad)
{
    if (!rad_dev)
        return fail(V1C_E_INVALID, "v1c_plan_run_auto: bad arguments");
    return run_auto(p, stream, units, n_units, rad_dev, n_rad, 0);
}

// ... and with the estimates taken from the units' own sources by the same call: apply()'s `get_radius_smart(radius, images)` over the
// images it is about to remap (remapper.py:379-380), threshold as get_radius's parameter (transformer.py:109: default 10).  Two launches
// in all -- one workgroup that scans every image's centre line and sets the scale, then the remap.
extern "C" int v1c_plan_run_auto_images(v1c_plan* p, void* stream, const v1c_unit* units, int n_units, int threshold)
{
    return run_auto(p, stream, units, n_units, nullptr, 0, threshold);
}

extern "C" int v1c_plan_get_map(v1c_plan* p, void* stream, float* xmap, float* ymap, int64_t map_pitch,
                                const double* rot_or_null)
{
    if (!p || !xmap || !ymap)
        return fail(V1C_E_INVALID, "v1c_plan_get_map: bad arguments");
    if (map_pitch < (int64_t)p->ctx.g.dst_w * 4 || (map_pitch & 3))
        return fail(V1C_E_INVALID, "map_pitch too small or not a multiple of 4");
    if (rot_or_null && !p->chain_has_rot)
        return fail(V1C_E_INVALID, "rotation given but the plan's chain has no rotate stage");
    DeviceGuard dg(p->device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    UnitArgs ua;
    std::memset(&ua, 0, sizeof(ua));
    int mode = p->mode;
    if (rot_or_null) {
        ua.u[0].has_rot = 1;
        for (int q = 0; q < 9; q++)
            ua.u[0].rot[q] = rot_or_null[q];
        if (p->n_rot_stages > 1 || p->gen_mode != 0)
            mode = MODE_LITERAL;
    }
    HIP_TRY(launch_get_map(mode, p->ctx, ua, xmap, ymap, map_pitch, (hipStream_t)stream));
    return V1C_OK;
}

// ------------------------------------------------------------------------------------------
// one-shot entry points
// ------------------------------------------------------------------------------------------
struct PlanKey {
    std::string bytes;
    bool operator<(const PlanKey& o) const { return bytes < o.bytes; }
};

// Plans of the one-shot entry point: at most kFusedCacheSize, least recently used first out (a caller sweeping radii or sizes would
// otherwise keep every plan's tables, tile boxes and unit ring alive).  Entries are shared_ptrs: a call in flight on another thread
// keeps its plan alive past its eviction; v1c_plan_destroy (hipFree: device-synchronising) runs when the last holder lets go.
static constexpr size_t kFusedCacheSize = 32;
static std::mutex g_cache_mu;
static std::list<std::pair<PlanKey, std::shared_ptr<v1c_plan>>> g_cache_lru;  // front = most recently used
static std::map<PlanKey, std::list<std::pair<PlanKey, std::shared_ptr<v1c_plan>>>::iterator> g_cache;

extern "C" int v1c_plan_last_launch(const v1c_plan* p)
{
    if (!p)
        return fail(V1C_E_INVALID, "plan is NULL");
    return p->last_launch.load(std::memory_order_relaxed);
}

extern "C" int v1c_fused_cache_size(void)
{
    std::lock_guard<std::mutex> lk(g_cache_mu);
    return (int)g_cache.size();
}

extern "C" int v1c_remap_fused(int device, void* stream, const uint8_t* src, int src_h, int src_w, int64_t src_pitch,
                               int cn, uint8_t* dst, int dst_h, int dst_w, int64_t dst_pitch, const v1c_chain* chain,
                               int interp, int border_mode, const uint8_t border_val[4])
{
    int rc = validate_chain(chain);
    if (rc)
        return rc;
    PlanKey key;
    key.bytes.assign((const char*)chain, sizeof(v1c_chain));
    const int dims[9] = {device, src_h, src_w, dst_h, dst_w, cn, interp, border_mode, 0};
    key.bytes.append((const char*)dims, sizeof(dims));
    key.bytes.append((const char*)(border_val ? border_val : (const uint8_t*)"\0\0\0\0"), 4);
    std::shared_ptr<v1c_plan> p;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) {
            g_cache_lru.splice(g_cache_lru.begin(), g_cache_lru, it->second);  // most recently used
            p = it->second->second;
        }
    }
    if (!p) {
        v1c_plan* raw = nullptr;
        rc = v1c_plan_create(&raw, device, chain, src_h, src_w, dst_h, dst_w, cn, interp, border_mode, border_val);
        if (rc)
            return rc;
        p.reset(raw, [](v1c_plan* q) { (void)v1c_plan_destroy(q); });
        std::vector<std::shared_ptr<v1c_plan>> evicted;  // (destroyed outside the lock)
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            auto it = g_cache.find(key);
            if (it != g_cache.end()) {  // another thread won the race: use its plan
                p = it->second->second;
            } else {
                g_cache_lru.emplace_front(key, p);
                g_cache[key] = g_cache_lru.begin();
                while (g_cache.size() > kFusedCacheSize) {
                    evicted.push_back(g_cache_lru.back().second);
                    g_cache.erase(g_cache_lru.back().first);
                    g_cache_lru.pop_back();
                }
            }
        }
    }
    v1c_unit u;
    std::memset(&u, 0, sizeof(u));
    u.src = src, u.dst = dst, u.src_pitch = src_pitch, u.dst_pitch = dst_pitch;
    return v1c_plan_run(p.get(), stream, &u, 1);
}

extern "C" int v1c_remap_lut(int device, void* stream, const uint8_t* src, int src_h, int src_w, int64_t src_pitch,
                             int cn, uint8_t* dst, int dst_h, int dst_w, int64_t dst_pitch, const float* xmap,
                             const float* ymap, int64_t map_pitch, int interp, int border_mode,
                             const uint8_t border_val[4])
{
    int rc = validate_geom(src_h, src_w, dst_h, dst_w, cn, interp, border_mode);
    if (rc)
        return rc;
    if (!src || !dst || !xmap || !ymap)
        return fail(V1C_E_INVALID, "v1c_remap_lut: NULL pointer");
    if (map_pitch < (int64_t)dst_w * 4 || (map_pitch & 3))
        return fail(V1C_E_INVALID, "map_pitch too small or not a multiple of 4");
    if (src_pitch < (int64_t)src_w * cn || dst_pitch < (int64_t)dst_w * cn)
        return fail(V1C_E_INVALID, "pitch smaller than a row");
    DeviceGuard dg(device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");

    // the only plan state a LUT launch needs is the interpolation table: cache one per
    // (device, interp)
    static std::mutex mu;
    static std::map<std::pair<int, int>, const short*> itabs;
    KernelCtx c{};
    Geom& g = c.g;
    g.src_h = src_h, g.src_w = src_w, g.dst_h = dst_h, g.dst_w = dst_w, g.cn = cn;
    g.interp = interp == V1C_INTER_AREA ? V1C_INTER_LINEAR : interp;
    g.border = border_mode;
    for (int k = 0; k < 4; k++)
        g.cval[k] = border_val ? border_val[k] : 0;
    if (g.interp == V1C_INTER_CUBIC || g.interp == V1C_INTER_LANCZOS4) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = itabs.find({device, g.interp});
        if (it == itabs.end()) {
            const std::vector<short> tab = build_itab(g.interp);
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, tab.size() * sizeof(short)));
            HIP_TRY(hipMemcpy(d, tab.data(), tab.size() * sizeof(short), hipMemcpyHostToDevice));
            it = itabs.emplace(std::make_pair(device, g.interp), (const short*)d).first;
        }
        c.itab = it->second;
    }
    c.xmap = xmap, c.ymap = ymap, c.map_pitch = map_pitch;
    UnitArgs ua;
    std::memset(&ua, 0, sizeof(ua));
    ua.u[0].src = src, ua.u[0].dst = dst, ua.u[0].src_pitch = src_pitch, ua.u[0].dst_pitch = dst_pitch;
    HIP_TRY(launch_remap(MODE_LUT, c, ua, 1, (hipStream_t)stream));
    return V1C_OK;
}

// merge=True of apply_lr(), remapper.py:485-497 (labels :498-516 stay with the caller).
extern "C" int v1c_anaglyph(int device, void* stream, const uint8_t* left, int64_t left_pitch, const uint8_t* right,
                            int64_t right_pitch, int h, int w, double* out, int64_t out_pitch)
{
    if (!left || !right || !out || h <= 0 || w <= 0 || h > 65535)
        return fail(V1C_E_INVALID, "v1c_anaglyph: bad arguments");
    if (left_pitch < (int64_t)w * 3 || right_pitch < (int64_t)w * 3 || out_pitch < (int64_t)w * 24 || (out_pitch & 7) ||
        ((uintptr_t)out & 7))
        return fail(V1C_E_INVALID, "v1c_anaglyph: pitch smaller than a row, or the float64 output is not 8-byte aligned");
    DeviceGuard dg(device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    HIP_TRY(launch_anaglyph(left, left_pitch, right, right_pitch, h, w, out, out_pitch, (hipStream_t)stream));
    return V1C_OK;
}

// get_radius(), transformer.py:108-140: the scan of the centre row (landscape) or column runs on the device (kernels.hip: k_get_radius);
// out_dev[0] = (last fall - first rise) / 2 with the reference's sign quirk, out_dev[1] = 0, or 1 where the reference raises IndexError
// (no black border; out_dev[0] = NaN).  Nothing is synchronised, nothing is allocated: graph-capturable.
extern "C" int v1c_get_radius_async(int device, void* stream, const uint8_t* img, int h, int w, int64_t pitch, int cn, int threshold,
                                    double* out_dev)
{
    if (!img || !out_dev || h <= 0 || w <= 0 || cn <= 0 || cn > 4 || pitch < (int64_t)w * cn)
        return fail(V1C_E_INVALID, "v1c_get_radius_async: bad arguments");
    DeviceGuard dg(device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    HIP_TRY(launch_get_radius(img, h, w, pitch, cn, threshold, out_dev, (hipStream_t)stream));
    return V1C_OK;
}

// ... and with the value handed to the caller, as the reference returns it (a float): the kernel writes its two doubles into
// page-locked host memory of the calling thread, the call synchronises the stream.
extern "C" int v1c_get_radius(int device, void* stream, const uint8_t* img, int h, int w, int64_t pitch, int cn,
                              int threshold, double* radius)
{
    if (!img || !radius || h <= 0 || w <= 0 || cn <= 0 || cn > 4 || pitch < (int64_t)w * cn)
        return fail(V1C_E_INVALID, "v1c_get_radius: bad arguments");
    DeviceGuard dg(device);
    if (!dg.ok)
        return fail(V1C_E_NODEVICE, "hipSetDevice failed");
    struct Pinned {
        double* p = nullptr;
        ~Pinned()
        {
            if (p)
                (void)hipHostFree(p);
        }
    };
    static thread_local Pinned host;  // (mapped: the kernel stores straight into it)
    if (!host.p)
        HIP_TRY(hipHostMalloc((void**)&host.p, 2 * sizeof(double), hipHostMallocMapped | hipHostMallocPortable));
    double* dptr = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void**)&dptr, host.p, 0));
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(launch_get_radius(img, h, w, pitch, cn, threshold, dptr, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (host.p[1] != 0.0)
        return fail(V1C_E_INVALID, "no black border");  // the reference raises IndexError here
    *radius = host.p[0];
    return V1C_OK;
}
