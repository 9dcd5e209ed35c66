/*
 * vr180_oracle.c -- CPU ORACLE for the vr180-convert hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
 * this file.  The product (vr180_convert_amd/) never imports anything under oracle/.
 *
 * What it restates, in plain C, one function per reference stage (citations are into
 * /root/reference/src/vr180_convert/):
 *
 *   (1) the coordinate chain of get_map()            remapper.py:23-59, transformer.py:93-98
 *       evaluated LITERALLY per pixel in float64 in the reference's operation order
 *       (sqrt/atan2/cos/sin per PolarRoll stage, arccos/arctan2 per 3-D conversion), then
 *       cast to float32 like remapper.py:58.
 *       PINNED: tests/test_oracle_golden.py checks it against tests/golden/ (.npz files), which were
 *       produced by importing the reference itself (tests/golden/make_golden.py).
 *
 *   (2) cv2.remap for uint8 images                    call site remapper.py:388-398
 *       The algorithm lives in a third-party dependency that is NOT under /root/reference:
 *       opencv-python ^4.9.0.80 (pyproject.toml:33), locked 4.10.0.82 (poetry.lock:1014-1015),
 *       modules/imgproc/src/imgwarp.cpp (RemapInvoker, remapNearest, remapBilinear,
 *       remapBicubic, remapLanczos4, initInterTab1D/2D) and borderInterpolate().  cv2 is not
 *       installed in the build container or on the GPU box and the reference's tests assert no
 *       pixel value, so this part is a restatement of the published algorithm:
 *       *** PARITY UNPINNED at the cv2.remap boundary *** except for the coarse known-answer
 *       test on the reference's docs/_static example pair (tests/test_oracle_remap.py::test_known_answer_reference_docs_pair).
 *
 *   (3) get_radius()                                  transformer.py:108-140
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  -ffp-contract=off matters:
 * the reference's NumPy ufuncs and OpenCV's baseline x86-64 build round every multiply and add
 * separately.
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/vr180_remap.h" /* only for the v1c_chain POD + enum values */

#define ORC_PI_2 (M_PI / 2) /* np.pi / 2 */

static int g_threads = 1;

void orc_set_threads(int n)
{
    g_threads = n < 1 ? 1 : n;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * (1) coordinate chain
 * ---------------------------------------------------------------------------------------- */

/* equidistant_to_3d, transformer.py:483-508 */
static void to_3d(double x, double y, double v[3])
{
    double phi = atan2(x, y);
    double theta = sqrt(x * x + y * y);
    v[0] = sin(theta) * sin(phi);
    v[1] = sin(theta) * cos(phi);
    v[2] = cos(theta);
}

/* equidistant_from_3d, transformer.py:511-530 */
static void from_3d(const double v[3], double* x, double* y)
{
    double theta = acos(v[2]);
    double phi = atan2(v[0], v[1]);
    *x = theta * sin(phi);
    *y = theta * cos(phi);
}

/* theta' = f(theta) of one PolarRoll stage.
 * FisheyeEncoder.transform_polar :359-377, .inverse_transform_polar :379-397,
 * PolynomialScaler.transform_polar :448-451, RectilinearDecoder :338-347 */
static double radial_fn(const v1c_op* op, double t)
{
    switch (op->iparam) {
    case V1C_RAD_ENC_RECTILINEAR:   return atan(t);
    case V1C_RAD_ENC_STEREOGRAPHIC: return 2 * atan(t);
    case V1C_RAD_ENC_EQUIDISTANT:   return t * ORC_PI_2;
    case V1C_RAD_ENC_EQUISOLID:     return 2 * asin(t / sqrt(2.0));
    case V1C_RAD_ENC_ORTHOGRAPHIC:  return asin(t);
    case V1C_RAD_DEC_RECTILINEAR:   return tan(t);
    case V1C_RAD_DEC_STEREOGRAPHIC: return 2 * tan(t / 2);
    case V1C_RAD_DEC_EQUIDISTANT:   return t / ORC_PI_2;
    case V1C_RAD_DEC_EQUISOLID:     return sqrt(2.0) * sin(t / 2);
    case V1C_RAD_DEC_ORTHOGRAPHIC:  return sin(t);
    case V1C_RAD_POLYNOMIAL: {
        /* np.polyval(np.flip(coefs_reverse), theta): y = 0; for c in high..low: y = y*x + c */
        double y = 0.0;
        for (int k = op->nparam - 1; k >= 0; k--)
            y = y * t + op->p[k];
        return y;
    }
    case V1C_RAD_RECTDEC_FWD: return tan(t) * op->p[0];
    case V1C_RAD_RECTDEC_INV: return atan(t / op->p[0]);
    default: return NAN;
    }
}

static void chain_eval(const v1c_chain* ch, double* px, double* py)
{
    double x = *px, y = *py;
    for (int i = 0; i < ch->n_ops; i++) {
        const v1c_op* op = &ch->ops[i];
        switch (op->opcode) {
        case V1C_OP_NORMALIZE: /* transformer.py:162-163 */
            x = (x - op->p[0]) / op->p[2] * 2;
            y = (y - op->p[1]) / op->p[2] * 2;
            break;
        case V1C_OP_DENORMALIZE: /* :202-203 */
            x = x * op->p[0] + op->p[2];
            y = y * op->p[1] + op->p[3];
            break;
        case V1C_OP_DENORMALIZE_INV: /* :211-212 */
            x = (x - op->p[2]) / op->p[0];
            y = (y - op->p[3]) / op->p[1];
            break;
        case V1C_OP_ZOOM: /* :471-472 */
            x = x / op->p[0];
            y = y / op->p[0];
            break;
        case V1C_OP_ZOOM_INV: /* :478-479 */
            x = x * op->p[0];
            y = y * op->p[0];
            break;
        case V1C_OP_EQUIRECT_ENC: { /* :540-568 */
            double v[3];
            if (op->iparam) {
                double lat = y * ORC_PI_2, lon = x * ORC_PI_2;
                v[0] = cos(lat) * sin(lon);
                v[1] = sin(lat);
                v[2] = cos(lat) * cos(lon);
            } else {
                double lat = x * ORC_PI_2, lon = y * ORC_PI_2;
                v[0] = sin(lat);
                v[1] = cos(lat) * sin(lon);
                v[2] = cos(lat) * cos(lon);
            }
            from_3d(v, &x, &y);
            break;
        }
        case V1C_OP_EQUIRECT_DEC: { /* :570-584 */
            double v[3];
            to_3d(x, y, v);
            if (op->iparam) {
                double lat = asin(v[1]), lon = atan2(v[0], v[2]);
                x = lon / ORC_PI_2;
                y = lat / ORC_PI_2;
            } else {
                double lat = asin(v[0]), lon = atan2(v[1], v[2]);
                x = lat / ORC_PI_2;
                y = lon / ORC_PI_2;
            }
            break;
        }
        case V1C_OP_RADIAL: { /* PolarRollTransformer.transform :268-276 (= :278-286) */
            double theta = sqrt(x * x + y * y);
            double roll = atan2(y, x);
            theta = radial_fn(op, theta);
            x = theta * cos(roll);
            y = theta * sin(roll);
            break;
        }
        case V1C_OP_ROTATE: { /* Euclidean3DTransformer.transform :651-657, rotate_vectors :676 */
            double v[3], r[3];
            to_3d(x, y, v);
            for (int k = 0; k < 3; k++)
                r[k] = op->p[3 * k] * v[0] + op->p[3 * k + 1] * v[1] + op->p[3 * k + 2] * v[2];
            from_3d(r, &x, &y);
            break;
        }
        default:
            x = y = NAN;
        }
    }
    *px = x;
    *py = y;
}

/* get_map(), remapper.py:50-58: meshgrid of integer pixel indices -> chain -> float32.
 * The chain passed in already contains the Normalize / Denormalize stages get_map adds. */
int orc_get_map(const v1c_chain* ch, int out_w, int out_h, float* xmap, float* ymap)
{
    if (!ch || !xmap || !ymap || out_w <= 0 || out_h <= 0)
        return -1;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < out_h; j++) {
        for (int i = 0; i < out_w; i++) {
            double x = (double)i, y = (double)j;
            chain_eval(ch, &x, &y);
            xmap[(size_t)j * out_w + i] = (float)x; /* astype(np.float32): RNE */
            ymap[(size_t)j * out_w + i] = (float)y;
        }
    }
    return 0;
}

/* same, float64 out (diagnostics: distance of the fp64 coordinate to a 1/32-bucket edge) */
int orc_get_map_f64(const v1c_chain* ch, int out_w, int out_h, double* xmap, double* ymap)
{
    if (!ch || !xmap || !ymap || out_w <= 0 || out_h <= 0)
        return -1;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int j = 0; j < out_h; j++) {
        for (int i = 0; i < out_w; i++) {
            double x = (double)i, y = (double)j;
            chain_eval(ch, &x, &y);
            xmap[(size_t)j * out_w + i] = x;
            ymap[(size_t)j * out_w + i] = y;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (2) cv2.remap, uint8, CV_32FC1 maps -- OpenCV 4.10 imgwarp.cpp restated
 * ---------------------------------------------------------------------------------------- */
#define INTER_BITS 5
#define INTER_TAB_SIZE 32
#define INTER_TAB_SIZE2 (INTER_TAB_SIZE * INTER_TAB_SIZE)
#define COEF_BITS 15
#define COEF_SCALE 32768

/* cvRound(float): SSE cvtss2si semantics -- round-half-even, NaN / out of range -> INT_MIN */
static int cv_round_f(float v)
{
    if (!(v >= -2147483648.0f && v < 2147483648.0f))
        return INT_MIN;
    return (int)nearbyintf(v);
}

static short sat_short(int v)
{
    return (short)(v < SHRT_MIN ? SHRT_MIN : v > SHRT_MAX ? SHRT_MAX : v);
}

static uint8_t sat_u8(int v)
{
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

/* FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS> */
static uint8_t fixpt_cast(int v)
{
    return sat_u8((v + (1 << (COEF_BITS - 1))) >> COEF_BITS);
}

/* cv::borderInterpolate (modules/core/src/copy.cpp) */
static int border_interpolate(int p, int len, int border)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border == V1C_BORDER_REPLICATE)
        return p < 0 ? 0 : len - 1;
    if (border == V1C_BORDER_REFLECT || border == V1C_BORDER_REFLECT_101) {
        int delta = border == V1C_BORDER_REFLECT_101;
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
    if (border == V1C_BORDER_WRAP) {
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}

static int clip_i(int x, int a, int b)
{
    return x >= a ? (x < b ? x : b - 1) : a;
}

/* interpolateLinear / interpolateCubic / interpolateLanczos4 (imgwarp.cpp), float arithmetic */
static void interp_linear(float x, float* c)
{
    c[0] = 1.f - x;
    c[1] = x;
}

static void interp_cubic(float x, float* c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static void interp_lanczos4(float x, float* c)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[][2] = {{1, 0},  {-s45, -s45}, {0, 1},  {s45, -s45},
                                   {-1, 0}, {s45, s45},   {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f /* FLT_EPSILON */) {
        for (int i = 0; i < 8; i++)
            c[i] = 0;
        c[3] = 1;
        return;
    }
    float sum = 0;
    double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(x + 3 - i) * M_PI * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++)
        c[i] *= sum;
}

static int interp_ksize(int interp)
{
    return interp == V1C_INTER_LINEAR ? 2 : interp == V1C_INTER_CUBIC ? 4 : interp == V1C_INTER_LANCZOS4 ? 8 : 0;
}

/* initInterTab2D(method, fixpt=true): itab[INTER_TAB_SIZE2][ksize][ksize] int16.
 * `itab` must hold INTER_TAB_SIZE2*ksize*ksize + 8 shorts, zero-initialised by the caller: the
 * original's fix-up loop indexes the "central" taps ksize/2 .. ksize/2+1 and, for ksize == 2,
 * thereby peeks into the (still zero) next table entry. */
int orc_build_itab(int interp, short* itab)
{
    int ksize = interp_ksize(interp);
    if (!ksize || !itab)
        return -1;
    float tab1d[8 * INTER_TAB_SIZE];
    float scale = 1.f / INTER_TAB_SIZE;
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        float* t = tab1d + i * ksize;
        if (ksize == 2)
            interp_linear(i * scale, t);
        else if (ksize == 4)
            interp_cubic(i * scale, t);
        else
            interp_lanczos4(i * scale, t);
    }
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; k1++) {
                float vy = tab1d[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    float v = vy * tab1d[j * ksize + k2];
                    itab[k1 * ksize + k2] = sat_short(cv_round_f(v * COEF_SCALE));
                    isum += itab[k1 * ksize + k2];
                }
            }
            if (isum != COEF_SCALE) {
                int diff = isum - COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
                            mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
                            Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0)
                    itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else
                    itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    return 0;
}

typedef struct {
    const uint8_t* src;
    int h, w, cn;
    int64_t pitch;
    int border;
    uint8_t cval[4];
} orc_img;

/* one destination pixel, ksize x ksize taps, top-left tap at (sx, sy); follows the generic
 * (non-inlier) branch of remapBilinear / remapBicubic / remapLanczos4, whose integer result is
 * identical to the inlier branch's.  Returns 0 if the pixel is to be left untouched
 * (BORDER_TRANSPARENT). */
static int sample_taps(const orc_img* im, int ksize, const short* w, int sx, int sy, uint8_t* D)
{
    const int W = im->w, H = im->h, cn = im->cn;
    int border = im->border;
    if (ksize == 2) {
        if (border == V1C_BORDER_CONSTANT && (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0)) {
            for (int k = 0; k < cn; k++)
                D[k] = im->cval[k];
            return 1;
        }
        if (border == V1C_BORDER_TRANSPARENT &&
            ((unsigned)sx >= (unsigned)(W - 1) || (unsigned)sy >= (unsigned)(H - 1)))
            return 0;
    } else {
        int c = ksize / 2 - 1; /* offset of the centre tap: 1 (cubic) / 3 (lanczos4) */
        if (border == V1C_BORDER_TRANSPARENT &&
            ((unsigned)(sx + c) >= (unsigned)W || (unsigned)(sy + c) >= (unsigned)H))
            return 0;
        if (border == V1C_BORDER_TRANSPARENT)
            border = V1C_BORDER_REFLECT_101; /* borderType1 */
        if (border == V1C_BORDER_CONSTANT && (sx >= W || sx + ksize <= 0 || sy >= H || sy + ksize <= 0)) {
            for (int k = 0; k < cn; k++)
                D[k] = im->cval[k];
            return 1;
        }
    }
    int xs[8], ys[8];
    for (int i = 0; i < ksize; i++) {
        if (ksize == 2 && border == V1C_BORDER_REPLICATE) {
            xs[i] = clip_i(sx + i, 0, W);
            ys[i] = clip_i(sy + i, 0, H);
        } else {
            xs[i] = border_interpolate(sx + i, W, border);
            ys[i] = border_interpolate(sy + i, H, border);
        }
    }
    for (int k = 0; k < cn; k++) {
        int cv = im->cval[k];
        /* bicubic/lanczos: sum = cv*ONE + sum (S - cv)*w over in-bounds taps; bilinear:
         * sum of v*w with v = cval for out-of-bounds taps.  Equal because sum(w) == ONE. */
        int sum = 0;
        for (int i = 0; i < ksize; i++)
            for (int j = 0; j < ksize; j++) {
                int v = (xs[j] >= 0 && ys[i] >= 0) ? im->src[(int64_t)ys[i] * im->pitch + (int64_t)xs[j] * cn + k] : cv;
                sum += v * w[i * ksize + j];
            }
        D[k] = fixpt_cast(sum);
    }
    return 1;
}

/* cv::remap(src, dst, map1=xmap (CV_32FC1), map2=ymap (CV_32FC1), interpolation, borderMode,
 * borderValue) for CV_8UC{1,3,4}.  border_val[k] is the already-saturated Scalar component k
 * (Python int v -> (v,0,0,0); the host wrapper does that conversion). */
int orc_remap(const uint8_t* src, int src_h, int src_w, int64_t src_pitch, int cn,
              uint8_t* dst, int dst_h, int dst_w, int64_t dst_pitch,
              const float* xmap, const float* ymap, int64_t map_pitch_elems,
              int interp, int border, const uint8_t border_val[4])
{
    if (!src || !dst || !xmap || !ymap || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0)
        return -1;
    if (cn != 1 && cn != 3 && cn != 4)
        return -1;
    if (interp == V1C_INTER_AREA)
        interp = V1C_INTER_LINEAR;
    orc_img im = {src, src_h, src_w, cn, src_pitch, border, {0, 0, 0, 0}};
    for (int k = 0; k < cn; k++)
        im.cval[k] = border_val[k & 3];

    if (interp == V1C_INTER_NEAREST) {
#pragma omp parallel for schedule(static) num_threads(g_threads)
        for (int dy = 0; dy < dst_h; dy++) {
            for (int dx = 0; dx < dst_w; dx++) {
                /* RemapInvoker, planar float maps, nearest: saturate_cast<short>(float) */
                int sx = sat_short(cv_round_f(xmap[dy * map_pitch_elems + dx]));
                int sy = sat_short(cv_round_f(ymap[dy * map_pitch_elems + dx]));
                uint8_t* D = dst + dy * dst_pitch + (int64_t)dx * cn;
                const uint8_t* S;
                if ((unsigned)sx < (unsigned)src_w && (unsigned)sy < (unsigned)src_h) {
                    S = src + sy * src_pitch + (int64_t)sx * cn;
                } else if (border == V1C_BORDER_TRANSPARENT) {
                    continue;
                } else if (border == V1C_BORDER_REPLICATE) {
                    sx = clip_i(sx, 0, src_w);
                    sy = clip_i(sy, 0, src_h);
                    S = src + sy * src_pitch + (int64_t)sx * cn;
                } else if (border == V1C_BORDER_CONSTANT) {
                    S = im.cval;
                } else {
                    sx = border_interpolate(sx, src_w, border);
                    sy = border_interpolate(sy, src_h, border);
                    S = src + sy * src_pitch + (int64_t)sx * cn;
                }
                for (int k = 0; k < cn; k++)
                    D[k] = S[k];
            }
        }
        return 0;
    }

    int ksize = interp_ksize(interp);
    if (!ksize)
        return -1;
    short* itab = (short*)calloc((size_t)INTER_TAB_SIZE2 * ksize * ksize + 8, sizeof(short));
    if (!itab)
        return -1;
    orc_build_itab(interp, itab);
    const int off = ksize / 2 - 1; /* 0 / 1 / 3 */

#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int dy = 0; dy < dst_h; dy++) {
        for (int dx = 0; dx < dst_w; dx++) {
            /* RemapInvoker, planar float maps: 5 fractional bits */
            int sx = cv_round_f(xmap[dy * map_pitch_elems + dx] * INTER_TAB_SIZE);
            int sy = cv_round_f(ymap[dy * map_pitch_elems + dx] * INTER_TAB_SIZE);
            int a = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1));
            int ix = sat_short(sx >> INTER_BITS);
            int iy = sat_short(sy >> INTER_BITS);
            uint8_t* D = dst + dy * dst_pitch + (int64_t)dx * cn;
            sample_taps(&im, ksize, itab + (size_t)a * ksize * ksize, ix - off, iy - off, D);
        }
    }
    free(itab);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * (3) get_radius(), transformer.py:108-140
 * returns 0 and *radius, or -2 where the reference raises IndexError (no 0->1 / 1->0 edge)
 * ---------------------------------------------------------------------------------------- */
int orc_get_radius(const uint8_t* img, int h, int w, int64_t pitch, int cn, int threshold, double* radius)
{
    if (!img || !radius || h <= 0 || w <= 0 || cn <= 0)
        return -1;
    int n = w > h ? w : h; /* width > height: centre row, else centre column */
    int first_rise = -1, last_fall = -1, prev = 0;
    for (int i = 0; i < n; i++) {
        const uint8_t* p = w > h ? img + (int64_t)(h / 2) * pitch + (int64_t)i * cn
                                 : img + (int64_t)i * pitch + (int64_t)(w / 2) * cn;
        double s = 0;
        for (int k = 0; k < cn; k++)
            s += p[k];
        int black = (s / cn) < threshold; /* np.mean(axis=-1) < threshold */
        if (i > 0) {
            int d = black - prev; /* np.diff */
            if (d == 1 && first_rise < 0)
                first_rise = i - 1;
            if (d == -1)
                last_fall = i - 1;
        }
        prev = black;
    }
    if (first_rise < 0 || last_fall < 0)
        return -2;
    *radius = (last_fall - first_rise) / 2.0;
    return 0;
}
