/*
 * vr180_remap.h -- C ABI of the MI355X-native fisheye -> equirectangular remap engine.
 *
 * This is the drop-in boundary for the ONE hot path of 34j/vr180-convert:
 *
 *     apply_lr() -> apply() -> get_map() -> MultiTransformer.transform() -> cv2.remap()
 *     (reference: src/vr180_convert/remapper.py:406,324,23 ; transformer.py:93 ; remapper.py:388-398)
 *
 * The reference has no FFI of its own (it is pure Python over NumPy + OpenCV); what a
 * maintainer would bind is exactly the pair "evaluate the transformer chain on the output
 * grid" + "cv2.remap the image through it".  Every entry point below names the reference
 * lines it replaces.  Plain pointers and sizes only: no torch / numpy / C++ types.
 *
 * All `src`, `dst`, `xmap`, `ymap` pointers are DEVICE pointers (HBM) owned by the caller.
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue work
 * on that stream and return without synchronising, except where stated.
 *
 * Return value: 0 on success, a negative V1C_E_* code otherwise; v1c_last_error() returns
 * a thread-local human-readable message for the last failing call on this thread.
 */
#ifndef VR180_REMAP_H
#define VR180_REMAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V1C_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------------------- */
#define V1C_OK             0
#define V1C_E_INVALID     -1   /* bad argument (NULL pointer, size <= 0, unknown enum ...)   */
#define V1C_E_UNSUPPORTED -2   /* valid request the engine cannot lower (caller falls back
                                  to v1c_remap_lut with a map it computed itself)           */
#define V1C_E_HIP         -3   /* a HIP runtime call failed (message has hipGetErrorString)  */
#define V1C_E_NODEVICE    -4   /* no usable gfx950 device                                    */

/* ---- cv2 enum values the reference passes straight through ---------------------------
 * remapper.py:330-331 (defaults INTER_LANCZOS4 / BORDER_CONSTANT), cli.py:57-79 (mirrors). */
#define V1C_INTER_NEAREST  0
#define V1C_INTER_LINEAR   1
#define V1C_INTER_CUBIC    2
#define V1C_INTER_AREA     3   /* cv2.remap treats AREA as LINEAR */
#define V1C_INTER_LANCZOS4 4

#define V1C_BORDER_CONSTANT    0
#define V1C_BORDER_REPLICATE   1
#define V1C_BORDER_REFLECT     2
#define V1C_BORDER_WRAP        3
#define V1C_BORDER_REFLECT_101 4
#define V1C_BORDER_TRANSPARENT 5

/* ---- the lowered transformer chain ----------------------------------------------------
 * One v1c_op per stage of the reference's MultiTransformer (transformer.py:87-105), in
 * application order, INCLUDING the NormalizeTransformer that get_map() prepends and the
 * DenormalizeTransformer it appends (remapper.py:51-57).  Stage semantics follow the
 * reference line by line; see DESIGN.md "Op list".                                         */
#define V1C_MAX_OPS    16
#define V1C_MAX_PARAMS 16

enum v1c_opcode {
    /* x=(x-p0)/p2*2 ; y=(y-p1)/p2*2            NormalizeTransformer.transform  transformer.py:153-164
     * optional (nparam = 5): p3 <= row < p4 = the rows of the WHOLE output grid in this plan's row numbering, when the
     * plan serves a band of rows of a larger grid (one eye split over several GPUs): tables sized by the reach of the
     * output are then sized for the whole grid and every band evaluates what the unsplit plan evaluates            */
    V1C_OP_NORMALIZE = 1,
    /* x=x*p0+p2 ; y=y*p1+p3                    DenormalizeTransformer.transform         :197-204 */
    V1C_OP_DENORMALIZE = 2,
    /* x=(x-p2)/p0 ; y=(y-p3)/p1                DenormalizeTransformer.inverse_transform :206-213 */
    V1C_OP_DENORMALIZE_INV = 3,
    /* x=x/p0 ; y=y/p0                          ZoomTransformer.transform                :468-473 */
    V1C_OP_ZOOM = 4,
    /* x=x*p0 ; y=y*p0                          ZoomTransformer.inverse_transform        :475-480 */
    V1C_OP_ZOOM_INV = 5,
    /* iparam = is_latitude_y                   EquirectangularEncoder.transform         :540-568 */
    V1C_OP_EQUIRECT_ENC = 6,
    /* iparam = is_latitude_y                   EquirectangularEncoder.inverse_transform :570-584 */
    V1C_OP_EQUIRECT_DEC = 7,
    /* iparam = v1c_radial kind, nparam/p = its parameters
                                                PolarRollTransformer.transform           :268-286 */
    V1C_OP_RADIAL = 8,
    /* p[0..8] = row-major 3x3 M, v' = M v      Euclidean3DTransformer.transform         :651-657
       (M = numpy-quaternion as_rotation_matrix(q), built by the host: quat.py)                  */
    V1C_OP_ROTATE = 9
};

/* radial function theta' = f(theta) applied by a V1C_OP_RADIAL stage */
enum v1c_radial {
    /* FisheyeEncoder.transform_polar, transformer.py:359-377 */
    V1C_RAD_ENC_RECTILINEAR   = 1,  /* arctan(t)                  */
    V1C_RAD_ENC_STEREOGRAPHIC = 2,  /* 2*arctan(t)                */
    V1C_RAD_ENC_EQUIDISTANT   = 3,  /* t*(pi/2)                   */
    V1C_RAD_ENC_EQUISOLID     = 4,  /* 2*arcsin(t/sqrt(2))        */
    V1C_RAD_ENC_ORTHOGRAPHIC  = 5,  /* arcsin(t)                  */
    /* FisheyeEncoder.inverse_transform_polar (= FisheyeDecoder), transformer.py:379-397 */
    V1C_RAD_DEC_RECTILINEAR   = 6,  /* tan(t)                     */
    V1C_RAD_DEC_STEREOGRAPHIC = 7,  /* 2*tan(t/2)                 */
    V1C_RAD_DEC_EQUIDISTANT   = 8,  /* t/(pi/2)                   */
    V1C_RAD_DEC_EQUISOLID     = 9,  /* sqrt(2)*sin(t/2)           */
    V1C_RAD_DEC_ORTHOGRAPHIC  = 10, /* sin(t)                     */
    /* PolynomialScaler.transform_polar, transformer.py:448-451: p[0..nparam) = coefs_reverse
       (lowest order first); Horner from the highest, starting at 0 like np.polyval.              */
    V1C_RAD_POLYNOMIAL        = 11,
    /* RectilinearDecoder.transform_polar / inverse_transform_polar, transformer.py:338-347,
       p[0] = factor = 2*focal_length/sensor_width_mm                                             */
    V1C_RAD_RECTDEC_FWD       = 12, /* tan(t)*p0                  */
    V1C_RAD_RECTDEC_INV       = 13  /* arctan(t/p0)               */
};

typedef struct v1c_op {
    int32_t opcode;               /* enum v1c_opcode */
    int32_t iparam;               /* integer parameter (kind / flag) */
    int32_t nparam;               /* number of valid entries in p */
    int32_t reserved;
    double  p[V1C_MAX_PARAMS];
} v1c_op;

typedef struct v1c_chain {
    int32_t n_ops;
    int32_t reserved;
    v1c_op  ops[V1C_MAX_OPS];
} v1c_chain;

/* One independent unit of work: an eye of a frame.  remapper.py:388-398 iterates these as
 * `for img in images`; apply_lr concatenates two of them (remapper.py:517-518), which the
 * engine does in place by pointing `dst` at each half of the SBS buffer with dst_pitch = row
 * bytes of the whole SBS image.                                                             */
typedef struct v1c_unit {
    const uint8_t* src;           /* (src_h, src_w, cn) uint8, row pitch src_pitch bytes */
    uint8_t*       dst;           /* (dst_h, dst_w, cn) uint8, row pitch dst_pitch bytes */
    int64_t        src_pitch;
    int64_t        dst_pitch;
    /* optional per-unit rotation (row-major 3x3) REPLACING the matrix of the chain's first
       V1C_OP_ROTATE stage; ignored unless has_rot != 0.  BASELINE config 5: per-frame, per-eye
       calibration rotations (cli.py:308-319) share every other chain parameter.              */
    double         rot[9];
    int32_t        has_rot;
    int32_t        reserved;
} v1c_unit;

typedef struct v1c_plan v1c_plan;   /* opaque */

/* ---- entry points -------------------------------------------------------------------- */

/* ABI / build identification. */
int v1c_abi_version(void);
/* Number of visible HIP devices (<0 on error). Does not create a context. */
int v1c_device_count(void);
/* Thread-local message of the last failing call ("" if none). */
const char* v1c_last_error(void);

/* NUMERICAL CONTRACT of every entry point that evaluates a chain: coordinates in float64, cast to float32 like remapper.py:58; the
 * 1/32-pixel buckets cv2.remap derives from them equal the reference's, pixels equal the CPU oracle's byte for byte -- except at
 * pixels where the CHAIN is ill-conditioned (a map coordinate moving by >= 1e6 x the perturbation of the output position: the pole of
 * a rectilinear projection, stacked polynomials taking an angle to 1e12 rad): there the last bit of the platform's sin / cos / atan2
 * decides the bucket.  BORDER_CONSTANT / BORDER_TRANSPARENT outputs are unaffected (such coordinates lie 1e6+ px outside the source);
 * under the four source-reading border modes those pixels may differ from the reference's.  (And a coordinate whose exact value is
 * a tie of the float64 -> float32 rounding to within a few ulps can land in the neighbouring bucket: measure zero, seen once in 30 000
 * random chains.)  INTEGRATION.md section 2.                                                                                  */

/* Build a reusable plan for one (chain, geometry, interpolation, border) combination.
 * Replaces: chain construction + np.meshgrid + MultiTransformer.transform + astype(float32)
 * of get_map() (remapper.py:50-58) -- here nothing is materialised: the plan only holds the
 * O(W+H) separable tables and the O(1 KiB..32 KiB) radial table the fused kernel reads.
 * `chain` must start with the stage get_map prepends and end with the stage it appends.
 * Synchronous (uploads tables); not graph-capturable.  cn must be 1, 3 or 4.               */
int v1c_plan_create(v1c_plan** out, int device, const v1c_chain* chain,
                    int src_h, int src_w, int dst_h, int dst_w, int cn,
                    int interp, int border_mode, const uint8_t border_val[4]);
int v1c_plan_destroy(v1c_plan* plan);

/* Which device code path the plan selected: 0 = generic fp64 interpreter, 1 = fused
 * "ray" path (separable tables + radial table), 2 = fused "planar" path.  For tests/bench. */
int v1c_plan_path(const v1c_plan* plan);

/* Enqueue the fused chain+gather for n_units independent units: ONE launch for up to 256 units
 * (longer arrays: one launch per 256; 16 per launch where a fix-up pass or the generic kernels
 * are needed).  Replaces: get_map() + the cv.remap list comprehension, remapper.py:381-398, and
 * the SBS concatenate of apply_lr, remapper.py:517-518 (via dst/dst_pitch).
 * `units` is a HOST array, read before the call returns: up to 16 units travel in the launch's
 * kernel arguments; longer batches are copied into a slot of a plan-owned device ring by small
 * launches on `stream` in front of the remap launch.  The call is launch-only (no allocation,
 * no sync) and may be recorded into a graph (at most 4 recorded launches of more than 16 units
 * per plan: each keeps its ring slot).  One plan may be run from several threads / streams.   */
int v1c_plan_run(v1c_plan* plan, void* stream, const v1c_unit* units, int n_units);

/* v1c_plan_run with the radius read from DEVICE memory: radius="auto" -- the reference's default, remapper.py:333,416 -- without a host
 * round trip or a plan per image.  Replaces get_radius_smart("auto") (remapper.py:82-84: the max over the images of get_radius) feeding
 * get_map's DenormalizeTransformer(scale=(radius, radius)) (remapper.py:51-57).  `rad_dev`: n_rad (radius, status) pairs in device memory as
 * v1c_get_radius_async writes them; the launch uses max(radius) -- clamped to 4 x the larger source dimension in magnitude; if any status
 * is set (the reference raises IndexError there) the map is sent far outside the source: every pixel the border colour.  The plan's own radius is ignored, so one plan
 * serves every image of a stream.  At most 16 units; chains EquirectangularEncoder() * [one rotation] * radial stages whose table needs
 * no fix-up pass -- anything else returns V1C_E_UNSUPPORTED and the caller takes the radius to the host (v1c_get_radius).  Launch-only,
 * graph-capturable; one plan may be used from several streams (ordered by an event).                                              */
int v1c_plan_run_auto(v1c_plan* plan, void* stream, const v1c_unit* units, int n_units,
                      const double* rad_dev, int n_rad);

/* The same with the estimates taken from the units' OWN source images by the call itself: apply()'s
 * get_radius_smart("auto", images) over the images it is about to remap (remapper.py:379-380;
 * get_radius, transformer.py:108-140, `threshold` its parameter: the reference's default is 10).
 * Two launches in all: one workgroup scans the centre line of every unit's source and sets the
 * scale, then the remap.  Same chains, limits and error behaviour as v1c_plan_run_auto.          */
int v1c_plan_run_auto_images(v1c_plan* plan, void* stream, const v1c_unit* units, int n_units,
                             int threshold);

/* Hand the plan's capture-owned unit buffers out again (see v1c_plan_run: a recorded launch of more
 * than 16 units keeps one of 4 for the graph that replays it).  Call it once every graph that
 * recorded such a launch of this plan has been destroyed -- a process that re-captures over and
 * over (a resolution that changes back and forth) would otherwise get V1C_E_UNSUPPORTED from the
 * 5th such capture on.  Nothing in the reference corresponds to it (no graphs there).             */
int v1c_plan_release_captures(v1c_plan* plan);

/* Which kernels the most recent launch group of v1c_plan_run on this plan used: one of the V1C_LAUNCH_* values, | V1C_LAUNCH_FIXUP
 * when a fix-up pass followed; -1 before the first run.  The engine has several code paths for the same bytes (a generic per-pixel
 * kernel for everything, LDS-tiled kernels for what the plan could prove about the chain and the units); tests use this to make
 * sure a case meant for a tiled kernel was not served by the generic one.  For tests / bench.                                  */
enum {
    V1C_LAUNCH_GENERIC = 0, /* k_remap: fp64 interpreter or the ray path per pixel, samples from global memory */
    V1C_LAUNCH_TILE = 1,    /* BGR, the general tile kernel: k_ray_lin3_tile (pairs, bicubic / Lanczos4, NEAREST, odd cases) */
    V1C_LAUNCH_MIRROR = 2,  /* BGR pairs and single images of unrotated chains: k_ray_lin3_pair_mirror_seq / _raw */
    V1C_LAUNCH_CN = 3,      /* grayscale / BGRA with plan-time boxes: k_ray_lin_cn */
    V1C_LAUNCH_CN_ROT = 4,  /* grayscale / BGRA, units with a rotation of their own: k_ray_lin_cn without boxes */
    V1C_LAUNCH_BATCH = 5,   /* BGR bilinear batches sharing a map: k_ray_lin3_batch_lean_raw (+ k_ray_lin3_tile for its rest tiles) */
    V1C_LAUNCH_ROT_PAIR = 6, /* BGR bilinear units with a rotation of their own: k_ray_lin3_rot_pair_raw */
    V1C_LAUNCH_FIXUP = 0x100
};
int v1c_plan_last_launch(const v1c_plan* plan);

/* Evaluate only the coordinate chain on the output grid and store float32 maps (device
 * pointers, row pitch map_pitch bytes).  Replaces get_map(), remapper.py:23-59.  Used for
 * coordinate-parity tests and for callers that want the map itself.                         */
int v1c_plan_get_map(v1c_plan* plan, void* stream, float* xmap, float* ymap,
                     int64_t map_pitch, const double* rot_or_null);

/* One-shot convenience: plan lookup/creation in an internal cache keyed on every argument
 * but the pointers (the 32 most recently used plans are kept; an evicted plan is destroyed once
 * no call uses it any more), then v1c_plan_run on one unit.  Same replacement as above.       */
int v1c_remap_fused(int device, void* stream,
                    const uint8_t* src, int src_h, int src_w, int64_t src_pitch, int cn,
                    uint8_t* dst, int dst_h, int dst_w, int64_t dst_pitch,
                    const v1c_chain* chain, int interp, int border_mode,
                    const uint8_t border_val[4]);

/* Number of plans the one-shot cache of v1c_remap_fused holds right now (<= 32).  For tests. */
int v1c_fused_cache_size(void);

/* cv2.remap with caller-supplied float32 maps (device pointers).  Replaces the cv.remap call
 * itself, remapper.py:388-398, for transformer chains the engine cannot lower (user-defined
 * TransformerBase subclasses, README.md:204-219): the caller evaluates the chain.           */
int v1c_remap_lut(int device, void* stream,
                  const uint8_t* src, int src_h, int src_w, int64_t src_pitch, int cn,
                  uint8_t* dst, int dst_h, int dst_w, int64_t dst_pitch,
                  const float* xmap, const float* ymap, int64_t map_pitch,
                  int interp, int border_mode, const uint8_t border_val[4]);

/* Auto-radius estimate of one device-resident image, get_radius() transformer.py:108-140
 * (centre row / column scan on the device, threshold on the channel mean, sign quirk preserved).
 * Synchronous: returns the value through *radius.  Returns V1C_E_INVALID with the message
 * "no black border" where the reference raises IndexError.                                  */
int v1c_get_radius(int device, void* stream, const uint8_t* img, int h, int w,
                   int64_t pitch, int cn, int threshold, double* radius);
/* The same scan with the result left on the device and nothing synchronised (graph-capturable):
 * out_dev[0] = the radius, out_dev[1] = 0.0 -- or out_dev[0] = NaN, out_dev[1] = 1.0 where the
 * reference raises IndexError.  `out_dev`: two doubles in device (or mapped host) memory.  The
 * scan itself (both forms) is one small kernel: transformer.py:125-140 never leaves the GPU.  */
int v1c_get_radius_async(int device, void* stream, const uint8_t* img, int h, int w,
                         int64_t pitch, int cn, int threshold, double* out_dev);

/* merge=True of apply_lr(), remapper.py:485-497: red/cyan anaglyph of two remapped eyes, both
 * (h, w, 3) uint8 on the device; `out` is (h, w, 3) float64 with row pitch out_pitch BYTES:
 *   out[c] = (mean_L * colL[c] + mean_R * colR[c]) / 255,  colL = (0,128,255), colR = (255,128,0),
 *   mean = float64 mean of the 3 channels -- the reference's NumPy float64 expression, operation
 * by operation (no FMA contraction).  The cv.putText labels (:498-516) stay with the caller.   */
int v1c_anaglyph(int device, void* stream,
                 const uint8_t* left, int64_t left_pitch, const uint8_t* right, int64_t right_pitch,
                 int h, int w, double* out, int64_t out_pitch);

/* The int16 fixed-point weight table the engine uses for INTER_CUBIC (1024*4*4 entries) or
 * INTER_LANCZOS4 (1024*8*8 entries), laid out [fy*32+fx][ky][kx]: OpenCV's initInterTab2D
 * (SURVEY.md Appendix A item 3).  Host-only (no device needed); `out` is a HOST buffer.
 * Lets callers and tests inspect exactly what the kernels read.                            */
int v1c_build_itab(int interp, int16_t* out);

#ifdef __cplusplus
}
#endif
#endif /* VR180_REMAP_H */
