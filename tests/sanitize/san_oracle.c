/* san_oracle.c -- TEST INFRASTRUCTURE: drives the CPU oracle (oracle/vr180_oracle.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer (gcc -fsanitize=address,undefined; SURVEY.md 5
 * "sanitizers on the CPU restatement").  Exercises every opcode of the chain interpreter, every
 * interpolation x border mode x channel count of the cv2.remap restatement with coordinates far
 * outside the source, NaN and infinities, odd pitches, and get_radius on rows / columns.
 * Exit code 0 and an empty sanitizer report = pass (tests/test_sanitizers.py). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vr180_remap.h"

int orc_get_map(const v1c_chain* ch, int out_w, int out_h, float* xmap, float* ymap);
int orc_get_map_f64(const v1c_chain* ch, int out_w, int out_h, double* xmap, double* ymap);
int orc_build_itab(int interp, short* itab);
int orc_remap(const uint8_t* src, int src_h, int src_w, int64_t src_pitch, int cn, uint8_t* dst, int dst_h, int dst_w,
              int64_t dst_pitch, const float* xmap, const float* ymap, int64_t map_pitch_elems, int interp, int border,
              const uint8_t* cval);
int orc_get_radius(const uint8_t* img, int h, int w, int64_t pitch, int cn, int threshold, double* radius);
void orc_set_threads(int n);

static uint32_t rng_state = 12345u;
static uint32_t rnd(void)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

static void set_op(v1c_op* o, int opcode, int iparam, int n, const double* p)
{
    memset(o, 0, sizeof(*o));
    o->opcode = opcode, o->iparam = iparam, o->nparam = n;
    for (int i = 0; i < n; i++)
        o->p[i] = p[i];
}

int main(void)
{
    orc_set_threads(2);
    const int W = 37, H = 29;
    /* exact-size heap blocks: any out-of-bounds access is an ASan report */
    float* xm = malloc(sizeof(float) * W * H);
    float* ym = malloc(sizeof(float) * W * H);
    double* xd = malloc(sizeof(double) * W * H);
    double* yd = malloc(sizeof(double) * W * H);
    int fails = 0;

    /* 1. chains: Normalize, every radial kind, both equirect directions, rotate, zoom (+inverses), denormalize */
    const double norm[3] = {W / 2.0, H / 2.0, (double)(W < H ? W : H)};
    const double den[4] = {13.5, 13.5, 18, 14};
    const double poly[4] = {0.01, 1.0, -0.1, 0.02};
    const double rot[9] = {0.8, 0.0, 0.6, 0.0, 1.0, 0.0, -0.6, 0.0, 0.8};
    const double zoom[1] = {1.3}, fac[1] = {1.7};
    for (int kind = V1C_RAD_ENC_RECTILINEAR; kind <= V1C_RAD_RECTDEC_INV; kind++)
        for (int lat_y = 0; lat_y < 2; lat_y++) {
            v1c_chain ch;
            memset(&ch, 0, sizeof(ch));
            int n = 0;
            set_op(&ch.ops[n++], V1C_OP_NORMALIZE, 0, 3, norm);
            set_op(&ch.ops[n++], V1C_OP_EQUIRECT_ENC, lat_y, 0, NULL);
            set_op(&ch.ops[n++], V1C_OP_ROTATE, 0, 9, rot);
            if (kind == V1C_RAD_POLYNOMIAL)
                set_op(&ch.ops[n++], V1C_OP_RADIAL, kind, 4, poly);
            else if (kind >= V1C_RAD_RECTDEC_FWD)
                set_op(&ch.ops[n++], V1C_OP_RADIAL, kind, 1, fac);
            else
                set_op(&ch.ops[n++], V1C_OP_RADIAL, kind, 0, NULL);
            set_op(&ch.ops[n++], V1C_OP_ZOOM, 0, 1, zoom);
            set_op(&ch.ops[n++], V1C_OP_ZOOM_INV, 0, 1, zoom);
            set_op(&ch.ops[n++], V1C_OP_EQUIRECT_DEC, lat_y, 0, NULL);
            set_op(&ch.ops[n++], V1C_OP_EQUIRECT_ENC, lat_y, 0, NULL);
            set_op(&ch.ops[n++], V1C_OP_DENORMALIZE, 0, 4, den);
            set_op(&ch.ops[n++], V1C_OP_DENORMALIZE_INV, 0, 4, den);
            set_op(&ch.ops[n++], V1C_OP_DENORMALIZE, 0, 4, den);
            ch.n_ops = n;
            fails += orc_get_map(&ch, W, H, xm, ym) != 0;
            fails += orc_get_map_f64(&ch, W, H, xd, yd) != 0;
        }

    /* 2. remap: all interpolations x borders x channels; coordinates include far outside, NaN, +-inf, huge */
    short* itab = malloc(sizeof(short) * 1024 * 64);
    fails += orc_build_itab(V1C_INTER_CUBIC, itab) != 0;
    fails += orc_build_itab(V1C_INTER_LANCZOS4, itab) != 0;
    free(itab);
    const int SH = 23, SW = 31;
    for (int cn = 1; cn <= 4; cn++) {
        if (cn == 2)
            continue;
        const int64_t sp = (int64_t)SW * cn + 5, dp = (int64_t)W * cn + 3; /* odd pitches */
        uint8_t* src = malloc((size_t)sp * (SH - 1) + (size_t)SW * cn);    /* last row has no padding */
        uint8_t* dst = malloc((size_t)dp * (H - 1) + (size_t)W * cn);
        for (size_t i = 0; i < (size_t)sp * (SH - 1) + (size_t)SW * cn; i++)
            src[i] = (uint8_t)rnd();
        for (int i = 0; i < W * H; i++) {
            xm[i] = (float)((int)(rnd() % 6000) - 3000) / 50.0f; /* -60 .. 60 around a 31-wide source */
            ym[i] = (float)((int)(rnd() % 6000) - 3000) / 50.0f;
        }
        xm[0] = NAN, ym[1] = NAN, xm[2] = INFINITY, ym[3] = -INFINITY, xm[4] = 3e9f, ym[5] = -3e9f, xm[6] = 32767.9f,
        ym[7] = -32768.4f, xm[8] = 1e38f, ym[8] = -1e38f;
        const uint8_t cval[4] = {9, 8, 7, 6};
        static const int interps[] = {V1C_INTER_NEAREST, V1C_INTER_LINEAR, V1C_INTER_CUBIC, V1C_INTER_AREA, V1C_INTER_LANCZOS4};
        for (unsigned a = 0; a < sizeof(interps) / sizeof(interps[0]); a++)
            for (int border = V1C_BORDER_CONSTANT; border <= V1C_BORDER_TRANSPARENT; border++) {
                memset(dst, 0, (size_t)dp * (H - 1) + (size_t)W * cn);
                fails += orc_remap(src, SH, SW, sp, cn, dst, H, W, dp, xm, ym, W, interps[a], border, cval) != 0;
            }
        /* 1 x 1 and 2 x 2 sources: every tap of an 8 x 8 footprint is a border tap */
        for (int tiny = 1; tiny <= 2; tiny++)
            for (int border = V1C_BORDER_CONSTANT; border <= V1C_BORDER_TRANSPARENT; border++)
                fails += orc_remap(src, tiny, tiny, sp, cn, dst, H, W, dp, xm, ym, W, V1C_INTER_LANCZOS4, border, cval) != 0;
        free(src);
        free(dst);
    }

    /* 3. get_radius: landscape (row scan), portrait (column scan), no border (error path), cn 1 and 3 */
    for (int cn = 1; cn <= 3; cn += 2)
        for (int portrait = 0; portrait < 2; portrait++) {
            const int h = portrait ? 64 : 40, w = portrait ? 40 : 64;
            uint8_t* img = calloc((size_t)h * w * cn, 1);
            for (int j = 0; j < h; j++)
                for (int i = 0; i < w; i++) {
                    const double dx = i - w / 2.0, dy = j - h / 2.0;
                    if (dx * dx + dy * dy < 15.0 * 15.0)
                        for (int c = 0; c < cn; c++)
                            img[((size_t)j * w + i) * cn + c] = 200;
                }
            double r = 0;
            fails += orc_get_radius(img, h, w, (int64_t)w * cn, cn, 10, &r) != 0;
            memset(img, 255, (size_t)h * w * cn);
            fails += orc_get_radius(img, h, w, (int64_t)w * cn, cn, 10, &r) == 0; /* must report "no black border" */
            free(img);
        }
    free(xm), free(ym), free(xd), free(yd);
    if (fails) {
        fprintf(stderr, "san_oracle: %d calls returned an unexpected code\n", fails);
        return 1;
    }
    puts("san_oracle ok");
    return 0;
}
