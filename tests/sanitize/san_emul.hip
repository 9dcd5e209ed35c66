// san_emul.hip -- TEST INFRASTRUCTURE: the product's __host__ __device__ per-pixel code (interpreter,
// fused ray path, radial fit, cv2.remap-exact sampler: vr180_convert_amd/csrc/v1c_core.hpp, radial_fit.hpp)
// compiled for the HOST through tests/host_emul/emul.hip and run under AddressSanitizer +
// UndefinedBehaviorSanitizer (CPU only: GPU sanitizers are not available on this pool).
// Build: hipcc --cuda-host-only -fsanitize=address,undefined (tests/test_sanitizers.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../host_emul/emul.hip"


static uint32_t rng_state = 4242u;
static uint32_t rnd()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

static void set_op(v1c_op& o, int opcode, int iparam, std::initializer_list<double> p)
{
    std::memset(&o, 0, sizeof(o));
    o.opcode = opcode, o.iparam = iparam, o.nparam = (int)p.size();
    int i = 0;
    for (double v : p)
        o.p[i++] = v;
}

int main()
{
    int fails = 0;
    const int W = 48, H = 40;
    // exact-size heap blocks
    float* xm = (float*)std::malloc(sizeof(float) * W * H);
    float* ym = (float*)std::malloc(sizeof(float) * W * H);
    long long st[5];
    // ray-shaped chains (equidistant, polynomial with even terms -> w table, rotation), planar chains (fused since round 5: 4, 5), the
    // general modes (6: lat_x, 7: planar + rotation, 8: a rotation behind a radial stage) and a literal-only chain (9: a decoder last)
    for (int variant = 0; variant < 10; variant++) {
        v1c_chain ch;
        std::memset(&ch, 0, sizeof(ch));
        int n = 0;
        set_op(ch.ops[n++], V1C_OP_NORMALIZE, 0, {W / 2.0, H / 2.0, (double)H});
        if (variant < 4) {
            set_op(ch.ops[n++], V1C_OP_EQUIRECT_ENC, 1, {});
            if (variant & 1)
                set_op(ch.ops[n++], V1C_OP_ROTATE, 0, {0.8, 0.0, 0.6, 0.0, 1.0, 0.0, -0.6, 0.0, 0.8});
            if (variant & 2)
                set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_POLYNOMIAL, {0.0, 1.0, -0.1});
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_DEC_EQUIDISTANT, {});
        } else if (variant < 6) {
            set_op(ch.ops[n++], V1C_OP_RADIAL, variant == 4 ? V1C_RAD_ENC_ORTHOGRAPHIC : V1C_RAD_ENC_STEREOGRAPHIC, {});  // NaN producer
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_DEC_EQUIDISTANT, {});
        } else if (variant == 6) {
            set_op(ch.ops[n++], V1C_OP_EQUIRECT_ENC, 0, {});
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_DEC_EQUIDISTANT, {});
        } else if (variant == 7) {
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_ENC_EQUIDISTANT, {});
            set_op(ch.ops[n++], V1C_OP_ROTATE, 0, {0.8, 0.0, 0.6, 0.0, 1.0, 0.0, -0.6, 0.0, 0.8});
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_DEC_EQUIDISTANT, {});
        } else if (variant == 8) {
            set_op(ch.ops[n++], V1C_OP_EQUIRECT_ENC, 1, {});
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_POLYNOMIAL, {0.0, 1.0, -0.1});
            set_op(ch.ops[n++], V1C_OP_ROTATE, 0, {0.8, 0.0, 0.6, 0.0, 1.0, 0.0, -0.6, 0.0, 0.8});
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_DEC_EQUIDISTANT, {});
        } else {
            set_op(ch.ops[n++], V1C_OP_RADIAL, V1C_RAD_ENC_EQUIDISTANT, {});
            set_op(ch.ops[n++], V1C_OP_EQUIRECT_DEC, 1, {});
        }
        set_op(ch.ops[n++], V1C_OP_DENORMALIZE, 0, {20.0, 20.0, 24.0, 20.0});
        ch.n_ops = n;
        fails += emul_get_map(&ch, nullptr, W, H, 0, xm, ym, st) != 0;
        const int rc = emul_get_map(&ch, nullptr, W, H, 1, xm, ym, st);
        fails += variant < 9 ? rc != 0 : rc == 0;  // everything but the decoder chain has a fused form
        long long info[12];
        fails += emul_plan_info(&ch, W, H, info) != 0;
        if (rc == 0) {  // the host model of the tiles' table slices, entry sharing and m-polynomial lanes (partial tiles at these sizes)
            double lm[9];
            fails += emul_lane_model_all(&ch, W, H, 0, lm) != 0;
            fails += emul_tile_lane_model(&ch, W, H, 0, 0, 1, lm) != 0;
            double probe[32];
            fails += emul_lane_probe(&ch, W, H, H / 2, 0, probe) != 0;
        }
        if (variant == 1) {  // per-unit rotation override
            const double r2[9] = {1, 0, 0, 0, 0.96, -0.28, 0, 0.28, 0.96};
            fails += emul_get_map(&ch, r2, W, H, 1, xm, ym, st) != 0;
            double claims[11];
            fails += emul_unit_rotation_check(&ch, r2, W, H, claims) > 1;  // (1: not a classic chain)
        }
    }
    // sampler: every interpolation x border x cn, coordinates far outside / NaN / inf / huge, odd pitches;
    // the int16 tables are random in range (the sampler's indexing, not the weights, is under test)
    std::vector<short> itab(1024 * 64);
    for (auto& v : itab)
        v = (short)((int)(rnd() % 4096) - 1024);
    const int SH = 19, SW = 27;
    for (int i = 0; i < W * H; i++) {
        xm[i] = (float)((int)(rnd() % 6000) - 3000) / 50.0f;
        ym[i] = (float)((int)(rnd() % 6000) - 3000) / 50.0f;
    }
    xm[0] = NAN, ym[1] = NAN, xm[2] = INFINITY, ym[3] = -INFINITY, xm[4] = 3e9f, ym[5] = -3e9f, xm[6] = 32767.9f, ym[7] = -32768.4f;
    for (int cn : {1, 3, 4}) {
        const int64_t sp = (int64_t)SW * cn + 5, dp = (int64_t)W * cn + 3;
        uint8_t* src = (uint8_t*)std::malloc((size_t)sp * (SH - 1) + (size_t)SW * cn);
        uint8_t* dst = (uint8_t*)std::malloc((size_t)dp * (H - 1) + (size_t)W * cn);
        for (size_t i = 0; i < (size_t)sp * (SH - 1) + (size_t)SW * cn; i++)
            src[i] = (uint8_t)rnd();
        const uint8_t cval[4] = {9, 8, 7, 6};
        for (int interp : {V1C_INTER_NEAREST, V1C_INTER_LINEAR, V1C_INTER_CUBIC, V1C_INTER_AREA, V1C_INTER_LANCZOS4})
            for (int border = V1C_BORDER_CONSTANT; border <= V1C_BORDER_TRANSPARENT; border++) {
                fails += emul_remap(src, SH, SW, sp, cn, dst, H, W, dp, xm, ym, interp, border, cval, itab.data()) != 0;
                fails += emul_remap(src, 1, 1, sp, cn, dst, H, W, dp, xm, ym, interp, border, cval, itab.data()) != 0;
            }
        std::free(src);
        std::free(dst);
    }
    std::free(xm);
    std::free(ym);
    if (fails) {
        std::fprintf(stderr, "san_emul: %d calls returned an unexpected code\n", fails);
        return 1;
    }
    std::puts("san_emul ok");
    return 0;
}
