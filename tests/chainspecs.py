"""Neutral descriptions of the transformer chains and geometries the parity tests cover.

A *spec* is a list of tuples (see ``oracle/oracle.py::chain_from_spec``).  Three independent
consumers turn a spec into something executable: the reference itself (``tests/golden/
make_golden.py``, build container only), the C oracle (``oracle/oracle.py``) and the product
(``to_product`` below -> ``vr180_convert_amd`` transformer objects -> the product's own lowering).

Chains come from the reference's own tests (tests/test_remapper.py:42-109), from BASELINE.json's
configs and from SURVEY.md section 8c's list of edge cases.
"""
from __future__ import annotations

import math

import numpy as np

EQUI = ("fisheye_dec", "equidistant")


def ry(angle: float) -> list[list[float]]:
    """Rotation by `angle` about y: what quaternion.from_euler_angles(0, angle, 0) encodes
    (reference tests/test_remapper.py:80)."""
    c, s = math.cos(angle), math.sin(angle)
    return [[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]]


def rotvec_quat(r) -> tuple[float, float, float, float]:
    """quaternion.from_rotation_vector: q = exp(r/2) as (w, x, y, z)."""
    r = np.asarray(r, float)
    a = float(np.linalg.norm(r))
    if a == 0.0:
        return (1.0, 0.0, 0.0, 0.0)
    s = math.sin(a / 2) / a
    return (math.cos(a / 2), r[0] * s, r[1] * s, r[2] * s)


def half_quats(q) -> tuple[tuple, tuple]:
    """cli.py:308-319: half_q = sin(phi/2)/sin(phi) * q + 0.5 with phi = arccos(q.w); the left eye
    gets conj(half_q), the right eye half_q (both generally non-unit)."""
    w, x, y, z = q
    phi = math.acos(w)
    k = math.sin(phi / 2) / math.sin(phi)
    h = (k * w + 0.5, k * x, k * y, k * z)
    return (h[0], -h[1], -h[2], -h[3]), h


# name -> (spec, size_output (W, H), size_input (H_in, W_in), radius)
SMALL_CASES: dict[str, tuple] = {}


def _add(name, spec, out=(96, 96), inp=(96, 96), radius=None):
    if radius is None:
        radius = min(inp[0] / 2, inp[1] / 2)  # radius="max", remapper.py:86
    SMALL_CASES[name] = (spec, out, inp, radius)


# reference tests/test_remapper.py:42-74 (test_apply): 6 encoders * FisheyeDecoder("equidistant")
for _m in ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]:
    _add(f"apply_{_m}", [("fisheye_enc", _m), EQUI])
_add("apply_equirectangular", [("equirect_enc", True), EQUI])
# tests/test_remapper.py:77-91 (test_transformer)
_add("transformer_rotator", [("fisheye_enc", "equidistant"), ("rot", ry(math.pi / 4)), EQUI])
_add("transformer_poly", [("fisheye_enc", "equidistant"), ("poly", [0, 1, -0.1]), EQUI])
# tests/test_remapper.py:94-109 (test_lr)
_add("lr_rotator", [("equirect_enc", True), ("rot", ry(math.pi / 4)), EQUI])
_add("lr_poly_default", [("equirect_enc", True), ("poly", [0, 1]), EQUI])
# BASELINE configs at small size
_add("c2_poly", [("equirect_enc", True), ("poly", [0, 1, -0.1]), EQUI], out=(128, 128), inp=(128, 128))
_add("c4_rot_poly", [("equirect_enc", True), ("rot", ry(math.pi / 4)), ("poly", [0, 1, -0.1]), EQUI], out=(128, 128), inp=(128, 128))
_qL, _qR = half_quats(rotvec_quat([0.013, -0.021, 0.017]))
_add("c5_calib_left", [("equirect_enc", True), ("rot_quat", _qL), EQUI], out=(128, 128), inp=(128, 128))
_add("c5_calib_right", [("equirect_enc", True), ("rot_quat", _qR), EQUI], out=(128, 128), inp=(128, 128))
# SURVEY.md 8c edge cases
_add("equirect_lat_x", [("equirect_enc", False), EQUI])
_add("zoom", [("equirect_enc", True), ("zoom", 1.3), EQUI])
_add("zoom_inverse", [("equirect_enc", True), ("inverse", ("zoom", 1.3)), EQUI])
_add("rectilinear_decoder", [("equirect_enc", True), ("rectilinear_dec", 12.0, 17.3)])
_add("rectilinear_decoder_inv", [("inverse", ("rectilinear_dec", 12.0, 17.3)), EQUI])
_add("nonsquare", [("equirect_enc", True), EQUI], out=(160, 100), inp=(135, 240), radius=60.25)
_add("odd_sizes", [("equirect_enc", True), ("poly", [0, 1, -0.1]), EQUI], out=(129, 97), inp=(101, 77))
_add("neg_radius", [("equirect_enc", True), EQUI], radius=-47.5)
_add("poly_c0", [("equirect_enc", True), ("poly", [0.05, 1, -0.1]), EQUI])
_add("poly_negative", [("equirect_enc", True), ("poly", [0, -1]), EQUI])
_add("poly_signchange", [("equirect_enc", True), ("poly", [0.5, -1]), ("fisheye_dec", "stereographic")])
_add("rot_after_radial", [("equirect_enc", True), ("poly", [0, 1, -0.1]), ("rot", ry(0.3)), EQUI])
_add("equirect_decoder", [("fisheye_enc", "equidistant"), ("equirect_dec", True)])
_add("equirect_decoder_lat_x", [("fisheye_enc", "stereographic"), ("equirect_dec", False)])
_add("equisolid_dec", [("equirect_enc", True), ("fisheye_dec", "equisolid")])
_add("orthographic_dec", [("equirect_enc", True), ("fisheye_dec", "orthographic")])
_add("rectilinear_dec_fisheye", [("equirect_enc", True), ("zoom", 2.0), ("fisheye_dec", "rectilinear")])
_add("two_rotations", [("equirect_enc", True), ("rot", ry(0.4)), ("rot_quat", rotvec_quat([0.3, 0.1, -0.2])), ("poly", [0, 1, -0.1]), EQUI])
_add("back_hemisphere", [("equirect_enc", True), ("rot", ry(2.2)), EQUI])

# BASELINE.json configs at full size: (spec, size_output, size_input, radius); SURVEY.md 8d
FULL_CASES = {
    "C1": ([("equirect_enc", True), EQUI], (2048, 2048), (2048, 2048), 1024.0),
    "C2": ([("equirect_enc", True), ("poly", [0, 1, -0.1]), EQUI], (4096, 4096), (4096, 4096), 2048.0),
    "C3": ([("equirect_enc", True), EQUI], (2880, 2880), (2880, 2880), 1440.0),
    "C4": ([("equirect_enc", True), ("rot", ry(math.pi / 4)), ("poly", [0, 1, -0.1]), EQUI], (8192, 8192), (8192, 8192), 4096.0),
}
FULL_STRIDE = 256  # rows / columns kept verbatim in the fixtures

# Chains that do not start with EquirectangularEncoder(is_latitude_y=True), at sizes the tile kernels are selected for: the reference's
# own test chains (tests/test_remapper.py:42-91: five FisheyeEncoders, the rotator and the polynomial between equidistant encoder and
# decoder), is_latitude_y=False, a rotation behind a radial stage, a non-square planar output.  (spec, size_output, size_input, radius)
PLANAR_CASES = {f"apply_{m}": ([("fisheye_enc", m), EQUI], (1024, 1024), (1024, 1024), 512.0)
                for m in ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]}
PLANAR_CASES.update({
    "transformer_rotator": ([("fisheye_enc", "equidistant"), ("rot", ry(math.pi / 4)), EQUI], (1024, 1024), (1024, 1024), 512.0),
    "transformer_poly": ([("fisheye_enc", "equidistant"), ("poly", [0, 1, -0.1]), EQUI], (1024, 1024), (1024, 1024), 512.0),
    "equirect_lat_x": ([("equirect_enc", False), EQUI], (1024, 1024), (1024, 1024), 512.0),
    "equirect_lat_x_rot": ([("equirect_enc", False), ("rot", ry(0.3)), ("poly", [0, 1, -0.1]), EQUI], (1024, 1024), (1024, 1024), 512.0),
    "rot_after_radial": ([("equirect_enc", True), ("poly", [0, 1, -0.1]), ("rot", ry(0.3)), EQUI], (1024, 1024), (1024, 1024), 512.0),
    "planar_rot_small_angle": ([("fisheye_enc", "stereographic"), ("rot_quat", rotvec_quat([0.02, -0.05, 0.03])), ("fisheye_dec", "equisolid")],
                               (1024, 1024), (1024, 1024), 512.0),
    "planar_nonsquare": ([("fisheye_enc", "stereographic"), ("zoom", 1.25), EQUI], (1536, 1024), (1080, 1920), 540.0),
})


def c5_spec(frame: int, eye: int) -> list[tuple]:
    """BASELINE config 5: per-frame calibration quaternion, L = conj(half_q), R = half_q."""
    rng = np.random.default_rng(20240619 + frame)
    q = rotvec_quat(rng.normal(0, 0.02, 3))
    return [("equirect_enc", True), ("rot_quat", half_quats(q)[eye]), EQUI]


def to_product(spec):
    """spec -> product transformer object (vr180_convert_amd public classes)."""
    import vr180_convert_amd.transformer as T

    def one(item):
        kind, *a = item
        if kind == "inverse":
            return T.InverseTransformer(one(a[0]))
        if kind == "equirect_enc":
            return T.EquirectangularEncoder(*(a[:1]))
        if kind == "equirect_dec":
            return T.EquirectangularDecoder(*(a[:1]))
        if kind == "fisheye_enc":
            return T.FisheyeEncoder(a[0])
        if kind == "fisheye_dec":
            return T.FisheyeDecoder(a[0])
        if kind == "poly":
            return T.PolynomialScaler(a[0])
        if kind == "zoom":
            return T.ZoomTransformer(a[0])
        if kind == "rot":
            return T.Euclidean3DRotator(np.asarray(a[0], float))
        if kind == "rot_quat":
            return T.Euclidean3DRotator(tuple(a[0]))
        if kind == "rectilinear_dec":
            return T.RectilinearDecoder(a[0], a[1])
        raise ValueError(item)

    out = one(spec[0])
    for it in spec[1:]:
        out = out * one(it)
    return out


def buckets(m: np.ndarray) -> np.ndarray:
    """cv2's 5-bit fixed point: cvRound(coord * 32) with NaN / overflow -> INT_MIN
    (SURVEY.md Appendix A item 2).  float32 in, int32 out."""
    v = m.astype(np.float32) * np.float32(32)
    bad = ~(np.abs(v) < 2147483648.0)  # NaN or out of int range
    out = np.rint(np.where(bad, 0, v)).astype(np.int64)
    out[bad] = -(2**31)
    return out.astype(np.int32)
