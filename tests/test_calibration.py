"""rotation_match / rotation_match_robust (reference remapper.py:93-191) against vectors produced by
the reference itself (tests/golden/make_golden.py rotation_match) and the reference's own
self-consistency test (tests/test_remapper.py:118-130: rotate, recover, atol 1e-3)."""
import numpy as np

from vr180_convert_amd.calibration import rotate_vectors, rotation_match, rotation_match_robust
from vr180_convert_amd.quat import from_rotation_vector


def _same_rotation(q, want, atol):
    got = np.array([q.w, q.x, q.y, q.z])
    want = np.asarray(want, float)
    return np.allclose(got, want, atol=atol) or np.allclose(got, -want, atol=atol)  # q and -q: one rotation


def test_rotation_match_equals_reference_vectors(golden_dir):
    g = np.load(golden_dir / "rotation_match.npz")
    for k in range(3):
        q = rotation_match(g[f"a{k}"], g[f"b{k}"])
        assert _same_rotation(q, g[f"q{k}"], 1e-12), k


def test_rotation_match_recovers_a_rotation_like_the_reference_test():
    rng = np.random.default_rng(0)
    q = from_rotation_vector(rng.normal(0, 0.3, 3))
    pts = rng.normal(size=(50, 3))
    got = rotation_match(pts, rotate_vectors(q, pts))
    assert _same_rotation(got, [q.w, q.x, q.y, q.z], 1e-3)


def test_rotation_match_robust_drops_outliers():
    rng = np.random.default_rng(1)
    q = from_rotation_vector([0.1, -0.2, 0.05])
    a = rng.normal(size=(300, 3))
    b = rotate_vectors(q, a) + 1e-4 * rng.normal(size=a.shape)
    b[:20] += rng.normal(0, 1.0, (20, 3))  # gross outliers
    got, bad = rotation_match_robust(a, b)
    assert _same_rotation(got, [q.w, q.x, q.y, q.z], 1e-3)
    assert bad.shape == (300,) and bad[:20].all() and bad.sum() < 300


def _decoders():
    from vr180_convert_amd import transformer as T

    return {
        "equidistant": T.FisheyeDecoder("equidistant"),
        "stereographic_zoom": T.ZoomTransformer(1.2) * T.FisheyeDecoder("stereographic"),
        "tuple": (T.FisheyeDecoder("equisolid"), T.FisheyeDecoder("rectilinear")),
    }


def test_match_lr_equals_reference_vectors(golden_dir):
    """match_lr (reference remapper.py:251-321: matched pixels -> unit rays through the decoder's
    inverse) against rays the reference's own function produced (make_golden.py match_lr)."""
    from vr180_convert_amd.calibration import match_lr

    g = np.load(golden_dir / "match_lr.npz")
    for name, dec in _decoders().items():
        h, w, radius = g[f"{name}_geom"]
        img = np.zeros((int(h), int(w), 3), np.uint8)
        vl, vr = match_lr(dec, g[f"{name}_pl"], g[f"{name}_pr"], [img, img], radius=float(radius))
        assert vl.shape == g[f"{name}_vl"].shape
        np.testing.assert_allclose(vl, g[f"{name}_vl"], rtol=0, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(vr, g[f"{name}_vr"], rtol=0, atol=1e-14, err_msg=name)
    with np.testing.assert_raises(ValueError):
        match_lr(_decoders()["equidistant"], [(1, 2)], [(1, 2), (3, 4)], [img, img], radius=10.0)


def test_calibration_rotators_align_the_eyes():
    """cli.py:308-319: q rotates the left rays onto the right ones; the left eye gets conj(half_q),
    the right eye half_q (pseudo-halves, non-unit), which together undo q:
    R(half_q) R(half_q) == R(q) up to the pseudo-half's approximation, and R(conj) == R(half)^T."""
    from vr180_convert_amd.calibration import calibration_rotators
    from vr180_convert_amd.quat import as_rotation_matrix

    q = from_rotation_vector([0.03, -0.02, 0.04])
    ql, qr = calibration_rotators(q)
    assert abs(np.sqrt(qr.w**2 + qr.x**2 + qr.y**2 + qr.z**2) - 1.0) > 1e-6  # not unit, like the reference's
    ml, mr = as_rotation_matrix(ql), as_rotation_matrix(qr)
    np.testing.assert_allclose(ml, mr.T, atol=1e-15)
    np.testing.assert_allclose(mr @ mr, as_rotation_matrix(q), atol=1e-5)  # (a pseudo-half: exact to O(phi^2))
    # same numbers as tests/chainspecs.py::half_quats, which the C5 fixtures were generated with
    import chainspecs as CS

    want_l, want_r = CS.half_quats((q.w, q.x, q.y, q.z))
    np.testing.assert_allclose([ql.w, ql.x, ql.y, ql.z], want_l, atol=1e-16)
    np.testing.assert_allclose([qr.w, qr.x, qr.y, qr.z], want_r, atol=1e-16)


def test_calibration_names_live_in_remapper_like_the_reference():
    import vr180_convert_amd.remapper as R
    from vr180_convert_amd import calibration

    assert R.rotation_match is calibration.rotation_match and R.match_lr is calibration.match_lr
    assert R.rotation_match_robust is calibration.rotation_match_robust
