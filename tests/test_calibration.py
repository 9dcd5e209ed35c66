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


def test_fitted_quaternion_sign_never_doubles_the_per_eye_rotation():
    """q and -q are one rotation, but ``phi = arccos(q.w)`` (cli.py:308-311) is not sign-blind: rotation_match returns
    w >= 0 whatever LAPACK's eigenvector sign, and calibration_rotators gives each eye HALF the angle for q and for -q."""
    from vr180_convert_amd.calibration import calibration_rotators
    from vr180_convert_amd.quat import as_rotation_matrix, quaternion

    def angle(m):
        return float(np.arccos(np.clip((np.trace(m) - 1.0) / 2.0, -1.0, 1.0)))

    rng = np.random.default_rng(7)
    for _ in range(200):
        rv = rng.normal(0, 0.02, 3)
        q = from_rotation_vector(rv)
        pts = rng.normal(size=(40, 3))
        got = rotation_match(pts, rotate_vectors(q, pts))
        assert got.w >= 0
        theta = float(np.linalg.norm(rv))
        for qq in (got, quaternion(-got.w, -got.x, -got.y, -got.z)):
            ql, qr = calibration_rotators(qq)
            np.testing.assert_allclose(angle(as_rotation_matrix(qr)), theta / 2, rtol=1e-3, atol=1e-9)
            np.testing.assert_allclose(angle(as_rotation_matrix(ql)), theta / 2, rtol=1e-3, atol=1e-9)
            np.testing.assert_allclose(as_rotation_matrix(qr) @ as_rotation_matrix(qr), as_rotation_matrix(q), atol=1e-4)  # pseudo-half: O(phi^3)


def test_calibration_names_live_in_remapper_like_the_reference():
    import vr180_convert_amd.remapper as R
    from vr180_convert_amd import calibration

    assert R.rotation_match is calibration.rotation_match and R.match_lr is calibration.match_lr
    assert R.rotation_match_robust is calibration.rotation_match_robust


def test_quaternion_convention_is_pinned_by_the_references_own_rotation_match(golden_dir):
    """The reference's only rotation contract (tests/test_remapper.py:118-130): points rotated with
    numpy-quaternion's ``rotate_vectors(q, a)`` are recovered by ``rotation_match(a, b) == q``, so the
    quaternion -> rotation convention of the package IS the one rotation_match's NumPy math solves for
    (Hamilton product, active rotation: (0, R a) = q (0, a) q^-1, docs/math.md:7-24).  ``q0`` of the
    fixture is what the REFERENCE's rotation_match returned for the noise-free pair (a0, b0): this
    engine's ``as_rotation_matrix`` (quat.py, used by Euclidean3DRotator and every rotated plan) must
    carry a0 onto b0 -- a transposed (passive) or differently ordered convention fails by O(1)."""
    from vr180_convert_amd.quat import as_rotation_matrix, quaternion, rotate_vectors as rv

    g = np.load(golden_dir / "rotation_match.npz")
    a, b, q = g["a0"], g["b0"], g["q0"]
    m = as_rotation_matrix(quaternion(*q))
    np.testing.assert_allclose(m @ a.T, b.T, rtol=0, atol=1e-12)
    np.testing.assert_allclose(rv(quaternion(*q), a), b, rtol=0, atol=1e-12)  # the rotate_vectors call shape of transformer.py:676
    np.testing.assert_allclose(as_rotation_matrix(quaternion(*(-q))), m, rtol=0, atol=1e-15)  # q and -q: one rotation
    # non-unit quaternions are normalised, not rejected (the CLI's pseudo-half quaternions, cli.py:308-319)
    np.testing.assert_allclose(as_rotation_matrix(quaternion(*(1.7 * q))), m, rtol=0, atol=1e-14)
    # the noisy pairs: the reference's q is the least-squares rotation, so it must beat small perturbations of itself
    for k in (1, 2):
        ak, bk, qk = g[f"a{k}"], g[f"b{k}"], g[f"q{k}"]
        e0 = np.sum((ak @ as_rotation_matrix(quaternion(*qk)).T - bk) ** 2)
        rng = np.random.default_rng(k)
        for _ in range(8):
            qp = qk + 1e-3 * rng.normal(size=4)
            assert np.sum((ak @ as_rotation_matrix(quaternion(*qp)).T - bk) ** 2) > e0


def test_rotation_vector_and_euler_constructors_mean_what_the_reference_tests_assume():
    """from_rotation_vector(r): rotation by |r| about r, right-handed, active -- checked on the axes
    against the pinned matrix convention; from_euler_angles(0, b, 0) is the same rotation as
    from_rotation_vector([0, b, 0]) (tests/test_remapper.py:80 and tests/test_cli.py:33 use the two
    spellings for the one "pi/4 about y" rotation of the reference's test chains)."""
    from vr180_convert_amd.quat import as_rotation_matrix, from_euler_angles

    t = 0.3
    c, s = np.cos(t), np.sin(t)
    np.testing.assert_allclose(as_rotation_matrix(from_rotation_vector([0, 0, t])), [[c, -s, 0], [s, c, 0], [0, 0, 1]], atol=1e-15)
    np.testing.assert_allclose(as_rotation_matrix(from_rotation_vector([0, t, 0])), [[c, 0, s], [0, 1, 0], [-s, 0, c]], atol=1e-15)
    np.testing.assert_allclose(as_rotation_matrix(from_rotation_vector([t, 0, 0])), [[1, 0, 0], [0, c, -s], [0, s, c]], atol=1e-15)
    np.testing.assert_allclose(as_rotation_matrix(from_euler_angles(0.0, np.pi / 4, 0.0)),
                               as_rotation_matrix(from_rotation_vector([0, np.pi / 4, 0])), atol=1e-15)
    # z-y-z composition: R = Rz(alpha) Ry(beta) Rz(gamma)
    al, be, ga = 0.4, -0.7, 1.1
    rz = lambda x: as_rotation_matrix(from_rotation_vector([0, 0, x]))  # noqa: E731
    ry = lambda x: as_rotation_matrix(from_rotation_vector([0, x, 0]))  # noqa: E731
    np.testing.assert_allclose(as_rotation_matrix(from_euler_angles(al, be, ga)), rz(al) @ ry(be) @ rz(ga), atol=1e-15)
