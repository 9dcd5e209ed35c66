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
