"""Host-side logic of the product: the Transformer API mirror, its lowering, quaternion helpers
and sharding -- checked against data produced by the imported reference (tests/golden/*.npz)."""
import inspect
import math

import numpy as np
import pytest

import chainspecs as CS
import vr180_convert_amd as V
from vr180_convert_amd import _abi, chain as CH, quat as Q
from vr180_convert_amd import transformer as T
from vr180_convert_amd.sharding import shard_range, shard_sizes


def test_public_names_match_reference():
    # reference src/vr180_convert/__init__.py:17-32
    ref_all = ["TransformerBase", "ZoomTransformer", "MultiTransformer", "NormalizeTransformer", "PolarRollTransformer",
               "DenormalizeTransformer", "FisheyeDecoder", "FisheyeEncoder", "EquirectangularEncoder", "Euclidean3DRotator",
               "Euclidean3DTransformer", "apply", "apply_lr", "get_map"]
    for n in ref_all:
        assert n in V.__all__ and hasattr(V, n)
    for n in ["PolynomialScaler", "RectilinearDecoder", "InverseTransformer", "EquirectangularDecoder", "get_radius",
              "equidistant_to_3d", "equidistant_from_3d"]:
        assert hasattr(T, n)


def test_signatures_match_reference():
    # reference remapper.py:324-334, 406-418, 23-29 -- including the `boarder_*` spelling
    ap = inspect.signature(V.apply).parameters
    assert list(ap)[:8] == ["transformer", "in_paths", "out_paths", "size_output", "interpolation", "boarder_mode",
                            "boarder_value", "radius"]
    assert ap["size_output"].default == (2048, 2048) and ap["interpolation"].default == 4
    assert ap["boarder_mode"].default == 0 and ap["boarder_value"].default == 0 and ap["radius"].default == "auto"
    assert all(p.kind is inspect.Parameter.KEYWORD_ONLY for n, p in ap.items() if n != "transformer")
    lr = inspect.signature(V.apply_lr).parameters
    assert list(lr)[:10] == ["transformer", "left_path", "right_path", "out_path", "size_output", "interpolation",
                             "boarder_mode", "boarder_value", "radius", "merge"]
    assert lr["merge"].default is False
    gm = inspect.signature(V.get_map).parameters
    assert list(gm)[:4] == ["transformer", "radius", "size_input", "size_output"]


def test_mul_flattens():
    a, b, c = T.ZoomTransformer(2), T.PolynomialScaler(), T.FisheyeEncoder("equidistant")
    m = (a * b) * (c * a)
    assert isinstance(m, T.MultiTransformer) and m.transformers == [a, b, c, a]
    assert (a * (b * c)).transformers == [a, b, c] and ((a * b) * c).transformers == [a, b, c]


def test_stage_transforms_match_reference(golden_dir):
    g = np.load(golden_dir / "stages.npz")
    px, py = g["px"], g["py"]
    singles = {
        "zoom": T.ZoomTransformer(1.7), "poly": T.PolynomialScaler([0.1, 0.9, -0.05, 0.01]),
        "equirect": T.EquirectangularEncoder(), "equirect_x": T.EquirectangularEncoder(False),
        "rot": T.Euclidean3DRotator(np.array(CS.ry(0.5))), "rectdec": T.RectilinearDecoder(10.0, 36.0),
    }
    for m in ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]:
        singles[f"fe_{m}"] = T.FisheyeEncoder(m)
    with np.errstate(invalid="ignore"):
        for name, t in singles.items():
            fx, fy = t.transform(px, py)
            np.testing.assert_allclose(fx, g[f"{name}__fwd_x"], rtol=1e-13, atol=1e-15, equal_nan=True)
            np.testing.assert_allclose(fy, g[f"{name}__fwd_y"], rtol=1e-13, atol=1e-15, equal_nan=True)
            if name != "poly":
                ix, iy = t.inverse_transform(px, py)
                np.testing.assert_allclose(ix, g[f"{name}__inv_x"], rtol=1e-13, atol=1e-15, equal_nan=True)
                np.testing.assert_allclose(iy, g[f"{name}__inv_y"], rtol=1e-13, atol=1e-15, equal_nan=True)
    dn = T.DenormalizeTransformer(scale=(100.5, 99.0), center=(320, 241))
    assert np.array_equal(dn.transform(px, py)[0], g["denorm__fwd_x"])
    assert np.array_equal(dn.inverse_transform(px * 400, py * 400)[1], g["denorm__inv_y"])
    gx, gy = np.meshgrid(np.arange(31), np.arange(33))
    assert np.array_equal(T.NormalizeTransformer().transform(gx, gy)[0], g["norm__fwd_x"])
    assert np.array_equal(T.NormalizeTransformer(scale="max").transform(gx, gy)[1], g["normmax__fwd_y"])


def test_equidistant_3d_roundtrip(golden_dir):
    # reference tests/test_remapper.py:112-115
    g = np.load(golden_dir / "equidistant3d.npz")
    v = T.equidistant_to_3d(g["x"], g["y"])
    np.testing.assert_allclose(v, g["v"], rtol=1e-14, atol=1e-16)
    np.testing.assert_allclose(T.equidistant_from_3d(v), (g["x"], g["y"]))


def test_error_behaviour():
    x = np.zeros((2, 2))
    with pytest.raises(ValueError, match="Unknown mapping type"):
        T.FisheyeEncoder("bogus").transform(x, x)
    with pytest.raises(ValueError, match="Unknown mapping type"):
        (T.FisheyeEncoder("bogus") * T.ZoomTransformer(1)).lower((4, 4))
    with pytest.raises(NotImplementedError):
        T.PolynomialScaler().inverse_transform(x, x)
    with pytest.raises(NotImplementedError):
        T.InverseTransformer(T.PolynomialScaler()).lower((4, 4))
    with pytest.warns(UserWarning):
        T.RectilinearDecoder(10.0).factor
    with pytest.raises(ZeroDivisionError):
        T.Euclidean3DRotator((0, 0, 0, 0)).matrix


def test_get_radius_matches_reference(golden_dir):
    g = np.load(golden_dir / "radius.npz")
    assert T.get_radius(g["landscape_img"]) == float(g["landscape_radius"])
    assert T.get_radius(g["portrait_img"]) == float(g["portrait_radius"])
    assert T.get_radius(g["noisy_img"], threshold=25) == float(g["thr_radius"])
    with pytest.raises(IndexError):
        T.get_radius(np.full((64, 80, 3), 90, np.uint8))


def test_lowering_equals_oracle_chain(oracle_mod):
    """The product's lowering and the oracle's independent spec->chain builder must agree op by
    op (so that parity tests feed both sides the same chain by construction, not by sharing code)."""
    for name, (spec, out, inp, radius) in CS.SMALL_CASES.items():
        mine = CH.lower_for_get_map(CS.to_product(spec), radius=radius, size_input=inp, size_output=out)
        ref = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
        assert mine.n_ops == ref.n_ops, name
        for i in range(ref.n_ops):
            a, b = mine.ops[i], ref.ops[i]
            assert (a.opcode, a.iparam, a.nparam) == (b.opcode, b.iparam, b.nparam), (name, i)
            np.testing.assert_allclose(list(a.p[: a.nparam]), list(b.p[: b.nparam]), rtol=0, atol=1e-16, err_msg=name)


def test_not_lowerable_cases():
    class Mine(T.TransformerBase):
        def transform(self, x, y, **kw):
            return x * 2, y

        def inverse_transform(self, x, y, **kw):
            return x / 2, y

    class TweakedZoom(T.ZoomTransformer):
        def transform(self, x, y, **kw):
            return x, y

    for t in (Mine(), TweakedZoom(2.0), T.PolynomialScaler(list(range(20))), T.NormalizeTransformer(scale=(2, 3))):
        with pytest.raises(T.NotLowerable):
            CH.lower_for_get_map(T.EquirectangularEncoder() * t, radius=1.0, size_input=(8, 8), size_output=(8, 8))
    long_chain = T.ZoomTransformer(1.0)
    for _ in range(20):
        long_chain = long_chain * T.ZoomTransformer(1.0)
    with pytest.raises(T.NotLowerable):
        CH.lower_for_get_map(long_chain, radius=1.0, size_input=(8, 8), size_output=(8, 8))


def test_quaternion_helpers():
    q = Q.from_euler_angles(0.0, math.pi / 4, 0.0)
    np.testing.assert_allclose(q.components(), (math.cos(math.pi / 8), 0, math.sin(math.pi / 8), 0), atol=1e-16)
    np.testing.assert_allclose(Q.as_rotation_matrix(q), CS.ry(math.pi / 4), atol=1e-15)
    np.testing.assert_allclose(Q.from_rotation_vector([0, math.pi / 4, 0]).components(), q.components(), atol=1e-16)
    # non-unit quaternions are normalised; matrix input passes through; tuple == object
    h = 0.37 * q + 0.5
    m = Q.as_rotation_matrix(h)
    np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-15)
    assert np.array_equal(Q.as_rotation_matrix(h.components()), m)
    assert Q.as_rotation_matrix(m) is not None and np.array_equal(Q.as_rotation_matrix(m), m)
    # reference tests/test_remapper.py:118-130 self-consistency: rotate, then conj rotates back
    rng = np.random.default_rng(0)
    v = rng.random((100, 3))
    r = Q.from_rotation_vector([0.1, 0.2, 0.3])
    np.testing.assert_allclose(Q.rotate_vectors(r.conj(), Q.rotate_vectors(r, v)), v, atol=1e-15)
    # cli.py:308-319 half quaternions agree with the oracle-side restatement used by the fixtures
    qL, qR = CS.half_quats(r.components())
    phi = math.acos(r.w)
    hq = math.sin(phi / 2) / math.sin(phi) * r + 0.5
    np.testing.assert_allclose(hq.components(), qR, atol=1e-16)
    np.testing.assert_allclose(hq.conj().components(), qL, atol=1e-16)
    # the CLI's "half" quaternion is only approximately half the rotation (it is not renormalised
    # before 0.5 is added): twice the half rotation is close to, not equal to, the full one
    np.testing.assert_allclose(Q.as_rotation_matrix(hq) @ Q.as_rotation_matrix(hq), Q.as_rotation_matrix(r), atol=5e-3)


def test_sharding_partitions():
    for n in (0, 1, 7, 64, 256, 513):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert [i for b in blocks for i in b] == list(range(n))
            sizes = shard_sizes(n, w)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vr180_convert_amd import _native

    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(_native.EngineUnavailable):
        V.apply(T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant"), in_paths=img, radius="max", size_output=(16, 16))
    with pytest.raises(_native.EngineUnavailable):
        V.get_map(T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant"), radius=8.0, size_input=(16, 16), size_output=(16, 16))


def test_io_many_round_trip(tmp_path):
    """Image files either side of the path (reference remapper.py:373,402): batches are decoded /
    encoded on a thread pool; PNG is lossless, arrays are BGR like cv2's."""
    from vr180_convert_amd import _io

    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (37, 41, 3), dtype=np.uint8) for _ in range(5)]
    paths = [tmp_path / f"im{k}.png" for k in range(5)]
    _io.imwrite_many(paths, imgs)
    back = _io.imread_many([*paths[:4], imgs[4]])  # arrays pass through
    assert all(np.array_equal(a, b) for a, b in zip(back, imgs))
    assert _io.imread(tmp_path / "missing.png") is None  # cv2.imread's contract
    _io.imwrite(tmp_path / "q.jpg", imgs[0])
    assert _io.imread(tmp_path / "q.jpg").shape == (37, 41, 3)


def test_estimator_surface_of_the_transformers():
    """The reference's TransformerBase is an sklearn BaseEstimator / TransformerMixin (transformer.py:11-18): get_params / set_params /
    fit_transform and sklearn.base.clone work on the drop-in classes too."""
    t = T.PolynomialScaler([0, 1, -0.1])
    assert t.get_params() == {"coefs_reverse": [0, 1, -0.1]}
    assert t.set_params(coefs_reverse=[0, 2]) is t and t.coefs_reverse == [0, 2]
    with pytest.raises(ValueError, match="Invalid parameter"):
        t.set_params(nope=1)
    enc = T.FisheyeEncoder("equisolid")
    assert enc.get_params() == {"mapping_type": "equisolid"}
    inv = T.InverseTransformer(T.ZoomTransformer(2.0))
    p = inv.get_params()
    assert p["transformer__scale"] == 2.0 and isinstance(p["transformer"], T.ZoomTransformer)
    inv.set_params(transformer__scale=4.0)
    assert inv.transformer.scale == 4.0
    x, y = np.array([0.5, 1.0]), np.array([0.25, -1.0])
    fx, fy = T.ZoomTransformer(2.0).fit_transform(x, y)
    assert np.array_equal(fx, x / 2.0) and np.array_equal(fy, y / 2.0)
    chain = T.EquirectangularEncoder() * T.ZoomTransformer(1.5)
    assert list(chain.get_params(deep=False)) == ["transformers"]
    assert "ZoomTransformer(scale=1.5)" in repr(chain)
    try:
        from sklearn.base import clone
    except Exception:  # noqa: BLE001
        return
    c = clone(T.ZoomTransformer(3.0))
    assert isinstance(c, T.ZoomTransformer) and c.scale == 3.0


def test_the_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under vr180_convert_amd/ (Python or C++) names it, importing the package does not
    load it, and bench.py reaches it only inside its cpu_baseline leg."""
    import re
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    for f in list((root / "vr180_convert_amd").rglob("*.py")) + list((root / "vr180_convert_amd" / "csrc").glob("*.h*")) + \
            list((root / "vr180_convert_amd" / "csrc").glob("*.hip")):
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle|libvr180oracle|orc_[a-z_]+\(", text, flags=re.M), f
    code = "import sys; import vr180_convert_amd, vr180_convert_amd.remapper, vr180_convert_amd.sharding, vr180_convert_amd.cli; " \
           "print(sorted(m for m in sys.modules if m.split('.')[0] == 'oracle'))"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, check=True).stdout.strip()
    assert out == "[]", out
    bench = (root / "bench.py").read_text()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle", bench)]
    assert uses, "bench.py's cpu_baseline leg imports the oracle"
    for u in uses:  # every import sits inside a function whose name says cpu_baseline / parity check
        head = bench.rfind("\ndef ", 0, u)
        assert re.match(r"\ndef (cpu_baseline|_cpu_baseline|parity|_parity)", bench[head:head + 40]), bench[head:head + 60]
