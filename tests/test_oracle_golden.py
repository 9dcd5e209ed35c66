"""Pins the C oracle's coordinate chain to maps produced by the imported reference
(tests/golden/make_golden.py).  Criterion: identical cvRound(coord*32) buckets,
NaN <=> NaN, |delta| <= 1e-6 px; full-size configs additionally by SHA-256 of the bucket planes."""
import hashlib

import numpy as np
import pytest

import chainspecs as CS


def assert_maps_match(xm, ym, gx, gy, what=""):
    assert xm.shape == gx.shape and xm.dtype == np.float32
    for a, g in ((xm, gx), (ym, gy)):
        assert np.array_equal(np.isnan(a), np.isnan(g)), f"{what}: NaN pattern differs"
        assert np.array_equal(CS.buckets(a), CS.buckets(g)), f"{what}: 1/32-pixel buckets differ"
        fin = np.isfinite(g) & (np.abs(g) < 1e6)
        assert np.max(np.abs(a[fin].astype(np.float64) - g[fin]), initial=0.0) <= 1e-6, f"{what}: |delta| > 1e-6 px"


@pytest.mark.parametrize("name", list(CS.SMALL_CASES))
def test_oracle_small_maps(oracle_mod, golden_dir, name):
    g = np.load(golden_dir / "maps_small.npz")
    spec, out, inp, radius = CS.SMALL_CASES[name]
    xm, ym = oracle_mod.get_map(spec, radius=radius, size_input=inp, size_output=out)
    assert_maps_match(xm, ym, g[f"{name}__x"], g[f"{name}__y"], name)


@pytest.mark.parametrize("name", list(CS.FULL_CASES))
def test_oracle_full_size_buckets(oracle_mod, golden_dir, name):
    g = np.load(golden_dir / "maps_full.npz")
    spec, out, inp, radius = CS.FULL_CASES[name]
    xm, ym = oracle_mod.get_map(spec, radius=radius, size_input=inp, size_output=out)
    s = CS.FULL_STRIDE
    assert_maps_match(xm[::s], ym[::s], g[f"{name}__rows_x"], g[f"{name}__rows_y"], name + " rows")
    assert_maps_match(xm[:, ::s], ym[:, ::s], g[f"{name}__cols_x"], g[f"{name}__cols_y"], name + " cols")
    assert hashlib.sha256(CS.buckets(xm).tobytes()).digest() == g[f"{name}__sha_bx"].tobytes()
    assert hashlib.sha256(CS.buckets(ym).tobytes()).digest() == g[f"{name}__sha_by"].tobytes()


def test_oracle_c5_units(oracle_mod, golden_dir):
    g = np.load(golden_dir / "maps_c5.npz")
    for frame in (0, 1, 7):
        for eye in (0, 1):
            xm, ym = oracle_mod.get_map(CS.c5_spec(frame, eye), radius=96.0, size_input=(192, 192), size_output=(192, 192))
            assert_maps_match(xm, ym, g[f"f{frame}_e{eye}__x"], g[f"f{frame}_e{eye}__y"], f"c5 f{frame} e{eye}")


def test_oracle_get_radius(oracle_mod, golden_dir):
    g = np.load(golden_dir / "radius.npz")
    assert oracle_mod.get_radius(g["landscape_img"]) == float(g["landscape_radius"])
    assert oracle_mod.get_radius(g["portrait_img"]) == float(g["portrait_radius"])
    assert oracle_mod.get_radius(g["noisy_img"]) == float(g["noisy_radius"])
    assert oracle_mod.get_radius(g["noisy_img"], threshold=25) == float(g["thr_radius"])
    assert bool(g["noborder_raises"])
    with pytest.raises(IndexError):
        oracle_mod.get_radius(np.full((64, 80, 3), 90, np.uint8))
    assert oracle_mod.get_radius_smart("auto", [g["landscape_img"], g["noisy_img"]]) == float(g["smart_auto"])
    assert oracle_mod.get_radius_smart("max", [g["landscape_img"]]) == float(g["smart_max"])
    assert oracle_mod.get_radius_smart(33.5, [g["landscape_img"]]) == float(g["smart_num"])
