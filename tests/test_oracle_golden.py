"""Pins the C oracle's coordinate chain to maps produced by the imported reference
(tests/golden/make_golden.py).  Criterion: identical cvRound(coord*32) buckets,
NaN <=> NaN, |delta| <= 1e-6 px; full-size configs additionally by SHA-256 of the bucket planes."""
import hashlib

import numpy as np
import pytest

import chainspecs as CS


# Output pixels whose reference coordinate is a quotient of rounding residues (no libm-independent
# value exists).  equirect_decoder_lat_x, pixel (row 48, col 0): the point (-1, 0) lands exactly on
# the lat = -90 deg pole of EquirectangularEncoder.inverse_transform (transformer.py:580-583), where
# lon = arctan2(v_y ~ 6e-17, v_z ~ 6e-17).
SINGULAR = {"equirect_decoder_lat_x": [(48, 0)]}

FAR = float(2**24)  # beyond this a coordinate has no sub-pixel meaning left in float32


def assert_maps_match(xm, ym, gx, gy, what=""):
    """Identical buckets / NaN pattern / |delta| <= 1e-6 px wherever the reference's coordinates
    are numerically meaningful.  Where the reference itself is at |coord| >= 2^24 (a tan() pole:
    radial values ~1e18 times a cosine ~1e-16) the OTHER coordinate is rounding noise of whichever
    libm computed it; there only "still far outside" is required (even the sign of tan() next to its pole is noise) -- such pixels
    are outside the image for every border mode's purposes but WRAP/REFLECT, whose result is
    equally meaningless in the reference."""
    assert xm.shape == gx.shape and xm.dtype == np.float32
    with np.errstate(invalid="ignore"):
        far = (np.abs(gx) >= FAR) | (np.abs(gy) >= FAR)
    for (j, i) in SINGULAR.get(what.split(" ")[0], []):
        far[j, i] = True
        xm, ym = xm.copy(), ym.copy()
        xm[j, i], ym[j, i] = gx[j, i], gy[j, i]
    for a, g in ((xm, gx), (ym, gy)):
        assert np.array_equal(np.isnan(a), np.isnan(g)), f"{what}: NaN pattern differs"
        ok = ~far
        assert np.array_equal(CS.buckets(a)[ok], CS.buckets(g)[ok]), f"{what}: 1/32-pixel buckets differ"
        fin = ok & np.isfinite(g)
        assert np.max(np.abs(a[fin].astype(np.float64) - g[fin]), initial=0.0) <= 1e-6, f"{what}: |delta| > 1e-6 px"
        with np.errstate(invalid="ignore"):
            big = far & (np.abs(g) >= FAR)
            assert np.all(np.abs(a[big]) >= FAR / 2), f"{what}: far-outside pixels moved inside"


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


@pytest.mark.parametrize("name", list(CS.SMALL_CASES))
def test_numpy_chain_vs_goldens(golden_dir, name):
    """oracle/chain_numpy.py (the reference-equivalent NumPy CPU path timed by bench.py) against the
    reference's own maps."""
    from oracle import chain_numpy

    g = np.load(golden_dir / "maps_small.npz")
    spec, out, inp, radius = CS.SMALL_CASES[name]
    with np.errstate(all="ignore"):
        xm, ym = chain_numpy.get_map(spec, radius=radius, size_input=inp, size_output=out)
    assert_maps_match(xm, ym, g[f"{name}__x"], g[f"{name}__y"], name)
