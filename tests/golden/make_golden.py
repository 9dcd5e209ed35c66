#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

Run from the repo root:   python tests/golden/make_golden.py

The reference (`/root/reference/src/vr180_convert`) cannot be imported as is here: its modules do
`import cv2` and `from quaternion import ...` at the top, and neither third-party package is
installed.  Only NAMES are needed at import time (default arguments `cv.INTER_LANCZOS4`,
`cv.BORDER_CONSTANT`; the `quaternion` / `rotate_vectors` / `as_quat_array` symbols), so two empty
placeholder modules carrying those names are registered first.  They contain no arithmetic.

Everything the fixtures pin is computed by the reference's own code:
  * `vr180_convert.remapper.get_map`  (remapper.py:23-59)  -> float32 xmap / ymap
  * `vr180_convert.transformer.get_radius` (transformer.py:108-140)
  * `equidistant_to_3d` / `equidistant_from_3d` (transformer.py:483-530)
Rotations: `Euclidean3DRotator.transform_v` calls numpy-quaternion's `rotate_vectors`
(transformer.py:675-676), which is absent.  The fixtures therefore use a subclass of the
reference's own `Euclidean3DTransformer` whose `transform_v` applies a 3x3 matrix supplied by the
spec -- the reference's 2-D <-> 3-D conversions around the rotation are pinned, the
quaternion -> matrix convention is NOT ("parity unpinned" for that step).

Outputs are data (inputs + expected outputs) -- no reference source text is stored.
"""
from __future__ import annotations

import hashlib
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

_q = types.ModuleType("quaternion")
_q.quaternion = type("quaternion", (), {})
_q.rotate_vectors = None
_q.as_quat_array = None
sys.modules["quaternion"] = _q
_cv = types.ModuleType("cv2")
_cv.INTER_LANCZOS4, _cv.INTER_LINEAR, _cv.BORDER_CONSTANT = 4, 1, 0
sys.modules["cv2"] = _cv
sys.path.insert(0, "/root/reference/src")

import vr180_convert.transformer as RT  # noqa: E402
from vr180_convert.remapper import get_map, get_radius_smart  # noqa: E402

import chainspecs as CS  # noqa: E402
from oracle.oracle import quat_to_matrix  # noqa: E402  (Appendix-B formula; unpinned, see above)


class MatrixRotator(RT.Euclidean3DTransformer):
    """Reference 3-D plumbing (transformer.py:651-657) around a caller-supplied matrix."""

    def __init__(self, m):
        self.m = np.asarray(m, float).reshape(3, 3)

    def transform_v(self, v):
        return np.einsum("ij,...j->...i", self.m, v)

    def inverse_transform_v(self, v):
        return np.einsum("ji,...j->...i", self.m, v)


def to_reference(spec):
    def one(item):
        kind, *a = item
        if kind == "inverse":
            return RT.InverseTransformer(one(a[0]))
        if kind == "equirect_enc":
            return RT.EquirectangularEncoder(*(a[:1]))
        if kind == "equirect_dec":
            return RT.EquirectangularDecoder(*(a[:1]))
        if kind == "fisheye_enc":
            return RT.FisheyeEncoder(a[0])
        if kind == "fisheye_dec":
            return RT.FisheyeDecoder(a[0])
        if kind == "poly":
            return RT.PolynomialScaler(a[0])
        if kind == "zoom":
            return RT.ZoomTransformer(a[0])
        if kind == "rot":
            return MatrixRotator(a[0])
        if kind == "rot_quat":
            return MatrixRotator(quat_to_matrix(a[0]))
        if kind == "rectilinear_dec":
            return RT.RectilinearDecoder(a[0], a[1])
        raise ValueError(item)

    out = one(spec[0])
    for it in spec[1:]:
        out = out * one(it)
    return out


def main() -> None:
    out_dir = Path(__file__).resolve().parent
    import warnings

    warnings.simplefilter("ignore")  # arcsin > 1 etc. produce the NaNs we want to pin

    small = {}
    for name, (spec, out, inp, radius) in CS.SMALL_CASES.items():
        xm, ym = get_map(to_reference(spec), radius=radius, size_input=inp, size_output=out)
        assert xm.dtype == np.float32 and xm.shape == (out[1], out[0])
        small[f"{name}__x"] = xm
        small[f"{name}__y"] = ym
        print(f"small {name:28s} {xm.shape} nan={int(np.isnan(xm).sum())}")
    np.savez_compressed(out_dir / "maps_small.npz", **small)

    full = {}
    for name, (spec, out, inp, radius) in CS.FULL_CASES.items():
        xm, ym = get_map(to_reference(spec), radius=radius, size_input=inp, size_output=out)
        bx, by = CS.buckets(xm), CS.buckets(ym)
        full[f"{name}__sha_bx"] = np.frombuffer(hashlib.sha256(bx.tobytes()).digest(), np.uint8)
        full[f"{name}__sha_by"] = np.frombuffer(hashlib.sha256(by.tobytes()).digest(), np.uint8)
        s = CS.FULL_STRIDE
        full[f"{name}__rows_x"], full[f"{name}__rows_y"] = xm[::s].copy(), ym[::s].copy()
        full[f"{name}__cols_x"], full[f"{name}__cols_y"] = xm[:, ::s].copy(), ym[:, ::s].copy()
        print(f"full  {name} {xm.shape} sha_bx={hashlib.sha256(bx.tobytes()).hexdigest()[:16]}")
        del xm, ym, bx, by
    np.savez_compressed(out_dir / "maps_full.npz", **full)

    # C5: a few (frame, eye) units at reduced size; rotation differs per unit
    c5 = {}
    for frame in (0, 1, 7):
        for eye in (0, 1):
            xm, ym = get_map(to_reference(CS.c5_spec(frame, eye)), radius=96.0, size_input=(192, 192), size_output=(192, 192))
            c5[f"f{frame}_e{eye}__x"], c5[f"f{frame}_e{eye}__y"] = xm, ym
    np.savez_compressed(out_dir / "maps_c5.npz", **c5)

    # get_radius / get_radius_smart on synthetic discs
    rad = {}
    yy, xx = np.mgrid[0:120, 0:200]
    disc = (((xx - 100) ** 2 + (yy - 60) ** 2) <= 50**2)[..., None] * np.array([200, 180, 160], np.uint8)
    disc = disc.astype(np.uint8)
    rad["landscape_img"], rad["landscape_radius"] = disc, np.float64(RT.get_radius(disc))
    port = np.ascontiguousarray(disc.transpose(1, 0, 2))
    rad["portrait_img"], rad["portrait_radius"] = port, np.float64(RT.get_radius(port))
    rng = np.random.default_rng(5)
    noisy = disc.copy()
    noisy[60, 20:40] = rng.integers(0, 30, (20, 3), dtype=np.uint8)  # speckle around the threshold
    rad["noisy_img"], rad["noisy_radius"] = noisy, np.float64(RT.get_radius(noisy))
    rad["thr_radius"] = np.float64(RT.get_radius(noisy, threshold=25))
    full_img = np.full((64, 80, 3), 90, np.uint8)
    try:
        RT.get_radius(full_img)
        rad["noborder_raises"] = np.bool_(False)
    except IndexError:
        rad["noborder_raises"] = np.bool_(True)
    rad["smart_auto"] = np.float64(get_radius_smart("auto", [disc, noisy]))
    rad["smart_max"] = np.float64(get_radius_smart("max", [disc]))
    rad["smart_num"] = np.float64(get_radius_smart(33.5, [disc]))
    np.savez_compressed(out_dir / "radius.npz", **rad)

    # equidistant_to_3d / from_3d (reference tests/test_remapper.py:112-115) + the raw 3-D vectors
    rng = np.random.default_rng(11)
    x, y = rng.random((41, 40)), rng.random((41, 40))
    v = RT.equidistant_to_3d(x, y)
    bx, by = RT.equidistant_from_3d(v)
    np.savez_compressed(out_dir / "equidistant3d.npz", x=x, y=y, v=v, back_x=bx, back_y=by)

    # transform / inverse_transform of single stages on random points (host-logic parity)
    st = {}
    px, py = rng.random((33, 31)) * 1.2 - 0.1, rng.random((33, 31)) * 1.2 - 0.1
    st["px"], st["py"] = px, py
    singles = {
        "zoom": [("zoom", 1.7)],
        "poly": [("poly", [0.1, 0.9, -0.05, 0.01])],
        "equirect": [("equirect_enc", True)],
        "equirect_x": [("equirect_enc", False)],
        "rot": [("rot", CS.ry(0.5))],
        "rectdec": [("rectilinear_dec", 10.0, 36.0)],
    }
    for m in ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]:
        singles[f"fe_{m}"] = [("fisheye_enc", m)]
    for name, spec in singles.items():
        t = to_reference(spec)  # single item -> the bare reference transformer object
        fx, fy = t.transform(px, py)
        st[f"{name}__fwd_x"], st[f"{name}__fwd_y"] = fx, fy
        if name != "poly":
            ix, iy = t.inverse_transform(px, py)
            st[f"{name}__inv_x"], st[f"{name}__inv_y"] = ix, iy
    dn = RT.DenormalizeTransformer(scale=(100.5, 99.0), center=(320, 241))
    st["denorm__fwd_x"], st["denorm__fwd_y"] = dn.transform(px, py)
    st["denorm__inv_x"], st["denorm__inv_y"] = dn.inverse_transform(px * 400, py * 400)
    gx, gy = np.meshgrid(np.arange(31), np.arange(33))
    st["norm__fwd_x"], st["norm__fwd_y"] = RT.NormalizeTransformer().transform(gx, gy)
    st["normmax__fwd_x"], st["normmax__fwd_y"] = RT.NormalizeTransformer(scale="max").transform(gx, gy)
    np.savez_compressed(out_dir / "stages.npz", **st)
    print("done")


def rotation_match_golden() -> None:
    """`rotation_match` (remapper.py:93-143) on seeded point sets.  `as_quat_array` only wraps the
    reference's result (w, x, y, z) -- the placeholder returns the array, so the fixture holds what
    the reference computed; the sign of an eigenvector is arbitrary (q and -q are one rotation)."""
    import vr180_convert.remapper as RR

    RR.as_quat_array = lambda a: np.asarray(a, dtype=float)
    rng = np.random.default_rng(20240619)
    out = {}
    for k, (n, noise) in enumerate([(12, 0.0), (200, 1e-3), (64, 5e-2)]):
        a = rng.normal(size=(n, 3))
        a /= np.linalg.norm(a, axis=-1, keepdims=True)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.05, 1.0)
        m = quat_to_matrix([np.cos(ang / 2), *(np.sin(ang / 2) * axis)])
        b = a @ m.T + noise * rng.normal(size=(n, 3))
        q = np.asarray(RR.rotation_match(a, b), dtype=float)
        out[f"a{k}"], out[f"b{k}"], out[f"q{k}"] = a, b, q
    np.savez_compressed(Path(__file__).resolve().parent / "rotation_match.npz", **out)
    print("rotation_match.npz:", {k: v.shape for k, v in out.items()})


def match_lr_golden() -> None:
    """`match_lr` (remapper.py:251-321) run as the reference has it: matched pixel positions -> unit
    rays.  The function reads its two images with `cv.imread` only for their shape (centre) -- the
    radius is passed as a number -- so the cv2 placeholder gets an `imread` that returns an empty
    image of the shape encoded in the file name; no arithmetic of the path is replaced."""
    import vr180_convert.remapper as RR

    RR.cv.imread = lambda name: np.zeros(tuple(int(t) for t in Path(name).stem.split("x")) + (3,), np.uint8)
    rng = np.random.default_rng(20240620)
    out = {}
    cases = {
        "equidistant": (RT.FisheyeDecoder("equidistant"), (960, 1280), 470.0),
        "stereographic_zoom": (RT.ZoomTransformer(1.2) * RT.FisheyeDecoder("stereographic"), (777, 1033), 380.5),
        "tuple": ((RT.FisheyeDecoder("equisolid"), RT.FisheyeDecoder("rectilinear")), (600, 600), 300.0),
    }
    for name, (dec, (h, w), radius) in cases.items():
        n = 40
        pl = np.stack([rng.uniform(0.2 * w, 0.8 * w, n), rng.uniform(0.2 * h, 0.8 * h, n)], axis=1)
        pr = pl + rng.normal(0, 3.0, (n, 2))
        vl, vr = RR.match_lr(dec, pl, pr, [f"{h}x{w}.png", f"{h}x{w}.png"], radius=radius)
        out[f"{name}_pl"], out[f"{name}_pr"], out[f"{name}_vl"], out[f"{name}_vr"] = pl, pr, np.asarray(vl), np.asarray(vr)
        out[f"{name}_geom"] = np.array([h, w, radius], dtype=float)
    np.savez_compressed(Path(__file__).resolve().parent / "match_lr.npz", **out)
    print("match_lr.npz:", {k: v.shape for k, v in out.items()})


def planar_golden() -> None:
    """get_map of the reference for chainspecs.PLANAR_CASES at full size: SHA-256 of the 1/32-pixel bucket planes, every 256th row and
    column verbatim, the NaN count (FisheyeEncoder("orthographic") / ("equisolid") beyond their domain, transformer.py:370-372)."""
    import warnings

    warnings.simplefilter("ignore")
    out = {}
    for name, (spec, size_out, size_in, radius) in CS.PLANAR_CASES.items():
        xm, ym = get_map(to_reference(spec), radius=radius, size_input=size_in, size_output=size_out)
        bx, by = CS.buckets(xm), CS.buckets(ym)
        out[f"{name}__sha_bx"] = np.frombuffer(hashlib.sha256(bx.tobytes()).digest(), np.uint8)
        out[f"{name}__sha_by"] = np.frombuffer(hashlib.sha256(by.tobytes()).digest(), np.uint8)
        s = CS.FULL_STRIDE
        out[f"{name}__rows_x"], out[f"{name}__rows_y"] = xm[::s].copy(), ym[::s].copy()
        out[f"{name}__cols_x"], out[f"{name}__cols_y"] = xm[:, ::s].copy(), ym[:, ::s].copy()
        out[f"{name}__nan"] = np.int64(np.isnan(xm).sum())
        print(f"planar {name:24s} {xm.shape} nan={int(np.isnan(xm).sum())} sha_bx={hashlib.sha256(bx.tobytes()).hexdigest()[:16]}")
    np.savez_compressed(Path(__file__).resolve().parent / "maps_planar.npz", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "planar":
        planar_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "rotation_match":
        rotation_match_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "match_lr":
        match_lr_golden()
    else:
        main()
        rotation_match_golden()
        match_lr_golden()
        planar_golden()
