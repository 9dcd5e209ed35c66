"""NumPy restatement of the reference's coordinate chain -- TEST / BASELINE INFRASTRUCTURE ONLY.

Array-at-a-time float64 ufunc passes over a meshgrid, like the reference's get_map()
(remapper.py:23-59 driving transformer.py:93-98): this is the "reference-equivalent" CPU path whose
speed bench.py reports beside the C port (SURVEY.md 8d, row "CPU baseline (A)").  Functional
style (a spec interpreter), no classes; every stage cites the reference lines it follows.
Pinned by tests/test_oracle_golden.py::test_numpy_chain_vs_goldens.
"""
from __future__ import annotations

import numpy as np

HALF_PI = np.pi / 2


def _to_3d(x, y):  # equidistant_to_3d, transformer.py:502-507
    phi = np.arctan2(x, y)
    theta = np.sqrt(x**2 + y**2)
    st = np.sin(theta)
    return st * np.sin(phi), st * np.cos(phi), np.cos(theta)


def _from_3d(vx, vy, vz):  # equidistant_from_3d, transformer.py:526-530
    theta = np.arccos(vz)
    phi = np.arctan2(vx, vy)
    return theta * np.sin(phi), theta * np.cos(phi)


def _polar(x, y, fn):  # PolarRollTransformer.transform, transformer.py:268-276
    theta = np.sqrt(x**2 + y**2)
    roll = np.arctan2(y, x)
    theta = fn(theta)
    return theta * np.cos(roll), theta * np.sin(roll)


_ENC = {  # FisheyeEncoder.transform_polar, transformer.py:359-372
    "rectilinear": np.arctan,
    "stereographic": lambda t: 2 * np.arctan(t),
    "equidistant": lambda t: t * HALF_PI,
    "equisolid": lambda t: 2 * np.arcsin(t / np.sqrt(2)),
    "orthographic": np.arcsin,
}
_DEC = {  # FisheyeEncoder.inverse_transform_polar, transformer.py:379-392
    "rectilinear": np.tan,
    "stereographic": lambda t: 2 * np.tan(t / 2),
    "equidistant": lambda t: t / HALF_PI,
    "equisolid": lambda t: np.sqrt(2) * np.sin(t / 2),
    "orthographic": np.sin,
}


def _stage(item, x, y, inverse=False):
    kind, *a = item
    if kind == "inverse":
        return _stage(a[0], x, y, not inverse)
    if kind in ("equirect_enc", "equirect_dec"):
        lat_y = bool(a[0]) if a else True
        fwd = (kind == "equirect_enc") != inverse
        if fwd:  # EquirectangularEncoder.transform, transformer.py:540-568
            lat, lon = ((y, x) if lat_y else (x, y))
            lat, lon = lat * HALF_PI, lon * HALF_PI
            across, along, forward = np.cos(lat) * np.sin(lon), np.sin(lat), np.cos(lat) * np.cos(lon)
            return _from_3d(across, along, forward) if lat_y else _from_3d(along, across, forward)
        vx, vy, vz = _to_3d(x, y)  # .inverse_transform, transformer.py:570-584
        if lat_y:
            return np.arctan2(vx, vz) / HALF_PI, np.arcsin(vy) / HALF_PI
        return np.arcsin(vx) / HALF_PI, np.arctan2(vy, vz) / HALF_PI
    if kind in ("fisheye_enc", "fisheye_dec"):
        fwd = (kind == "fisheye_enc") != inverse
        return _polar(x, y, (_ENC if fwd else _DEC)[a[0]])
    if kind == "poly":  # PolynomialScaler.transform_polar, transformer.py:448-451
        if inverse:
            raise NotImplementedError("PolynomialScaler does not support inverse transform.")
        return _polar(x, y, lambda t: np.polyval(np.flip(np.asarray(a[0], float)), t))
    if kind == "zoom":  # ZoomTransformer, transformer.py:468-480
        return (x * a[0], y * a[0]) if inverse else (x / a[0], y / a[0])
    if kind in ("rot", "rot_quat"):  # Euclidean3DTransformer.transform, transformer.py:651-657
        from .oracle import quat_to_matrix

        m = np.asarray(a[0], float).reshape(3, 3) if kind == "rot" else quat_to_matrix(a[0])
        vx, vy, vz = _to_3d(x, y)
        r = [m[k, 0] * vx + m[k, 1] * vy + m[k, 2] * vz for k in range(3)]
        return _from_3d(*r)
    if kind == "rectilinear_dec":  # RectilinearDecoder, transformer.py:338-347
        factor = 2 * float(a[0]) / float(a[1])
        return _polar(x, y, (lambda t: np.arctan(t / factor)) if inverse else (lambda t: np.tan(t) * factor))
    raise ValueError(f"unknown spec item {item!r}")


def get_map(spec, *, radius, size_input, size_output=(2048, 2048)):
    """get_map(), remapper.py:50-58, for a chain spec (see oracle.chain_from_spec)."""
    W, H = size_output
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    s = min(W, H)
    x, y = (x - W / 2) / s * 2, (y - H / 2) / s * 2  # NormalizeTransformer.transform, :162-163
    for item in spec:
        x, y = _stage(item, x, y)
    x = x * radius + size_input[1] // 2  # DenormalizeTransformer.transform, :202-203
    y = y * radius + size_input[0] // 2
    return x.astype(np.float32), y.astype(np.float32)
