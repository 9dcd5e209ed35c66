"""ctypes front end of the C oracle (``oracle/vr180_oracle.c``) -- TEST INFRASTRUCTURE ONLY.

Chains are described by a neutral *spec* (a list of tuples, see ``chain_from_spec``) so that the
oracle does not depend on the product's transformer classes or on its lowering code:

* ``tests/golden/make_golden.py`` turns a spec into REFERENCE objects (in the build container),
* this module turns it into the ``v1c_chain`` POD the C oracle interprets,
* the tests turn it into product transformer objects and let the product lower them.

Reference citations: get_map wrapping ``remapper.py:50-58``; quaternion -> matrix follows
numpy-quaternion ``as_rotation_matrix`` as restated in SURVEY.md Appendix B (parity unpinned: the
package is not installed here and the reference's tests only use it self-consistently).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from pathlib import Path
from typing import Any, Sequence

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libvr180oracle.so"

MAX_OPS = 16
MAX_PARAMS = 16

OP_NORMALIZE, OP_DENORMALIZE, OP_DENORMALIZE_INV, OP_ZOOM, OP_ZOOM_INV = 1, 2, 3, 4, 5
OP_EQUIRECT_ENC, OP_EQUIRECT_DEC, OP_RADIAL, OP_ROTATE = 6, 7, 8, 9
_FISHEYE = ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]
RAD_POLYNOMIAL, RAD_RECTDEC_FWD, RAD_RECTDEC_INV = 11, 12, 13

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101, BORDER_TRANSPARENT = range(6)


class Op(C.Structure):
    _fields_ = [
        ("opcode", C.c_int32),
        ("iparam", C.c_int32),
        ("nparam", C.c_int32),
        ("reserved", C.c_int32),
        ("p", C.c_double * MAX_PARAMS),
    ]


class Chain(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("reserved", C.c_int32), ("ops", Op * MAX_OPS)]


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (``make -C oracle``) if the .so is missing or stale."""
    src = _HERE / "vr180_oracle.c"
    hdr = _HERE.parent / "include" / "vr180_remap.h"
    stale = (
        force
        or not _LIB_PATH.exists()
        or _LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)
    )
    if stale:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libvr180oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        L.orc_get_map.argtypes = [C.POINTER(Chain), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_get_map_f64.argtypes = [C.POINTER(Chain), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_remap.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int,
            C.c_void_p, C.c_int, C.c_int, C.c_int64,
            C.c_void_p, C.c_void_p, C.c_int64,
            C.c_int, C.c_int, C.c_void_p,
        ]
        L.orc_build_itab.argtypes = [C.c_int, C.c_void_p]
        L.orc_get_radius.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.orc_set_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ----------------------------------------------------------------------------- chain building
def quat_to_matrix(q: Sequence[float]) -> np.ndarray:
    """(w, x, y, z) -> 3x3, non-unit quaternions normalised."""
    w, x, y, z = (float(v) for v in q)
    n = w * w + x * x + y * y + z * z
    if n == 0.0:
        raise ZeroDivisionError("zero quaternion")
    if abs(n - 1.0) < np.finfo(float).eps:
        return np.array(
            [
                [1 - 2 * (y**2 + z**2), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                [2 * (x * y + z * w), 1 - 2 * (x**2 + z**2), 2 * (y * z - x * w)],
                [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x**2 + y**2)],
            ]
        )
    return np.array(
        [
            [1 - 2 * (y**2 + z**2) / n, 2 * (x * y - z * w) / n, 2 * (x * z + y * w) / n],
            [2 * (x * y + z * w) / n, 1 - 2 * (x**2 + z**2) / n, 2 * (y * z - x * w) / n],
            [2 * (x * z - y * w) / n, 2 * (y * z + x * w) / n, 1 - 2 * (x**2 + y**2) / n],
        ]
    )


def _op(opcode: int, iparam: int = 0, params: Sequence[float] = ()) -> Op:
    if len(params) > MAX_PARAMS:
        raise ValueError("too many parameters for one op")
    o = Op()
    o.opcode, o.iparam, o.nparam = opcode, iparam, len(params)
    for i, v in enumerate(params):
        o.p[i] = float(v)
    return o


def _spec_item(item: tuple, inverse: bool = False) -> Op:
    kind, *a = item
    if kind == "inverse":
        return _spec_item(a[0], not inverse)
    if kind == "equirect_enc":
        lat_y = int(bool(a[0])) if a else 1
        return _op(OP_EQUIRECT_DEC if inverse else OP_EQUIRECT_ENC, lat_y)
    if kind == "equirect_dec":
        lat_y = int(bool(a[0])) if a else 1
        return _op(OP_EQUIRECT_ENC if inverse else OP_EQUIRECT_DEC, lat_y)
    if kind in ("fisheye_enc", "fisheye_dec"):
        fwd = (kind == "fisheye_enc") != inverse
        return _op(OP_RADIAL, _FISHEYE.index(a[0]) + (1 if fwd else 6))
    if kind == "poly":
        if inverse:
            raise NotImplementedError("PolynomialScaler does not support inverse transform.")
        return _op(OP_RADIAL, RAD_POLYNOMIAL, a[0])
    if kind == "zoom":
        return _op(OP_ZOOM_INV if inverse else OP_ZOOM, 0, [a[0]])
    if kind == "rot":
        return _op(OP_ROTATE, 0, np.asarray(a[0], dtype=float).reshape(9))
    if kind == "rot_quat":
        return _op(OP_ROTATE, 0, quat_to_matrix(a[0]).reshape(9))
    if kind == "rectilinear_dec":
        factor = 2 * float(a[0]) / float(a[1])
        return _op(OP_RADIAL, RAD_RECTDEC_INV if inverse else RAD_RECTDEC_FWD, [factor])
    raise ValueError(f"unknown spec item {item!r}")


def chain_from_spec(spec: Sequence[tuple], *, radius: float, size_input: tuple[int, int], size_output: tuple[int, int]) -> Chain:
    """get_map's wrapping, remapper.py:50-57: Normalize() * T * Denormalize((r, r), (W_in//2, H_in//2)).

    ``size_input`` is (H_in, W_in), ``size_output`` is (W, H) -- the reference's conventions.
    """
    W, H = size_output
    ops = [_op(OP_NORMALIZE, 0, [W / 2, H / 2, min(W, H)])]
    ops += [_spec_item(it) for it in spec]
    ops.append(_op(OP_DENORMALIZE, 0, [radius, radius, size_input[1] // 2, size_input[0] // 2]))
    if len(ops) > MAX_OPS:
        raise ValueError("chain too long")
    ch = Chain()
    ch.n_ops = len(ops)
    for i, o in enumerate(ops):
        ch.ops[i] = o
    return ch


# ----------------------------------------------------------------------------- entry points
def get_map(spec, *, radius: float, size_input: tuple[int, int], size_output: tuple[int, int] = (2048, 2048), f64: bool = False,
            out: tuple[np.ndarray, np.ndarray] | None = None):
    ch = spec if isinstance(spec, Chain) else chain_from_spec(spec, radius=radius, size_input=size_input, size_output=size_output)
    W, H = size_output
    dt = np.float64 if f64 else np.float32
    if out is not None:  # reuse the caller's (already touched) buffers
        xm, ym = out
        assert xm.shape == ym.shape == (H, W) and xm.dtype == ym.dtype == dt and xm.flags.c_contiguous and ym.flags.c_contiguous
    else:
        xm = np.empty((H, W), dt)
        ym = np.empty((H, W), dt)
    fn = lib().orc_get_map_f64 if f64 else lib().orc_get_map
    rc = fn(C.byref(ch), W, H, xm.ctypes.data, ym.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"orc_get_map failed ({rc})")
    return xm, ym


def border_scalar(value: Any) -> np.ndarray:
    """Python borderValue -> saturated uint8[4] the way cv2 builds its Scalar: an int sets only
    component 0; a tuple sets the leading components."""
    vals = [value] if np.isscalar(value) else list(value)
    out = np.zeros(4, np.uint8)
    for i, v in enumerate(vals[:4]):
        out[i] = int(min(255, max(0, round(float(v)))))  # saturate_cast<uchar>(double): round half even
    return out


def remap(src: np.ndarray, xmap: np.ndarray, ymap: np.ndarray, interpolation: int = INTER_LINEAR,
          border_mode: int = BORDER_CONSTANT, border_value: Any = 0, dst: np.ndarray | None = None) -> np.ndarray:
    """cv2.remap(src, xmap, ymap, interpolation=, borderMode=, borderValue=) for uint8 images."""
    if src.dtype != np.uint8:
        raise TypeError("uint8 only")
    s = src if src.ndim == 3 else src[..., None]
    if s.strides[2] != 1 or s.strides[1] != s.shape[2]:
        s = np.ascontiguousarray(s)
    xmap = np.ascontiguousarray(xmap, np.float32)
    ymap = np.ascontiguousarray(ymap, np.float32)
    H, W = xmap.shape
    cn = s.shape[2]
    if dst is None:
        dst = np.zeros((H, W, cn), np.uint8)
    bv = border_scalar(border_value)
    rc = lib().orc_remap(
        s.ctypes.data, s.shape[0], s.shape[1], s.strides[0], cn,
        dst.ctypes.data, H, W, dst.strides[0],
        xmap.ctypes.data, ymap.ctypes.data, W,
        int(interpolation), int(border_mode), bv.ctypes.data,
    )
    if rc != 0:
        raise RuntimeError(f"orc_remap failed ({rc})")
    return dst if src.ndim == 3 else dst[..., 0]


def build_itab(interpolation: int) -> np.ndarray:
    k = {INTER_LINEAR: 2, INTER_CUBIC: 4, INTER_LANCZOS4: 8}[interpolation]
    buf = np.zeros(1024 * k * k + 8, np.int16)
    lib().orc_build_itab(int(interpolation), buf.ctypes.data)
    return buf[: 1024 * k * k].reshape(1024, k, k)


def get_radius(img: np.ndarray, threshold: int = 10) -> float:
    """get_radius(), transformer.py:108-140; raises IndexError like the reference."""
    img = np.ascontiguousarray(img)
    r = C.c_double()
    rc = lib().orc_get_radius(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], img.shape[2], threshold, C.byref(r))
    if rc == -2:
        raise IndexError("index 0 is out of bounds for axis 0 with size 0")
    if rc != 0:
        raise RuntimeError("orc_get_radius failed")
    return r.value


def get_radius_smart(radius, images) -> float:
    """get_radius_smart(), remapper.py:62-90."""
    if isinstance(radius, str) and radius == "auto":
        return max(get_radius(im) for im in images)
    if isinstance(radius, str) and radius == "max":
        return min(images[0].shape[0] / 2, images[0].shape[1] / 2)
    return radius


def apply(spec, images: Sequence[np.ndarray], *, size_output=(2048, 2048), interpolation=INTER_LANCZOS4,
          border_mode=BORDER_CONSTANT, border_value=0, radius="auto") -> list[np.ndarray]:
    """apply(), remapper.py:365-398 on in-memory images: one map, remap each image."""
    r = get_radius_smart(radius, images)
    xm, ym = get_map(spec, radius=r, size_input=(images[0].shape[0], images[0].shape[1]), size_output=size_output)
    return [remap(im, xm, ym, interpolation, border_mode, border_value) for im in images]


def apply_lr(spec, left: np.ndarray, right: np.ndarray, **kw) -> np.ndarray:
    """apply_lr(), remapper.py:460-484,517-518 (merge=False): per-eye or shared map, SBS concat."""
    if isinstance(spec, tuple):
        ims = [apply(s, [im], **kw)[0] for s, im in zip(spec, (left, right))]
    else:
        ims = apply(spec, [left, right], **kw)
    return np.concatenate(ims, axis=1)


def anaglyph(left: np.ndarray, right: np.ndarray) -> np.ndarray:
    """``merge=True`` of apply_lr, reference remapper.py:485-497, as the NumPy expression the
    reference evaluates (float64): per-eye channel mean times a colour, summed, / 255.  The
    cv.putText labels (:498-516) are not part of it."""
    colors = [(0, 128, 255), (255, 128, 0)]
    combine = np.mean(left, axis=-1)[..., None] * np.array(colors[0]).reshape([1] * (left.ndim - 1) + [3]) + (
        np.mean(right, axis=-1)[..., None] * np.array(colors[1]).reshape([1] * (right.ndim - 1) + [3])
    )
    combine /= 255
    return combine
