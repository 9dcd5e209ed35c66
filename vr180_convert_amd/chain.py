"""Transformer algebra of the remap engine and its lowering to the C-ABI op list.

Host-side mirror of the reference's ``vr180_convert/transformer.py`` public surface: same class
names, constructor arguments, ``transform`` / ``inverse_transform`` semantics, ``*`` composition
and error behaviour, so code (and tests) written against the reference read the same here.  What
is new is :meth:`TransformerBase.lower`: every built-in stage knows how to describe itself as
``v1c_op`` records (``include/vr180_remap.h``) so that the whole chain can be evaluated per output
pixel inside one HIP kernel instead of as ~50 full-image NumPy passes.

The NumPy ``transform`` methods below are the public coordinate API (calibration code calls
``inverse_transform`` on a handful of points; users call ``transform`` on their own subclasses,
reference README.md:204-219).  They are NOT a fallback of the image path: ``apply`` / ``apply_lr``
lower the chain and run it on the GPU, and refuse to run without the HIP library.
"""
from __future__ import annotations

import warnings
from abc import ABCMeta, abstractmethod
from dataclasses import dataclass, field
from typing import Any, Generic, Literal, Sequence, TypeVar

import numpy as np
from numpy.typing import NDArray

from . import _abi
from .quat import as_rotation_matrix

_HALF_PI = np.pi / 2
_MAPPINGS = ("rectilinear", "stereographic", "equidistant", "equisolid", "orthographic")


class NotLowerable(Exception):
    """The stage has no ``v1c_op`` form (user subclass, unsupported parameters)."""


class TransformerBase(metaclass=ABCMeta):
    """Base class for transformers (reference transformer.py:14-81)."""

    @abstractmethod
    def transform(self, x: NDArray, y: NDArray, **kwargs: Any) -> tuple[NDArray, NDArray]:
        """Map x (left-right) / y (up-down) coordinates forward."""

    @abstractmethod
    def inverse_transform(self, x: NDArray, y: NDArray, **kwargs: Any) -> tuple[NDArray, NDArray]:
        """Map coordinates backward."""

    # ---- the estimator surface the reference inherits from sklearn.base.BaseEstimator / TransformerMixin (transformer.py:11-18) ----
    # (restated here, without the dependency: parameters = the constructor's arguments, nested ones as <name>__<sub>)
    @classmethod
    def _param_names(cls) -> list[str]:
        import inspect

        if cls.__init__ is object.__init__:
            return []
        sig = inspect.signature(cls.__init__)
        return sorted(p.name for p in sig.parameters.values() if p.name != "self" and p.kind not in (p.VAR_KEYWORD, p.VAR_POSITIONAL))

    def get_params(self, deep: bool = True) -> dict:
        """Constructor parameters by name; with ``deep`` also those of parameter values that are estimators themselves."""
        out: dict = {}
        for name in self._param_names():
            value = getattr(self, name)
            if deep and hasattr(value, "get_params") and not isinstance(value, type):
                out.update((f"{name}__{k}", v) for k, v in value.get_params().items())
            out[name] = value
        return out

    def set_params(self, **params: Any) -> "TransformerBase":
        """Set constructor parameters (``name`` or nested ``name__sub``); unknown names raise ValueError.  Returns self."""
        valid = self.get_params(deep=True)
        nested: dict = {}
        for key, value in params.items():
            name, delim, sub = key.partition("__")
            if name not in valid:
                raise ValueError(f"Invalid parameter {name!r} for estimator {self}. Valid parameters are: {self._param_names()!r}.")
            if delim:
                nested.setdefault(name, {})[sub] = value
            else:
                setattr(self, name, value)
                valid[name] = value
        for name, sub in nested.items():
            valid[name].set_params(**sub)
        return self

    def fit(self, *args: Any, **kwargs: Any) -> "TransformerBase":
        """Nothing to learn (the reference leaves ``fit`` commented out, transformer.py:21-22); present so that ``fit_transform`` works."""
        return self

    def fit_transform(self, x: NDArray, y: NDArray, **kwargs: Any) -> tuple[NDArray, NDArray]:
        return self.fit(x, y, **kwargs).transform(x, y, **kwargs)

    def __mul__(self, other: "TransformerBase") -> "MultiTransformer":
        # flattening composition, left operand applied first (transformer.py:71-81)
        left = self.transformers if isinstance(self, MultiTransformer) else [self]
        right = other.transformers if isinstance(other, MultiTransformer) else [other]
        return MultiTransformer(transformers=[*left, *right])

    # ---- lowering ---------------------------------------------------------------------------
    def lower(self, out_shape: tuple[int, int]) -> list[_abi.Op]:
        """Ops equivalent to ``transform`` on an (H, W) = ``out_shape`` grid."""
        raise NotLowerable(type(self).__name__)

    def lower_inverse(self, out_shape: tuple[int, int]) -> list[_abi.Op]:
        """Ops equivalent to ``inverse_transform``."""
        raise NotLowerable(type(self).__name__)

    def _is_builtin(self, cls: type) -> bool:
        # a user subclass may override anything: only exact built-in types are lowered
        return type(self) is cls


T = TypeVar("T", bound=TransformerBase)


@dataclass
class MultiTransformer(TransformerBase):
    """Applies several transformers in sequence (transformer.py:87-105)."""

    transformers: list

    def transform(self, x, y, **kwargs):
        for t in self.transformers:
            x, y = t.transform(x, y, **kwargs)
        return x, y

    def inverse_transform(self, x, y, **kwargs):
        for t in reversed(self.transformers):
            x, y = t.inverse_transform(x, y, **kwargs)
        return x, y

    def lower(self, out_shape):
        ops: list[_abi.Op] = []
        for t in self.transformers:
            ops += t.lower(out_shape)
        return ops

    def lower_inverse(self, out_shape):
        ops: list[_abi.Op] = []
        for t in reversed(self.transformers):
            ops += t.lower_inverse(out_shape)
        return ops


def get_radius(input: NDArray, *, threshold: int = 10) -> float:
    """Estimate the fisheye circle radius from the centre row / column (transformer.py:108-140).

    Sign quirk and ``IndexError`` on a missing black border are the reference's.
    """
    height, width = input.shape[:2]
    line = input[height // 2, :, :] if width > height else input[:, width // 2, :]
    black = (np.mean(line, axis=-1) < threshold).astype(int)
    step = np.diff(black)
    first_rise = np.where(step == 1)[0][0]
    last_fall = np.where(step == -1)[0][-1]
    return (last_fall - first_rise) / 2


@dataclass
class NormalizeTransformer(TransformerBase):
    """Normalize pixel coordinates to [-1, 1] (transformer.py:143-177)."""

    center: tuple | None = None
    scale: Any = None  # (sx, sy) | "min" | "max" | None

    def _params(self, x):
        center = self.center or (x.shape[1] / 2, x.shape[0] / 2)
        if self.scale in ["min", None]:
            scale = min(x.shape[1], x.shape[0])
        elif self.scale == "max":
            scale = max(x.shape[1], x.shape[0])
        else:
            scale = self.scale
        return center, scale

    def transform(self, x, y, **kwargs):
        center, scale = self._params(x)
        return (x - center[0]) / scale * 2, (y - center[1]) / scale * 2

    def inverse_transform(self, x, y, **kwargs):
        center, scale = self._params(x)
        # the reference indexes the scale here (only a tuple scale works), transformer.py:175-176
        return x * scale[0] + center[0], y * scale[1] + center[1]

    def lower(self, out_shape):
        if not self._is_builtin(NormalizeTransformer):
            raise NotLowerable("NormalizeTransformer subclass")
        h, w = out_shape
        center = self.center or (w / 2, h / 2)
        if self.scale in ["min", None]:
            scale = min(w, h)
        elif self.scale == "max":
            scale = max(w, h)
        elif np.isscalar(self.scale):
            scale = self.scale
        else:
            raise NotLowerable("NormalizeTransformer with a tuple scale")
        return [_abi.op(_abi.OP_NORMALIZE, 0, [center[0], center[1], scale])]


@dataclass
class DenormalizeTransformer(TransformerBase):
    """[-1, 1] -> source pixel coordinates (transformer.py:188-213)."""

    scale: tuple
    center: tuple

    def transform(self, x, y, **kwargs):
        return x * self.scale[0] + self.center[0], y * self.scale[1] + self.center[1]

    def inverse_transform(self, x, y, **kwargs):
        return (x - self.center[0]) / self.scale[0], (y - self.center[1]) / self.scale[1]

    def _p(self):
        return [self.scale[0], self.scale[1], self.center[0], self.center[1]]

    def lower(self, out_shape):
        if not self._is_builtin(DenormalizeTransformer):
            raise NotLowerable("DenormalizeTransformer subclass")
        return [_abi.op(_abi.OP_DENORMALIZE, 0, self._p())]

    def lower_inverse(self, out_shape):
        if not self._is_builtin(DenormalizeTransformer):
            raise NotLowerable("DenormalizeTransformer subclass")
        return [_abi.op(_abi.OP_DENORMALIZE_INV, 0, self._p())]


class PolarRollTransformer(TransformerBase):
    """Transform expressed on (theta, roll) polar coordinates (transformer.py:216-286)."""

    @abstractmethod
    def transform_polar(self, theta: NDArray, roll: NDArray, **kwargs: Any) -> tuple[NDArray, NDArray]:
        """theta: distance / angle from the centre; roll: angle around it."""

    @abstractmethod
    def inverse_transform_polar(self, theta: NDArray, roll: NDArray, **kwargs: Any) -> tuple[NDArray, NDArray]:
        """Inverse of :meth:`transform_polar`."""

    @staticmethod
    def _through_polar(fn, x, y, kwargs):
        theta = np.sqrt(x**2 + y**2)
        roll = np.arctan2(y, x)
        theta, roll = fn(theta, roll, **kwargs)
        return theta * np.cos(roll), theta * np.sin(roll)

    def transform(self, x, y, **kwargs):
        return self._through_polar(self.transform_polar, x, y, kwargs)

    def inverse_transform(self, x, y, **kwargs):
        return self._through_polar(self.inverse_transform_polar, x, y, kwargs)


_SENSOR_WIDTHS_MM = {
    # https://en.wikipedia.org/wiki/Image_sensor_format#Table_of_sensor_formats_and_sizes
    "35mm": 36.0, "APS-H": 27.90, "APS-C": 23.6, "APS-C-Canon": 22.30, "MFT": 17.30,
    "1": 13.20, "1/1.12": 11.43, "1/1.2": 10.67, "1/1.33": 9.6, "1/1.6": 8.08,
    "1/1.7": 7.60, "1/1.8": 7.18, "1/2": 6.40, "1/2.3": 6.17,
}


@dataclass
class RectilinearDecoder(PolarRollTransformer):
    """theta -> tan(theta) * 2f / sensor_width (transformer.py:289-347)."""

    focal_length: float
    sensor_width: Any = "35mm"  # named format, or width in mm

    @property
    def sensor_width_mm(self) -> float:
        if self.sensor_width in ["35mm", "APS-C", "1/2.3"]:
            warnings.warn(
                "Sensor size may vary by about 0.2 mm depending on the camera model. "
                "To get very accurate results, consider setting the sensor width in mm manually.",
                UserWarning,
                stacklevel=2,
            )
        if isinstance(self.sensor_width, str):
            return _SENSOR_WIDTHS_MM[self.sensor_width]
        return self.sensor_width

    @property
    def factor(self) -> float:
        return 2 * self.focal_length / self.sensor_width_mm

    def transform_polar(self, theta, roll, **kwargs):
        return np.tan(theta) * self.factor, roll

    def inverse_transform_polar(self, theta, roll, **kwargs):
        return np.arctan(theta / self.factor), roll

    def lower(self, out_shape):
        if not self._is_builtin(RectilinearDecoder):
            raise NotLowerable("RectilinearDecoder subclass")
        return [_abi.op(_abi.OP_RADIAL, _abi.RAD_RECTDEC_FWD, [self.factor])]

    def lower_inverse(self, out_shape):
        if not self._is_builtin(RectilinearDecoder):
            raise NotLowerable("RectilinearDecoder subclass")
        return [_abi.op(_abi.OP_RADIAL, _abi.RAD_RECTDEC_INV, [self.factor])]


def _unknown_mapping(mapping_type) -> ValueError:
    return ValueError(
        f"Unknown mapping type: {mapping_type}, "
        "should be one of 'rectilinear', 'stereographic', 'equidistant', 'equisolid', 'orthographic'."
    )


@dataclass
class FisheyeEncoder(PolarRollTransformer):
    """Fisheye projection models (transformer.py:350-397)."""

    mapping_type: str

    def transform_polar(self, theta, roll, **kwargs):
        """[-1, 1] -> [-pi/2, pi/2]."""
        m = self.mapping_type
        if m == "rectilinear":
            return np.arctan(theta), roll
        if m == "stereographic":
            return 2 * np.arctan(theta), roll
        if m == "equidistant":
            return theta * _HALF_PI, roll
        if m == "equisolid":
            return 2 * np.arcsin(theta / np.sqrt(2)), roll
        if m == "orthographic":
            return np.arcsin(theta), roll
        raise _unknown_mapping(m)

    def inverse_transform_polar(self, theta, roll, **kwargs):
        """[-pi/2, pi/2] -> [-1, 1]."""
        m = self.mapping_type
        if m == "rectilinear":
            return np.tan(theta), roll
        if m == "stereographic":
            return 2 * np.tan(theta / 2), roll
        if m == "equidistant":
            return theta / _HALF_PI, roll
        if m == "equisolid":
            return np.sqrt(2) * np.sin(theta / 2), roll
        if m == "orthographic":
            return np.sin(theta), roll
        raise _unknown_mapping(m)

    def _kind(self, base: int) -> int:
        if self.mapping_type not in _MAPPINGS:
            raise _unknown_mapping(self.mapping_type)
        return base + _MAPPINGS.index(self.mapping_type)

    def lower(self, out_shape):
        if not self._is_builtin(FisheyeEncoder):
            raise NotLowerable("FisheyeEncoder subclass")
        return [_abi.op(_abi.OP_RADIAL, self._kind(_abi.RAD_ENC_RECTILINEAR))]

    def lower_inverse(self, out_shape):
        if not self._is_builtin(FisheyeEncoder):
            raise NotLowerable("FisheyeEncoder subclass")
        return [_abi.op(_abi.OP_RADIAL, self._kind(_abi.RAD_DEC_RECTILINEAR))]


@dataclass
class InverseTransformer(TransformerBase, Generic[T]):
    """Swaps transform() and inverse_transform() of the wrapped transformer (transformer.py:400-415)."""

    transformer: Any

    def transform(self, x, y, **kwargs):
        return self.transformer.inverse_transform(x, y, **kwargs)

    def inverse_transform(self, x, y, **kwargs):
        return self.transformer.transform(x, y, **kwargs)

    def lower(self, out_shape):
        return self.transformer.lower_inverse(out_shape)

    def lower_inverse(self, out_shape):
        return self.transformer.lower(out_shape)


def FisheyeDecoder(mapping_type: str) -> InverseTransformer:
    """Decodes a fisheye image: ``InverseTransformer(FisheyeEncoder(...))`` (transformer.py:418-437)."""
    return InverseTransformer(FisheyeEncoder(mapping_type))


@dataclass
class PolynomialScaler(PolarRollTransformer):
    """theta -> polynomial(theta); ``coefs_reverse`` lowest order first (transformer.py:440-458)."""

    coefs_reverse: Sequence[float] = field(default_factory=lambda: [0, 1])

    def transform_polar(self, theta, roll, **kwargs):
        return np.polyval(np.flip(self.coefs_reverse), theta), roll

    def inverse_transform_polar(self, theta, roll, **kwargs):
        raise NotImplementedError("PolynomialScaler does not support inverse transform.")

    def lower(self, out_shape):
        if not self._is_builtin(PolynomialScaler):
            raise NotLowerable("PolynomialScaler subclass")
        coefs = [float(c) for c in self.coefs_reverse]
        if len(coefs) > _abi.MAX_PARAMS:
            raise NotLowerable("polynomial with more than %d coefficients" % _abi.MAX_PARAMS)
        return [_abi.op(_abi.OP_RADIAL, _abi.RAD_POLYNOMIAL, coefs)]

    def lower_inverse(self, out_shape):
        raise NotImplementedError("PolynomialScaler does not support inverse transform.")


@dataclass
class ZoomTransformer(TransformerBase):
    """x / scale, y / scale (transformer.py:461-480)."""

    scale: float

    def transform(self, x, y, **kwargs):
        return x / self.scale, y / self.scale

    def inverse_transform(self, x, y, **kwargs):
        return x * self.scale, y * self.scale

    def lower(self, out_shape):
        if not self._is_builtin(ZoomTransformer):
            raise NotLowerable("ZoomTransformer subclass")
        return [_abi.op(_abi.OP_ZOOM, 0, [self.scale])]

    def lower_inverse(self, out_shape):
        if not self._is_builtin(ZoomTransformer):
            raise NotLowerable("ZoomTransformer subclass")
        return [_abi.op(_abi.OP_ZOOM_INV, 0, [self.scale])]


def equidistant_to_3d(x: NDArray, y: NDArray) -> NDArray:
    """Equidistant-fisheye plane point -> unit vector; z forward, x right, y up (transformer.py:483-508)."""
    phi = np.arctan2(x, y)
    theta = np.sqrt(x**2 + y**2)
    sin_t = np.sin(theta)
    return np.stack([sin_t * np.sin(phi), sin_t * np.cos(phi), np.cos(theta)], axis=-1)


def equidistant_from_3d(v: NDArray) -> tuple[NDArray, NDArray]:
    """Unit vector -> equidistant-fisheye plane point (transformer.py:511-530)."""
    theta = np.arccos(v[..., 2])
    phi = np.arctan2(v[..., 0], v[..., 1])
    return theta * np.sin(phi), theta * np.cos(phi)


@dataclass
class EquirectangularEncoder(TransformerBase):
    """Longitude / latitude plane -> equidistant-fisheye plane (transformer.py:533-584)."""

    is_latitude_y: bool = True

    def transform(self, x, y, **kwargs):
        lat, lon = (y, x) if self.is_latitude_y else (x, y)
        lat = lat * _HALF_PI
        lon = lon * _HALF_PI
        across = np.cos(lat) * np.sin(lon)
        along = np.sin(lat)
        forward = np.cos(lat) * np.cos(lon)
        parts = [across, along, forward] if self.is_latitude_y else [along, across, forward]
        return equidistant_from_3d(np.stack(parts, axis=-1))

    def inverse_transform(self, x, y, **kwargs):
        v = equidistant_to_3d(x, y)
        if self.is_latitude_y:
            lat = np.arcsin(v[..., 1])
            lon = np.arctan2(v[..., 0], v[..., 2])
            return lon / _HALF_PI, lat / _HALF_PI
        lat = np.arcsin(v[..., 0])
        lon = np.arctan2(v[..., 1], v[..., 2])
        return lat / _HALF_PI, lon / _HALF_PI

    def lower(self, out_shape):
        if not self._is_builtin(EquirectangularEncoder):
            raise NotLowerable("EquirectangularEncoder subclass")
        return [_abi.op(_abi.OP_EQUIRECT_ENC, int(bool(self.is_latitude_y)))]

    def lower_inverse(self, out_shape):
        if not self._is_builtin(EquirectangularEncoder):
            raise NotLowerable("EquirectangularEncoder subclass")
        return [_abi.op(_abi.OP_EQUIRECT_DEC, int(bool(self.is_latitude_y)))]


def EquirectangularDecoder(is_latitude_y: bool = True) -> InverseTransformer:
    """``InverseTransformer(EquirectangularEncoder(...))`` (transformer.py:587-604)."""
    return InverseTransformer(EquirectangularEncoder(is_latitude_y))


class Euclidean3DTransformer(TransformerBase):
    """Transform applied to the 3-D unit vector (transformer.py:607-665)."""

    @abstractmethod
    def transform_v(self, v: NDArray) -> NDArray:
        """Transform unit vectors (last axis = xyz)."""

    @abstractmethod
    def inverse_transform_v(self, v: NDArray) -> NDArray:
        """Inverse of :meth:`transform_v`."""

    def transform(self, x, y, **kwargs):
        return equidistant_from_3d(self.transform_v(equidistant_to_3d(x, y)))

    def inverse_transform(self, x, y, **kwargs):
        # sic: the reference applies transform_v here as well (transformer.py:659-665)
        return equidistant_from_3d(self.transform_v(equidistant_to_3d(x, y)))


@dataclass
class Euclidean3DRotator(Euclidean3DTransformer):
    """Rotate the unit vectors (transformer.py:668-679).

    ``rotation`` may be anything with ``w, x, y, z`` attributes (numpy-quaternion's
    ``quaternion``, :class:`vr180_convert_amd.quat.quaternion`), a ``(w, x, y, z)`` sequence or
    a 3x3 matrix.  Non-unit quaternions are normalised the way numpy-quaternion's
    ``as_rotation_matrix`` does.
    """

    rotation: Any

    @property
    def matrix(self) -> NDArray:
        return as_rotation_matrix(self.rotation)

    def transform_v(self, v):
        return np.einsum("ij,...j->...i", self.matrix, v)

    def inverse_transform_v(self, v):
        return np.einsum("ji,...j->...i", self.matrix, v)

    def lower(self, out_shape):
        if not self._is_builtin(Euclidean3DRotator):
            raise NotLowerable("Euclidean3DRotator subclass")
        return [_abi.op(_abi.OP_ROTATE, 0, self.matrix.reshape(9))]

    # inverse_transform == transform in the reference (see Euclidean3DTransformer)
    lower_inverse = lower


def lower_for_get_map(transformer: TransformerBase, *, radius: float, size_input: tuple[int, int],
                      size_output: tuple[int, int], row_band: tuple[int, int] | None = None) -> _abi.Chain:
    """The chain get_map() evaluates (remapper.py:50-57):
    ``NormalizeTransformer() * transformer * DenormalizeTransformer((r, r), (W_in // 2, H_in // 2))``
    lowered for an output grid of ``size_output`` = (W, H); ``size_input`` = (H_in, W_in).
    ``row_band`` = (r0, r1): the chain of output rows r0 .. r1 - 1 only, as a grid of its own (one eye's rows split
    over several GPUs): the same Normalize with its centre moved up by r0 rows -- row j' of the band
    gets ((j' + r0) - H / 2) / s * 2 from the exactly representable difference (j' - (H / 2 - r0)), i.e. the very
    numbers of row j' + r0 of the full grid.
    Raises :class:`NotLowerable` when a stage has no op form or the chain is too long."""
    w, h = size_output
    if row_band is None:
        norm, out_shape = NormalizeTransformer(), (h, w)
    else:
        r0, r1 = row_band
        if not 0 <= r0 < r1 <= h:
            raise ValueError("row_band outside the output grid")
        norm, out_shape = NormalizeTransformer(center=(w / 2, h / 2 - r0), scale=min(w, h)), (r1 - r0, w)
    full = (
        norm
        * transformer
        * DenormalizeTransformer(scale=(radius, radius), center=(size_input[1] // 2, size_input[0] // 2))
    )
    ops = full.lower(out_shape)
    if row_band is not None and ops and ops[0].opcode == _abi.OP_NORMALIZE and ops[0].nparam == 3:
        # the rows of the WHOLE grid in the band's row numbering (p[3] <= j' < p[4]): what a plan sizes from the reach of the output --
        # the radial table of a planar chain covers m = xn^2 + yn^2 up to the corners -- it sizes for the whole grid, so that every band
        # evaluates the very polynomials the unsplit plan does (include/vr180_remap.h: V1C_OP_NORMALIZE)
        ops[0].p[3], ops[0].p[4], ops[0].nparam = float(-r0), float(h - r0), 5
    if len(ops) > _abi.MAX_OPS:
        raise NotLowerable(f"chain has {len(ops)} stages, the op list holds {_abi.MAX_OPS}")
    return _abi.chain(ops)
