"""Minimal quaternion helpers so the engine never needs the ``numpy-quaternion`` package.

The reference gets its rotations from numpy-quaternion (``transformer.py:10,676``; tests use
``from_euler_angles`` / ``from_rotation_vector``, tests/test_remapper.py:80,121; the CLI builds
non-unit "half" quaternions, cli.py:308-319).  That package is not installed in the build or the
GPU image, so its conventions are restated here from SURVEY.md Appendix B; objects of the real
package (anything with ``w, x, y, z``) are accepted wherever a rotation is expected.
"""
from __future__ import annotations

import math
from typing import Any

import numpy as np

_EPS = float(np.finfo(float).eps)


class quaternion:
    """w + xi + yj + zk with the handful of operations the reference's call sites use."""

    __slots__ = ("w", "x", "y", "z")

    def __init__(self, w: float, x: float, y: float, z: float):
        self.w, self.x, self.y, self.z = float(w), float(x), float(y), float(z)

    def components(self) -> tuple[float, float, float, float]:
        return (self.w, self.x, self.y, self.z)

    def conj(self) -> "quaternion":
        return quaternion(self.w, -self.x, -self.y, -self.z)

    conjugate = conj

    def norm(self) -> float:
        """Cayley norm w^2+x^2+y^2+z^2 (numpy-quaternion's ``norm`` is the SQUARED length)."""
        return self.w**2 + self.x**2 + self.y**2 + self.z**2

    def inverse(self) -> "quaternion":
        n = self.norm()
        return quaternion(self.w / n, -self.x / n, -self.y / n, -self.z / n)

    def __neg__(self):
        return quaternion(-self.w, -self.x, -self.y, -self.z)

    def __mul__(self, other):
        if isinstance(other, quaternion):
            a, b = self, other
            return quaternion(
                a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w,
            )
        return quaternion(self.w * other, self.x * other, self.y * other, self.z * other)

    def __rmul__(self, other):
        return quaternion(self.w * other, self.x * other, self.y * other, self.z * other)

    def __add__(self, other):
        if isinstance(other, quaternion):
            return quaternion(self.w + other.w, self.x + other.x, self.y + other.y, self.z + other.z)
        return quaternion(self.w + other, self.x, self.y, self.z)  # scalar adds to w (cli.py:311)

    __radd__ = __add__

    def __repr__(self) -> str:
        return f"quaternion({self.w}, {self.x}, {self.y}, {self.z})"


def _wxyz(q: Any) -> tuple[float, float, float, float]:
    if all(hasattr(q, a) for a in "wxyz"):
        return float(q.w), float(q.x), float(q.y), float(q.z)
    a = np.asarray(q, dtype=float)
    if a.shape == (4,):
        return tuple(float(v) for v in a)  # type: ignore[return-value]
    raise TypeError("rotation must have w,x,y,z attributes, be a (w,x,y,z) sequence or a 3x3 matrix")


def as_rotation_matrix(q: Any) -> np.ndarray:
    """3x3 matrix of a rotation given as quaternion-like, (w,x,y,z) or 3x3 (returned as is).

    Non-unit quaternions are normalised, a zero quaternion raises ``ZeroDivisionError``
    (numpy-quaternion ``as_rotation_matrix``).
    """
    if not all(hasattr(q, a) for a in "wxyz"):
        a = np.asarray(q, dtype=float)
        if a.shape == (3, 3):
            return a
    w, x, y, z = _wxyz(q)
    n = w * w + x * x + y * y + z * z
    if n == 0.0:
        raise ZeroDivisionError("cannot build a rotation from the zero quaternion")
    s = 2.0 if abs(n - 1.0) < _EPS else 2.0 / n
    return np.array(
        [
            [1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w)],
            [s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w)],
            [s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y)],
        ]
    )


def rotate_vectors(R: Any, v: np.ndarray, axis: int = -1) -> np.ndarray:
    """Rotate vectors whose xyz components lie along ``axis``."""
    m = as_rotation_matrix(R)
    v = np.asarray(v, dtype=float)
    return np.moveaxis(np.tensordot(m, v, axes=(-1, axis)), 0, axis % v.ndim)


def from_rotation_vector(rot: Any) -> quaternion:
    """q = exp(r / 2)."""
    r = np.asarray(rot, dtype=float)
    angle = float(np.linalg.norm(r))
    if angle == 0.0:
        return quaternion(1.0, 0.0, 0.0, 0.0)
    s = math.sin(angle / 2) / angle
    return quaternion(math.cos(angle / 2), r[0] * s, r[1] * s, r[2] * s)


def from_euler_angles(alpha: float, beta: float, gamma: float) -> quaternion:
    """z-y-z Euler angles: R = Rz(alpha) Ry(beta) Rz(gamma)."""
    return quaternion(
        math.cos(beta / 2) * math.cos((alpha + gamma) / 2),
        -math.sin(beta / 2) * math.sin((alpha - gamma) / 2),
        math.sin(beta / 2) * math.cos((alpha - gamma) / 2),
        math.cos(beta / 2) * math.sin((alpha + gamma) / 2),
    )


def as_quat_array(a: Any) -> quaternion:
    """[w, x, y, z] -> quaternion (remapper.py:143)."""
    w, x, y, z = (float(v) for v in a)
    return quaternion(w, x, y, z)


def allclose(a: quaternion, b: quaternion, atol: float = 1e-8) -> bool:
    return bool(np.allclose(_wxyz(a), _wxyz(b), atol=atol, rtol=0))
