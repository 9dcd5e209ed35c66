"""Rotation estimation of the reference's calibration path: the closed-form pieces.

``rotation_match`` / ``rotation_match_robust`` (reference remapper.py:93-191) produce the per-pair
quaternions that ``remap_tensors(..., rotations=...)`` consumes (BASELINE config 5).  They are one
3x3 correlation and one symmetric 4x4 eigenproblem on the host.  Feature detection and matching
(``match_points``, remapper.py:194-248: cv2.AKAZE + BFMatcher) needs OpenCV and is not mirrored;
everything downstream of the matched points is: ``match_lr`` (points -> unit rays through the
decoder's inverse, remapper.py:251-321), the rotation fit, and ``calibration_rotators`` (the
pseudo-half quaternions the CLI inserts per eye, cli.py:308-319).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Any, Sequence

import numpy as np

from .chain import DenormalizeTransformer, TransformerBase, equidistant_to_3d
from .quat import as_rotation_matrix, quaternion

LOG = logging.getLogger(__name__)


def rotate_vectors(q: Any, v: np.ndarray) -> np.ndarray:
    """numpy-quaternion's ``rotate_vectors(q, v)`` for one rotation: ``v`` (..., 3) rotated by the
    (normalised) quaternion -- the matrix form the engine uses everywhere (quat.as_rotation_matrix)."""
    return np.einsum("ij,...j->...i", as_rotation_matrix(q), np.asarray(v, dtype=float))


def _horn_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Symmetric 4x4 matrix N (order w, x, y, z) with ``sum_k b_k . R(q) a_k == q^T N q`` for unit q:
    expanding ``R(q) a = q (0, a) q^-1`` makes every entry a signed sum of entries of the 3x3
    correlation ``M = sum_k a_k b_k^T`` (Horn 1987, closed-form absolute orientation)."""
    m = a.T @ b
    (sxx, sxy, sxz), (syx, syy, syz), (szx, szy, szz) = m
    return np.array([
        [sxx + syy + szz, syz - szy, szx - sxz, sxy - syx],
        [syz - szy, sxx - syy - szz, sxy + syx, szx + sxz],
        [szx - sxz, sxy + syx, syy - sxx - szz, syz + szy],
        [sxy - syx, szx + sxz, syz + szy, szz - sxx - syy],
    ])


def rotation_match(points_to_be_rotated: np.ndarray, points: np.ndarray) -> quaternion:
    """Quaternion q minimising ``E(q) = sum_k |R(q) a_k - b_k|^2`` (reference remapper.py:93-143, derivation
    docs/math.md:28-59: E is the quadratic form q^T B q, minimised by the eigenvector of B's smallest
    eigenvalue).  Since ``|R a - b|^2 = |a|^2 + |b|^2 - 2 b.(R a)``, ``B = sum_k (|a_k|^2 + |b_k|^2) I - 2 N``
    with Horn's matrix N of the 3x3 correlation of the two point sets: the minimiser of E is the eigenvector
    of N's LARGEST eigenvalue -- one 3x3 product and one symmetric 4x4 eigenproblem (``eigh``), whatever
    the number of points.  q and -q are one rotation, and LAPACK returns either; the result is canonicalised to
    ``w >= 0`` because the CLI's half-rotation (``calibration_rotators``: ``phi = arccos(q.w)``) is sign-sensitive --
    with w < 0 it applies the FULL rotation to each eye instead of half (the reference takes whichever sign
    ``np.linalg.eig`` happens to return, remapper.py:140-143)."""
    a = np.asarray(points_to_be_rotated, dtype=float).reshape(-1, 3)
    b = np.asarray(points, dtype=float).reshape(-1, 3)
    if a.shape != b.shape:
        raise ValueError("point sets must have the same shape")
    lam, vec = np.linalg.eigh(_horn_matrix(a, b))  # ascending eigenvalues
    w, x, y, z = vec[:, -1]
    if w < 0 or (w == 0 and (x, y, z) < (0, 0, 0)):
        w, x, y, z = -w, -x, -y, -z
    if LOG.isEnabledFor(logging.DEBUG):
        e_min = float(np.sum(a * a) + np.sum(b * b) - 2.0 * lam[-1])  # = smallest eigenvalue of B = min E
        LOG.debug("Error: %s", np.sqrt(max(e_min, 0.0)) / max(len(b), 1))
    return quaternion(w, x, y, z)


def rotation_match_robust(points_to_be_rotated: np.ndarray, points: np.ndarray, n_iter: int = 15,
                          quantile: float = 0.9) -> tuple[quaternion, np.ndarray]:
    """``rotation_match`` with iterated trimming (reference remapper.py:146-191): fit, measure every
    surviving pair's residual ``|R(q) a - b|``, discard the pairs above the ``quantile`` of the residuals,
    refit -- ``n_iter`` fits in all, the last one is not followed by a trim.  Returns the final quaternion
    and a mask over the ORIGINAL points that is True for every discarded pair."""
    a = np.asarray(points_to_be_rotated, dtype=float).reshape(-1, 3)
    b = np.asarray(points, dtype=float).reshape(-1, 3)
    alive = np.arange(len(b))  # indices (into the original arrays) of the pairs still in the fit
    q = quaternion(1, 0, 0, 0)
    for it in range(n_iter):
        q = rotation_match(a[alive], b[alive])
        if it + 1 == n_iter:
            break
        residual = np.linalg.norm(a[alive] @ as_rotation_matrix(q).T - b[alive], axis=-1)
        keep = residual <= np.quantile(residual, quantile)
        LOG.debug("Removed %d outliers, %d points left.", int((~keep).sum()), int(keep.sum()))
        alive = alive[keep]
    discarded = np.ones(len(b), dtype=bool)
    discarded[alive] = False
    return q, discarded


def match_lr(decoder: TransformerBase | tuple[TransformerBase, TransformerBase], points_l: Sequence[tuple[float, float]],
             points_r: Sequence[tuple[float, float]], in_paths: Sequence[Any], *, radius: float | str = "auto") -> tuple[np.ndarray, np.ndarray]:
    """Matched pixel positions of the two eyes -> unit rays, the arguments of ``rotation_match`` /
    ``rotation_match_robust`` (reference remapper.py:251-321): the points go back through
    ``(decoder * DenormalizeTransformer((r, r), (W // 2, H // 2))).inverse_transform`` as float32
    coordinates and through ``equidistant_to_3d``.  ``in_paths``: the two images (paths or arrays);
    only their shape (centre) and, for ``radius="auto"`` / ``"max"``, their pixels are used."""
    from . import _io
    from .remapper import get_radius_smart

    if len(points_l) != len(points_r):
        raise ValueError("The number of points must be the same.")
    images = [_io.imread(p) if isinstance(p, (str, Path)) else p for p in in_paths]
    center = (images[0].shape[1] // 2, images[0].shape[0] // 2)
    radius_ = get_radius_smart(radius, images)

    def rays(dec: TransformerBase, pts: Any) -> np.ndarray:
        pts_ = np.array(pts)
        x, y = pts_[:, 0].astype(np.float32), pts_[:, 1].astype(np.float32)
        x, y = (dec * DenormalizeTransformer(scale=(radius_, radius_), center=center)).inverse_transform(x, y)
        return equidistant_to_3d(x, y)

    if isinstance(decoder, tuple):
        return rays(decoder[0], points_l), rays(decoder[1], points_r)
    v = rays(decoder, np.concatenate([points_l, points_r], axis=0))
    return v[: len(points_l)], v[len(points_l):]


def calibration_rotators(q: Any) -> tuple[quaternion, quaternion]:
    """The two quaternions the reference's CLI wraps in ``Euclidean3DRotator`` for the left and the
    right eye (cli.py:308-319): ``half_q = sin(phi / 2) / sin(phi) * q + 0.5`` with
    ``phi = arccos(q.w)`` -- generally NOT unit (the rotator normalises) --
    left: ``conj(half_q)``, right: ``half_q``.  The formula halves the rotation only for ``w >= 0``
    (for w < 0, phi > pi / 2 and ``k q + 0.5`` normalises to the full rotation or worse), so the equivalent
    quaternion -q is used then: every sign of a fitted q gives each eye half the inter-eye rotation."""
    w, x, y, z = (q.w, q.x, q.y, q.z) if hasattr(q, "w") else tuple(float(c) for c in q)
    if w < 0:
        w, x, y, z = -w, -x, -y, -z
    phi = np.arccos(w)
    k = np.sin(phi / 2) / np.sin(phi)
    half = quaternion(k * w + 0.5, k * x, k * y, k * z)
    return half.conj(), half
