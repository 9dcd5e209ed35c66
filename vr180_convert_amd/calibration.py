"""Rotation estimation of the reference's calibration path: the closed-form pieces.

``rotation_match`` / ``rotation_match_robust`` (reference remapper.py:93-191) produce the per-pair
quaternions that ``remap_tensors(..., rotations=...)`` consumes (BASELINE config 5).  They are a few
4x4 products on the host, exactly like the reference.  Feature detection and matching
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


def rotation_match(points_to_be_rotated: np.ndarray, points: np.ndarray) -> quaternion:
    """Quaternion minimising ``sum |R a_k - b_k|^2`` (reference remapper.py:93-143;
    https://lisyarus.github.io/blog/posts/3d-shape-matching-with-quaternions.html):
    E = sum |q a_k - b_k q|^2 = q^T B q with B = sum S_k^T S_k, S_k = Rmul(a_k) - Lmul(b_k); the
    minimiser is the eigenvector of the smallest eigenvalue of the symmetric 4x4 matrix B."""
    a = np.asarray(points_to_be_rotated, dtype=float)
    b = np.asarray(points, dtype=float)
    a = np.concatenate([a, np.zeros_like(a[..., :1])], axis=1)
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    b = np.concatenate([b, np.zeros_like(b[..., :1])], axis=1)
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    right_mult_matrix = np.array([[aw, -az, ay, -ax], [az, aw, -ax, -ay], [-ay, ax, aw, -az], [ax, ay, az, aw]])
    left_mult_matrix = np.array([[bw, bz, -by, -bx], [-bz, bw, bx, -by], [by, -bx, bw, -bz], [bx, by, bz, bw]])
    S = right_mult_matrix - left_mult_matrix
    B = np.einsum("jik,jlk->il", S, S)
    eigenvalues, eigenvectors = np.linalg.eig(B)  # (the reference's call; B is symmetric)
    q = eigenvectors[:, np.argmin(eigenvalues)]
    LOG.debug("Error: %s", np.sqrt(max(float(np.min(eigenvalues.real)), 0.0)) / len(points))
    q = np.real(q)
    return quaternion(q[3], q[0], q[1], q[2])  # eigenvector is (x, y, z, w)


def rotation_match_robust(points_to_be_rotated: np.ndarray, points: np.ndarray, n_iter: int = 15,
                          quantile: float = 0.9) -> tuple[quaternion, np.ndarray]:
    """``rotation_match`` repeated while dropping the worst ``1 - quantile`` of the residuals
    (reference remapper.py:146-191).  Returns the quaternion and the mask of removed points."""
    a = np.asarray(points_to_be_rotated, dtype=float)
    b = np.asarray(points, dtype=float)
    bad_idx = np.full(len(b), False)
    q = quaternion(1, 0, 0, 0)
    for i in range(n_iter):
        q = rotation_match(points_to_be_rotated=a, points=b)
        if i == n_iter - 1:
            break
        error = np.linalg.norm(rotate_vectors(q, a) - b, axis=-1)
        threshold = np.quantile(error, quantile)
        bad_idx_current = error > threshold
        bad_idx[~bad_idx] = bad_idx_current
        a = a[~bad_idx_current]
        b = b[~bad_idx_current]
        LOG.debug("Removed %d outliers, %d points left.", int(bad_idx_current.sum()), len(b))
    return q, bad_idx


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
    left: ``conj(half_q)``, right: ``half_q``."""
    w, x, y, z = (q.w, q.x, q.y, q.z) if hasattr(q, "w") else tuple(float(c) for c in q)
    phi = np.arccos(w)
    k = np.sin(phi / 2) / np.sin(phi)
    half = quaternion(k * w + 0.5, k * x, k * y, k * z)
    return half.conj(), half
