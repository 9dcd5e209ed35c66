"""The two OpenCV-only pieces of the reference's calibration front end: AKAZE feature matching between the eyes
(``match_points``, reference remapper.py:194-248) and the click-to-pick window (cli.py:82-113).  Both are pure cv2
calls with no arithmetic of their own; they are imported lazily by the CLI (``--automatch fm`` / ``gui``) and raise
``ImportError`` where cv2 is not installed (the build and GPU images).  Everything downstream of the matched points
-- rays, rotation fit, per-eye rotators -- is in calibration.py and needs no OpenCV."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Sequence

import numpy as np


def match_points(image_l: np.ndarray, image_r: np.ndarray, *, scale: float = 1.0):
    """Brute-force matched AKAZE keypoints of the two eyes, in full-resolution pixel coordinates; features are
    detected on images resized by ``scale``.  Returns ``(points_l, points_r, extras)`` with ``extras`` = the
    keypoints, matches and (resized) images for drawing the match picture."""
    import cv2 as cv

    if scale != 1:
        image_l, image_r = (cv.resize(im, (int(im.shape[1] * scale), int(im.shape[0] * scale))) for im in (image_l, image_r))
    detector = cv.AKAZE_create()
    (kp_l, des_l), (kp_r, des_r) = (detector.detectAndCompute(im, None) for im in (image_l, image_r))
    matches = cv.BFMatcher().match(des_l, des_r)
    pts_l = np.array([kp_l[m.queryIdx].pt for m in matches], dtype=float) / scale
    pts_r = np.array([kp_r[m.trainIdx].pt for m in matches], dtype=float) / scale
    return pts_l, pts_r, dict(kp_l=kp_l, kp_r=kp_r, matches=matches, image_l=image_l, image_r=image_r)


def pick_points(images: Sequence[Any]) -> list[tuple[int, int]]:
    """Show the images one after the other full screen and return the position clicked in each."""
    import cv2 as cv

    shown = [cv.imread(Path(im).as_posix()) if isinstance(im, (str, Path)) else im for im in images]
    window = "Select position"
    cv.namedWindow(window, cv.WND_PROP_FULLSCREEN)
    cv.setWindowProperty(window, cv.WND_PROP_FULLSCREEN, cv.WINDOW_FULLSCREEN)
    clicks: list[tuple[int, int]] = []
    cv.setMouseCallback(window, lambda ev, x, y, flags, param: clicks.append((x, y)) if ev == cv.EVENT_LBUTTONDOWN else None)
    for k, im in enumerate(shown):
        cv.imshow(window, im)
        while len(clicks) <= k:
            cv.waitKey(10)
    cv.destroyAllWindows()
    return clicks[: len(shown)]
