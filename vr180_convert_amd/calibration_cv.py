"""The front ends of the reference that are OpenCV / libxmp calls from end to end, imported lazily:

* ``match_points`` -- AKAZE keypoints + brute-force matching of the two eyes (reference remapper.py:194-248), what ``--automatch fm``
  feeds into ``match_lr`` / ``rotation_match_robust``;
* ``pick_points_gui`` -- the full-screen window in which the user clicks matching points (cli.py:82-113), ``--automatch gui``;
* ``draw_match_image`` -- the ``--savematch`` picture: 100 sampled inlier matches side by side (cli.py:299-304, 362-365);
* ``write_vr180_xmp`` -- the ``xmp`` command's Google VR180 photo metadata (GPano / GImage XMP, cli.py:439-540).

None of them is on the remap path and none has arithmetic of its own to restate: they hand images to ``cv2`` / ``libxmp`` and results
on.  The engine's GPU image ships neither library, so every function raises ``OptionalDependencyMissing`` -- with the package to
install -- when its import fails; the CLI turns that into a usage error.  ``tests/test_cli.py`` drives all four through stand-in
modules that record the calls (the glue -- option parsing, scaling, file names, XMP properties -- is what is tested; the libraries
are theirs).
"""
from __future__ import annotations

import base64
import logging
import random
from pathlib import Path
from tempfile import NamedTemporaryFile
from typing import Any, Sequence

import numpy as np

LOG = logging.getLogger(__name__)


class OptionalDependencyMissing(ImportError):
    """cv2 / libxmp is not installed."""


def _cv2():
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        raise OptionalDependencyMissing("this option needs OpenCV (pip install opencv-python): " + repr(e)) from e
    return cv2


def match_points(image1: np.ndarray, image2: np.ndarray, *, scale: float = 1):
    """Keypoint matches between the two eyes (reference remapper.py:194-248): AKAZE on both images (resized by ``scale`` first),
    one brute-force match per descriptor of image 1.  Returns ``(points1, points2, keypoints1, keypoints2, matches, image1, image2)``
    -- the points in ORIGINAL pixel units, the images as matched (resized)."""
    cv = _cv2()
    if scale != 1:
        image1 = cv.resize(image1, (int(image1.shape[1] * scale), int(image1.shape[0] * scale)))
        image2 = cv.resize(image2, (int(image2.shape[1] * scale), int(image2.shape[0] * scale)))
    detector = cv.AKAZE_create()
    kp1, des1 = detector.detectAndCompute(image1, None)
    kp2, des2 = detector.detectAndCompute(image2, None)
    matches = cv.BFMatcher().match(des1, des2)
    p1 = np.array([kp1[m.queryIdx].pt for m in matches], dtype=float)
    p2 = np.array([kp2[m.trainIdx].pt for m in matches], dtype=float)
    if scale != 1:
        p1, p2 = p1 / scale, p2 / scale
    return p1, p2, np.array(kp1), np.array(kp2), np.array(matches), image1, image2


def pick_points_gui(images: Sequence[Any]) -> list[tuple[int, int]]:
    """One left click per image, the images shown one after the other in a full-screen window (cli.py:82-113).  ``images``: paths or
    arrays -- the CLI passes [left, right] * n for n matched pairs."""
    cv = _cv2()
    shown = [cv.imread(Path(im).as_posix()) if isinstance(im, (str, Path)) else im for im in images]
    title = "Select position"
    cv.namedWindow(title, cv.WND_PROP_FULLSCREEN)
    cv.setWindowProperty(title, cv.WND_PROP_FULLSCREEN, cv.WINDOW_FULLSCREEN)
    clicks: list[tuple[int, int]] = []

    def on_mouse(event: int, x: int, y: int, flags: int, param: Any) -> None:
        if event == cv.EVENT_LBUTTONDOWN:
            LOG.info(f"Position {len(clicks)}: ({x}, {y})")
            clicks.append((x, y))

    cv.setMouseCallback(title, on_mouse)
    for k, im in enumerate(shown):
        cv.imshow(title, im)
        while len(clicks) <= k:
            cv.waitKey(10)
    cv.destroyAllWindows()
    return clicks[: len(shown)]


def draw_match_image(img_l: np.ndarray, kp_l: Any, img_r: np.ndarray, kp_r: Any, matches: np.ndarray, discarded: np.ndarray,
                     n: int = 100) -> np.ndarray:
    """``--savematch``: ``n`` randomly sampled matches that survived the robust fit, drawn side by side (cli.py:299-304)."""
    cv = _cv2()
    inliers = list(np.asarray(matches)[~np.asarray(discarded, dtype=bool)])
    return cv.drawMatches(img_l, kp_l, img_r, kp_r, random.sample(inliers, min(n, len(inliers))), None)  # noqa: S311


# Google's namespaces (cli.py:492-498)
XMP_GIMAGE = "http://ns.google.com/photos/1.0/image/"
XMP_GPANO = "http://ns.google.com/photos/1.0/panorama/"
XMP_NOTE = "http://ns.adobe.com/xmp/note/"


def vr180_xmp_properties(width: int, height: int) -> list[tuple[str, str, Any]]:
    """(namespace, name, value) of the GPano block the reference writes for a (height, width) side-by-side image (cli.py:500-512):
    the left half is the cropped area of a full panorama of the SBS image's size; ints are written with set_property_int."""
    return [
        (XMP_GPANO, "UsePanoramaViewer", "True"),
        (XMP_GPANO, "ProjectionType", "equirectangular"),
        (XMP_GPANO, "CroppedAreaImageWidthPixels", width / 2),
        (XMP_GPANO, "CroppedAreaImageHeightPixels", height),
        (XMP_GPANO, "CroppedAreaLeftPixels", width / 4),
        (XMP_GPANO, "CroppedAreaTopPixels", 0),
        (XMP_GPANO, "FullPanoWidthPixels", width),
        (XMP_GPANO, "FullPanoHeightPixels", height),
        (XMP_GPANO, "PosePitchDegrees", 0),
        (XMP_GPANO, "PoseRollDegrees", 0),
        (XMP_GPANO, "InitialViewHeadingDegrees", 180),
    ]


def write_vr180_xmp(in_path: Path) -> Path:
    """``<name>.xmp<ext>`` next to ``in_path``: the LEFT half of the side-by-side image with the right half embedded as base64 in
    ``GImage:Data`` and the GPano block above (cli.py:467-540).  Returns the written path."""
    try:
        from libxmp import XMPFiles, XMPMeta
    except Exception as e:  # noqa: BLE001
        raise OptionalDependencyMissing("the xmp command needs python-xmp-toolkit (pip install python-xmp-toolkit; it loads the exempi "
                                        "library): " + repr(e)) from e
    from . import _io

    image = _io.imread(in_path)
    if image is None:
        raise ValueError(f"cannot read {in_path}")
    height, width = image.shape[0], image.shape[1]
    left_path = in_path.with_suffix(f".xmp{in_path.suffix}")
    with NamedTemporaryFile(suffix=left_path.suffix) as right_file:
        _io.imwrite(left_path, np.ascontiguousarray(image[:, : width // 2]))
        _io.imwrite(right_file.name, np.ascontiguousarray(image[:, width // 2:]))
        xmpfile = XMPFiles(file_path=left_path.as_posix(), open_forupdate=True)
        meta = XMPMeta()
        XMPMeta.register_namespace(XMP_GIMAGE, "GImage")
        XMPMeta.register_namespace(XMP_GPANO, "GPano")
        XMPMeta.register_namespace(XMP_NOTE, "xmpNote")
        for ns, name, value in vr180_xmp_properties(width, height):
            if isinstance(value, str):
                meta.set_property(ns, name, value)
            else:
                meta.set_property_int(ns, name, value)
        meta.set_property(XMP_GIMAGE, "Mime", "image/jpeg")
        meta.set_property(XMP_GIMAGE, "Data", base64.b64encode(Path(right_file.name).read_bytes()).decode())
        meta.set_property(XMP_NOTE, "HasExtendedXMP", "06A56CB0A1A7FAFDA459CA3FAA14B474")
        if not xmpfile.can_put_xmp(meta):
            raise ValueError(f"Cannot put XMP to {in_path}")
        xmpfile.put_xmp(meta)
        xmpfile.close_file()
    return left_path
