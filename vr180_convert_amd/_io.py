"""Image file I/O either side of the hot path (reference remapper.py:373,402,453,519 use
``cv.imread`` / ``cv.imwrite``).  cv2 is used when it is importable, Pillow otherwise; arrays are
BGR like cv2's.  Codec work is outside the measured path.

Two additions for batch work, where codecs -- not the remap -- set the end-to-end time:
* ``.npy`` files are read (memory-mapped) and written as raw uint8 arrays in cv2 channel order: the codec-free
  format for frame sequences (the GPU image ships no device-side JPEG / PNG codec: no rocJPEG, no torchvision);
* large PNGs are written by the multi-threaded encoder of ``_png.py`` when cv2 is absent, and PNGs it wrote are read back by its
  multi-threaded decoder (a private chunk holds the band directory; every other reader sees an ordinary PNG)."""
from __future__ import annotations

from pathlib import Path
from typing import Any

import numpy as np

PARALLEL_PNG_MIN_BYTES = 1 << 20  # below this a single zlib stream is as fast

try:  # pragma: no cover - not installed in the build / GPU image
    import cv2 as _cv
except Exception:  # noqa: BLE001
    _cv = None


def imread(path: Any):
    """BGR uint8 (H, W, 3) array, or ``None`` when the file cannot be read (cv2.imread's contract)."""
    p = Path(path).as_posix()
    if p.lower().endswith(".npy"):
        try:
            a = np.load(p, mmap_mode="r")
        except (OSError, ValueError):
            return None
        return a if a.dtype == np.uint8 and a.ndim in (2, 3) else None
    if _cv is not None:
        return _cv.imread(p)
    if p.lower().endswith(".png"):
        # PNGs of this package's writer carry a band directory: inflated on all cores (any other PNG: None, Pillow below)
        from . import _png

        a = _png.read(p)
        if a is not None:  # cv2.imread's default flag (IMREAD_COLOR): always 3 channels
            return a if a.ndim == 3 and a.shape[2] == 3 else np.repeat(a[..., None], 3, axis=2) if a.ndim == 2 else np.ascontiguousarray(a[..., :3])
    from PIL import Image

    try:
        with Image.open(p) as im:
            rgb = np.asarray(im.convert("RGB"))
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[..., ::-1])


def imwrite(path: Any, image: np.ndarray) -> bool:
    p = Path(path).as_posix()
    if image.dtype != np.uint8:
        image = np.clip(np.rint(image), 0, 255).astype(np.uint8)  # cv2.imwrite converts with saturation
    if p.lower().endswith(".npy"):
        np.save(p, np.ascontiguousarray(image))
        return True
    if _cv is not None:
        return bool(_cv.imwrite(p, image))
    if p.lower().endswith(".png") and image.size >= PARALLEL_PNG_MIN_BYTES and image.ndim in (2, 3):
        from . import _png

        _png.write(p, image, level=1)  # cv2.imwrite's default compression level
        return True
    from PIL import Image

    arr = image if image.ndim == 2 else image[..., ::-1] if image.shape[2] == 3 else image[..., [2, 1, 0, 3]]
    # cv2.imwrite's defaults: PNG compression level 1 (IMWRITE_PNG_COMPRESSION), JPEG quality 95
    ext = Path(p).suffix.lower()
    opts = {"compress_level": 1} if ext == ".png" else {"quality": 95} if ext in (".jpg", ".jpeg") else {}
    Image.fromarray(np.ascontiguousarray(arr)).save(p, **opts)
    return True


def imread_many(paths: list) -> list:
    """``imread`` of several files on a small thread pool (the codecs release the GIL); entries that
    are not paths are passed through."""
    todo = [i for i, q in enumerate(paths) if isinstance(q, (str, Path))]
    if len(todo) < 2:
        return [imread(q) if isinstance(q, (str, Path)) else q for q in paths]
    from concurrent.futures import ThreadPoolExecutor

    out = list(paths)
    with ThreadPoolExecutor(max_workers=min(8, len(todo))) as pool:
        for i, img in zip(todo, pool.map(imread, [paths[i] for i in todo])):
            out[i] = img
    return out


def imwrite_many(paths: list, images: list) -> None:
    if len(paths) < 2:
        for q, im in zip(paths, images):
            imwrite(q, im)
        return
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(8, len(paths))) as pool:
        list(pool.map(imwrite, paths, images))


def draw_anaglyph_labels(combine: np.ndarray) -> np.ndarray:
    """The "L" / "R" labels of apply_lr(merge=True) (reference remapper.py:498-516): they need
    cv2.putText and are drawn only when cv2 is importable.  The anaglyph itself is computed on the
    device (remapper.anaglyph_tensors)."""
    colors = [(0, 128, 255), (255, 128, 0)]
    if _cv is not None:  # pragma: no cover
        _cv.putText(combine, "L", (0, len(combine[1]) // 10), _cv.FONT_HERSHEY_SIMPLEX, len(combine) // 1000, colors[0], 2, _cv.LINE_AA)
        _cv.putText(combine, "R", (len(combine[1]) // 2, len(combine[0]) // 10), _cv.FONT_HERSHEY_SIMPLEX, len(combine) // 1000, colors[1], 2, _cv.LINE_AA)
    else:
        global _LABEL_WARNED
        if not _LABEL_WARNED:
            _LABEL_WARNED = True
            import warnings

            warnings.warn("apply_lr(merge=True): cv2 is not installed, so the 'L' / 'R' labels the reference draws with "
                          "cv2.putText (remapper.py:498-516) are missing from the anaglyph", RuntimeWarning, stacklevel=3)
    return combine


_LABEL_WARNED = False
