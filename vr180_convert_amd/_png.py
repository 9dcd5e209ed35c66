"""Multi-threaded PNG writer for the image I/O step behind the hot path (the reference writes its
results with ``cv.imwrite``, default extension PNG: remapper.py:402,519, cli.py:39).

An 8192 x 4096 side-by-side result is 100 MB of pixels; a single zlib stream at cv2's default level takes seconds on
one core while the remap takes 50 microseconds.  Here the scanlines are cut into bands, every band is deflated on its
own thread (zlib releases the GIL) as a raw deflate segment that ends on a byte boundary (``Z_SYNC_FLUSH``), and
the segments are concatenated into ONE valid zlib stream (header + segments + Adler-32 of all scanlines): any PNG
reader decodes it, pixels are stored losslessly as always.  The GPU image has no device-side codec (no rocJPEG /
nvJPEG equivalent, no torchvision), so this is the file path; ``.npy`` is the codec-free one (_io.py).
"""
from __future__ import annotations

import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Any

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"


def _chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def encode(image: np.ndarray, *, level: int = 1, threads: int | None = None, band_rows: int | None = None, up_filter: bool = True) -> bytes:
    """PNG bytes of a uint8 image in cv2 channel order: (H, W) gray, (H, W, 3) BGR or (H, W, 4) BGRA.
    ``up_filter``: scanline filter 2 ("Up": each byte minus the one above it, one vectorised subtraction) instead
    of 0 ("None") -- photographs deflate about a third smaller and faster."""
    if image.dtype != np.uint8 or image.ndim not in (2, 3):
        raise TypeError("PNG encoder takes uint8 (H, W[, C]) arrays")
    a = image if image.ndim == 3 else image[..., None]
    h, w, cn = a.shape
    if cn not in (1, 3, 4) or h == 0 or w == 0:
        raise ValueError("1, 3 or 4 channels and a non-empty image")
    color_type = {1: 0, 3: 2, 4: 6}[cn]
    # scanlines: filter byte + pixels in RGB(A) order
    lines = np.empty((h, 1 + w * cn), np.uint8)
    lines[:, 0] = 2 if up_filter else 0
    body = lines[:, 1:].reshape(h, w, cn)
    if cn == 1:
        body[...] = a
    else:
        body[..., 0], body[..., 1], body[..., 2] = a[..., 2], a[..., 1], a[..., 0]
        if cn == 4:
            body[..., 3] = a[..., 3]
    if up_filter and h > 1:
        flat = lines[:, 1:]
        flat[1:] -= flat[:-1].copy()  # uint8 wrap-around = the filter's modulo-256 difference (row 0: the row above is zero)
    nthreads = threads or min(32, os.cpu_count() or 1)
    rows = band_rows or max(16, -(-h // (4 * nthreads)))
    bands = [(r, min(r + rows, h)) for r in range(0, h, rows)]

    def deflate(k: int) -> bytes:
        r0, r1 = bands[k]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(lines[r0:r1]) + c.flush(zlib.Z_FINISH if k == len(bands) - 1 else zlib.Z_SYNC_FLUSH)

    if len(bands) > 1 and nthreads > 1:
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            parts = list(pool.map(deflate, range(len(bands))))
    else:
        parts = [deflate(k) for k in range(len(bands))]
    adler = 1
    for r0, r1 in bands:
        adler = zlib.adler32(lines[r0:r1], adler)
    stream = b"\x78\x01" + b"".join(parts) + struct.pack(">I", adler & 0xFFFFFFFF)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)
    return _SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", stream) + _chunk(b"IEND", b"")


def write(path: Any, image: np.ndarray, **kw: Any) -> None:
    Path(path).write_bytes(encode(image, **kw))
