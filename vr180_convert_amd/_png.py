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
    # band directory for decode(): an ancillary, private, unsafe-to-copy chunk other readers skip.  Every band was deflated from an
    # empty window, so each segment inflates on its own.
    index = struct.pack(">BBI", 1, 2 if up_filter else 0, len(bands))
    off = 2
    for (r0, r1), part in zip(bands, parts):
        index += struct.pack(">III", r0, r1, off)
        off += len(part)
    return _SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(_INDEX_CHUNK, index) + _chunk(b"IDAT", stream) + _chunk(b"IEND", b"")


_INDEX_CHUNK = b"vrBD"


def decode(data: bytes, *, threads: int | None = None):
    """The image of a PNG written by ``encode`` (cv2 channel order), its bands inflated in parallel -- or ``None`` for every other
    PNG (no band directory, several IDAT chunks, other bit depths / filters, anything inconsistent): the caller then uses a general
    decoder.  The Adler-32 of the zlib stream is verified."""
    if len(data) < 8 + 25 or data[:8] != _SIGNATURE:
        return None
    pos, ihdr, index, idat = 8, None, None, None
    while pos + 12 <= len(data):
        (n,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n:
            return None
        if kind == b"IHDR":
            ihdr = body
        elif kind == _INDEX_CHUNK:
            if zlib.crc32(kind + body) & 0xFFFFFFFF != struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]:
                return None
            index = body
        elif kind == b"IDAT":
            if idat is not None:
                return None
            idat = memoryview(data)[pos + 8:pos + 8 + n]
        elif kind == b"IEND":
            break
        pos += 12 + n
    if ihdr is None or index is None or idat is None or len(ihdr) != 13 or len(index) < 6:
        return None
    w, h, depth, color_type, comp, flt, lace = struct.unpack(">IIBBBBB", ihdr)
    cn = {0: 1, 2: 3, 6: 4}.get(color_type)
    version, ftype, nb = struct.unpack(">BBI", index[:6])
    if depth != 8 or cn is None or comp or flt or lace or version != 1 or ftype not in (0, 2) or len(index) != 6 + 12 * nb or nb == 0:
        return None
    bands = [struct.unpack(">III", index[6 + 12 * k:18 + 12 * k]) for k in range(nb)]
    stride = 1 + w * cn
    end = len(idat) - 4
    ok = bands[0][0] == 0 and bands[-1][1] == h and all(a[1] == b[0] for a, b in zip(bands, bands[1:])) and \
        all(r0 < r1 and 2 <= off <= end for r0, r1, off in bands) and all(a[2] <= b[2] for a, b in zip(bands, bands[1:]))
    if not ok or len(idat) < 6:
        return None
    lines = np.empty((h, stride), np.uint8)

    def inflate(k: int) -> bool:
        r0, r1, off = bands[k]
        stop = bands[k + 1][2] if k + 1 < nb else end
        try:
            raw = zlib.decompressobj(-15).decompress(idat[off:stop])
        except zlib.error:
            return False
        if len(raw) != (r1 - r0) * stride:
            return False
        lines[r0:r1] = np.frombuffer(raw, np.uint8).reshape(r1 - r0, stride)
        return True

    nthreads = threads or min(32, os.cpu_count() or 1)
    if nb > 1 and nthreads > 1:
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            good = all(pool.map(inflate, range(nb)))
    else:
        good = all(inflate(k) for k in range(nb))
    if not good or not (lines[:, 0] == ftype).all():
        return None
    adler = 1
    for r0, r1, _ in bands:
        adler = zlib.adler32(lines[r0:r1], adler)
    if adler & 0xFFFFFFFF != struct.unpack(">I", idat[end:])[0]:
        return None
    flat = lines[:, 1:]
    if ftype == 2 and h > 1:
        # "Up": every byte was stored minus the one above it -- a running sum down the columns (modulo 256), in column blocks
        step = max(4096, -(-flat.shape[1] // max(nthreads, 1)))
        cols = [(c, min(c + step, flat.shape[1])) for c in range(0, flat.shape[1], step)]

        def unfilter(c: tuple) -> None:
            np.add.accumulate(flat[:, c[0]:c[1]], axis=0, dtype=np.uint8, out=flat[:, c[0]:c[1]])

        if len(cols) > 1 and nthreads > 1:
            with ThreadPoolExecutor(max_workers=nthreads) as pool:
                list(pool.map(unfilter, cols))
        else:
            for c in cols:
                unfilter(c)
    body = flat.reshape(h, w, cn)
    if cn == 1:
        return np.ascontiguousarray(body[..., 0])
    out = np.empty((h, w, cn), np.uint8)
    out[..., 0], out[..., 1], out[..., 2] = body[..., 2], body[..., 1], body[..., 0]
    if cn == 4:
        out[..., 3] = body[..., 3]
    return out


def write(path: Any, image: np.ndarray, **kw: Any) -> None:
    Path(path).write_bytes(encode(image, **kw))


def read(path: Any, **kw: Any):
    """``decode`` of a file; ``None`` when it is not one of this writer's PNGs (or cannot be read)."""
    try:
        return decode(Path(path).read_bytes(), **kw)
    except OSError:
        return None
