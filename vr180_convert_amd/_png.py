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
MAX_IMAGE_BYTES = 1 << 31  # raw size the reader accepts (a 16384 x 8192 BGR side-by-side frame is 0.4 GB)


def _chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def _adler32_combine(a1: int, a2: int, len2: int) -> int:
    """Adler-32 of a concatenation from the checksums of its two parts (zlib's adler32_combine)."""
    base = 65521
    rem = len2 % base
    s1 = a1 & 0xFFFF
    s2 = (rem * s1) % base
    s1 += (a2 & 0xFFFF) + base - 1
    s2 += ((a1 >> 16) & 0xFFFF) + ((a2 >> 16) & 0xFFFF) + base - rem
    if s1 >= base:
        s1 -= base
    if s1 >= base:
        s1 -= base
    if s2 >= base << 1:
        s2 -= base << 1
    if s2 >= base:
        s2 -= base
    return s1 | (s2 << 16)


def _to_png_order(dst: np.ndarray, a: np.ndarray) -> None:
    """cv2 channel order (gray, BGR, BGRA) -> PNG's (gray, RGB, RGBA), and back: the swap is its own inverse"""
    cn = a.shape[2]
    if cn == 1:
        dst[...] = a
    else:
        dst[..., 0], dst[..., 1], dst[..., 2] = a[..., 2], a[..., 1], a[..., 0]
        if cn == 4:
            dst[..., 3] = a[..., 3]


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
    nthreads = threads or min(32, os.cpu_count() or 1)
    rows = band_rows or max(16, -(-h // (4 * nthreads)))
    bands = [(r, min(r + rows, h)) for r in range(0, h, rows)]
    stride = 1 + w * cn

    def band(k: int):
        # everything a band needs on its own thread (numpy and zlib release the GIL): scanlines = filter byte + pixels in RGB(A)
        # order, the "Up" differences (uint8 wrap-around = the filter's modulo-256 difference; the row above row 0 is zero),
        # the deflate segment from an empty window and the band's Adler-32
        r0, r1 = bands[k]
        lines = np.empty((r1 - r0, stride), np.uint8)
        lines[:, 0] = 2 if up_filter else 0
        body = lines[:, 1:].reshape(r1 - r0, w, cn)
        _to_png_order(body, a[r0:r1])
        if up_filter:
            flat = lines[:, 1:]
            if r1 - r0 > 1:
                flat[1:] -= flat[:-1].copy()
            if r0 > 0:
                above = np.empty((1, w, cn), np.uint8)
                _to_png_order(above, a[r0 - 1:r0])
                flat[0] -= above.reshape(-1)
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        part = c.compress(lines) + c.flush(zlib.Z_FINISH if k == len(bands) - 1 else zlib.Z_SYNC_FLUSH)
        return part, zlib.adler32(lines), lines.size

    if len(bands) > 1 and nthreads > 1:
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            done = list(pool.map(band, range(len(bands))))
    else:
        done = [band(k) for k in range(len(bands))]
    parts = [d[0] for d in done]
    adler = 1
    for _, ad, n in done:
        adler = _adler32_combine(adler, ad, n)
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
    # The file is not trusted: a forged band directory must not turn header dimensions into a huge allocation or a band into a zip bomb
    # (Pillow, which this reader stands in for, refuses images beyond its decompression-bomb limit).  Deflate never expands beyond
    # ~1032 : 1, so a genuine image's raw size is bounded by its compressed size too; every band is inflated to at most its own
    # (r1 - r0) * stride bytes + 1.
    if h * stride > MAX_IMAGE_BYTES or h * stride > 1100 * len(idat) + 1024:
        return None
    lines = np.empty((h, stride), np.uint8)
    sums = [0] * nb

    def inflate(k: int) -> bool:
        r0, r1, off = bands[k]
        stop = bands[k + 1][2] if k + 1 < nb else end
        want = (r1 - r0) * stride
        try:
            dec = zlib.decompressobj(-15)
            raw = dec.decompress(idat[off:stop], want + 1)  # (bounded: a band that inflates to more is not one of ours)
        except zlib.error:
            return False
        if len(raw) != want or dec.unconsumed_tail:
            return False
        sums[k] = zlib.adler32(raw)
        lines[r0:r1] = np.frombuffer(raw, np.uint8).reshape(r1 - r0, stride)
        return bool((lines[r0:r1, 0] == ftype).all())

    nthreads = threads or min(32, os.cpu_count() or 1)
    pool = ThreadPoolExecutor(max_workers=nthreads) if nthreads > 1 else None

    def run(fn, items):
        return list(pool.map(fn, items)) if pool is not None and len(items) > 1 else [fn(i) for i in items]

    try:
        if not all(run(inflate, list(range(nb)))):
            return None
        adler = 1
        for k, (r0, r1, _) in enumerate(bands):
            adler = _adler32_combine(adler, sums[k], (r1 - r0) * stride)
        if adler & 0xFFFFFFFF != struct.unpack(">I", idat[end:])[0]:
            return None
        flat = lines[:, 1:]
        if ftype == 2 and h > 1:
            # "Up": every byte was stored minus the one above it -- a running sum down the columns (modulo 256), in column blocks
            step = max(4096, -(-flat.shape[1] // (2 * nthreads)))

            def unfilter(c0: int) -> None:
                np.add.accumulate(flat[:, c0:c0 + step], axis=0, dtype=np.uint8, out=flat[:, c0:c0 + step])

            run(unfilter, list(range(0, flat.shape[1], step)))
        body = flat.reshape(h, w, cn)
        if cn == 1:
            return np.ascontiguousarray(body[..., 0])
        out = np.empty((h, w, cn), np.uint8)
        rstep = max(16, -(-h // (4 * nthreads)))

        def swap(r0: int) -> None:
            _to_png_order(out[r0:r0 + rstep], body[r0:r0 + rstep])

        run(swap, list(range(0, h, rstep)))
        return out
    finally:
        if pool is not None:
            pool.shutdown()


def write(path: Any, image: np.ndarray, **kw: Any) -> None:
    Path(path).write_bytes(encode(image, **kw))


def read(path: Any, **kw: Any):
    """``decode`` of a file; ``None`` when it is not one of this writer's PNGs (or cannot be read)."""
    try:
        return decode(Path(path).read_bytes(), **kw)
    except (OSError, MemoryError):
        return None
