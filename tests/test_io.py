"""Image I/O either side of the path: the multi-threaded PNG writer decodes to the same pixels
with an independent reader (Pillow), .npy round trips, cv2's imread contract (None for unreadable files)."""
import numpy as np
import pytest

from vr180_convert_amd import _io, _png


@pytest.mark.parametrize("shape", [(1, 1, 3), (37, 53, 3), (64, 48), (200, 300, 4), (1030, 517, 3)])
@pytest.mark.parametrize("threads", [1, 4])
def test_parallel_png_decodes_with_pillow(tmp_path, shape, threads):
    from PIL import Image

    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    if img.ndim == 3:
        img[: shape[0] // 2] = img[0, 0]  # a compressible half
    p = tmp_path / "x.png"
    _png.write(p, img, threads=threads, band_rows=16)
    with Image.open(p) as im:
        im.load()
        got = np.asarray(im)
    want = img if img.ndim == 2 else img[..., ::-1] if shape[2] == 3 else img[..., [2, 1, 0, 3]]
    assert got.shape == want.shape and np.array_equal(got, want)
    back = _io.imread(p)  # cv2.imread's default: always 3 channels, BGR
    assert np.array_equal(back, np.repeat(img[..., None], 3, axis=2) if img.ndim == 2 else img[..., :3])
    # the writer's own reader (band directory chunk, bands inflated in parallel): the array as written, for either filter
    assert np.array_equal(_png.read(p, threads=threads), img)
    assert np.array_equal(_png.decode(_png.encode(img, threads=threads, band_rows=7, up_filter=False), threads=threads), img)


def test_parallel_png_reader_leaves_foreign_and_damaged_files_to_the_general_decoder(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    # a PNG of another writer: no band directory
    Image.fromarray(img[..., ::-1].copy()).save(tmp_path / "pil.png")
    assert _png.read(tmp_path / "pil.png") is None and np.array_equal(_io.imread(tmp_path / "pil.png"), img)
    good = _png.encode(img, band_rows=16)
    assert np.array_equal(_png.decode(good), img)
    at = good.index(b"vrBD")
    # a directory that does not match the stream (an offset moved), a damaged band, a damaged checksum, a truncated file
    moved = bytearray(good)
    moved[at + 4 + 6 + 12 + 11] ^= 1
    assert _png.decode(bytes(moved)) is None  # (the chunk's CRC no longer matches)
    idat = good.index(b"IDAT")
    hurt = bytearray(good)
    hurt[idat + 40] ^= 0x55
    assert _png.decode(bytes(hurt)) is None
    tail = bytearray(good)
    tail[-17] ^= 1  # last byte of the Adler-32
    assert _png.decode(bytes(tail)) is None
    assert _png.decode(good[: len(good) // 2]) is None and _png.decode(b"") is None
    assert _png.read(tmp_path / "missing.png") is None


def test_imwrite_uses_the_parallel_encoder_for_large_pngs_and_round_trips(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (700, 900, 3), dtype=np.uint8)  # 1.9 MB > PARALLEL_PNG_MIN_BYTES
    p = tmp_path / "big.png"
    assert _io.imwrite(p, img)
    assert np.array_equal(_io.imread(p), img)
    small = img[:20, :30]
    assert _io.imwrite(tmp_path / "small.png", small) and np.array_equal(_io.imread(tmp_path / "small.png"), small)
    # float input is saturated like cv2.imwrite does
    f = np.array([[[-5.0, 100.4, 300.0]]])
    _io.imwrite(tmp_path / "f.png", f)
    assert np.array_equal(_io.imread(tmp_path / "f.png"), [[[0, 100, 255]]])


def test_npy_round_trip_and_unreadable_files(tmp_path):
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    p = tmp_path / "frame.npy"
    assert _io.imwrite(p, img[:, ::-1])  # a non-contiguous view
    got = _io.imread(p)
    assert np.array_equal(got, img[:, ::-1])
    assert _io.imread(tmp_path / "missing.png") is None and _io.imread(tmp_path / "missing.npy") is None
    np.save(tmp_path / "float.npy", np.zeros((4, 4), np.float32))
    assert _io.imread(tmp_path / "float.npy") is None  # images are uint8


def test_parallel_png_reader_refuses_forged_band_directories(tmp_path):
    """A crafted file must come back as None (-> the general decoder), not as a huge allocation or an inflated zip bomb: header
    dimensions far beyond what the compressed bytes could hold, and a band whose stream inflates to more than its rows."""
    import struct
    import zlib

    from vr180_convert_amd import _png

    img = (np.arange(64 * 48 * 3) % 251).astype(np.uint8).reshape(64, 48, 3)
    good = _png.encode(img, level=1)
    assert np.array_equal(_png.decode(good), img)

    def chunks(data):
        pos, out = 8, []
        while pos < len(data):
            n, kind = struct.unpack(">I4s", data[pos:pos + 8])
            out.append((kind, data[pos + 8:pos + 8 + n]))
            pos += 12 + n
        return out

    def build(parts):
        b = data_sig
        for kind, body in parts:
            b += struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
        return b

    data_sig = good[:8]
    parts = chunks(good)
    # (1) header says 30000 x 30000 for the same few hundred compressed bytes, band directory stretched to match
    forged = []
    for kind, body in parts:
        if kind == b"IHDR":
            body = struct.pack(">IIBBBBB", 30000, 30000, 8, 2, 0, 0, 0)
        elif kind == b"vrBD":
            ver, ftype, nb = struct.unpack(">BBI", body[:6])
            bands = [list(struct.unpack(">III", body[6 + 12 * k:18 + 12 * k])) for k in range(nb)]
            bands[-1][1] = 30000
            body = struct.pack(">BBI", ver, ftype, nb) + b"".join(struct.pack(">III", *b) for b in bands)
        forged.append((kind, body))
    assert _png.decode(build(forged)) is None
    # (2) a band's deflate stream replaced by one that inflates to 50 MB of zeros
    bomb = zlib.compressobj(9, zlib.DEFLATED, -15)
    payload = bomb.compress(bytes(50 << 20)) + bomb.flush()
    forged = []
    for kind, body in parts:
        if kind == b"IDAT":
            body = body[:2] + payload + body[-4:]
        elif kind == b"vrBD":
            ver, ftype, nb = struct.unpack(">BBI", body[:6])
            body = struct.pack(">BBI", ver, ftype, 1) + struct.pack(">III", 0, 64, 2)
        forged.append((kind, body))
    assert _png.decode(build(forged)) is None
    p = tmp_path / "bomb.png"
    p.write_bytes(build(forged))
    assert _png.read(p) is None
