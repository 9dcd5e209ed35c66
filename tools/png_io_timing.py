"""Time the PNG writer / reader of vr180_convert_amd._png against Pillow on an 8192 x 4096 side-by-side frame
(codecs are outside the measured path; this is the end-to-end apply_lr(png -> png) figure of HISTORY.md 6)."""
import io
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from vr180_convert_amd import _png  # noqa: E402


def photo_like(h, w, seed):
    """smooth gradients + fine noise: deflates about like a photograph (level 1: ~45 % of the raw size)"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = np.stack([128 + 100 * np.sin(x / 311 + k) * np.cos(y / 173 - k) for k in range(3)], axis=-1)
    return np.clip(base + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t)
    return min(ts), r


img = photo_like(4096, 8192, 1)
te, data = best(lambda: _png.encode(img))
td, back = best(lambda: _png.decode(data))
assert np.array_equal(back, img)
from PIL import Image  # noqa: E402


def pillow():
    with Image.open(io.BytesIO(data)) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


tp, back2 = best(pillow)
assert np.array_equal(back2, img)
print(f"8192 x 4096 BGR, {len(data) / 1e6:.1f} MB PNG: encode {te * 1e3:.0f} ms, decode (band directory, threads) {td * 1e3:.0f} ms, "
      f"Pillow decode {tp * 1e3:.0f} ms")
