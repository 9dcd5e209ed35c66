"""Seeded synthetic fisheye frames for tests and benchmarks (SURVEY.md 8d "Synthetic pixels").

``noise_disc``: uniform uint8 noise inside the inscribed circle, black outside -- the worst case
for parity (any 1/32-pixel bucket flip changes the output by several levels).
``pattern``: smooth rings + spokes, black outside -- in the spirit of the reference's
``generate_test_image`` (testing.py:11-61, which needs cv2 drawing and is not restated).
"""
from __future__ import annotations

import numpy as np

SEED = 20240619


def _disc_mask(h: int, w: int) -> np.ndarray:
    yy, xx = np.ogrid[:h, :w]
    r = min(h, w) / 2
    return (xx - w / 2 + 0.5) ** 2 + (yy - h / 2 + 0.5) ** 2 <= r * r


def noise_disc(h: int, w: int, frame: int = 0, cn: int = 3) -> np.ndarray:
    rng = np.random.default_rng(SEED + frame)
    img = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
    img[~_disc_mask(h, w)] = 0
    return img


def pattern(h: int, w: int, rings: int = 10, spokes: int = 24) -> np.ndarray:
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    dx, dy = xx - w / 2 + 0.5, yy - h / 2 + 0.5
    r = np.hypot(dx, dy) / (min(h, w) / 2)
    ang = np.arctan2(dy, dx)
    b = 127.5 * (1 + np.cos(2 * np.pi * rings * r))
    g = 127.5 * (1 + np.cos(spokes * ang))
    rch = 255 * np.clip(1 - r, 0, 1)
    img = np.stack([b, g, rch], axis=-1)
    img[r > 1] = 0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def noise_disc_torch(h: int, w: int, frame: int, device, cn: int = 3):
    """Same distribution generated on the device (bench-sized inputs without a PCIe upload; NOT
    bit-identical to ``noise_disc`` -- parity tests use the numpy version)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(SEED + frame)
    img = torch.randint(0, 256, (h, w, cn), dtype=torch.uint8, device=device, generator=g)
    yy = torch.arange(h, device=device, dtype=torch.float32).view(h, 1)
    xx = torch.arange(w, device=device, dtype=torch.float32).view(1, w)
    r = min(h, w) / 2
    mask = (xx - w / 2 + 0.5) ** 2 + (yy - h / 2 + 0.5) ** 2 <= r * r
    return img * mask.unsqueeze(-1).to(torch.uint8)
