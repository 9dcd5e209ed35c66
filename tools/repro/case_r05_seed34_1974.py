"""Reproduction of tools/fuzz.py --seed 34 case 1974 (round 5): a zoom stage in front of two rotations (gen_mode 2), bicubic,
BORDER_TRANSPARENT, a 154 x 1427 output from a 192 x 192 source.  Prints the pixels of unit 0 that differ from the oracle."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import chainspecs as CS  # noqa: E402
import vr180_convert_amd as V  # noqa: E402
from oracle import oracle as O  # noqa: E402
from vr180_convert_amd import remapper as R  # noqa: E402

spec = [('equirect_enc', True), ('zoom', 1.7738941798002221),
        ('rot', [[0.9982728060815481, 0.058690132718145105, 0.002621633002223754], [-0.05872470490000005, 0.9981458957992685, 0.01600561568587129],
                 [-0.0016774005126221434, -0.016131925508209265, 0.9998684650027312]]),
        ('rot', [[0.9999105236968615, -0.00792577709040616, 0.010776207949987819], [0.007884260865708426, 0.999961353887725, 0.003889622299068912],
                 [-0.010806619770753778, -0.003804311835424161, 0.999934369936642]]), ('fisheye_dec', 'equidistant')]
wo, ho, ws, hs, radius = 154, 1427, 192, 192, 96.0
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
img = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
fill = rng.integers(0, 256, (ho, wo, 3), dtype=np.uint8)
xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
t = CS.to_product(spec)
import os
for interp in [int(v) for v in os.environ.get("INTERPS", "2,4,1").split(",")]:
    for border in (5, 0):
        want = O.remap(img, xm, ym, interp, border, (1, 2, 3), dst=fill.copy())
        dst = torch.from_numpy(fill.copy()).to(dev)
        V.remap_tensors(t, [torch.from_numpy(img).to(dev)], [dst], radius=radius, interpolation=interp, boarder_mode=border, boarder_value=(1, 2, 3))
        got = dst.cpu().numpy()
        d = np.argwhere((got != want).any(axis=2))
        print(f"interp {interp} border {border} kinds {R.last_launch_kinds()}: {len(d)} pixels differ", [(int(j), int(i), float(xm[j, i]), float(ym[j, i])) for j, i in d[:6]], flush=True)
# which source pixel did the kernel read?  INTER_NEAREST of an image that encodes its own coordinates
yy, xx = np.mgrid[:hs, :ws]
pos = np.stack([xx, yy, np.zeros_like(xx)], axis=2).astype(np.uint8)
want = O.remap(pos, xm, ym, 0, 0, (255, 255, 255), dst=fill.copy())
dst = torch.from_numpy(fill.copy()).to(dev)
V.remap_tensors(t, [torch.from_numpy(pos).to(dev)], [dst], radius=radius, interpolation=0, boarder_mode=0, boarder_value=(255, 255, 255))
got = dst.cpu().numpy()
d = np.argwhere((got != want).any(axis=2))
print("nearest on a position image:", R.last_launch_kinds(), len(d), "pixels differ")
for j, i in d[:8]:
    print(f"  ({j}, {i}) map ({xm[j, i]:.4f}, {ym[j, i]:.4f}) kernel read {got[j, i].tolist()} oracle {want[j, i].tolist()}")
for i in range(88, 100):
    print(f"  row 872 col {i}: map ({xm[872, i]:.3f}, {ym[872, i]:.3f}) kernel {got[872, i].tolist()}")
