#!/usr/bin/env python3
"""Per-phase cycle sums of the tile kernel (diagnostic build -DV1C_STAMPS, V1C_LIB=<that build>):
python3 tools/stamps.py [C2]   -- shares of a wave's lifetime per phase (lane 0 of every wave)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd import _native  # noqa: E402
from vr180_convert_amd.remapper import _PLANS  # noqa: E402
from vr180_convert_amd.synth import noise_disc_torch  # noqa: E402
from vr180_convert_amd.transformer import EquirectangularEncoder, FisheyeDecoder, PolynomialScaler  # noqa: E402

dev = torch.device("cuda", 0)
n = 4096
t = EquirectangularEncoder() * PolynomialScaler([0, 1, -0.1]) * FisheyeDecoder("equidistant")
l, r = noise_disc_torch(n, n, 0, dev), noise_disc_torch(n, n, 1, dev)
out = torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)
for _ in range(3):
    V.apply_lr_tensors(t, l, r, out=out, size_output=(n, n), interpolation=1, radius="max")
torch.cuda.synchronize()
plan = next(iter(_PLANS.values()))
lib = _native.lib()
buf = (C.c_ulonglong * 8)()
lib.v1c_debug_read_stamps(plan._h, buf)  # clears
reps = 20
for _ in range(reps):
    V.apply_lr_tensors(t, l, r, out=out, size_output=(n, n), interpolation=1, radius="max")
torch.cuda.synchronize()
lib.v1c_debug_read_stamps(plan._h, buf)
names = ["setup + issue loads", "wait loads + LDS stores", "barrier", "coordinates", "tap reads", "blend+store eye 0", "blend+store eye 1", "-"]
tot = sum(buf)
for k in range(7):  # (shares only: the counter's unit is not the shader clock)
    print(f"{names[k]:26s} {100.0 * buf[k] / tot:5.1f} %")
