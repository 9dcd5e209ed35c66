#!/usr/bin/env python3
"""End-to-end rate of apply() on HOST-resident images (PCIe both ways included; never bench.py's
`value`): python3 tools/host_io_bench.py [--frames 16] [--size 2880] [--interp 1]
Run twice to compare: V1C_HOST_PIPELINE=0 python3 tools/host_io_bench.py"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd.synth import noise_disc  # noqa: E402
from vr180_convert_amd.transformer import EquirectangularEncoder, FisheyeDecoder  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--size", type=int, default=2880)
ap.add_argument("--interp", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
t = EquirectangularEncoder() * FisheyeDecoder("equidistant")
imgs = [noise_disc(a.size, a.size, f) for f in range(a.frames)]
V.apply(t, in_paths=imgs[:2], size_output=(a.size, a.size), interpolation=a.interp, radius="max")  # plan + warm-up
best = 1e9
for _ in range(a.reps):
    t0 = time.perf_counter()
    out = V.apply(t, in_paths=imgs, size_output=(a.size, a.size), interpolation=a.interp, radius="max")
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
mpx = a.frames * a.size * a.size / 1e6
gb = a.frames * a.size * a.size * 3 * 2 / 1e9
print(f"pipeline={os.environ.get('V1C_HOST_PIPELINE', '1')} frames={a.frames} size={a.size}: {best * 1e3:.1f} ms  "
      f"{mpx / best / 1e3:.2f} Gpx/s  {gb / best:.1f} GB/s over PCIe (in + out)")
