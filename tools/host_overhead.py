#!/usr/bin/env python3
"""Where the host time of a device-resident apply_lr_tensors call goes (cProfile over 3000 calls of the
C1 shape, kernel 25 us): python3 tools/host_overhead.py"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd.synth import noise_disc_torch  # noqa: E402
from vr180_convert_amd.transformer import EquirectangularEncoder, FisheyeDecoder  # noqa: E402

dev = torch.device("cuda", 0)
n = 2048
t = EquirectangularEncoder() * FisheyeDecoder("equidistant")
sets = [(noise_disc_torch(n, n, k, dev), noise_disc_torch(n, n, k + 50, dev), torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)) for k in range(4)]


def step(i):
    a, b, o = sets[i % 4]
    V.apply_lr_tensors(t, a, b, out=o, size_output=(n, n), interpolation=1, radius="max")


for i in range(50):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3000):
    step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e6 * (t1 - t0) / 3000:.1f} us/call, with drain {1e6 * (t2 - t0) / 3000:.1f} us/call")
pr = cProfile.Profile()
pr.enable()
for i in range(3000):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
