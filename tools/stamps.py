#!/usr/bin/env python3
"""Per-phase cycle sums of the tile kernel (diagnostic build -DV1C_STAMPS, V1C_LIB=<that build>):
python3 tools/stamps.py [workload ...]   -- shares of a wave's lifetime per phase (lane 0 of every wave) for bench.py's workloads
(default C2R; the pair code of k_ray_lin3_tile: rotated / general-mode / K x K pairs -- the mirror and batch kernels carry no stamps)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd import _native  # noqa: E402
from vr180_convert_amd import remapper as R  # noqa: E402
from vr180_convert_amd.synth import noise_disc_torch  # noqa: E402

dev = torch.device("cuda", 0)
names = ["setup + issue loads", "wait loads + LDS stores", "barrier", "coordinates", "tap reads / unit 0", "blend+store eye 0 / unit 1", "blend+store eye 1", "-"]
for wl in sys.argv[1:] or ["C2R"]:
    cfg = bench.WORKLOADS[wl]
    n = cfg["size"]
    t = bench.build_transformer(cfg)
    l, r = noise_disc_torch(n, n, 0, dev), noise_disc_torch(n, n, 1, dev)
    out = torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)
    R.clear_caches()

    def call():
        V.apply_lr_tensors(t, l, r, out=out, size_output=(n, n), interpolation=cfg["interp"], radius=n / 2)

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    plan = next(iter(R._PLANS.values()))
    lib = _native.lib()
    buf = (C.c_ulonglong * 8)()
    lib.v1c_debug_read_stamps(plan._h, buf)  # clears
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    lib.v1c_debug_read_stamps(plan._h, buf)
    tot = sum(buf) or 1
    print(f"== {wl} ({R.last_launch_kinds()})")
    for k in range(7):  # (shares only: the counter's unit is not the shader clock)
        print(f"{names[k]:28s} {100.0 * buf[k] / tot:5.1f} %   {buf[k] / 20:14.0f}")
