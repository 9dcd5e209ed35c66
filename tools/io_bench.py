#!/usr/bin/env python3
"""End-to-end time of apply_lr with files either side: C2-sized pair (2 x 4096^2 in, 8192 x 4096 out)
for PNG / JPEG / NPY, per stage -- what a CLI user of the reference waits for.  python3 tools/io_bench.py [size]"""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd import _io, _png  # noqa: E402
from vr180_convert_amd.synth import pattern  # noqa: E402
from vr180_convert_amd.transformer import EquirectangularEncoder, FisheyeDecoder, PolynomialScaler  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = EquirectangularEncoder() * PolynomialScaler([0, 1, -0.1]) * FisheyeDecoder("equidistant")
img = pattern(n, n)  # a smooth test card (photo-like compressibility), not noise
rng = np.random.default_rng(0)
img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, img.shape), 0, 255).astype(np.uint8)  # sensor-like noise
dev = torch.device("cuda", 0)
res = {"size": n, "formats": {}}


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best, out


with tempfile.TemporaryDirectory(dir=(sys.argv[2] if len(sys.argv) > 2 else None)) as d:
    d = Path(d)
    V.apply_lr(t, left_path=img, right_path=img, out_path=None, size_output=(n, n), interpolation=1, radius="max")  # plan + warm-up
    import resource

    fsize = resource.getrlimit(resource.RLIMIT_FSIZE)[0]
    res["rlimit_fsize"] = fsize
    for ext in ("png", "jpg", "npy"):
        if ext == "npy" and fsize != resource.RLIM_INFINITY and fsize < 6 * n * n + 4096:
            res["formats"][ext] = {"skipped": f"RLIMIT_FSIZE {fsize} < the {6 * n * n} byte raw result"}
            continue
        left, right, out = d / f"l.{ext}", d / f"r.{ext}", d / f"out.{ext}"
        _io.imwrite(left, img), _io.imwrite(right, img)
        t_read, (a, b) = timed(lambda: _io.imread_many([left, right]))
        t_total, _ = timed(lambda: V.apply_lr(t, left_path=left, right_path=right, out_path=out, size_output=(n, n), interpolation=1, radius="max"))
        sbs = np.array(_io.imread(out))  # (a copy: .npy files come back memory-mapped, and the file is rewritten next)
        t_write, _ = timed(lambda: _io.imwrite(out, sbs))
        res["formats"][ext] = {"read_2_files_s": round(t_read, 3), "write_sbs_s": round(t_write, 3), "apply_lr_total_s": round(t_total, 3),
                               "out_file_mb": round(out.stat().st_size / 1e6, 1), "end_to_end_mpx_s": round(2 * n * n / 1e6 / t_total, 1)}
    # the PNG writers side by side on the SBS result
    sbs = np.array(_io.imread(d / "out.png"))
    from PIL import Image

    t_pil, _ = timed(lambda: Image.fromarray(np.ascontiguousarray(sbs[..., ::-1])).save(d / "pil.png", compress_level=1), reps=2)
    t_par, _ = timed(lambda: _png.write(d / "par.png", sbs, level=1), reps=2)
    res["png_writer"] = {"pillow_level1_s": round(t_pil, 3), "parallel_level1_s": round(t_par, 3),
                         "pillow_mb": round((d / "pil.png").stat().st_size / 1e6, 1), "parallel_mb": round((d / "par.png").stat().st_size / 1e6, 1)}
    # device-resident remap alone, for scale
    lt, rt = torch.from_numpy(img).to(dev), torch.from_numpy(img).to(dev)
    o = torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)
    t_k, _ = timed(lambda: [V.apply_lr_tensors(t, lt, rt, out=o, size_output=(n, n), interpolation=1, radius="max") for _ in range(100)], reps=3)
    res["device_resident_remap_s"] = round(t_k / 100, 6)
print(json.dumps(res, indent=1))
