"""Time of a bicubic / Lanczos4 apply_lr pair against the radius: f * n / 2 for f = 1 (the image circle touches the frame of the source --
radius="max" -- and whole tiles next to the poles take the per-pixel patch path) down to 0.9.  HISTORY.md 4.5; profiles/r04d_final/kxk_border_tiles.log."""
import sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import chainspecs as CS
import vr180_convert_amd as V
from vr180_convert_amd import remapper
from vr180_convert_amd.synth import noise_disc_torch
dev = torch.device("cuda", 0)
spec = [("equirect_enc", True), CS.EQUI]
t = CS.to_product(spec)
for n in (2048, 4096):
    a, b = noise_disc_torch(n, n, 1, dev), noise_disc_torch(n, n, 2, dev)
    out = torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)
    for interp in (4, 2):
        for rf in (1.0, 0.999, 0.995, 0.99, 0.95, 0.9):
            r = rf * n / 2
            for _ in range(3):
                V.apply_lr_tensors(t, a, b, out=out, size_output=(n, n), interpolation=interp, radius=r)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                V.apply_lr_tensors(t, a, b, out=out, size_output=(n, n), interpolation=interp, radius=r)
            e1.record(); torch.cuda.synchronize()
            print(f"n={n} interp={interp} radius={rf:.3f} x n/2: {e0.elapsed_time(e1) / 20:.4f} ms {remapper.last_launch_kinds()}", flush=True)
