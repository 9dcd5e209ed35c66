"""Plan creation time for a stream of images whose radius moves by half a pixel (radius="auto"): the first plan of a chain fits its radial tables,
the following ones take them from the per-process cache (HISTORY.md 6).  V1C_LIB=<other build> for a comparison."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import chainspecs as CS
import vr180_convert_amd as V
from vr180_convert_amd import remapper as R
from vr180_convert_amd.synth import noise_disc_torch
dev = torch.device("cuda", 0)
for name, n in (("C2", 4096), ("C1", 2048), ("C4", 4096)):
    spec = CS.FULL_CASES[name][0]
    t = CS.to_product(spec)
    a, b = noise_disc_torch(n, n, 1, dev), noise_disc_torch(n, n, 2, dev)
    ms = []
    for k in range(6):
        R.clear_caches()
        V.apply_lr_tensors(t, a, b, size_output=(n, n), interpolation=1 if name != "C4" else 4, radius=n / 2 - 0.5 * k)
        torch.cuda.synchronize()
        ms.append(round(sum(p.create_ms for p in R._PLANS.values()), 2))
    print(name, "plan_create_ms for radii r, r-0.5, ...:", ms, flush=True)
