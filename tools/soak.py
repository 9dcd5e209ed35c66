"""Soak of the LDS-DMA kernels: every step's output compared on the device with the first output of the same inputs."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import vr180_convert_amd as V
from vr180_convert_amd.synth import noise_disc_torch
from vr180_convert_amd.transformer import EquirectangularEncoder, FisheyeDecoder, PolynomialScaler
dev = torch.device("cuda", 0)
def pair(n, poly, steps):
    t = EquirectangularEncoder()
    if poly: t = t * PolynomialScaler(poly)
    t = t * FisheyeDecoder("equidistant")
    sets = [(noise_disc_torch(n, n, k, dev), noise_disc_torch(n, n, k + 50, dev)) for k in range(3)]
    refs = [V.apply_lr_tensors(t, a, b, size_output=(n, n), interpolation=1, radius="max").clone() for a, b in sets]
    out = torch.empty_like(refs[0]); bad = 0
    # interleave with an unrelated kernel now and then to vary the timing
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for i in range(steps):
        a, b = sets[i % 3]
        V.apply_lr_tensors(t, a, b, out=out, size_output=(n, n), interpolation=1, radius="max")
        if i % 7 == 0: junk.add_(1)
        bad += int(not torch.equal(out, refs[i % 3]))
    return bad
def batch(n, frames, steps):
    t = EquirectangularEncoder() * FisheyeDecoder("equidistant")
    ins = [noise_disc_torch(n, 2 * n, 10 + f, dev) for f in range(frames)]
    srcs = [v for fr in ins for v in (fr[:, :n], fr[:, n:])]
    def run():
        outs = [torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev) for _ in range(frames)]
        dsts = [v for fr in outs for v in (fr[:, :n], fr[:, n:])]
        V.remap_tensors(t, srcs, dsts, radius=n / 2, interpolation=1)
        return torch.stack(outs)
    ref = run().clone(); bad = 0
    for i in range(steps):
        bad += int(not torch.equal(run(), ref))
    return bad
t0 = time.time()
print("C2-like pairs 4096:", pair(4096, [0, 1, -0.1], 1500), "bad of 1500", flush=True)
print("C1-like pairs 2048:", pair(2048, None, 3000), "bad of 3000", flush=True)
print("pairs 1024 poly:", pair(1024, [0, 1, -0.1], 4000), "bad of 4000", flush=True)
print("C3-like batches 2880 x 8 frames:", batch(2880, 8, 300), "bad of 300", flush=True)
print("batches 1440 x 5 frames:", batch(1440, 5, 1000), "bad of 1000", flush=True)
print(f"{time.time() - t0:.0f} s")
