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
def single(n, steps):
    """one image: the one-eye instantiation of k_ray_lin3_pair_mirror_raw"""
    t = EquirectangularEncoder() * FisheyeDecoder("equidistant")
    srcs = [noise_disc_torch(n, n, 20 + k, dev) for k in range(3)]
    def run(src):
        dst = torch.empty((n, n, 3), dtype=torch.uint8, device=dev)
        V.remap_tensors(t, [src], [dst], radius=n / 2, interpolation=1)
        return dst
    refs = [run(s).clone() for s in srcs]; bad = 0
    for i in range(steps):
        bad += int(not torch.equal(run(srcs[i % 3]), refs[i % 3]))
    return bad
def rotated(n, frames, steps):
    """per-unit calibration rotations: k_ray_lin3_rot_pair_raw (two units per workgroup, fence-less box reduction + LDS-DMA)"""
    import numpy as np
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector
    from vr180_convert_amd.transformer import Euclidean3DRotator
    t = EquirectangularEncoder() * Euclidean3DRotator((1.0, 0.0, 0.0, 0.0)) * FisheyeDecoder("equidistant")
    rng = np.random.default_rng(3)
    rots = [as_rotation_matrix(from_rotation_vector(rng.normal(0, 0.02, 3))) for _ in range(2 * frames)]
    ins = [noise_disc_torch(n, 2 * n, 40 + f, dev) for f in range(frames)]
    srcs = [v for fr in ins for v in (fr[:, :n], fr[:, n:])]
    def run():
        outs = [torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev) for _ in range(frames)]
        dsts = [v for fr in outs for v in (fr[:, :n], fr[:, n:])]
        V.remap_tensors(t, srcs, dsts, radius=n / 2, interpolation=1, rotations=rots)
        return torch.stack(outs)
    ref = run().clone(); bad = 0
    for i in range(steps):
        bad += int(not torch.equal(run(), ref))
    return bad
def cn_batch(n, cn, units, steps, interp=1, rot=False):
    """grayscale / BGRA units sharing the map: k_ray_lin_cn (double-buffered LDS-DMA boxes behind bare barriers); `interp` 0 / 4: its
    NEAREST / Lanczos4 forms; `rot`: a rotation per unit (boxes reduced in the kernel)"""
    import numpy as np
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector
    from vr180_convert_amd.transformer import Euclidean3DRotator
    t = EquirectangularEncoder() * PolynomialScaler([0, 1, -0.1]) * FisheyeDecoder("equidistant")
    rots = None
    if rot:
        t = EquirectangularEncoder() * Euclidean3DRotator((1.0, 0.0, 0.0, 0.0)) * FisheyeDecoder("equidistant")
        rng = np.random.default_rng(4)
        rots = [as_rotation_matrix(from_rotation_vector(rng.normal(0, 0.02, 3))) for _ in range(units)]
    g = torch.Generator(device=dev).manual_seed(5)
    srcs = [torch.randint(0, 256, (n, n, cn), dtype=torch.uint8, device=dev, generator=g) for _ in range(units)]
    def run():
        dsts = [torch.empty((n, n, cn), dtype=torch.uint8, device=dev) for _ in range(units)]
        V.remap_tensors(t, srcs, dsts, radius=n / 2, interpolation=interp, rotations=rots)
        return torch.stack(dsts)
    ref = run().clone(); bad = 0
    for i in range(steps):
        bad += int(not torch.equal(run(), ref))
    return bad
import os
REPS = int(os.environ.get("SOAK_REPS", "1"))  # SOAK_REPS=<n>: the whole list n times
t0 = time.time()
for rep in range(REPS):
    print("gray 2048 x 5 units:", cn_batch(2048, 1, 5, 1000), "bad of 1000", flush=True)
    print("BGRA 2048 x 2 units:", cn_batch(2048, 4, 2, 1000), "bad of 1000", flush=True)
    print("BGRA 1024 x 7 units:", cn_batch(1024, 4, 7, 1000), "bad of 1000", flush=True)
    print("gray 1024 x 4 units NEAREST:", cn_batch(1024, 1, 4, 500, interp=0), "bad of 500", flush=True)
    print("BGRA 1024 x 3 units Lanczos4:", cn_batch(1024, 4, 3, 300, interp=4), "bad of 300", flush=True)
    # (K x K taps take the units two at a time, both boxes resident, one fetch of the weight rows: pairs, an odd count, many)
    print("gray 1024 x 2 units Lanczos4:", cn_batch(1024, 1, 2, 500, interp=4), "bad of 500", flush=True)
    print("gray 1024 x 7 units bicubic:", cn_batch(1024, 1, 7, 300, interp=2), "bad of 300", flush=True)
    print("BGRA 1024 x 16 units bicubic:", cn_batch(1024, 4, 16, 150, interp=2), "bad of 150", flush=True)
    print("BGRA 1024 x 6 units, a rotation each:", cn_batch(1024, 4, 6, 500, rot=True), "bad of 500", flush=True)
    print("single images 2048:", single(2048, 2000), "bad of 2000", flush=True)
    print("rotated units 1440 x 16 frames (32 units: one launch through the unit ring):", rotated(1440, 16, 300), "bad of 300", flush=True)
    print("batches 1024 x 12 frames (24 units through the ring):", batch(1024, 12, 500), "bad of 500", flush=True)
    print("C5-like rotated units 1920 x 8 frames:", rotated(1920, 8, 400), "bad of 400", flush=True)
    print("rotated units 1024 x 5 frames (odd unit count per launch group):", rotated(1024, 5, 800), "bad of 800", flush=True)
    print("C2-like pairs 4096:", pair(4096, [0, 1, -0.1], 1500), "bad of 1500", flush=True)
    print("C1-like pairs 2048:", pair(2048, None, 3000), "bad of 3000", flush=True)
    print("pairs 1024 poly:", pair(1024, [0, 1, -0.1], 4000), "bad of 4000", flush=True)
    print("C3-like batches 2880 x 8 frames:", batch(2880, 8, 300), "bad of 300", flush=True)
    print("batches 1440 x 5 frames:", batch(1440, 5, 1000), "bad of 1000", flush=True)
print(f"{time.time() - t0:.0f} s")
