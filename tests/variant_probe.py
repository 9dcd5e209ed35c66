"""Helper of test_gpu_parity.py::test_kernel_variants_bit_exact: one pair, one odd batch and one
per-unit-rotation batch through the product, compared with the oracle; run in a subprocess because
the engine reads V1C_UPB / V1C_DISABLE_* once per process.  Prints 'OK' on success."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]

import chainspecs as CS  # noqa: E402
import vr180_convert_amd as V  # noqa: E402
from oracle import oracle as O  # noqa: E402
from vr180_convert_amd.synth import noise_disc  # noqa: E402

O.build()
dev = torch.device("cuda", 0)
size = 320
poly = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
equi = [("equirect_enc", True), CS.EQUI]
for interp in (1, 2):
    # pair (apply_lr): PAIR instantiation unless V1C_UPB = 1
    l, r = noise_disc(size, size, 1), noise_disc(size, size, 2)
    got = V.apply_lr_tensors(CS.to_product(poly), torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev),
                             size_output=(size, size), interpolation=interp, radius="max").cpu().numpy()
    assert np.array_equal(got, O.apply_lr(poly, l, r, size_output=(size, size), interpolation=interp, radius="max")), ("pair", interp)
    # odd batch sharing one map: batch loop / several workgroup groups
    imgs = [noise_disc(size, size, 10 + f) for f in range(5)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.empty_like(s) for s in srcs]
    V.remap_tensors(CS.to_product(equi), srcs, dsts, radius=size / 2, interpolation=interp)
    want = O.apply(equi, imgs, size_output=(size, size), interpolation=interp, radius=size / 2)
    for f in range(5):
        assert np.array_equal(dsts[f].cpu().numpy(), want[f]), ("batch", interp, f)
    # per-unit rotations: in-kernel boxes
    from vr180_convert_amd import transformer as T

    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(4)]
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    dsts = [torch.empty_like(s) for s in srcs[:4]]
    V.remap_tensors(base, srcs[:4], dsts, radius=size / 2, interpolation=interp, rotations=quats)
    for f in range(4):
        want = O.apply(CS.c5_spec(f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=interp, radius=size / 2)[0]
        assert np.array_equal(dsts[f].cpu().numpy(), want), ("rot", interp, f)
print("OK")
