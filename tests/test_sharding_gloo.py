"""N > 1 path on CPU: two gloo ranks partition the frames of a batch exactly like bench.py's
ranks do on RCCL, each processes ITS frames (here: through the oracle, standing in for the GPU),
and the world agrees on totals with no data-path collective -- only the bench's timing / checksum
reductions."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, n_frames, ret):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from vr180_convert_amd.sharding import shard_range
        from vr180_convert_amd.synth import noise_disc
        import bench

        spec = [("equirect_enc", True), ("fisheye_dec", "equidistant")]
        mine = shard_range(n_frames, rank, world)
        checksum = 0
        for f in mine:
            sbs = noise_disc(32, 64, frame=f)
            out = O.apply_lr(spec, sbs[:, :32], sbs[:, 32:], size_output=(32, 32), interpolation=1, radius="max")
            checksum += int(out.astype(np.int64).sum())
        # the same reductions bench.py performs: max of the step time, sums of units / checksums
        t_max = bench.allreduce_max(float(rank + 1), torch.device("cpu"))
        tot = torch.tensor([len(mine), checksum], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ret[rank] = (list(mine), t_max, int(tot[0]), int(tot[1]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_partition_and_agree():
    n_frames = 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_frames, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0[0] == [0, 1, 2] and r1[0] == [3, 4]
    assert r0[1] == r1[1] == 2.0  # MAX over ranks
    assert r0[2] == r1[2] == n_frames
    # single-process reference of the checksum
    sys.path.insert(0, str(ROOT))
    from oracle import oracle as O
    from vr180_convert_amd.synth import noise_disc

    spec = [("equirect_enc", True), ("fisheye_dec", "equidistant")]
    total = 0
    for f in range(n_frames):
        sbs = noise_disc(32, 64, frame=f)
        total += int(O.apply_lr(spec, sbs[:, :32], sbs[:, 32:], size_output=(32, 32), interpolation=1, radius="max").astype(np.int64).sum())
    assert r0[3] == r1[3] == total
