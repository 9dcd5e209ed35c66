"""N > 1 path on CPU: two gloo ranks run the PRODUCT's multi-GPU host logic -- the partition
(``sharding.plan_shards``), chain lowering / plan grouping (``remapper.group_units``) and the
marshalling of the ``v1c_unit`` records (``remapper.marshal_units``) through
``sharding.build_rank_job`` / ``run_rank_job`` -- on host tensors.  Only the device launch is
replaced: an executor interprets the marshalled records (pointers, pitches, per-unit rotations) with
the oracle, i.e. it reads and writes exactly the bytes the GPU launch would.  The world then agrees on
totals with no data-path collective -- only the bench's timing / checksum reductions."""
import ctypes as C
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
SIZE = 32
POLY = [0, 1, -0.1]


def _chain_objects(with_rot):
    from vr180_convert_amd.transformer import EquirectangularEncoder, Euclidean3DRotator, FisheyeDecoder, PolynomialScaler

    t = EquirectangularEncoder()
    if with_rot:
        t = t * Euclidean3DRotator((1.0, 0.0, 0.0, 0.0))
    return t * PolynomialScaler(POLY) * FisheyeDecoder("equidistant")


def _rot(frame, eye):
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector

    rng = np.random.default_rng(77 + 2 * frame + eye)
    return as_rotation_matrix(from_rotation_vector(rng.normal(0, 0.05, 3)))


def _oracle_executor(interp):
    """Stands in for the device launch: consumes a LaunchGroup's chain and its marshalled v1c_unit array."""
    from oracle import oracle as O
    from vr180_convert_amd import _abi

    def launch(group, units, n):
        h_in, w_in = group.src_hw
        for k in range(n):
            u = units[k]
            ch = O.Chain.from_buffer_copy(bytes(group.chain))
            if u.has_rot:
                (i,) = [i for i in range(ch.n_ops) if ch.ops[i].opcode == _abi.OP_ROTATE]
                for q in range(9):
                    ch.ops[i].p[q] = u.rot[q]
            src = np.lib.stride_tricks.as_strided(
                np.ctypeslib.as_array(C.cast(u.src, C.POINTER(C.c_uint8)), shape=(h_in * u.src_pitch,)),
                shape=(h_in, w_in, 3), strides=(u.src_pitch, 3, 1))
            dst = np.lib.stride_tricks.as_strided(
                np.ctypeslib.as_array(C.cast(u.dst, C.POINTER(C.c_uint8)), shape=(SIZE * u.dst_pitch,)),
                shape=(SIZE, SIZE, 3), strides=(u.dst_pitch, 3, 1))
            xm, ym = O.get_map(ch, radius=0, size_input=(h_in, w_in), size_output=(SIZE, SIZE))
            O.remap(src, xm, ym, interp, dst=dst)

    return launch


def _expected(frame_np, frame, with_rot):
    from oracle import oracle as O

    def spec(eye):
        s = [("equirect_enc", True)]
        if with_rot:
            s.append(("rot", _rot(frame, eye)))
        return s + [("poly", POLY), ("fisheye_dec", "equidistant")]

    left, right = frame_np[:, :SIZE], frame_np[:, SIZE:]
    sp = (spec(0), spec(1)) if with_rot else spec(0)
    return O.apply_lr(sp, left, right, size_output=(SIZE, SIZE), interpolation=1, radius="max")


def _worker(rank, world, port, n_frames, with_rot, ret):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from vr180_convert_amd.sharding import build_rank_job, plan_shards, run_rank_job, split_sbs
        from vr180_convert_amd.synth import noise_disc

        shard = plan_shards(n_frames, world)[rank]
        # this rank's frames as host tensors: sources are the column halves of the SBS input (pitched views,
        # remapper.py:455-456), outputs the halves of the SBS result (remapper.py:517-518 as a pitch)
        ins = {f: torch.from_numpy(noise_disc(SIZE, 2 * SIZE, frame=f)) for f in shard.frames}
        outs = {f: torch.zeros((SIZE, 2 * SIZE, 3), dtype=torch.uint8) for f in shard.frames}
        sources = {(f, e): split_sbs(ins[f])[e] for f, e in shard.units}
        outputs = {(f, e): outs[f][:, e * SIZE:(e + 1) * SIZE] for f, e in shard.units}
        rots = {u: _rot(*u) for u in shard.units} if with_rot else None
        job = build_rank_job(_chain_objects(with_rot), shard, sources, outputs, radius=SIZE / 2, size_output=(SIZE, SIZE),
                             rotations=rots)
        paths = run_rank_job(job, interpolation=1, launch=_oracle_executor(1))
        n_groups = len(job.groups)
        # every unit of the shard was marshalled exactly once; per-unit rotations share ONE plan group
        assert sorted(u for ids in job.unit_ids for u in ids) == sorted(shard.units)
        assert n_groups == (1 if shard.units else 0) and len(paths) == n_groups
        if with_rot and shard.units:
            assert job.groups[0].rots is not None and all(job.units[0][k].has_rot for k in range(len(shard.units)))
        ok = True
        for f in shard.frames:
            want = _expected(ins[f].numpy(), f, with_rot)
            for e in (0, 1):
                if (f, e) in shard.units:
                    ok &= bool(np.array_equal(outs[f].numpy()[:, e * SIZE:(e + 1) * SIZE], want[:, e * SIZE:(e + 1) * SIZE]))
        checksum = sum(int(outs[f].numpy().astype(np.int64).sum()) for f in shard.frames)
        # the same reductions bench.py performs: max of the step time, sums of units / checksums
        t_max = bench.allreduce_max(float(rank + 1), torch.device("cpu"))
        tot = torch.tensor([len(shard.units), checksum, int(ok)], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ret[rank] = (list(shard.units), t_max, int(tot[0]), int(tot[1]), int(tot[2]))
    finally:
        dist.destroy_process_group()


def _run(n_frames, with_rot):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, n_frames, with_rot, ret), nprocs=2, join=True)
    return ret[0], ret[1]


def _total_checksum(n_frames, with_rot):
    sys.path.insert(0, str(ROOT))
    from vr180_convert_amd.synth import noise_disc

    return sum(int(_expected(noise_disc(SIZE, 2 * SIZE, frame=f), f, with_rot).astype(np.int64).sum()) for f in range(n_frames))


def test_two_ranks_partition_frames_and_agree():
    n_frames = 5
    r0, r1 = _run(n_frames, with_rot=False)
    assert r0[0] == [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)] and r1[0] == [(3, 0), (3, 1), (4, 0), (4, 1)]
    assert r0[1] == r1[1] == 2.0  # MAX over ranks
    assert r0[2] == r1[2] == 2 * n_frames
    assert r0[4] == r1[4] == 2  # both ranks bit-equal to the oracle's apply_lr on their frames
    assert r0[3] == r1[3] == _total_checksum(n_frames, False)


def test_single_pair_splits_eyes_over_two_ranks():
    """a single L+R pair on 2 GPUs = one eye per GPU."""
    r0, r1 = _run(1, with_rot=False)
    assert r0[0] == [(0, 0)] and r1[0] == [(0, 1)]
    assert r0[2] == 2 and r0[4] == 2
    # each rank wrote only its half: the halves' checksums add up to the whole frame's
    assert r0[3] == _total_checksum(1, False)


def test_per_unit_rotations_travel_with_the_units():
    """BASELINE config 5: per-frame, per-eye calibration rotations share one plan, the matrices are
    marshalled into the unit records."""
    n_frames = 3
    r0, r1 = _run(n_frames, with_rot=True)
    assert r0[2] == 2 * n_frames and r0[4] == r1[4] == 2
    assert r0[3] == _total_checksum(n_frames, True)


def _band_worker(rank, world, port, virtual_world, ret):
    """Two gloo processes play `virtual_world` ranks of a row-band split of ONE pair (rank r of the process takes the virtual
    ranks r, r + world, ...): the product's partition (plan_band_shards), band lowering and marshalling (build_band_job) and
    run_band_job with an executor that interprets the marshalled records with the oracle."""
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from vr180_convert_amd.sharding import build_band_job, plan_band_shards, run_band_job
        from vr180_convert_amd.synth import noise_disc

        frame = noise_disc(SIZE, 2 * SIZE, frame=0)
        eyes = {(0, e): torch.from_numpy(np.ascontiguousarray(frame[:, e * SIZE:(e + 1) * SIZE])) for e in (0, 1)}
        sbs = torch.zeros((SIZE, 2 * SIZE, 3), dtype=torch.uint8)
        work = plan_band_shards(1, virtual_world, SIZE)
        mine = [u for v in range(rank, virtual_world, world) for u in work[v]]
        outputs = {u: sbs[u[2]:u[3], u[1] * SIZE:(u[1] + 1) * SIZE] for u in mine}
        job = build_band_job(_chain_objects(False), mine, eyes, outputs, radius=SIZE / 2, size_output=(SIZE, SIZE))

        def launch(chain, units, band):
            f, e, r0, r1 = band
            u = units[0]
            src = np.lib.stride_tricks.as_strided(
                np.ctypeslib.as_array(C.cast(u.src, C.POINTER(C.c_uint8)), shape=(SIZE * u.src_pitch,)),
                shape=(SIZE, SIZE, 3), strides=(u.src_pitch, 3, 1))
            dst = np.lib.stride_tricks.as_strided(
                np.ctypeslib.as_array(C.cast(u.dst, C.POINTER(C.c_uint8)), shape=((r1 - r0) * u.dst_pitch,)),
                shape=(r1 - r0, SIZE, 3), strides=(u.dst_pitch, 3, 1))
            xm, ym = O.get_map(O.Chain.from_buffer_copy(bytes(chain)), radius=0, size_input=(SIZE, SIZE), size_output=(SIZE, r1 - r0))
            O.remap(src, xm, ym, 1, dst=dst)

        assert run_band_job(job, interpolation=1, launch=launch) == ["custom"] * len(mine)
        want = _expected(frame, 0, False)
        ok = all(np.array_equal(sbs.numpy()[r0:r1, e * SIZE:(e + 1) * SIZE], want[r0:r1, e * SIZE:(e + 1) * SIZE]) for _, e, r0, r1 in mine)
        # rows this process did not own stay untouched: the parts add up to the whole frame with no exchange
        tot = torch.tensor([len(mine), int(sbs.numpy().astype(np.int64).sum()), int(ok)], dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ret[rank] = (mine, int(tot[0]), int(tot[1]), int(tot[2]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("virtual_world", [2, 4])
def test_single_pair_splits_rows_into_bands_over_ranks(virtual_world):
    """a single pair on 4 / 8 GPUs = bands of output rows per GPU (what bench.py --split bands times)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_band_worker, args=(2, port, virtual_world, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert sorted(r0[0] + r1[0]) == [(0, e, a, b) for e in (0, 1) for a, b in ([(0, SIZE)] if virtual_world == 2 else [(0, 16), (16, 32)])]
    assert r0[1] == r1[1] == len(r0[0]) + len(r1[0]) and r0[3] == 2
    assert r0[2] == _total_checksum(1, False)


@pytest.mark.parametrize("n_frames,world", [(0, 1), (1, 1), (1, 2), (1, 8), (3, 4), (5, 2), (8, 8), (64, 8), (257, 8)])
def test_plan_shards_partition_exactly(n_frames, world):
    sys.path.insert(0, str(ROOT))
    from vr180_convert_amd.sharding import plan_shards

    shards = plan_shards(n_frames, world)
    assert len(shards) == world and [s.rank for s in shards] == list(range(world))
    flat = [u for s in shards for u in s.units]
    assert flat == [(f, e) for f in range(n_frames) for e in (0, 1)]  # exact, ordered, disjoint
    sizes = [len(s.units) for s in shards]
    if n_frames >= world:
        assert all(len(set(f for f, _ in s.units)) * 2 == len(s.units) for s in shards)  # both eyes of a frame together
        assert max(sizes) - min(sizes) <= 2
    else:
        assert max(sizes) <= 2 and sizes == sorted(sizes, reverse=True)


@pytest.mark.parametrize("height,n", [(4096, 4), (100, 3), (30, 8), (16, 4), (2880, 7)])
def test_row_bands_cover_the_rows(height, n):
    sys.path.insert(0, str(ROOT))
    from vr180_convert_amd.sharding import plan_band_shards, row_bands

    bands = row_bands(height, n)
    assert bands[0][0] == 0 and bands[-1][1] == height and len(bands) <= n
    assert all(a[1] == b[0] for a, b in zip(bands, bands[1:])) and all(r1 > r0 for r0, r1 in bands)
    assert all(r0 % 16 == 0 for r0, _ in bands)
    work = plan_band_shards(1, 2 * n, height)
    flat = sorted(u for rank in work for u in rank)
    assert flat == [(0, e, r0, r1) for e in (0, 1) for (r0, r1) in bands]
    assert max(len(r) for r in work) == 1  # one band per rank


def test_row_band_chain_reproduces_the_rows_of_the_full_map(emul_lib, oracle_mod):
    """One eye's output rows split over GPUs: the chain of a band -- the same Normalize with its
    centre moved up by r0 rows -- gives rows r0 .. r1 - 1 of the full map bit for bit (product lowering + the
    product's per-pixel code compiled for the host; literal interpreter and fused ray path)."""
    sys.path.insert(0, str(ROOT))
    import ctypes as C

    import chainspecs as CS
    from test_host_emul import emul_map
    from vr180_convert_amd.chain import lower_for_get_map

    O = oracle_mod
    W, H = 96, 80
    # (round 5: planar chains too -- their radial table covers m = xn^2 + yn^2 up to the corners of the WHOLE grid, which a band learns
    #  from its Normalize stage, so that every band evaluates the unsplit plan's polynomials -- and the general modes)
    for spec in ([("equirect_enc", True), ("rot", CS.ry(0.3)), ("poly", [0, 1, -0.1]), CS.EQUI],
                 [("fisheye_enc", "stereographic"), ("poly", [0, 1, -0.1]), CS.EQUI],
                 [("fisheye_enc", "equidistant"), ("rot", CS.ry(0.3)), CS.EQUI],
                 [("equirect_enc", False), CS.EQUI]):
        t = CS.to_product(spec)
        full = O.Chain.from_buffer_copy(bytes(lower_for_get_map(t, radius=41.5, size_input=(90, 100), size_output=(W, H))))
        for r0, r1 in ((32, 64), (0, 16), (64, 80)):
            band = O.Chain.from_buffer_copy(bytes(lower_for_get_map(t, radius=41.5, size_input=(90, 100), size_output=(W, H), row_band=(r0, r1))))
            for mode in (0, 1):
                rc, fx, fy, _ = emul_map(emul_lib, full, W, H, mode)
                rc2, bx, by, _ = emul_map(emul_lib, band, W, r1 - r0, mode)
                assert rc == 0 and rc2 == 0, (spec, mode)
                assert np.array_equal(bx.view(np.uint32), fx[r0:r1].view(np.uint32)) and np.array_equal(by.view(np.uint32), fy[r0:r1].view(np.uint32)), (spec, r0, mode)
    t = CS.to_product([("equirect_enc", True), ("rot", CS.ry(0.3)), ("poly", [0, 1, -0.1]), CS.EQUI])
    with pytest.raises(ValueError):
        lower_for_get_map(t, radius=41.5, size_input=(90, 100), size_output=(W, H), row_band=(64, 96))
