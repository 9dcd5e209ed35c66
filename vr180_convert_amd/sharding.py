"""Multi-GPU dispatch of the hot path: units over devices, no data-path collective.

Every (frame, eye) is an independent unit; the reference loops over them serially
(remapper.py:388-398, the two eyes :448-484).  Here they are dealt to the GPUs of a node
(SURVEY.md 8e):

* ``shard_range`` / ``plan_shards`` -- the partition.  With at least as many frames as ranks every
  rank takes a contiguous, balanced block of FRAMES (both eyes of a frame end up in one side-by-side
  buffer, so they stay together); with fewer frames than ranks -- a single L+R pair on 2 GPUs -- the
  EYES are dealt, one eye per GPU; with more ranks than eyes (a single pair on 4 / 8 GPUs) ``plan_band_shards``
  cuts every eye's output rows into bands (a band = an ordinary plan of a chain whose Normalize centre is moved,
  ``chain.lower_for_get_map(row_band=...)``).  The partition needs no communication: every rank computes it.
* ``build_rank_job`` -- host logic of one rank's share: lowered chains, plan groups and the marshalled
  ``v1c_unit`` records (pointers / pitches into the rank's source views and SBS halves).  No device
  is involved, so the multi-rank CPU tests (gloo, tests/test_sharding_gloo.py) run exactly this code.
* ``run_rank_job`` -- launches a job on this process's device (one process per GPU: ``bench.py``,
  torch.distributed.run).
* ``remap_sharded`` -- one process driving several GPUs: a worker thread and stream per device, host
  images staged through page-locked rings (``_hostpipe``), results assembled side by side.

Nothing here exchanges pixels between GPUs; the only collective a multi-rank run needs is whatever the
caller uses to agree on timing or checksums (bench.py: one MAX).
"""
from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Any, Callable, Sequence

import numpy as np


def shard_range(n_items: int, rank: int, world_size: int) -> range:
    """Contiguous balanced block of ``range(n_items)`` owned by ``rank`` (first ranks get the
    remainder).  Blocks of all ranks partition ``range(n_items)`` exactly."""
    if world_size < 1 or not 0 <= rank < world_size or n_items < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_sizes(n_items: int, world_size: int) -> list[int]:
    return [len(shard_range(n_items, r, world_size)) for r in range(world_size)]


@dataclass(frozen=True)
class Shard:
    """The units of one rank: ``(frame, eye)`` pairs, eye 0 = left, 1 = right."""

    rank: int
    world: int
    units: tuple[tuple[int, int], ...]

    @property
    def frames(self) -> list[int]:
        return sorted({f for f, _ in self.units})


def plan_shards(n_frames: int, world: int) -> list[Shard]:
    """Deal ``n_frames`` L+R frames to ``world`` ranks (see module docstring).  Ranks beyond
    ``2 * n_frames`` get an empty shard."""
    if n_frames < 0 or world < 1:
        raise ValueError("bad shard arguments")
    shards = []
    for r in range(world):
        if n_frames >= world:
            units = tuple((f, e) for f in shard_range(n_frames, r, world) for e in (0, 1))
        else:
            flat = [(f, e) for f in range(n_frames) for e in (0, 1)]
            units = tuple(flat[i] for i in shard_range(len(flat), r, min(world, max(len(flat), 1)))) if r < len(flat) else ()
        shards.append(Shard(r, world, units))
    return shards


def row_bands(height: int, n: int, align: int = 16) -> list[tuple[int, int]]:
    """``n`` bands of output rows covering ``range(height)``, balanced, cut on multiples of ``align`` (the kernels'
    tile height) where the height allows; fewer than ``n`` when there are not enough rows."""
    if height <= 0 or n < 1:
        raise ValueError("bad band arguments")
    blocks = -(-height // align)
    n = min(n, blocks)
    cuts = [min(height, align * (blocks * k // n)) for k in range(n + 1)]
    return [(cuts[k], cuts[k + 1]) for k in range(n) if cuts[k + 1] > cuts[k]]


def plan_band_shards(n_frames: int, world: int, out_height: int) -> list[list[tuple[int, int, int, int]]]:
    """More ranks than eyes (a single pair on 4 or 8 GPUs): every eye's output ROWS are split into
    ``world // (2 * n_frames)`` bands and rank r gets the r-th ``(frame, eye, row0, row1)`` of the list ordered by
    frame, eye, band; surplus ranks get nothing.  Each rank needs the whole source eye of its band; nothing is
    exchanged between GPUs."""
    if n_frames < 1 or world < 1:
        raise ValueError("bad shard arguments")
    per_eye = max(1, world // (2 * n_frames))
    units = [(f, e, r0, r1) for f in range(n_frames) for e in (0, 1) for (r0, r1) in row_bands(out_height, per_eye)]
    out: list[list[tuple[int, int, int, int]]] = [[] for _ in range(world)]
    for k, u in enumerate(units):
        out[k % world].append(u)
    return out


def split_sbs(frame: Any):
    """One frame -> (left, right) source views: an (H, 2W, C) side-by-side array is split by column
    slicing exactly like the reference's ``left_path == right_path`` branch (remapper.py:448-456: the
    halves are ``[:, :W // 2]`` and ``[:, W // 2:]``, non-contiguous views); a (left, right) pair is
    passed through."""
    if isinstance(frame, (tuple, list)):
        left, right = frame
        return left, right
    w = frame.shape[1]
    return frame[:, : w // 2], frame[:, w // 2:]


@dataclass
class RankJob:
    """Host-side description of one rank's launches: ``groups`` are ``remapper.LaunchGroup`` records,
    ``units[g]`` the marshalled ``v1c_unit`` array of group ``g``; ``unit_ids[g][k]`` names the
    ``(frame, eye)`` of its k-th unit."""

    shard: Shard
    groups: list
    units: list
    unit_ids: list
    dst_wh: tuple[int, int]
    cn: int
    host_mapped: list


def build_rank_job(transformer: Any, shard: Shard, sources: dict, outputs: dict, *, radius: float,
                   size_output: tuple[int, int], rotations: dict | None = None, device: Any = None) -> RankJob:
    """Host logic of one rank.  ``sources[(frame, eye)]`` / ``outputs[(frame, eye)]`` are the rank's
    (H, W, C) uint8 source views and the views of the SBS halves they are written into (torch tensors:
    on the rank's device for a real run, host tensors in the CPU tests); ``rotations[(frame, eye)]``
    optionally replaces the rotation of the chain's single ``Euclidean3DRotator`` per unit (cli.py:308-319).
    ``transformer`` is one chain or an (L, R) pair (remapper.py:460-473)."""
    from . import remapper as R
    from .chain import Euclidean3DRotator, MultiTransformer
    from .quat import as_rotation_matrix

    ids = list(shard.units)
    if not ids:
        return RankJob(shard, [], [], [], tuple(size_output), 0, [])
    srcs = [sources[u] for u in ids]
    dsts = [outputs[u] for u in ids]
    cn = int(srcs[0].shape[2])
    if isinstance(transformer, tuple):
        per_unit: Any = [transformer[e] for _, e in ids]
    else:
        per_unit = transformer
    if rotations is not None:
        # per-unit rotations: every unit's chain = the shared chain with its rotator replaced; group_units
        # finds that they differ in the rotation only and lets them share one plan
        def with_rot(t, rot):
            stages = t.transformers if isinstance(t, MultiTransformer) else [t]
            idx = [i for i, s in enumerate(stages) if type(s) is Euclidean3DRotator]
            if len(idx) != 1:
                raise ValueError("rotations= needs a chain with exactly one Euclidean3DRotator")
            new = list(stages)
            new[idx[0]] = Euclidean3DRotator(as_rotation_matrix(rot))
            return MultiTransformer(transformers=new)

        base = per_unit if isinstance(per_unit, list) else [per_unit] * len(ids)
        per_unit = [with_rot(t, rotations[u]) if rotations.get(u) is not None else t for t, u in zip(base, ids)]
    size_in = None if isinstance(per_unit, list) else (int(srcs[0].shape[0]), int(srcs[0].shape[1]))
    groups, host_mapped = R.group_units(per_unit, srcs, dsts, radius=radius, size_input=size_in)
    dst_wh = (int(size_output[0]), int(size_output[1]))
    units = [R.marshal_units(g.srcs, g.dsts, g.rots, src_hw=g.src_hw, dst_wh=dst_wh, cn=cn, device=device) for g in groups]
    unit_ids = [[ids[k] for k in g.index] for g in groups]
    return RankJob(shard, groups, units, unit_ids, dst_wh, cn, [(ids[k], t, s) for k, t, s in host_mapped])


def run_rank_job(job: RankJob, *, interpolation: int, boarder_mode: int = 0, boarder_value: Any = 0,
                 launch: Callable | None = None) -> list[str]:
    """Launch every group of ``job`` on the current device / stream.  ``launch(group, units, n)`` replaces
    the device launch (the CPU tests pass an executor that interprets the marshalled records)."""
    from . import remapper as R

    if job.host_mapped:
        raise NotImplementedError("sharded runs need lowerable chains (user subclasses: use apply())")
    paths = []
    for g, units in zip(job.groups, job.units):
        if launch is not None:
            launch(g, units, len(g.srcs))
            paths.append("custom")
            continue
        dev = g.srcs[0].device
        plan = R._plan_for(g.chain, src_hw=g.src_hw, dst_wh=job.dst_wh, cn=job.cn, interpolation=interpolation,
                           border_mode=boarder_mode, border_value=boarder_value, device=dev)
        plan.run_units(units, len(g.srcs))
        paths.append(plan.path)
    return paths


def remap_sharded(transformer: Any, frames: Sequence[Any], *, size_output: tuple[int, int] = (2048, 2048),
                  interpolation: int = 4, boarder_mode: int = 0, boarder_value: Any = 0,
                  radius: float | str = "auto", rotations: Sequence[Any] | None = None,
                  devices: Sequence[int] | None = None) -> list[np.ndarray]:
    """``apply_lr`` for a batch of frames on several GPUs of one node, driven by this process.

    ``frames``: host arrays, each an (H, 2W, C) side-by-side frame or a ``(left, right)`` pair;
    ``rotations``: optional ``(rot_left, rot_right)`` per frame (matrices or quaternions) replacing the
    chain's single rotator.  Returns one (size_output[1], 2 * size_output[0], C) uint8 array per frame
    (remapper.py:517-518).  Units go to devices by ``plan_shards``; every device has its own worker thread,
    stream and page-locked staging ring; no pixels move between GPUs.  ``radius`` follows ``apply_lr``:
    per frame, "auto" / "max" from that frame's eyes (remapper.py:474-484) -- pass a number for a stream
    of frames so that they share one plan."""
    import torch

    from . import _hostpipe
    from . import remapper as R

    n = len(frames)
    if n == 0:
        return []
    if not torch.cuda.is_available():
        from ._native import EngineUnavailable

        raise EngineUnavailable("no HIP device visible: vr180_convert_amd has no CPU fallback")
    devs = list(range(torch.cuda.device_count())) if devices is None else [int(d) for d in devices]
    if not devs:
        raise ValueError("no devices")
    eyes = [split_sbs(f) for f in frames]
    for left, right in eyes:
        for im in (left, right):
            if not isinstance(im, np.ndarray) or im.dtype != np.uint8 or im.ndim != 3:
                raise TypeError("remap_sharded takes host uint8 (H, W, C) arrays")
    w, h = size_output
    cn = eyes[0][0].shape[2]
    radii = [R.get_radius_smart(radius, list(e)) if not isinstance(transformer, tuple) else None for e in eyes]
    outs = [np.empty((h, 2 * w, cn), np.uint8) for _ in range(n)]
    errors: list[BaseException] = []
    if len(devs) >= 4 * n and not isinstance(transformer, tuple) and rotations is None:
        # more devices than eyes: split every eye's output rows into bands (plan_band_shards)
        return _remap_banded(transformer, eyes, radii, outs, devs, size_output=size_output, interpolation=interpolation,
                             boarder_mode=boarder_mode, boarder_value=boarder_value)
    shards = [s for s in plan_shards(n, len(devs)) if s.units]

    def worker(shard: Shard, dev_index: int) -> None:
        try:
            dev = torch.device("cuda", dev_index)
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                # units that share (radius, source shape) go through one staging ring and one plan
                buckets: dict[Any, list] = {}
                for f, e in shard.units:
                    im = eyes[f][e]
                    r = radii[f] if radii[f] is not None else R.get_radius_smart(radius, [im])
                    buckets.setdefault((float(r), im.shape), []).append((f, e))
                for (r, _), ids in buckets.items():
                    images = [eyes[f][e] for f, e in ids]

                    def remap(srcs_g, dsts_g, idx, ids=ids, r=r):
                        gshard = Shard(shard.rank, shard.world, tuple(ids[i] for i in idx))
                        if boarder_mode == 5:  # BORDER_TRANSPARENT: the ring's reused slots must not show through skipped pixels
                            for d in dsts_g:
                                d.zero_()
                        job = build_rank_job(transformer, gshard, dict(zip(gshard.units, srcs_g)), dict(zip(gshard.units, dsts_g)),
                                             radius=r, size_output=size_output, device=dev,
                                             rotations=None if rotations is None else
                                             {(f, e): rotations[f][e] for f, e in gshard.units})
                        run_rank_job(job, interpolation=interpolation, boarder_mode=boarder_mode, boarder_value=boarder_value)

                    res = _hostpipe.run(images, dev, (h, w, cn), remap, with_index=True)
                    for (f, e), a in zip(ids, res):
                        np.copyto(outs[f][:, e * w:(e + 1) * w], a)  # np.concatenate(axis=1), remapper.py:518
        except BaseException as ex:  # noqa: BLE001 - re-raised in the calling thread
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(s, devs[s.rank]), name=f"v1c-dev{devs[s.rank]}") for s in shards]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return outs


@dataclass
class BandJob:
    """Host-side description of one rank's share of a row-band split (``plan_band_shards``): per band the lowered chain of
    that band (``chain.lower_for_get_map(row_band=...)``: the same Normalize with its centre moved up by r0 rows) and the
    marshalled one-unit ``v1c_unit`` record (whole source eye -> rows r0 .. r1 - 1 of the eye's half of the output)."""

    bands: list        # (frame, eye, r0, r1)
    chains: list
    units: list
    srcs: list
    src_hws: list      # (H, W) of every band's own source
    dst_w: int
    cn: int


def build_band_job(transformer: Any, bands: Sequence[tuple[int, int, int, int]], sources: dict, outputs: dict, *, radius: float,
                   size_output: tuple[int, int], device: Any = None, size_input: tuple[int, int] | None = None) -> BandJob:
    """Host logic of one rank of a row-band split (a single pair on 4 / 8 GPUs).  ``sources[(frame, eye)]``: the
    whole source eye (every rank uploads the eyes of its bands itself; nothing is exchanged); ``outputs[(frame, eye, r0, r1)]``:
    the (r1 - r0, W, C) view the band is written into.  ``size_input``: the (H, W) the Denormalize centre is taken from -- the
    frame's LEFT eye, ``images[0]`` of the call (remapper.py:385) -- whichever eyes' bands this rank holds (default: the first
    band's source, right when both eyes share a shape).  No device is involved: the gloo tests run exactly this."""
    from . import remapper as R
    from .chain import lower_for_get_map

    bands = list(bands)
    if not bands:
        return BandJob([], [], [], [], [], int(size_output[0]), 0)
    w = int(size_output[0])
    first = sources[(bands[0][0], bands[0][1])]
    centre_hw = (int(first.shape[0]), int(first.shape[1])) if size_input is None else (int(size_input[0]), int(size_input[1]))
    cn = int(first.shape[2])
    chains, units, srcs, src_hws = [], [], [], []
    for f, e, r0, r1 in bands:
        src, dst = sources[(f, e)], outputs[(f, e, r0, r1)]
        src_hw = (int(src.shape[0]), int(src.shape[1]))  # every band against its OWN source (the eyes of a frame may differ in shape)
        if int(src.shape[2]) != cn or tuple(int(v) for v in dst.shape[:2]) != (r1 - r0, w):
            raise ValueError("band sources must share a channel count and band outputs must be (r1 - r0, W, C) views")
        chains.append(lower_for_get_map(transformer[e] if isinstance(transformer, tuple) else transformer, radius=radius, size_input=centre_hw,
                                        size_output=tuple(size_output), row_band=(r0, r1)))
        units.append(R.marshal_units([src], [dst], None, src_hw=src_hw, dst_wh=(w, r1 - r0), cn=cn, device=device))
        srcs.append(src)
        src_hws.append(src_hw)
    return BandJob(bands, chains, units, srcs, src_hws, w, cn)


def run_band_job(job: BandJob, *, interpolation: int, boarder_mode: int = 0, boarder_value: Any = 0, launch: Callable | None = None) -> list[str]:
    """Launch every band of ``job`` on the current device / stream (``launch(chain, units, band)`` replaces the device launch in
    the CPU tests)."""
    from . import remapper as R

    paths = []
    for (f, e, r0, r1), chain, units, src, src_hw in zip(job.bands, job.chains, job.units, job.srcs, job.src_hws):
        if launch is not None:
            launch(chain, units, (f, e, r0, r1))
            paths.append("custom")
            continue
        plan = R._plan_for(chain, src_hw=src_hw, dst_wh=(job.dst_w, r1 - r0), cn=job.cn, interpolation=interpolation,
                           border_mode=boarder_mode, border_value=boarder_value, device=src.device)
        plan.run_units(units, 1)
        paths.append(plan.path)
    return paths


def _remap_banded(transformer, eyes, radii, outs, devs, *, size_output, interpolation, boarder_mode, boarder_value):
    """Worker per device, each remapping bands of output rows of whole source eyes it uploads itself."""
    import torch

    from . import remapper as R

    w, h = size_output
    cn = eyes[0][0].shape[2]
    work = plan_band_shards(len(eyes), len(devs), h)
    errors: list[BaseException] = []

    def worker(units, dev_index):
        try:
            dev = torch.device("cuda", dev_index)
            torch.cuda.set_device(dev)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                by_frame: dict = {}
                for u in units:
                    by_frame.setdefault(u[0], []).append(u)
                for f, bands in by_frame.items():  # (the radius is per frame: remapper.py:474-484)
                    sources = {(f, e): R._to_device(eyes[f][e], dev) for e in sorted({b[1] for b in bands})}
                    outputs = {b: torch.empty((b[3] - b[2], w, cn), dtype=torch.uint8, device=dev) for b in bands}
                    if boarder_mode == 5:  # BORDER_TRANSPARENT: skipped pixels must be deterministic
                        for d in outputs.values():
                            d.zero_()
                    job = build_band_job(transformer, bands, sources, outputs, radius=radii[f], size_output=size_output, device=dev,
                                         size_input=eyes[f][0].shape[:2])
                    run_band_job(job, interpolation=interpolation, boarder_mode=boarder_mode, boarder_value=boarder_value)
                    for (_, e, r0, r1), d in outputs.items():
                        np.copyto(outs[f][r0:r1, e * w:(e + 1) * w], d.cpu().numpy())
        except BaseException as ex:  # noqa: BLE001 - re-raised in the calling thread
            errors.append(ex)

    threads = [threading.Thread(target=worker, args=(u, devs[k]), name=f"v1c-band-dev{devs[k]}") for k, u in enumerate(work) if u]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return outs
