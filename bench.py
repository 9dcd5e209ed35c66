#!/usr/bin/env python3
"""Benchmark of the hot path: dual-fisheye -> side-by-side equirect remap (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A step = one apply_lr-shaped pass over one batch of synthetic input that is already resident in
HBM: L+R 4096x4096 fisheye -> 8192x4096 SBS equirect, EquirectangularEncoder *
PolynomialScaler([0,1,-0.1]) * FisheyeDecoder("equidistant"), INTER_LINEAR, BORDER_CONSTANT,
radius "max" (BASELINE.json configs[1], "C2").  With N > 1 every rank (one process per GPU,
launched by torch.distributed.run) remaps its own L+R pair per step: the path shards by frame
with no data-path collective, so scaling is weak and the value is N pairs per step time.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused remap launch):
algorithmic bytes per launch (3 B * (source + destination pixels), both eyes: SURVEY.md 8d)
over its average duration: two events on the launch stream bracket the K timed steps.  `cpu_baseline` is the
oracle (plain-C port of the reference path, oracle/) timed on this box's host cores on the same
workload, rank 0 at N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # name: (eye size, transformer spec, interpolation)
    "C2": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=1,
               desc="L+R 4096x4096 fisheye -> 8192x4096 SBS equirect, PolynomialScaler([0,1,-0.1]), bilinear"),
    "C1": dict(size=2048, poly=None, rot=None, interp=1,
               desc="L+R 2048x2048 fisheye -> 4096x2048 SBS equirect, equidistant, bilinear"),
    "C4": dict(size=8192, poly=[0, 1, -0.1], rot="ry45", interp=4,
               desc="L+R 8192x8192 -> 16384x8192 SBS, Euler rotation + PolynomialScaler, Lanczos4"),
    # batch shapes: `frames` SBS frames per GPU per step (BASELINE configs 3 and 5 shard 8 resp. 32 per GPU)
    "C3": dict(size=2880, poly=None, rot=None, interp=1, frames=8,
               desc="8 SBS frames 5760x2880 per GPU (of 64 over 8 GPUs), equidistant, bilinear"),
    "C5": dict(size=3840, poly=None, rot="calib", interp=1, frames=32,
               desc="32 SBS frames 7680x3840 per GPU (of 256 over 8 GPUs), per-frame per-eye calibration rotation, bilinear"),
}


def allreduce_max(value: float, device: torch.device) -> float:
    """MAX over ranks of a host scalar (identity when not distributed)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_transformer(cfg):
    import math

    from vr180_convert_amd.quat import from_euler_angles
    from vr180_convert_amd.transformer import (EquirectangularEncoder, Euclidean3DRotator, FisheyeDecoder,
                                               PolynomialScaler)

    t = EquirectangularEncoder()
    if cfg["rot"] == "ry45":
        t = t * Euclidean3DRotator(from_euler_angles(0.0, math.pi / 4, 0.0))
    if cfg["rot"] == "calib":
        t = t * Euclidean3DRotator((1.0, 0.0, 0.0, 0.0))  # replaced per unit, see calib_rotations()
    if cfg["poly"] is not None:
        t = t * PolynomialScaler(cfg["poly"])
    return t * FisheyeDecoder("equidistant")


def calib_rotations(frame: int):
    """BASELINE config 5: q = from_rotation_vector(N(0, 0.02)^3) per frame; the left
    eye gets conj(half_q), the right eye half_q with half_q = sin(phi/2)/sin(phi)*q + 0.5 (cli.py:308-319)."""
    import math

    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector

    rng = np.random.default_rng(20240619 + frame)
    q = from_rotation_vector(rng.normal(0, 0.02, 3))
    phi = math.acos(q.w)
    half = math.sin(phi / 2) / math.sin(phi) * q + 0.5
    return as_rotation_matrix(half.conj()), as_rotation_matrix(half)


def oracle_spec(cfg):
    import math

    spec = [("equirect_enc", True)]
    if cfg["rot"] == "ry45":
        c, s = math.cos(math.pi / 4), math.sin(math.pi / 4)
        spec.append(("rot", [[c, 0, s], [0, 1, 0], [-s, 0, c]]))
    if cfg["poly"] is not None:
        spec.append(("poly", cfg["poly"]))
    spec.append(("fisheye_dec", "equidistant"))
    return spec


def cpu_baseline(cfg, left: np.ndarray, right: np.ndarray, gpu_out: np.ndarray | None):
    """Oracle (C port of the reference path) on this box's host cores, same workload; also the
    bit-for-bit parity of the GPU result on the bench inputs."""
    from oracle import oracle as O

    O.build()
    cores = os.cpu_count() or 1
    O.set_threads(cores)
    size = cfg["size"]
    spec = oracle_spec(cfg)
    # output buffers are allocated (and touched) once: the timed passes measure arithmetic, not
    # first-touch page faults of 100 MB of fresh memory per call
    xm, ym = np.zeros((size, size), np.float32), np.zeros((size, size), np.float32)
    out = np.zeros((size, 2 * size, 3), np.uint8)
    halves = [np.zeros((size, size, 3), np.uint8), np.zeros((size, size, 3), np.uint8)]

    def one_pass():
        # apply_lr with a shared transformer: ONE map (remapper.py:381-386), remap per eye (:388-398),
        # concatenate (:518)
        O.get_map(spec, radius=size / 2, size_input=(size, size), size_output=(size, size), out=(xm, ym))
        O.remap(left, xm, ym, cfg["interp"], dst=halves[0])
        O.remap(right, xm, ym, cfg["interp"], dst=halves[1])
        out[:, :size], out[:, size:] = halves[0], halves[1]

    one_pass()  # warm (table build, thread pool)
    times = []
    t_end = time.perf_counter() + 12.0
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
        t0 = time.perf_counter()
        one_pass()
        times.append(time.perf_counter() - t0)
    best = min(times)
    mpx = 2 * size * size / 1e6
    res = {
        "value": round(mpx / best, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"{len(times)} full passes of the bench workload ({mpx:.1f} Mpx each: fp64 chain per pixel + "
                  f"fixed-point remap, both eyes, one shared map), best of them, OpenMP {cores} threads",
    }
    # (A) reference-equivalent path: NumPy chain (single-threaded float64 ufunc passes, like the
    # reference's get_map) on a 1024-row band of the same map + the C remap of that band; and
    # (B) remap only with a precomputed map (SURVEY.md 8d "CPU baseline")
    try:
        from oracle import chain_numpy

        band = 1024 if size >= 1024 else size
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            xm, ym = chain_numpy.get_map(spec, radius=size / 2, size_input=(size, size), size_output=(size, band))
        t_np = time.perf_counter() - t0
        bd = [np.zeros((band, size, 3), np.uint8), np.zeros((band, size, 3), np.uint8)]
        O.remap(left, xm, ym, cfg["interp"], dst=bd[0])  # warm
        t0 = time.perf_counter()
        O.remap(left, xm, ym, cfg["interp"], dst=bd[0])
        O.remap(right, xm, ym, cfg["interp"], dst=bd[1])
        t_rm = time.perf_counter() - t0
        res["numpy_chain_plus_remap"] = {
            "value": round(2 * size * band / 1e6 / (t_np + t_rm), 3), "unit": "Mpixels/s",
            "sample": f"top {band} rows of the workload's map: NumPy float64 chain (1 thread) {t_np:.2f} s + C remap of both eyes "
                      f"({cores} threads) {t_rm:.3f} s",
        }
        res["remap_only"] = {"value": round(2 * size * band / 1e6 / t_rm, 1), "unit": "Mpixels/s",
                             "sample": "same band, map precomputed"}
    except Exception as e:  # noqa: BLE001 - the extra baselines must never break the bench line
        res["numpy_chain_plus_remap"] = {"error": repr(e)}
    parity = None
    if gpu_out is not None:
        diff = np.abs(gpu_out.astype(np.int16) - out.astype(np.int16))
        parity = {"max_abs_diff": int(diff.max()), "bytes_differing": int((diff != 0).sum()), "bytes": int(diff.size)}
    return res, parity


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C2", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    # V1C_BENCH_REHEARSAL=1: rehearse the multi-rank path on a box with fewer GPUs than ranks (gloo for the
    # barrier / MAX reduction, ranks share the devices there are) -- never a measurement
    rehearsal = os.environ.get("V1C_BENCH_REHEARSAL", "0") == "1"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if rehearsal else local_rank)
    torch.cuda.set_device(dev)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import vr180_convert_amd as V
    from vr180_convert_amd import _native
    from vr180_convert_amd.synth import noise_disc

    _native.lib()  # fail loudly if the HIP engine is missing
    cfg = WORKLOADS[args.workload]
    size = cfg["size"]
    transformer = build_transformer(cfg)
    frames = cfg.get("frames", 0)
    left_h = right_h = None
    if frames:
        # batch of SBS frames resident in HBM; units are the column halves (pitched views), each
        # eye is written straight into its half of the output SBS frame
        from vr180_convert_amd.synth import noise_disc_torch

        ins = [noise_disc_torch(size, 2 * size, rank * frames + f, dev) for f in range(frames)]
        outs = [torch.empty((size, 2 * size, 3), dtype=torch.uint8, device=dev) for _ in range(frames)]
        srcs = [v for fr in ins for v in (fr[:, :size], fr[:, size:])]
        dsts = [v for fr in outs for v in (fr[:, :size], fr[:, size:])]
        rots = None
        if cfg["rot"] == "calib":
            rots = [m for f in range(frames) for m in calib_rotations(rank * frames + f)]
        sbs = outs[0]

        def step():
            V.remap_tensors(transformer, srcs, dsts, radius=size / 2, interpolation=cfg["interp"], rotations=rots)
    else:
        # seeded noise-disc frames; frame index = rank so ranks hold different pixels
        left_h, right_h = noise_disc(size, size, 2 * rank), noise_disc(size, size, 2 * rank + 1)
        left, right = torch.from_numpy(left_h).to(dev), torch.from_numpy(right_h).to(dev)
        sbs = torch.empty((size, 2 * size, 3), dtype=torch.uint8, device=dev)

        def step():
            V.apply_lr_tensors(transformer, left, right, out=sbs, size_output=(size, size), interpolation=cfg["interp"],
                               radius="max")

    def barrier():
        if world > 1:
            dist.barrier()

    # Setup, not a step: ~25 ms of device work so that the shader clock has left its idle state
    # before the W warm-up steps (a step is ~0.07 ms; the CPU-side input synthesis above leaves the
    # GPU idle for about a second).  The timed region below is exactly K steps.
    spin = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for _ in range(40):
        spin.add_(1)
    del spin
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    # Two events on the launch stream bracket the K steps: (e1 - e0) / K is the average GPU time of
    # a step's launch(es) including the gaps between them.  (An event pair around every step would
    # add two packets per launch and, whenever the host is the slower side, measure host latency.)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = allreduce_max(elapsed, dev)
    kernel_ms = e0.elapsed_time(e1) / args.steps
    kernel_ms_max = allreduce_max(kernel_ms, dev)

    units = 2 * max(frames, 1)
    px_per_step = units * size * size * world
    value = px_per_step * args.steps / elapsed / 1e6
    # all eyes of a step: source read once + destination written once (a batch is ceil(units/16) launches:
    # the figure is per step, i.e. per group of launches, for batch workloads)
    alg_bytes = units * 3 * (size * size + size * size)
    achieved = alg_bytes / (kernel_ms_max * 1e-3) / 1e9

    if rank == 0:
        from vr180_convert_amd.remapper import _PLANS

        paths = sorted({p.path for p in _PLANS.values()})
        # HBM bytes per launch from the PMC counters: collected by tools/profile.sh in separate
        # rocprofv3 --pmc passes of this very command (a profiler cannot run inside the timed run)
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic_latest.json"
        if tfile.exists():
            try:
                t = json.loads(tfile.read_text())
                if t.get("workload") == args.workload:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "Mpixels/s dual-fisheye->SBS-equirect remap; achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (seeded uint8 noise inside the fisheye circle, black outside), resident in HBM"
                    + (" -- REHEARSAL: ranks share devices, gloo; not a measurement" if rehearsal else ""),
            "config": {"workload": f"{args.workload}: {cfg['desc']}", "units_per_step_per_gpu": units,
                       "arithmetic": "f64 coordinates (fused chain), u8 pixels with int32 fixed-point blend",
                       "sharding": "frames over ranks, no collective", "kernel_path": paths},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel_ms": round(kernel_ms_max, 4), "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu_baseline and not frames:
            torch.cuda.synchronize(dev)
            cb, parity = cpu_baseline(cfg, left_h, right_h, sbs.cpu().numpy())
            line["cpu_baseline"] = cb
            line["parity_vs_oracle"] = parity
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
