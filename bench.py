#!/usr/bin/env python3
"""Benchmark of the hot path: dual-fisheye -> side-by-side equirect remap (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A step = one apply_lr-shaped pass over one batch of synthetic input that is already resident in
HBM: L+R 4096x4096 fisheye -> 8192x4096 SBS equirect, EquirectangularEncoder *
PolynomialScaler([0,1,-0.1]) * FisheyeDecoder("equidistant"), INTER_LINEAR, BORDER_CONSTANT,
radius "max" (BASELINE.json configs[1], "C2").  With N > 1 every rank (one process per GPU,
launched by torch.distributed.run) remaps its own L+R pair per step: the path shards by frame
with no data-path collective, so scaling is weak and the value is N pairs per step time.

`python bench.py --gpus N` started WITHOUT torch.distributed.run starts its N ranks itself: a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...` is spawned before this process has
made any HIP call, its output is passed through and its exit code returned.

The inputs / outputs of a step rotate over >= 3 buffer sets of > 640 MiB in total, so no step finds
its source or destination lines in the 256 MiB Infinity Cache (L3-cold, like a stream of frames).
Before the W warm-up steps the workload's own launch is replayed untimed for 0.5 s: clocks and the HIP
launch queue are then in the state of a running stream, whatever K and W are (see main()).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fused remap launch):
algorithmic bytes per launch (3 B * (source + destination pixels), both eyes: SURVEY.md 8d)
over its average duration: two events on the launch stream bracket the K timed steps.  `cpu_baseline` is the
oracle (plain-C port of the reference path, oracle/) timed on this box's host cores on the same
workload, rank 0 at N=1 only.  `roofline.traffic` is measured by this very run: rank 0 at N=1 runs
two short `rocprofv3 --pmc` child passes of itself (FETCH_SIZE, WRITE_SIZE; separate passes, gfx950
corrections of MI355X_MICROARCH.md) after the timed region; `traffic_source` says "live" or names
the committed file it fell back to.  `cold` is what a first call costs (plan creation + first launch).
`sustained` = a second leg of >= 1000 steps (>= 0.25 s) behind the timed region, bracketed the same way and NOT part of
`value`: what a stream of frames sees, next to the driver's short window (`window_over_sustained` = their ratio).
"""
from __future__ import annotations

import argparse
import csv
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ROTATE_MIN_BYTES = 640 << 20  # > 2x the 256 MiB Infinity Cache
ROTATE_MIN_SETS = 3
PMC_STEPS, PMC_WARMUP = 6, 2  # length of a rocprofv3 --pmc child pass
SUSTAINED_MIN_STEPS = 1000  # the second, untimed-for-`value` leg of every run (`sustained` in the line)

WORKLOADS = {
    # name: (eye size, transformer spec, interpolation)
    "C2": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=1,
               desc="L+R 4096x4096 fisheye -> 8192x4096 SBS equirect, PolynomialScaler([0,1,-0.1]), bilinear"),
    "C1": dict(size=2048, poly=None, rot=None, interp=1,
               desc="L+R 2048x2048 fisheye -> 4096x2048 SBS equirect, equidistant, bilinear"),
    # BASELINE config 1 as it is written: ONE 2048x2048 image through apply() (a single-unit launch: no second eye to share
    # the coordinates with)
    "C1S": dict(size=2048, poly=None, rot=None, interp=1, single=True,
                desc="single 2048x2048 fisheye -> 2048x2048 equirect, equidistant, bilinear (BASELINE config 1)"),
    "C4": dict(size=8192, poly=[0, 1, -0.1], rot="ry45", interp=4,
               desc="L+R 8192x8192 -> 16384x8192 SBS, Euler rotation + PolynomialScaler, Lanczos4"),
    # (not a BASELINE config: a rotated bilinear pair -- the pair kernels the unrotated configs above do not reach; tools/ab.sh)
    "C2R": dict(size=4096, poly=[0, 1, -0.1], rot="ry45", interp=1,
                desc="L+R 4096x4096 -> 8192x4096 SBS, Euler rotation + PolynomialScaler, bilinear (A/B only)"),
    "C2NN": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=0,
                 desc="L+R 4096x4096 -> 8192x4096 SBS, PolynomialScaler, INTER_NEAREST (not a BASELINE config: the bilinear tile kernels "
                      "with coordinates 32 * cvRound(x))"),
    # configurations off the BASELINE list -- reporting only: gray / BGRA run k_ray_lin_cn (raw LDS-DMA boxes; HISTORY.md 4.5), bilinear
    # BORDER_TRANSPARENT the BGR tile kernels
    "C2G": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=1, cn=1,
                desc="L+R 4096x4096 GRAY -> 8192x4096 SBS, PolynomialScaler, bilinear (k_ray_lin_cn since r03; not a BASELINE config)"),
    "C2A": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=1, cn=4,
                desc="L+R 4096x4096 BGRA -> 8192x4096 SBS, PolynomialScaler, bilinear (k_ray_lin_cn since r03; not a BASELINE config)"),
    "C2GN": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=0, cn=1,
                 desc="L+R 4096x4096 GRAY -> 8192x4096 SBS, PolynomialScaler, INTER_NEAREST (k_ray_lin_cn since r04; not a BASELINE config)"),
    "C2AN": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=0, cn=4,
                 desc="L+R 4096x4096 BGRA -> 8192x4096 SBS, PolynomialScaler, INTER_NEAREST (k_ray_lin_cn since r04; not a BASELINE config)"),
    "C2GL": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=4, cn=1,
                 desc="L+R 4096x4096 GRAY -> 8192x4096 SBS, PolynomialScaler, INTER_LANCZOS4 (k_ray_lin_cn K = 8 since r04; not a BASELINE config)"),
    "C2AL": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=4, cn=4,
                 desc="L+R 4096x4096 BGRA -> 8192x4096 SBS, PolynomialScaler, INTER_LANCZOS4 (k_ray_lin_cn K = 8 since r04; not a BASELINE config)"),
    "C2C": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=2,
                desc="L+R 4096x4096 -> 8192x4096 SBS, PolynomialScaler, INTER_CUBIC (the K x K tile kernel with 4 x 4 taps; not a BASELINE config)"),
    "C2L": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=4,
                desc="L+R 4096x4096 -> 8192x4096 SBS, PolynomialScaler, INTER_LANCZOS4 (the reference's default interpolation at C2's size; not a BASELINE config)"),
    "C1L": dict(size=2048, poly=None, rot=None, interp=4,
                desc="L+R 2048x2048 -> 4096x2048 SBS, equidistant, INTER_LANCZOS4: the reference's call with every default (cli.py:230-231 "
                     "transformer, remapper.py:412-413 size and interpolation); not a BASELINE config"),
    "C1L995": dict(size=2048, poly=None, rot=None, interp=4, rscale=0.995,
                   desc="C1L with the image circle 0.5 % inside the frame: no K x K footprint crosses the border of the source (A/B only)"),
    "C2T": dict(size=4096, poly=[0, 1, -0.1], rot=None, interp=1, border=5,
                desc="L+R 4096x4096 -> 8192x4096 SBS, PolynomialScaler, bilinear, BORDER_TRANSPARENT (tile kernels since r03; not a BASELINE config)"),
    "C2N": dict(size=4080, poly=[0, 1, -0.1], rot=None, interp=1,
                desc="L+R 4080x4080 -> 8160x4080 SBS (rows do not mirror about a tile boundary), bilinear (A/B only)"),
    # batch shapes: `frames` SBS frames per GPU per step (BASELINE configs 3 and 5 shard 8 resp. 32 per GPU)
    "C3": dict(size=2880, poly=None, rot=None, interp=1, frames=8,
               desc="8 SBS frames 5760x2880 per GPU (of 64 over 8 GPUs), equidistant, bilinear"),
    "C5G": dict(size=1920, poly=None, rot="calib", interp=1, frames=8, cn=1,
                desc="8 GRAY SBS frames 3840x1920 per GPU, per-frame per-eye calibration rotation, bilinear (k_ray_lin_cn without plan-time boxes "
                     "since r04; not a BASELINE config)"),
    "C5A": dict(size=1920, poly=None, rot="calib", interp=1, frames=8, cn=4,
                desc="8 BGRA SBS frames 3840x1920 per GPU, per-frame per-eye calibration rotation, bilinear (k_ray_lin_cn without plan-time boxes "
                     "since r04; not a BASELINE config)"),
    "C5": dict(size=3840, poly=None, rot="calib", interp=1, frames=32,
               desc="32 SBS frames 7680x3840 per GPU (of 256 over 8 GPUs), per-frame per-eye calibration rotation, bilinear"),
    # The reference's own test calls (tests/test_remapper.py:42-109): 256 x 256 outputs, the default Lanczos4, radius "max"
    "C0": dict(size=256, poly=None, rot=None, interp=4,
               desc="L+R 256x256 -> 512x256 SBS, equidistant, INTER_LANCZOS4: the size of the reference's own test calls "
                    "(tests/test_remapper.py:73,90,108); not a BASELINE config"),
    "C0B": dict(size=256, poly=None, rot=None, interp=1,
                desc="L+R 256x256 -> 512x256 SBS, equidistant, bilinear (the reference's test size; not a BASELINE config)"),
}
# Chains that do not start with EquirectangularEncoder(is_latitude_y=True) -- 7 of the reference's 10 test chains
# (tests/test_remapper.py:42-91: fisheye -> fisheye re-projection, SURVEY.md 8a "planar mode"), each as a 2048 x 2048 pair,
# bilinear and Lanczos4.  `spec`: the neutral chain description of oracle/oracle.py (chain_from_spec).
_RY45 = [[math.cos(math.pi / 4), 0.0, math.sin(math.pi / 4)], [0.0, 1.0, 0.0], [-math.sin(math.pi / 4), 0.0, math.cos(math.pi / 4)]]
_SPEC_CHAINS = {
    "P1": ("FisheyeEncoder('rectilinear') * FisheyeDecoder('equidistant')",
           [("fisheye_enc", "rectilinear"), ("fisheye_dec", "equidistant")]),
    "P2": ("FisheyeEncoder('equidistant') * Euclidean3DRotator(Ry pi/4) * FisheyeDecoder('equidistant')",
           [("fisheye_enc", "equidistant"), ("rot", _RY45), ("fisheye_dec", "equidistant")]),
    "P3": ("EquirectangularEncoder(is_latitude_y=False) * FisheyeDecoder('equidistant')",
           [("equirect_enc", False), ("fisheye_dec", "equidistant")]),
    "P4": ("FisheyeEncoder('equidistant') * PolynomialScaler([0, 1, -0.1]) * FisheyeDecoder('equidistant')",
           [("fisheye_enc", "equidistant"), ("poly", [0, 1, -0.1]), ("fisheye_dec", "equidistant")]),
}
for _n, (_d, _sp) in _SPEC_CHAINS.items():
    WORKLOADS[_n] = dict(size=2048, poly=None, rot=None, interp=1, spec=_sp,
                         desc=f"L+R 2048x2048 -> 4096x2048 SBS, {_d}, bilinear (a chain of the reference's tests; not a BASELINE config)")
    WORKLOADS[_n + "L"] = dict(size=2048, poly=None, rot=None, interp=4, spec=_sp,
                               desc=f"L+R 2048x2048 -> 4096x2048 SBS, {_d}, INTER_LANCZOS4 (a chain of the reference's tests; not a BASELINE config)")


def allreduce_max(value: float, device: torch.device) -> float:
    """MAX over ranks of a host scalar (identity when not distributed)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allgather_scalar(value: float, device: torch.device, world: int) -> list[float]:
    """One host scalar per rank, on every rank (tensor all_gather: RCCL on the device, gloo on the host)."""
    import torch.distributed as dist

    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    where = device if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=where)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def build_transformer(cfg):
    import math

    from vr180_convert_amd.quat import from_euler_angles
    from vr180_convert_amd.transformer import (EquirectangularEncoder, Euclidean3DRotator, FisheyeDecoder,
                                               PolynomialScaler)

    if cfg.get("spec") is not None:
        return spec_to_product(cfg["spec"])
    t = EquirectangularEncoder()
    if cfg["rot"] == "ry45":
        t = t * Euclidean3DRotator(from_euler_angles(0.0, math.pi / 4, 0.0))
    if cfg["rot"] == "calib":
        t = t * Euclidean3DRotator((1.0, 0.0, 0.0, 0.0))  # replaced per unit, see calib_rotations()
    if cfg["poly"] is not None:
        t = t * PolynomialScaler(cfg["poly"])
    return t * FisheyeDecoder("equidistant")


def spec_to_product(spec):
    """Neutral chain description (oracle/oracle.py: chain_from_spec) -> the product's transformer objects."""
    import vr180_convert_amd.transformer as T

    def one(item):
        kind, *a = item
        if kind == "equirect_enc":
            return T.EquirectangularEncoder(*a[:1])
        if kind == "fisheye_enc":
            return T.FisheyeEncoder(a[0])
        if kind == "fisheye_dec":
            return T.FisheyeDecoder(a[0])
        if kind == "poly":
            return T.PolynomialScaler(a[0])
        if kind == "rot":
            return T.Euclidean3DRotator(np.asarray(a[0], float))
        raise ValueError(item)

    out = one(spec[0])
    for it in spec[1:]:
        out = out * one(it)
    return out


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

    if cfg.get("spec") is not None:
        return list(cfg["spec"])
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
    single = bool(cfg.get("single"))
    eyes = 1 if single else 2
    # output buffers are allocated (and touched) once: the timed passes measure arithmetic, not
    # first-touch page faults of 100 MB of fresh memory per call
    xm, ym = np.zeros((size, size), np.float32), np.zeros((size, size), np.float32)
    cn = left.shape[2]
    out = np.zeros((size, eyes * size, cn), np.uint8)
    halves = [np.zeros((size, size, cn), np.uint8), np.zeros((size, size, cn), np.uint8)]

    def one_pass():
        # apply_lr with a shared transformer: ONE map (remapper.py:381-386), remap per eye (:388-398),
        # concatenate (:518); apply() of one image: the map and one remap
        O.get_map(spec, radius=size / 2, size_input=(size, size), size_output=(size, size), out=(xm, ym))
        O.remap(left, xm, ym, cfg["interp"], cfg.get("border", 0), dst=halves[0])
        if single:
            out[:] = halves[0]
            return
        O.remap(right, xm, ym, cfg["interp"], cfg.get("border", 0), dst=halves[1])
        out[:, :size], out[:, size:] = halves[0], halves[1]

    one_pass()  # warm (table build, thread pool)
    times = []
    t_end = time.perf_counter() + 12.0
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
        t0 = time.perf_counter()
        one_pass()
        times.append(time.perf_counter() - t0)
    best = min(times)
    mpx = eyes * size * size / 1e6
    res = {
        "value": round(mpx / best, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
        "sample": f"{len(times)} full passes of the bench workload ({mpx:.1f} Mpx each: fp64 chain per pixel + "
                  f"fixed-point remap, {'one image' if single else 'both eyes, one shared map'}), best of them, OpenMP {cores} threads",
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
        bd = [np.zeros((band, size, cn), np.uint8), np.zeros((band, size, cn), np.uint8)]
        O.remap(left, xm, ym, cfg["interp"], dst=bd[0])  # warm
        t0 = time.perf_counter()
        O.remap(left, xm, ym, cfg["interp"], dst=bd[0])
        if not single:
            O.remap(right, xm, ym, cfg["interp"], dst=bd[1])
        t_rm = time.perf_counter() - t0
        res["numpy_chain_plus_remap"] = {
            "value": round(eyes * size * band / 1e6 / (t_np + t_rm), 3), "unit": "Mpixels/s",
            "sample": f"top {band} rows of the workload's map: NumPy float64 chain (1 thread) {t_np:.2f} s + C remap of {'the image' if single else 'both eyes'} "
                      f"({cores} threads) {t_rm:.3f} s",
        }
        res["remap_only"] = {"value": round(eyes * size * band / 1e6 / t_rm, 1), "unit": "Mpixels/s",
                             "sample": "same band, map precomputed"}
    except Exception as e:  # noqa: BLE001 - the extra baselines must never break the bench line
        res["numpy_chain_plus_remap"] = {"error": repr(e)}
    parity = None
    if gpu_out is not None:
        diff = np.abs(gpu_out.astype(np.int16) - out.astype(np.int16))
        parity = {"max_abs_diff": int(diff.max()), "bytes_differing": int((diff != 0).sum()), "bytes": int(diff.size)}
    return res, parity


def spawn_ranks(args, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process
    (this process has made no HIP call yet and never makes one), pass its output through, return its
    exit code.  Never exec / re-exec."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


PMC_CHILD_ARGS: list[str] = []  # main(): what besides --workload selects the launch (--split)


def _pmc_pass(counter: str, workload: str, timeout_s: float, want_all: bool = False):
    """One `rocprofv3 --pmc <counters>` child pass of this bench (short, nothing else profiled): mean
    counter value of the dominant remap kernel, or None.  The child is a fresh process; this one only waits.
    `counter` may name several counters of one pass (space separated); with `want_all` the result is
    (kernel, {counter: sum over the dominant kernel's dispatches}, dispatches) instead of (kernel, per-step value)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None
    names = counter.split()
    tmp = tempfile.mkdtemp(prefix="v1c_pmc_")
    env = dict(os.environ, V1C_BENCH_CHILD="1", TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [exe, "--pmc", *names, "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable,
           str(Path(__file__).resolve()), "--workload", workload, "--steps", str(PMC_STEPS), "--warmup", str(PMC_WARMUP), "--no-cpu-baseline",
           "--traffic", "none", "--no-cold-extra", "--no-condition", "--no-sustained", *PMC_CHILD_ARGS]
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)  # exactly the process group started above
            p.wait()
            return None
        rows = []
        for f in Path(tmp).rglob("*counter_collection.csv"):
            rows += [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") in names and
                     ("k_ray" in r.get("Kernel_Name", "") or "k_remap" in r.get("Kernel_Name", ""))]
        if not rows:
            return None
        by_kernel: dict[str, dict[str, list[float]]] = {}
        for r in rows:
            by_kernel.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        # the dominant kernel = the one with the largest total of the first counter; per STEP = total / steps of the child
        kern = max(by_kernel, key=lambda k: sum(by_kernel[k].get(names[0], [0.0])))
        if want_all:
            return kern, {c: sum(v) for c, v in by_kernel[kern].items()}, len(by_kernel[kern].get(names[0], []))
        return kern, sum(sum(v.get(names[0], [])) for v in by_kernel.values()) / (1 + PMC_WARMUP + PMC_STEPS)  # + the cold first step
    except Exception:  # noqa: BLE001 - traffic must never break the bench line
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_valu_per_wave(workload: str):
    """VALU instructions per wave of the dominant kernel, from one `--pmc SQ_INSTS_VALU SQ_WAVES` child pass of this run."""
    r = _pmc_pass("SQ_INSTS_VALU SQ_WAVES", workload, 240.0, want_all=True)
    if r is None or not r[1].get("SQ_WAVES"):
        return None
    return {"kernel": r[0].split("(")[0], "valu_instr_per_wave": r[1]["SQ_INSTS_VALU"] / r[1]["SQ_WAVES"],
            "waves_per_launch": r[1]["SQ_WAVES"] / max(r[2], 1)}


def measure_traffic(workload: str):
    """HBM bytes per step (= per launch for the pair workloads) from the PMC counters, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes, both
    in KiB; gfx950 tallies 64 B per 128-B read request, so FETCH_SIZE is doubled; WRITE_SIZE is exact."""
    f = _pmc_pass("FETCH_SIZE", workload, 240.0)
    if f is None:
        return None
    w = _pmc_pass("WRITE_SIZE", workload, 240.0)
    if w is None:
        return None
    return {"hbm_bytes_per_step": int(f[1] * 1024 * 2 + w[1] * 1024), "fetch_bytes_corrected": int(f[1] * 1024 * 2),
            "write_bytes": int(w[1] * 1024), "kernel": f[0].split("(")[0]}


def cold_call(cfg, dev, V, transformer_builder):
    """What the reference's own use case (one apply_lr per process, cli.py) pays on top of the kernel:
    plan creation (host radial fit + validation, tile boxes, tables) and the first launch, on fresh
    geometry.  Wall time, synchronised on both sides."""
    from vr180_convert_amd import remapper as R
    from vr180_convert_amd.synth import noise_disc_torch

    size = cfg["size"]
    left, right = noise_disc_torch(size, size, 1000, dev), noise_disc_torch(size, size, 1001, dev)
    sbs = torch.empty((size, 2 * size, 3), dtype=torch.uint8, device=dev)
    t = transformer_builder(cfg)
    R.clear_caches()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    V.apply_lr_tensors(t, left, right, out=sbs, size_output=(size, size), interpolation=cfg["interp"], radius="max")
    torch.cuda.synchronize(dev)
    first = (time.perf_counter() - t0) * 1e3
    create = sum(p.create_ms for p in R._PLANS.values())
    t0 = time.perf_counter()
    V.apply_lr_tensors(t, left, right, out=sbs, size_output=(size, size), interpolation=cfg["interp"], radius="max")
    torch.cuda.synchronize(dev)
    second = (time.perf_counter() - t0) * 1e3
    return {"plan_create_ms": round(create, 3), "first_call_ms": round(first, 3), "second_call_ms": round(second, 3)}


def new_radius_cost(cfg, dev, V, transformer_builder):
    """What radius="auto" -- the reference's default, remapper.py:333,416 -- costs per image when the image circle changes from image to
    image: the device-resident form (estimate, maximum, scale and remap on the stream: no synchronisation, one plan for every radius)
    against the exact form (estimates brought to the host, a plan per new radius).  Wall time per apply_lr_tensors call, ms; discs of
    different radii in turn so that every call sees a radius its predecessor did not."""
    from vr180_convert_amd import remapper as R

    size = cfg["size"]
    yy, xx = torch.meshgrid(torch.arange(size, device=dev), torch.arange(size, device=dev), indexing="ij")
    rr = (xx - size / 2) ** 2 + (yy - size / 2) ** 2
    base = torch.randint(40, 256, (size, size, 3), dtype=torch.uint8, device=dev)
    n_img = 22  # (a first pair to create the plan, then 10 timed pairs)
    imgs = [base * (rr <= (size / 2 - 3 - 1.5 * k) ** 2)[..., None].to(torch.uint8) for k in range(n_img)]
    sbs = torch.empty((size, 2 * size, 3), dtype=torch.uint8, device=dev)
    t = transformer_builder(cfg)
    out = {}
    for key, on_dev in (("new_radius_ms", True), ("new_radius_exact_ms", False)):
        try:
            V.apply_lr_tensors(t, imgs[0], imgs[1], out=sbs, size_output=(size, size), interpolation=cfg["interp"], radius="auto", auto_radius_on_device=on_dev)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(2, n_img, 2):
                V.apply_lr_tensors(t, imgs[k], imgs[k + 1], out=sbs, size_output=(size, size), interpolation=cfg["interp"], radius="auto",
                                   auto_radius_on_device=on_dev)
            torch.cuda.synchronize(dev)
            out[key] = round((time.perf_counter() - t0) * 1e3 / ((n_img - 2) // 2), 4)
        except Exception as e:  # noqa: BLE001 - must never break the bench line
            out[key] = repr(e)
    out["new_radius_note"] = ("radius='auto' per L+R pair whose image circle differs from the previous pair's, wall ms per call: on the device "
                              "(v1c_plan_run_auto_images: no sync, one plan) / exact (estimates to the host: a sync; the launch that reads the radius from "
                              "device memory for a circle that moved, a plan of its own for one that repeats)")
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="C2", choices=list(WORKLOADS))
    ap.add_argument("--split", default="frames", choices=["frames", "eyes", "bands"],
                    help="how the job is dealt to the ranks (SURVEY.md 8e): frames = every rank its own frames (weak scaling, the "
                         "default); eyes = ONE L+R pair, one eye per rank (N <= 2); bands = ONE pair, every eye's output rows cut into "
                         "N / 2 bands (strong scaling; every rank holds the whole source eye of its bands, nothing is exchanged)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic", default="live", choices=["live", "file", "none"],
                    help="roofline.traffic: measured by rocprofv3 --pmc child passes of this run (N=1), read from "
                         "profiles/traffic_latest.json, or omitted")
    ap.add_argument("--no-cold-extra", action="store_true", help="skip the 8192x8192 Lanczos4 cold-call measurement")
    ap.add_argument("--no-rotate", action="store_true", help="A/B only: replay ONE buffer set (L3-warm for small workloads)")
    ap.add_argument("--streams", type=int, default=1,
                    help="A/B only: deal the steps round-robin to this many HIP streams (independent frames overlap: one launch's tail "
                         "fills with the next one's head); the default, one stream, is what every reported number uses")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the second, >= 1000-step leg behind the timed region (`sustained` in the line)")
    ap.add_argument("--no-condition", action="store_true",
                    help="skip the 0.5 s clock / launch-queue conditioning (the --pmc child passes: counters do not depend on clocks)")
    args = ap.parse_args()
    if args.no_condition:  # (the counter passes: a few launches are all they need)
        args.no_sustained = True

    PMC_CHILD_ARGS[:] = ["--split", args.split]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves, as a child, before any HIP call
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different job")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    # V1C_BENCH_REHEARSAL=1: rehearse the multi-rank path on a box with fewer GPUs than ranks (gloo for the
    # barrier / MAX reduction, ranks share the devices there are) -- never a measurement
    rehearsal = os.environ.get("V1C_BENCH_REHEARSAL", "0") == "1"
    if not rehearsal and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} HIP devices")
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
    from vr180_convert_amd import remapper as R
    from vr180_convert_amd.sharding import (Shard, build_band_job, build_rank_job, plan_band_shards, plan_shards, run_band_job,
                                            run_rank_job, shard_range)
    from vr180_convert_amd.synth import noise_disc, noise_disc_torch

    _native.lib()  # fail loudly if the HIP engine is missing
    cfg = WORKLOADS[args.workload]
    size = cfg["size"]
    transformer = build_transformer(cfg)
    frames = cfg.get("frames", 0)
    single = bool(cfg.get("single"))
    strong = args.split != "frames"
    if strong and (frames or single):
        raise SystemExit("--split eyes / bands divide ONE L+R pair: use a pair workload (C2, C4, C1, C2R, C2N)")
    units = 1 if single else 2 * max(frames, 1)
    cn, border = int(cfg.get("cn", 3)), int(cfg.get("border", 0))
    set_bytes = units * cn * (size * size + size * size)
    my_units: list = []
    if strong:
        # ONE pair for the whole job (strong scaling): this rank's part of it by the product's partition
        if args.split == "eyes":
            if world > 2:
                raise SystemExit("--split eyes deals the two eyes of one pair: at most 2 ranks (more: --split bands)")
            my_units = list(plan_shards(1, world)[rank].units)
            my_px = len(my_units) * size * size
        else:
            my_units = list(plan_band_shards(1, world, size)[rank])
            my_px = sum(r1 - r0 for _, _, r0, r1 in my_units) * size
        my_eyes = sorted({u[1] for u in my_units})
        # bytes this rank moves per step: the source rows its output rows read are at most its whole eyes; algorithmic = the
        # job's 6 B per output pixel times this rank's pixels
        set_bytes = max(6 * my_px, 1)
    nsets = 1 if (args.no_rotate or (strong and not my_units)) else max(ROTATE_MIN_SETS, -(-ROTATE_MIN_BYTES // set_bytes))
    left_h = right_h = None
    sets = []
    if strong:
        left_h, right_h = noise_disc(size, size, 0), noise_disc(size, size, 1)
        for k in range(nsets):
            eyes_d = {}
            for e in my_eyes:
                eyes_d[e] = torch.from_numpy((left_h, right_h)[e]).to(dev) if k == 0 else noise_disc_torch(size, size, e + 100003 * k, dev)
            sbs = torch.zeros((size, 2 * size, 3), dtype=torch.uint8, device=dev)
            sources = {(0, e): eyes_d[e] for e in my_eyes}
            if args.split == "eyes":
                outputs = {(0, e): sbs[:, e * size:(e + 1) * size] for _, e in my_units}
                job = build_rank_job(transformer, Shard(rank, world, tuple(my_units)), sources, outputs, radius=size / 2,
                                     size_output=(size, size), device=dev)
            else:
                outputs = {u: sbs[u[2]:u[3], u[1] * size:(u[1] + 1) * size] for u in my_units}
                job = build_band_job(transformer, my_units, sources, outputs, radius=size / 2, size_output=(size, size), device=dev)
            sets.append(dict(job=job, sbs=sbs, keep=(eyes_d, sources, outputs)))

        def step(i: int):
            # the launch side of the gloo-tested host logic (tests/test_sharding_gloo.py): build_*_job above, run_*_job here
            job = sets[i % nsets]["job"]
            if args.split == "eyes":
                run_rank_job(job, interpolation=cfg["interp"])
            else:
                run_band_job(job, interpolation=cfg["interp"])
    elif single:
        left_h = noise_disc(size, size, 2 * rank)
        for k in range(nsets):
            src = torch.from_numpy(left_h).to(dev) if k == 0 else noise_disc_torch(size, size, 2 * rank + 100003 * k, dev)
            sets.append(dict(left=src, sbs=torch.empty((size, size, 3), dtype=torch.uint8, device=dev)))

        def step(i: int):
            b = sets[i % nsets]
            V.remap_tensors(transformer, [b["left"]], [b["sbs"]], radius=size / 2, interpolation=cfg["interp"])
    elif frames:
        # batch of SBS frames resident in HBM; units are the column halves (pitched views), each eye is
        # written straight into its half of the output SBS frame.  The job's frames*world frames are
        # dealt to the ranks by the product's partition (sharding.shard_range): this rank's block
        mine = list(shard_range(frames * world, rank, world))
        assert len(mine) == frames
        rots = None
        if cfg["rot"] == "calib":
            rots = [m for f in mine for m in calib_rotations(f)]
        for k in range(nsets):
            ins = [noise_disc_torch(size, 2 * size, f + 100003 * k, dev, cn) for f in mine]
            outs = [torch.empty((size, 2 * size, cn), dtype=torch.uint8, device=dev) for _ in mine]
            sets.append(dict(srcs=[v for fr in ins for v in (fr[:, :size], fr[:, size:])],
                             dsts=[v for fr in outs for v in (fr[:, :size], fr[:, size:])], sbs=outs[0]))

        def step(i: int):
            b = sets[i % nsets]
            V.remap_tensors(transformer, b["srcs"], b["dsts"], radius=size / 2, interpolation=cfg["interp"], rotations=rots)
    else:
        # seeded noise-disc frames; frame index = rank so ranks hold different pixels;
        # set 0 is the numpy-seeded pair the CPU baseline / parity check uses, the others are generated on the device
        left_h, right_h = noise_disc(size, size, 2 * rank, cn), noise_disc(size, size, 2 * rank + 1, cn)
        for k in range(nsets):
            if k == 0:
                left, right = torch.from_numpy(left_h).to(dev), torch.from_numpy(right_h).to(dev)
            else:
                left, right = (noise_disc_torch(size, size, 2 * rank + e + 100003 * k, dev, cn) for e in (0, 1))
            # (zeros: BORDER_TRANSPARENT leaves skipped pixels as they are -- the oracle starts from zeros too)
            sets.append(dict(left=left, right=right, sbs=torch.zeros((size, 2 * size, cn), dtype=torch.uint8, device=dev)))

        def step(i: int):
            b = sets[i % nsets]
            V.apply_lr_tensors(transformer, b["left"], b["right"], out=b["sbs"], size_output=(size, size),
                               interpolation=cfg["interp"], radius=("max" if "rscale" not in cfg else cfg["rscale"] * size / 2), boarder_mode=border)

    def barrier():
        if world > 1:
            dist.barrier()

    # cold cost of THIS workload: the first step creates the plan (wall time, synchronised)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    step(0)
    torch.cuda.synchronize(dev)
    first_call_ms = (time.perf_counter() - t0) * 1e3
    plan_create_ms = sum(p.create_ms for p in R._PLANS.values())

    # Setup, not steps: condition the GPU before the W warm-up steps.  On this pool the card comes out of its idle
    # power state slowly (the CPU-side input synthesis above leaves it idle for about a second) and its clocks follow
    # the KIND of work: after 0.5 s of a memory-bound filler kernel 20 timed C2 steps still ran at half speed
    # (0.105 ms against 0.053 ms after ~5 ms of the remap itself), and the HIP runtime stalls once for ~35 ms when
    # its launch queue first gets ~60 packets deep.  So the workload's own launch is replayed, untimed, for 0.5 s
    # (first one burst of 256 launches, then bursts of 64) -- the state a stream of frames runs in.  Nothing of it is
    # measured; the timed region below is exactly K steps after the W warm-up steps.
    # (ranks are aligned first: input synthesis and plan creation take different times on different ranks, and a rank
    # that waited idle at the barrier in front of the timed region would enter it with its clocks down again)
    torch.cuda.synchronize(dev)
    barrier()
    t_cond = time.perf_counter()
    n_cond, burst = 0, 256
    while not args.no_condition:
        for _ in range(burst):
            step(n_cond)
            n_cond += 1
        torch.cuda.synchronize(dev)
        burst = 64
        if time.perf_counter() - t_cond >= 0.5:
            break
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    # Two events on the launch stream bracket the K steps: (e1 - e0) / K is the average GPU time of
    # a step's launch(es) including the gaps between them.  (An event pair around every step would
    # add two packets per launch and, whenever the host is the slower side, measure host latency.)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()  # (behind the opening event's own host cost: the K steps start here)
    if args.streams > 1:
        # (consecutive steps use different buffer sets, so they are independent; the side streams start behind e0 and the
        # launch stream joins them in front of e1: the bracket still covers exactly the K steps)
        side = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
        for st in side:
            st.wait_event(e0)
        for i in range(args.steps):
            with torch.cuda.stream(side[i % args.streams]):
                step(args.warmup + i)
        for st in side:
            torch.cuda.current_stream(dev).wait_stream(st)
    else:
        for i in range(args.steps):
            step(args.warmup + i)
    e1.record()
    # (the host polls the closing event before it synchronises: a blocking hipDeviceSynchronize wakes tens of microseconds
    # after the GPU is done, which is several per cent of a 20-step region of 50 us kernels)
    while not e1.query():
        pass
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = allreduce_max(elapsed, dev)
    kernel_ms = e0.elapsed_time(e1) / args.steps
    kernel_ms_max = allreduce_max(kernel_ms, dev)
    per_rank_ms = allgather_scalar(kernel_ms, dev, world)

    # Second leg, NOT part of `value`: the same step for >= 1000 steps (>= 0.25 s), bracketed the same way.  A short timed window
    # (the driver's --steps 20) sits 3 - 5 % above what a stream of frames sustains (the card's clocks have not settled into the load:
    # round 3's verdict); the line carries both so that neither has to be taken on trust.
    sustained = None
    if not args.no_sustained:
        if args.steps >= SUSTAINED_MIN_STEPS:
            sus_steps, sus_ms, sus_wall = args.steps, kernel_ms_max, elapsed / args.steps * 1e3
        else:
            sus_steps = max(SUSTAINED_MIN_STEPS, int(math.ceil(0.25 / max(kernel_ms_max * 1e-3, 1e-6))))
            sus_steps = min(sus_steps, 20000)
            barrier()
            torch.cuda.synchronize(dev)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            ts = time.perf_counter()
            for i in range(sus_steps):
                step(args.warmup + args.steps + i)
            s1.record()
            while not s1.query():
                pass
            torch.cuda.synchronize(dev)
            barrier()
            sus_wall = allreduce_max(time.perf_counter() - ts, dev) / sus_steps * 1e3
            sus_ms = allreduce_max(s0.elapsed_time(s1) / sus_steps, dev)
        sustained = {"steps": sus_steps, "ms_per_step": round(sus_wall, 4), "kernel_ms": round(sus_ms, 4),
                     "frac": round(set_bytes / (sus_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "value": round((2 * size * size if strong else units * size * size * world) / (sus_wall * 1e-3) / 1e6, 1),
                     "window_over_sustained": round(kernel_ms_max / sus_ms, 4)}

    px_per_step = 2 * size * size if strong else units * size * size * world
    value = px_per_step * args.steps / elapsed / 1e6
    # all eyes of a step: source read once + destination written once, cn bytes per pixel (a batch is ceil(units/16) launches:
    # the figure is per step, i.e. per group of launches, for batch workloads)
    alg_bytes = set_bytes
    achieved = alg_bytes / (kernel_ms_max * 1e-3) / 1e9

    if rank == 0:
        paths = sorted({p.path for p in R._PLANS.values()})
        line = {
            "metric": "Mpixels/s dual-fisheye->SBS-equirect remap; achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (seeded uint8 noise inside the fisheye circle, black outside), resident in HBM, "
                    f"{nsets} input/output buffer sets rotated per step ({nsets * set_bytes / 2**20:.0f} MiB: L3-cold)"
                    + (" -- REHEARSAL: ranks share devices, gloo; not a measurement" if rehearsal else ""),
            "config": {"workload": f"{args.workload}: {cfg['desc']}", "units_per_step_per_gpu": units,
                       "arithmetic": "f64 coordinates (fused chain), u8 pixels with int32 fixed-point blend",
                       "sharding": {"frames": "frames over ranks (sharding.shard_range), no collective",
                                    "eyes": "ONE pair, one eye per rank (sharding.plan_shards / build_rank_job / run_rank_job), no collective",
                                    "bands": "ONE pair, output rows of every eye in N / 2 bands (sharding.plan_band_shards / build_band_job / "
                                             "run_band_job); every rank holds the whole source eye of its bands (upload not timed), "
                                             "no collective"}[args.split],
                       "split": args.split, "kernel_path": paths,
                       # which kernel family served the launches of this workload (v1c_plan_last_launch): the measured launch is the
                       # tiled kernel the workload is meant for, not the generic one
                       "kernels": sorted({p.last_launch() for p in R._PLANS.values() if p.last_launch()}),
                       "buffer_sets": nsets},
            # what the process group reported (a SCALE record then shows by itself that RCCL saw N ranks)
            "backend": (dist.get_backend() if world > 1 and dist.is_initialized() else "none (single process)"),
            "world_size_seen": (dist.get_world_size() if world > 1 and dist.is_initialized() else 1),
            "per_rank_kernel_ms": [round(ms, 4) for ms in per_rank_ms],
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": None,
                         "kernel_ms": round(kernel_ms_max, 4), "algorithmic_bytes_per_launch": alg_bytes,
                         "per_gpu_frac": [round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) for ms in per_rank_ms]},
            "cold": {"plan_create_ms": round(plan_create_ms, 3), "first_call_ms": round(first_call_ms, 3)},
        }
        if sustained is not None:
            line["sustained"] = sustained
        child = os.environ.get("V1C_BENCH_CHILD") == "1" or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if cfg["interp"] in (2, 4) and not frames and not single:
            # INTER_CUBIC / INTER_LANCZOS4 are not HBM-bound: K x K taps x 3 channels x 2 eyes exact integer multiply-accumulates per
            # output position, one v_perm_b32 + one v_dot2_i32_i16 per two of them (OpenCV's int16 weights admit no cheaper exact form:
            # HISTORY.md 4.5).  Ceiling = the VALU issue time of the kernel's instruction stream: waves x VALU instructions per wave x 4
            # cycles per wave-instruction on 1024 SIMDs at the part's MAXIMUM clock, 2.4 GHz (MI355X_MICROARCH.md; under load it holds 1.9 - 2.3: a
            # ceiling at a sustained clock -- 2.1 GHz until round 5 -- put bicubic above 1).  The instruction count is MEASURED by this
            # run (a `--pmc SQ_INSTS_VALU SQ_WAVES` child pass of the same workload: "source": "live"); without rocprofv3 the object
            # says so and carries no ceiling.  `frac` stays the HBM fraction (the metric); `valu_int` says how close the launch is to
            # what actually bounds it.
            line["roofline"]["bound"] = "valu-int"
            cyc, clk = 4.0, 2.4e9
            vi = None
            if world == 1 and not child and args.traffic == "live":
                vi = measure_valu_per_wave(args.workload)
            if vi is not None:
                floor_ms = vi["waves_per_launch"] * vi["valu_instr_per_wave"] * cyc / (1024 * clk) * 1e3
                line["roofline"]["valu_int"] = {"source": "live", "floor_ms": round(floor_ms, 4), "frac_of_ceiling": round(floor_ms / kernel_ms_max, 4),
                                                "valu_instr_per_wave": round(vi["valu_instr_per_wave"], 1), "waves_per_launch": round(vi["waves_per_launch"], 1),
                                                "kernel": vi["kernel"], "cycles_per_wave_instr": cyc, "clock_hz": clk,
                                                "note": "exact int16 tap arithmetic on VALU bounds this launch, not HBM; achieved / peak / frac "
                                                        "are still the HBM figures of the metric"}
            else:
                line["roofline"]["valu_int"] = {"source": "unmeasured (no rocprofv3 --pmc pass in this run)", "floor_ms": None, "frac_of_ceiling": None}
        if world == 1 and not child:
            if not frames:
                torch.cuda.synchronize(dev)
                gpu_out = sets[0]["sbs"].cpu().numpy()  # set 0 = the numpy-seeded pair (image)
            if args.traffic == "live":
                t = measure_traffic(args.workload)
                if t is not None:
                    line["roofline"]["traffic"] = t["hbm_bytes_per_step"]
                    line["roofline"]["traffic_source"] = "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run"
                    line["roofline"]["traffic_detail"] = t
            if line["roofline"]["traffic"] is None and args.traffic != "none":
                tfile = ROOT / "profiles" / "traffic_latest.json"
                try:
                    t = json.loads(tfile.read_text()).get(args.workload)
                    if t:
                        line["roofline"]["traffic"] = t.get("hbm_bytes_per_step")
                        line["roofline"]["traffic_source"] = f"file: profiles/traffic_latest.json ({t.get('tag')})"
                except Exception:  # noqa: BLE001
                    pass
            if not args.no_cold_extra and args.workload != "C4":
                try:
                    line["cold_C4_8192_lanczos4"] = cold_call(WORKLOADS["C4"], dev, V, build_transformer)
                except Exception as e:  # noqa: BLE001
                    line["cold_C4_8192_lanczos4"] = {"error": repr(e)}
            if not args.no_cold_extra and not frames and not single and not strong and cn == 3 and cfg.get("spec") is None:
                line["cold"].update(new_radius_cost(cfg, dev, V, build_transformer))
            if not args.no_cpu_baseline and not frames:
                cb, parity = cpu_baseline(cfg, left_h, right_h, gpu_out)
                line["cpu_baseline"] = cb
                line["parity_vs_oracle"] = parity
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
