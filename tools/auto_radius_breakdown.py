"""Where the time of apply_lr_tensors(radius="auto", auto_radius_on_device=True) goes: wall per call as bench.py's
`cold.new_radius_ms` takes it (5 calls behind a synchronisation), the host's share (200 calls queued without a synchronisation:
wall / call, and the stream's own time between two events), and the launches alone (the call recorded into a graph, replayed).
Under `rocprofv3 --kernel-trace --stats` the per-kernel averages.  V1C_LIB=<other build> for a comparison."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import chainspecs as CS  # noqa: E402
import vr180_convert_amd as V  # noqa: E402
from vr180_convert_amd import remapper as R  # noqa: E402

dev = torch.device("cuda", 0)
CASES = (("C1L", "C1", 2048, 4), ("C2", "C2", 4096, 1), ("C2L", "C2", 4096, 4))
only = sys.argv[1:] or [c[0] for c in CASES]
for name, chain, n, interp in CASES:
    if name not in only:
        continue
    t = CS.to_product(CS.FULL_CASES[chain][0])
    yy, xx = torch.meshgrid(torch.arange(n, device=dev), torch.arange(n, device=dev), indexing="ij")
    rr = (xx - n / 2) ** 2 + (yy - n / 2) ** 2
    base = torch.randint(40, 256, (n, n, 3), dtype=torch.uint8, device=dev)
    imgs = [base * (rr <= (n / 2 - 3 - 1.5 * k) ** 2)[..., None].to(torch.uint8) for k in range(12)]
    sbs = torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev)
    del yy, xx, rr

    def call(k, on_dev=True):
        V.apply_lr_tensors(t, imgs[k % 12], imgs[(k + 1) % 12], out=sbs, size_output=(n, n), interpolation=interp, radius="auto",
                           auto_radius_on_device=on_dev)

    def two_step(k, _=None):  # (round 5's first form: an estimate launch per image, the patch launch, the remap)
        srcs = [imgs[k % 12], imgs[(k + 1) % 12]]
        R.remap_tensors_auto(t, srcs, [sbs[:, :n], sbs[:, n:]], rad=R.auto_radius_tensor(srcs), interpolation=interp, size_input=(n, n))

    row = {}
    for key, fn in (("two_step", two_step), ("images", call)):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            fn(2 * k)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        row[f"queued200_{key}"] = {"host_ms": round(t_host * 1e3 / 200, 4), "stream_ms": round(e0.elapsed_time(e1) / 200, 4)}
        t0 = time.perf_counter()
        for k in range(2, 12, 2):
            fn(k)
        torch.cuda.synchronize()
        row[f"wall5_{key}_ms"] = round((time.perf_counter() - t0) * 1e3 / 5, 4)
    for key, on_dev in (("device", True), ("exact", False)):
        call(0, on_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(2, 12, 2):
            call(k, on_dev)
        torch.cuda.synchronize()
        row[f"wall5_{key}_ms"] = round((time.perf_counter() - t0) * 1e3 / 5, 4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for k in range(200):
        call(2 * k)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    row["queued200_host_ms"] = round(t_host * 1e3 / 200, 4)
    row["queued200_wall_ms"] = round((time.perf_counter() - t0) * 1e3 / 200, 4)
    row["queued200_stream_ms"] = round(e0.elapsed_time(e1) / 200, 4)
    s = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        call(0)
    graph.replay()
    torch.cuda.synchronize()
    e0.record()
    for k in range(200):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    row["graph_replay_ms"] = round(e0.elapsed_time(e1) / 200, 4)
    # the planned launch of the same pair (radius known: what the box-less kernels are measured against)
    V.apply_lr_tensors(t, imgs[0], imgs[1], out=sbs, size_output=(n, n), interpolation=interp, radius=n / 2 - 3)
    torch.cuda.synchronize()
    e0.record()
    for k in range(200):
        V.apply_lr_tensors(t, imgs[0], imgs[1], out=sbs, size_output=(n, n), interpolation=interp, radius=n / 2 - 3)
    e1.record()
    torch.cuda.synchronize()
    row["planned_stream_ms"] = round(e0.elapsed_time(e1) / 200, 4)
    print(name, row, flush=True)
