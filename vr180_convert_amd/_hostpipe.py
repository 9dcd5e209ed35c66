"""Host-resident batches through the device without idle PCIe: the image I/O step either side of
the hot path (the reference holds its images as NumPy arrays between ``cv.imread``
and ``cv.imwrite``, remapper.py:371-402).

``apply()`` on NumPy inputs used to upload every image from pageable memory, launch, and download
every result -- three serial phases, each a blocking copy through the driver's bounce buffers
(2.6 Gpixel/s on a C2 pair).  Here the images of one call are cut into groups that share launches;
three streams run host->device copies, the remap and device->host copies of different groups at
the same time:

* inputs are copied into page-locked staging tensors by a small thread pool (NumPy releases the GIL
  for the copy), one group ahead of the transfers (page-locking the caller's arrays in place with
  ``hipHostRegister`` was 15 % faster but leaves sticky HIP errors behind when two small arrays share
  a page, which then surface in unrelated torch calls);
* results are written by the device straight into page-locked arrays that are handed back to the
  caller as ``numpy`` views (no extra host copy);
* device buffers form a ring of ``SLOTS`` groups; events order copy-in -> remap -> copy-out per
  slot and guard the reuse of a slot.

Nothing here computes pixels: the remap itself is ``remap_tensors`` (one plan, one map per call).
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Any, Callable, Sequence

import numpy as np
import torch

SLOTS = 3
COPY_THREADS = 8
GROUP = 4  # images per group: large enough for the shared-map batch kernel, small enough to pipeline


MAX_PINNED_BYTES = 4 << 30  # results are returned as page-locked arrays: beyond this the plain path is used


def enabled(n_images: int, result_bytes: int = 0) -> bool:
    return n_images >= 2 and result_bytes <= MAX_PINNED_BYTES and os.environ.get("V1C_HOST_PIPELINE", "1") != "0"


def run(images: Sequence[Any], dev: torch.device, out_hw_c: tuple[int, int, int],
        remap: Callable[..., None], with_index: bool = False) -> list[np.ndarray]:
    """``images``: uint8 (H, W, C) ndarrays of one shape (host).  ``remap(srcs, dsts)`` launches the
    remap of one group on the current stream (``with_index``: ``remap(srcs, dsts, idx)`` with the
    positions of the group's images in ``images``).  Returns one page-locked-backed ndarray per image."""
    n = len(images)
    h, w, cn = out_hw_c
    in_shape = tuple(np.asarray(images[0]).shape)
    main = torch.cuda.current_stream(dev)
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    groups = [list(range(i, min(i + GROUP, n))) for i in range(0, n, GROUP)]
    nslots = min(SLOTS, len(groups))
    stage = [[torch.empty(in_shape, dtype=torch.uint8, pin_memory=True) for _ in range(GROUP)] for _ in range(nslots)]
    src_dev = [[torch.empty(in_shape, dtype=torch.uint8, device=dev) for _ in range(GROUP)] for _ in range(nslots)]
    dst_dev = [[torch.empty((h, w, cn), dtype=torch.uint8, device=dev) for _ in range(GROUP)] for _ in range(nslots)]
    ev_in = [torch.cuda.Event() for _ in range(nslots)]     # the slot's staging buffers have been read
    ev_done = [torch.cuda.Event() for _ in range(nslots)]   # remap of the slot's group finished (sources free)
    ev_out = [torch.cuda.Event() for _ in range(nslots)]    # results of the slot's group are on the host (dsts free)
    results = [torch.empty((h, w, cn), dtype=torch.uint8, pin_memory=True) for _ in range(n)]

    def fill(slot: int, k: int, i: int) -> None:  # pageable (possibly strided) array -> page-locked staging
        np.copyto(stage[slot][k].numpy(), np.asarray(images[i]))

    s_in.wait_stream(main)
    with ThreadPoolExecutor(max_workers=COPY_THREADS) as pool:
        pending = [pool.submit(fill, 0, k, i) for k, i in enumerate(groups[0])]
        try:
            for g, idx in enumerate(groups):
                slot = g % nslots
                for f in pending:
                    f.result()
                with torch.cuda.stream(s_in):
                    if g >= nslots:
                        s_in.wait_event(ev_done[slot])
                    for k in range(len(idx)):
                        src_dev[slot][k].copy_(stage[slot][k], non_blocking=True)
                    ev_in[slot].record(s_in)
                # the staging copies of the next group run on the pool while this group is in flight
                pending = []
                if g + 1 < len(groups):
                    nslot = (g + 1) % nslots
                    if g + 1 >= nslots:
                        ev_in[nslot].synchronize()  # its staging buffers were last read by group g + 1 - nslots
                    pending = [pool.submit(fill, nslot, k, i) for k, i in enumerate(groups[g + 1])]
                main.wait_event(ev_in[slot])
                if g >= nslots:
                    main.wait_event(ev_out[slot])
                if with_index:
                    remap(src_dev[slot][: len(idx)], dst_dev[slot][: len(idx)], idx)
                else:
                    remap(src_dev[slot][: len(idx)], dst_dev[slot][: len(idx)])
                ev_done[slot].record(main)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_done[slot])
                    for k, i in enumerate(idx):
                        results[i].copy_(dst_dev[slot][k], non_blocking=True)
                    ev_out[slot].record(s_out)
            s_out.synchronize()
            main.wait_stream(s_out)
        finally:
            for f in pending:
                f.cancel()
            torch.cuda.synchronize(dev)
    return [r.numpy() for r in results]
