"""Host-resident batches through the device without idle PCIe: the image I/O step either side of
the hot path (the reference holds its images as NumPy arrays between ``cv.imread``
and ``cv.imwrite``, remapper.py:371-402).

``apply()`` on NumPy inputs used to upload every image from pageable memory, launch, and download
every result -- three serial phases, each a blocking copy through the driver's bounce buffers
(2.6 Gpixel/s on a C2 pair).  Here the images of one call are cut into groups that share launches;
three streams run host->device copies, the remap and device->host copies of different groups at
the same time:

* inputs are page-locked in place (``hipHostRegister`` on the caller's array, released afterwards)
  when they are C-contiguous, else copied once into a page-locked staging tensor;
* results are written by the device straight into page-locked arrays that are handed back to the
  caller as ``numpy`` views (no extra host copy);
* device buffers form a ring of ``SLOTS`` groups; events order copy-in -> remap -> copy-out per
  slot and guard the reuse of a slot.

Nothing here computes pixels: the remap itself is ``remap_tensors`` (one plan, one map per call).
"""
from __future__ import annotations

import os
from typing import Any, Callable, Sequence

import numpy as np
import torch

SLOTS = 3
GROUP = 4  # images per group: large enough for the shared-map batch kernel, small enough to pipeline


def enabled(n_images: int) -> bool:
    return n_images >= 2 and os.environ.get("V1C_HOST_PIPELINE", "1") != "0"


class _Registered:
    """Page-locks a C-contiguous ndarray in place for the duration of the call."""

    def __init__(self) -> None:
        self._ptrs: list[int] = []
        self._rt = torch.cuda.cudart()

    def try_pin(self, a: np.ndarray) -> torch.Tensor | None:
        if not a.flags.c_contiguous or not a.flags.writeable or a.nbytes == 0:  # (torch wraps writable arrays only)
            return None
        ptr = a.ctypes.data
        try:
            rc = self._rt.cudaHostRegister(ptr, a.nbytes, 0)
        except Exception:  # noqa: BLE001 -- older torch builds: fall back to the staging copy
            return None
        if int(rc) != 0:
            return None
        self._ptrs.append(ptr)
        return torch.from_numpy(a)

    def release(self) -> None:
        for p in self._ptrs:
            try:
                self._rt.cudaHostUnregister(p)
            except Exception:  # noqa: BLE001
                pass
        self._ptrs.clear()


def run(images: Sequence[Any], dev: torch.device, out_hw_c: tuple[int, int, int],
        remap: Callable[[list[torch.Tensor], list[torch.Tensor]], None]) -> list[np.ndarray]:
    """``images``: uint8 (H, W, C) ndarrays of one shape (host).  ``remap(srcs, dsts)`` launches the
    remap of one group on the current stream.  Returns one page-locked-backed ndarray per image."""
    n = len(images)
    h, w, cn = out_hw_c
    in_shape = tuple(np.asarray(images[0]).shape)
    main = torch.cuda.current_stream(dev)
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    groups = [list(range(i, min(i + GROUP, n))) for i in range(0, n, GROUP)]
    nslots = min(SLOTS, len(groups))
    src_dev = [[torch.empty(in_shape, dtype=torch.uint8, device=dev) for _ in range(GROUP)] for _ in range(nslots)]
    dst_dev = [[torch.empty((h, w, cn), dtype=torch.uint8, device=dev) for _ in range(GROUP)] for _ in range(nslots)]
    ev_in = [torch.cuda.Event() for _ in range(nslots)]
    ev_done = [torch.cuda.Event() for _ in range(nslots)]   # remap of the slot's group finished (sources free)
    ev_out = [torch.cuda.Event() for _ in range(nslots)]    # results of the slot's group are on the host (dsts free)
    results = [torch.empty((h, w, cn), dtype=torch.uint8, pin_memory=True) for _ in range(n)]
    reg = _Registered()
    staged: list[torch.Tensor] = []  # keeps staging tensors alive until the copies have run
    s_in.wait_stream(main)
    try:
        for g, idx in enumerate(groups):
            slot = g % nslots
            with torch.cuda.stream(s_in):
                if g >= nslots:
                    s_in.wait_event(ev_done[slot])
                for k, i in enumerate(idx):
                    a = np.asarray(images[i])
                    host = reg.try_pin(a)
                    if host is None:  # views / registration refused: one copy into page-locked staging
                        host = torch.empty(in_shape, dtype=torch.uint8, pin_memory=True)
                        host.numpy()[...] = a
                        staged.append(host)
                    src_dev[slot][k].copy_(host, non_blocking=True)
                ev_in[slot].record(s_in)
            main.wait_event(ev_in[slot])
            if g >= nslots:
                main.wait_event(ev_out[slot])
            remap(src_dev[slot][: len(idx)], dst_dev[slot][: len(idx)])
            ev_done[slot].record(main)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_done[slot])
                for k, i in enumerate(idx):
                    results[i].copy_(dst_dev[slot][k], non_blocking=True)
                ev_out[slot].record(s_out)
        s_out.synchronize()
        main.wait_stream(s_out)
        s_in.synchronize()
    finally:
        torch.cuda.synchronize(dev)
        reg.release()
    return [r.numpy() for r in results]
