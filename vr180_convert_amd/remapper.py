"""``get_map`` / ``apply`` / ``apply_lr`` of the reference, running on MI355X.

Drop-in for ``vr180_convert/remapper.py`` (same names, keyword signatures incl. the ``boarder_*``
spelling, return types and error behaviour).  Instead of evaluating the transformer chain as
NumPy passes over a meshgrid and handing a float32 map to ``cv2.remap`` (remapper.py:50-58,
388-398), the chain is lowered once (``chain.lower_for_get_map``) and evaluated per output pixel
inside the HIP kernel that also does cv2.remap's fixed-point gather; eyes are written straight
into their half of the side-by-side buffer (remapper.py:517-518).

Inputs may be paths, ``numpy`` arrays (uploaded, results come back as ``numpy``) or CUDA
``torch`` tensors (device-resident in and out -- what ``bench.py`` times).
"""
from __future__ import annotations

import ctypes as C
import threading
import time
from collections import OrderedDict
import logging
from logging import getLogger
from pathlib import Path
from typing import Any, Literal, Sequence

import numpy as np
import torch
from numpy.typing import NDArray

from . import _abi, _hostpipe, _io, _native
from .chain import NotLowerable, TransformerBase, get_radius, lower_for_get_map
from .chain import DenormalizeTransformer, NormalizeTransformer

LOG = getLogger(__name__)

INTER_LANCZOS4 = _abi.INTER_LANCZOS4
BORDER_CONSTANT = _abi.BORDER_CONSTANT


# --------------------------------------------------------------------------------------------
# device plumbing
# --------------------------------------------------------------------------------------------
def _device(device: Any = None) -> torch.device:
    if not torch.cuda.is_available():
        raise _native.EngineUnavailable(
            "no HIP device visible: vr180_convert_amd runs the remap on an MI355X and has no CPU fallback"
        )
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device)
    if d.type != "cuda":
        raise ValueError("device must be a cuda (HIP) device")
    return torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())


try:  # the raw handle of the current stream without building a torch.cuda.Stream object (4 us per call)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def _stream_ptr(dev: torch.device) -> int:
    if _raw_stream is not None and dev.index is not None:
        return int(_raw_stream(dev.index))
    return int(torch.cuda.current_stream(dev).cuda_stream)


def border_scalar(value: Any) -> np.ndarray:
    """Python ``borderValue`` -> saturated uint8[4] the way cv2 fills its Scalar: a bare int sets
    only component 0, a tuple sets the leading components."""
    vals = [value] if np.isscalar(value) else list(value)
    out = np.zeros(4, np.uint8)
    for i, v in enumerate(vals[:4]):
        out[i] = int(min(255, max(0, np.rint(float(v)))))
    return out


def _check_image_tensor(t: torch.Tensor, what: str) -> None:
    if t.dtype != torch.uint8 or t.dim() != 3:
        raise TypeError(f"{what} must be a uint8 (H, W, C) tensor")
    cn = t.shape[2]
    if (cn > 1 and t.stride(2) != 1) or (t.shape[1] > 1 and t.stride(1) != cn) or (
        t.shape[0] > 1 and t.stride(0) < t.shape[1] * cn
    ):
        raise ValueError(f"{what}: pixels must be contiguous within a row (column-sliced views are fine)")


def marshal_units(srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], rots: Sequence[Any] | None, *,
                  src_hw: tuple[int, int], dst_wh: tuple[int, int], cn: int, device: torch.device | None):
    """The ``v1c_unit`` array of a launch (include/vr180_remap.h): one record per eye -- pointers and row
    pitches of its source view and of ITS half of the side-by-side output (remapper.py:517-518 becomes a
    pitch), optionally the 3x3 rotation replacing the chain's.  Validates shapes, layout and device;
    ``device=None`` skips the device check (the multi-rank CPU tests marshal host tensors)."""
    n = len(srcs)
    units = (_abi.Unit * n)()
    for k, (s, d) in enumerate(zip(srcs, dsts)):
        _check_image_tensor(s, "src")
        _check_image_tensor(d, "dst")
        if tuple(s.shape) != (*src_hw, cn) or tuple(d.shape) != (dst_wh[1], dst_wh[0], cn):
            raise ValueError(f"unit {k}: tensor shapes {tuple(s.shape)} -> {tuple(d.shape)} do not match the plan")
        if device is not None and (s.device != device or d.device != device):
            raise ValueError(f"unit {k}: tensors must live on {device}")
        units[k].src, units[k].dst = s.data_ptr(), d.data_ptr()
        units[k].src_pitch, units[k].dst_pitch = s.stride(0), d.stride(0)
        if rots is not None and rots[k] is not None:
            m = np.asarray(rots[k], dtype=np.float64).reshape(9)
            units[k].has_rot = 1
            for q in range(9):
                units[k].rot[q] = m[q]
    return units


class Plan:
    """Owner of one ``v1c_plan`` (see include/vr180_remap.h: v1c_plan_create)."""

    def __init__(self, chain: _abi.Chain, *, src_hw, dst_wh, cn, interpolation, border_mode, border_value, device):
        self.device = _device(device)
        self.src_hw, self.dst_wh, self.cn = tuple(src_hw), tuple(dst_wh), int(cn)
        self._h = C.c_void_p()
        self._memo: "OrderedDict[tuple, Any]" = OrderedDict()  # marshalled unit arrays of recent buffer sets
        bv = border_scalar(border_value)
        t0 = time.perf_counter()
        rc = _native.lib().v1c_plan_create(
            C.byref(self._h), self.device.index, C.byref(chain), src_hw[0], src_hw[1], dst_wh[1], dst_wh[0],
            cn, int(interpolation), int(border_mode), bv.ctypes.data,
        )
        _native.check(rc, "v1c_plan_create")
        self.create_ms = (time.perf_counter() - t0) * 1e3  # synchronous: host analysis + table uploads + tile boxes

    @property
    def path(self) -> str:
        """'ray' (fused separable tables + radial table), 'planar' (the same kernels on the normalised plane point: chains that do not start
        with an EquirectangularEncoder) or 'literal' (fp64 interpreter)."""
        return {1: "ray", 2: "planar"}.get(_native.lib().v1c_plan_path(self._h), "literal")

    def last_launch(self) -> str:
        """Kernel family of this plan's most recent launch group (``v1c_plan_last_launch``): 'generic', 'tile', 'mirror', 'batch',
        'rot_pair', 'cn' or 'cn_rot', with '+fixup' when a fix-up pass followed; '' before the first run.  For tests and the bench."""
        k = _native.lib().v1c_plan_last_launch(self._h)
        if k < 0:
            return ""
        return _abi.LAUNCH_NAMES.get(k & 0xFF, "?") + ("+fixup" if k & _abi.LAUNCH_FIXUP else "")

    def run(self, srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], rots: Sequence[Any] | None = None) -> None:
        n = len(srcs)
        if n == 0 or len(dsts) != n or (rots is not None and len(rots) != n):
            raise ValueError("srcs / dsts / rots lengths differ or are empty")
        # a steady stream of calls on the same buffers (video frames written in place, bench steps):
        # the validated, marshalled unit array of the previous call is reused
        sig = None
        if rots is None:
            try:
                sig = tuple((s.data_ptr(), s.stride(), s.shape, s.dtype, d.data_ptr(), d.stride(), d.shape, d.dtype) for s, d in zip(srcs, dsts))
            except AttributeError:
                sig = None
            # (a few buffer sets are remembered: double / triple buffering rotates them; dict get / set of one key
            # are atomic under the GIL, an entry is written once and never mutated)
            units = self._memo.get(sig) if sig is not None else None
            if units is not None:
                rc = _native.lib().v1c_plan_run(self._h, _stream_ptr(self.device), units, n)
                _native.check(rc, "v1c_plan_run")
                return
        units = marshal_units(srcs, dsts, rots, src_hw=self.src_hw, dst_wh=self.dst_wh, cn=self.cn, device=self.device)
        rc = _native.lib().v1c_plan_run(self._h, _stream_ptr(self.device), units, n)
        _native.check(rc, "v1c_plan_run")
        if sig is not None:
            self._memo[sig] = units
            while len(self._memo) > 64:
                try:
                    self._memo.popitem(last=False)
                except KeyError:  # another thread emptied it
                    break

    def run_auto(self, srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], rad: torch.Tensor | None = None,
                 rots: Sequence[Any] | None = None, threshold: int = 10) -> None:
        """``v1c_plan_run_auto``: the launch with the radius read from device memory -- ``rad`` = float64 ``(n, 2)`` (radius, status)
        pairs as ``v1c_get_radius_async`` writes them; the launch uses their maximum.  ``rad=None``: the estimates are taken from
        ``srcs`` themselves by the same call (``v1c_plan_run_auto_images``, ``threshold`` = get_radius's).  Raises NotImplementedError
        for chains / geometries the device-resident form does not serve (the caller then takes the radius to the host)."""
        units = marshal_units(srcs, dsts, rots, src_hw=self.src_hw, dst_wh=self.dst_wh, cn=self.cn, device=self.device)
        if rad is None:
            rc = _native.lib().v1c_plan_run_auto_images(self._h, _stream_ptr(self.device), units, len(srcs), int(threshold))
            _native.check(rc, "v1c_plan_run_auto_images")
            return
        if rad.dtype != torch.float64 or rad.dim() != 2 or rad.shape[1] != 2 or not rad.is_contiguous() or rad.device != self.device:
            raise ValueError("rad must be a contiguous float64 (n, 2) tensor on the plan's device")
        rc = _native.lib().v1c_plan_run_auto(self._h, _stream_ptr(self.device), units, len(srcs), rad.data_ptr(), int(rad.shape[0]))
        _native.check(rc, "v1c_plan_run_auto")

    def run_units(self, units: Any, n: int) -> None:
        """Launch an already marshalled ``v1c_unit`` array (``marshal_units``) on the current stream."""
        rc = _native.lib().v1c_plan_run(self._h, _stream_ptr(self.device), units, int(n))
        _native.check(rc, "v1c_plan_run")

    def release_captures(self) -> None:
        """Hand the plan's capture-owned unit buffers out again (``v1c_plan_release_captures``): a graph-captured launch of more than
        16 units keeps one of 4; call this once the graphs that recorded them are destroyed."""
        _native.check(_native.lib().v1c_plan_release_captures(self._h), "v1c_plan_release_captures")

    def get_map(self, rot: Any = None) -> tuple[torch.Tensor, torch.Tensor]:
        w, h = self.dst_wh
        xm = torch.empty((h, w), dtype=torch.float32, device=self.device)
        ym = torch.empty((h, w), dtype=torch.float32, device=self.device)
        r = None if rot is None else np.ascontiguousarray(np.asarray(rot, np.float64).reshape(9))
        rc = _native.lib().v1c_plan_get_map(
            self._h, _stream_ptr(self.device), xm.data_ptr(), ym.data_ptr(), xm.stride(0) * 4,
            None if r is None else r.ctypes.data,
        )
        _native.check(rc, "v1c_plan_get_map")
        return xm, ym

    def __del__(self):
        try:
            if self._h:
                _native.lib().v1c_plan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass


_PLANS: "OrderedDict[tuple, Plan]" = OrderedDict()
_PLAN_CACHE_SIZE = 32
_PLANS_LOCK = threading.Lock()  # one worker thread per device (sharding.py) shares this cache


_AUTO_LAST: dict = {}  # (chain, geometry) -> the radius its last radius="auto" call found on the host (_remap_host_radius)
_TLS = threading.local()  # .plans: the plans the calling thread's last remap_tensors ran (last_launch_kinds)


def last_launch_kinds() -> list[str]:
    """Kernel family per launch group of the calling thread's most recent ``remap_tensors`` / ``apply_lr_tensors`` call
    (``Plan.last_launch``: 'generic', 'tile', 'mirror', 'batch', 'rot_pair', 'cn', 'cn_rot', '+fixup' appended when a fix-up pass followed).  The engine
    serves the same bytes through several kernels; tests use this to make sure a case meant for a tiled kernel reached it."""
    return [p.last_launch() for p in getattr(_TLS, "plans", [])]


def last_auto_radius_form() -> str:
    """Which form the calling thread's most recent ``apply_lr_tensors(radius="auto")`` took: 'device' (``remap_tensors_auto``: the radius
    never left the GPU) or 'exact' (estimates brought to the host); '' before the first such call.  For tests and the bench."""
    return getattr(_TLS, "auto_form", "")


def clear_caches() -> None:
    """Forget every plan and lowered chain (cold-call measurements, tests)."""
    with _PLANS_LOCK:
        _PLANS.clear()
        _LOWERED.clear()
        _LAST_SHARED[0] = None
        _AUTO_LAST.clear()


def _plan_for(chain: _abi.Chain, *, src_hw, dst_wh, cn, interpolation, border_mode, border_value, device) -> Plan:
    dev = _device(device)
    key = (chain.key(), tuple(src_hw), tuple(dst_wh), cn, int(interpolation), int(border_mode),
           border_scalar(border_value).tobytes(), dev.index)
    with _PLANS_LOCK:
        plan = _PLANS.get(key)
        if plan is not None:
            _PLANS.move_to_end(key)
            return plan
    # created outside the lock (tens of ms: other devices' threads keep going); a racing duplicate is dropped
    plan = Plan(chain, src_hw=src_hw, dst_wh=dst_wh, cn=cn, interpolation=interpolation,
                border_mode=border_mode, border_value=border_value, device=dev)
    with _PLANS_LOCK:
        plan = _PLANS.setdefault(key, plan)
        _PLANS.move_to_end(key)
        while len(_PLANS) > _PLAN_CACHE_SIZE:
            _PLANS.popitem(last=False)
    return plan


def _split_single_rotation(chain: _abi.Chain) -> tuple[_abi.Chain, np.ndarray | None]:
    """Chains that differ only in the matrix of their single rotate stage (BASELINE config 5:
    per-frame, per-eye calibration, cli.py:308-319) share one plan: returns the chain with that
    matrix blanked plus the matrix, which then travels per unit (v1c_unit.rot).  Chains with no
    or several rotate stages are returned unchanged."""
    idx = [i for i in range(chain.n_ops) if chain.ops[i].opcode == _abi.OP_ROTATE]
    if len(idx) != 1:
        return chain, None
    i = idx[0]
    rot = np.array(chain.ops[i].p[:9], dtype=np.float64)
    blank = _abi.Chain.from_buffer_copy(bytes(chain))
    for q in range(9):
        blank.ops[i].p[q] = 1.0 if q in (0, 4, 8) else 0.0
    return blank, rot


def _host_map(transformer: TransformerBase, *, radius, size_input, size_output):
    """The reference's own recipe (remapper.py:50-58) for chains that cannot be lowered: user
    subclasses only expose NumPy ``transform``, so the map has to be computed by calling it."""
    xmap, ymap = np.meshgrid(np.arange(size_output[0]), np.arange(size_output[1]))
    full = (
        NormalizeTransformer()
        * transformer
        * DenormalizeTransformer(scale=(radius, radius), center=(size_input[1] // 2, size_input[0] // 2))
    )
    xmap, ymap = full.transform(xmap, ymap)
    return xmap.astype(np.float32), ymap.astype(np.float32)


_LOWERED: "OrderedDict[tuple, _abi.Chain]" = OrderedDict()
_LAST_SHARED: list = [None]  # [(key, Plan)] of the last remap_tensors call with one shared transformer


class _NoKey(Exception):
    """The transformer holds state the memo key cannot capture exactly: lower it on every call."""


_FIELD_NAMES: dict = {}  # type -> tuple of dataclass field names (built-in stages), None (a foreign dataclass), False (no dataclass)


def _param_key(obj: Any):
    """Exact, hashable image of a built-in transformer's parameters: floats by their bits
    (``float.hex``), arrays by dtype / shape / bytes, quaternion-likes by their four components,
    dataclass stages field by field.  ``repr`` is NOT used: NumPy prints arrays with 8 digits and
    honours ``np.set_printoptions``, numpy-quaternion prints ``%.15g`` -- two rotations 2e-9
    apart would share a key and the second call would silently reuse the first one's plan.
    (Runs on every call of the device-resident API: exact-type tests first, field names cached per class.)"""
    t = type(obj)
    if t is float:
        return ("f", obj.hex())
    if t is int:
        return ("i", obj)
    if obj is None or t is bool or t is str or t is bytes:
        return obj
    names = _FIELD_NAMES.get(t, 0)
    if names == 0:
        import dataclasses

        if dataclasses.is_dataclass(t):
            # a user subclass may carry state outside its fields
            names = tuple(f.name for f in dataclasses.fields(t)) if t.__module__ == TransformerBase.__module__ else None
        else:
            names = False
        _FIELD_NAMES[t] = names
    if names is None:
        raise _NoKey
    if names is not False:
        return (t.__name__, tuple([_param_key(getattr(obj, n)) for n in names]))
    if isinstance(obj, (bool, str, bytes)):
        return obj
    if isinstance(obj, (int, np.integer)):
        return ("i", int(obj))
    if isinstance(obj, (float, np.floating)):
        return ("f", float(obj).hex())
    if isinstance(obj, np.ndarray):
        if obj.dtype == object:
            raise _NoKey
        a = np.ascontiguousarray(obj)
        return ("nd", a.dtype.str, a.shape, a.tobytes())
    if isinstance(obj, (list, tuple)):
        return (t.__name__, tuple([_param_key(v) for v in obj]))
    if all(hasattr(obj, c) for c in "wxyz"):
        return ("q", tuple(float(getattr(obj, c)).hex() for c in "wxyz"))
    raise _NoKey


def _transformer_key(t: Any):
    try:
        return _param_key(t)
    except (_NoKey, TypeError, ValueError):
        return None


def _lower_cached(t: TransformerBase, *, radius, size_input, size_output) -> _abi.Chain:
    """``lower_for_get_map`` memoised on the chain's exact parameter set (``_param_key``): a steady
    stream of identical calls -- frames of a video, bench steps -- lowers once.  Chains holding
    anything the key cannot capture exactly (user subclasses, object arrays) are lowered every time."""
    k = _transformer_key(t)
    if k is None:
        return lower_for_get_map(t, radius=radius, size_input=size_input, size_output=size_output)
    key = (k, float(radius).hex(), tuple(size_input), tuple(size_output))
    with _PLANS_LOCK:  # (the per-device worker threads of sharding.remap_sharded share this cache)
        ch = _LOWERED.get(key)
    if ch is None:
        ch = lower_for_get_map(t, radius=radius, size_input=size_input, size_output=size_output)
        with _PLANS_LOCK:
            _LOWERED[key] = ch
            while len(_LOWERED) > 64:
                _LOWERED.popitem(last=False)
    return ch


def remap_tensors(
    transformer: TransformerBase | Sequence[TransformerBase],
    srcs: Sequence[torch.Tensor],
    dsts: Sequence[torch.Tensor],
    *,
    radius: float,
    interpolation: int = INTER_LANCZOS4,
    boarder_mode: int = BORDER_CONSTANT,
    boarder_value: Any = 0,
    size_input: tuple[int, int] | None = None,
    rotations: Sequence[Any] | None = None,
) -> list[str]:
    """Device-resident core of ``apply``: remap every ``srcs[k]`` into ``dsts[k]`` (same device,
    same shapes) on the current stream.  ``transformer`` is one chain shared by all units
    (remapper.py:381-398) or one per unit.  ``rotations`` (optional, one 3x3 matrix or quaternion
    per unit) replaces the rotation of the chain's single ``Euclidean3DRotator`` per unit -- the
    per-frame, per-eye calibration of the CLI (cli.py:308-319) without lowering a chain per unit.
    Returns the code path used per launch group ('ray' / 'literal' / 'lut').  Nothing is
    synchronised."""
    n = len(srcs)
    if n == 0:
        return []
    if rotations is not None:
        return _remap_with_rotations(transformer, srcs, dsts, rotations, radius=radius, interpolation=interpolation,
                                     boarder_mode=boarder_mode, boarder_value=boarder_value, size_input=size_input)
    # one shared transformer, same geometry and parameters as the previous call: straight to its plan
    # (the key is the chain's exact parameter set, see _param_key)
    memo_key = None
    if not isinstance(transformer, (list, tuple)) and len(dsts) == n and isinstance(srcs[0], torch.Tensor) and isinstance(dsts[0], torch.Tensor):
        tk = _transformer_key(transformer)
        if tk is not None and srcs[0].dim() == 3 and dsts[0].dim() == 3:
            try:
                memo_key = (tk, float(radius).hex(), srcs[0].shape, dsts[0].shape, int(interpolation), int(boarder_mode),
                            border_scalar(boarder_value).tobytes(), None if size_input is None else tuple(size_input), srcs[0].device)
            except (TypeError, ValueError):
                memo_key = None
            last = _LAST_SHARED[0]  # (one read: another thread may replace the entry at any time)
            if memo_key is not None and last is not None and last[0] == memo_key:
                plan = last[1]
                plan.run(srcs, dsts, None)
                _TLS.plans = [plan]
                return [plan.path_cached]
    dev = srcs[0].device
    cn = int(srcs[0].shape[2])
    dst_wh = (int(dsts[0].shape[1]), int(dsts[0].shape[0]))
    groups, host_mapped = group_units(transformer, srcs, dsts, radius=radius, size_input=size_input)
    paths: list[str] = []
    _TLS.plans = []
    for k, t, size_in_k in host_mapped:
        xm, ym = _host_map(t, radius=radius, size_input=size_in_k, size_output=dst_wh)
        xm_d, ym_d = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        bv = border_scalar(boarder_value)
        _check_image_tensor(srcs[k], "src")
        _check_image_tensor(dsts[k], "dst")
        rc = _native.lib().v1c_remap_lut(
            dev.index, _stream_ptr(dev), srcs[k].data_ptr(), srcs[k].shape[0], srcs[k].shape[1], srcs[k].stride(0), cn,
            dsts[k].data_ptr(), dst_wh[1], dst_wh[0], dsts[k].stride(0), xm_d.data_ptr(), ym_d.data_ptr(),
            xm_d.stride(0) * 4, int(interpolation), int(boarder_mode), bv.ctypes.data,
        )
        _native.check(rc, "v1c_remap_lut")
        paths.append("lut")
    for g in groups:
        plan = _plan_for(g.chain, src_hw=g.src_hw, dst_wh=dst_wh, cn=cn, interpolation=interpolation,
                         border_mode=boarder_mode, border_value=boarder_value, device=dev)
        plan.run(g.srcs, g.dsts, g.rots)
        _TLS.plans.append(plan)
        paths.append(plan.path)
        if memo_key is not None and len(groups) == 1 and not host_mapped and g.rots is None and len(g.srcs) == n:
            plan.path_cached = paths[-1]
            _LAST_SHARED[0] = (memo_key, plan)
    return paths


class LaunchGroup:
    """Units of one call that share a plan: ``chain`` is what the plan is created from (the group's one
    rotation baked in, or blanked when rotations differ per unit and travel in ``rots``)."""

    __slots__ = ("chain", "src_hw", "srcs", "dsts", "rots", "index")

    def __init__(self, chain, src_hw, srcs, dsts, rots, index):
        self.chain, self.src_hw, self.srcs, self.dsts, self.rots, self.index = chain, src_hw, srcs, dsts, rots, index


def group_units(transformer, srcs, dsts, *, radius: float, size_input: tuple[int, int] | None = None):
    """Host logic of a call, no device involved: lower the chain of every unit (one shared transformer or
    one per unit), split units into groups that share a plan -- same chain up to the matrix of a single
    rotate stage, same source size -- and list the units whose chain cannot be lowered.

    Returns ``(groups, host_mapped)``: ``LaunchGroup`` records in first-appearance order and
    ``(unit index, transformer, size_input)`` for units that take the map from their own ``transform()``.
    Used by ``remap_tensors`` and by the multi-GPU dispatcher (sharding.py), whose CPU tests run it as is."""
    n = len(srcs)
    per_unit = list(transformer) if isinstance(transformer, (list, tuple)) else [transformer] * n
    if len(per_unit) != n or len(dsts) != n:
        raise ValueError("need one transformer and one dst per src")
    src_hw = (int(srcs[0].shape[0]), int(srcs[0].shape[1]))
    dst_wh = (int(dsts[0].shape[1]), int(dsts[0].shape[0]))
    acc: "OrderedDict[bytes, dict]" = OrderedDict()
    lowered: dict[Any, Any] = {}
    host_mapped = []
    for k, t in enumerate(per_unit):
        # the Denormalize centre is (W_in // 2, H_in // 2) of the image the map is FOR: with one shared
        # transformer that is images[0] (remapper.py:385), with per-eye transformers every eye is its own
        # apply() call and uses its own shape (remapper.py:460-473) -- odd-width SBS files split into W // 2
        # and W - W // 2 columns
        size_in_k = tuple(size_input) if size_input is not None else (
            src_hw if not isinstance(transformer, (list, tuple)) else (int(srcs[k].shape[0]), int(srcs[k].shape[1])))
        lk = (id(t), size_in_k)
        if lk not in lowered:
            try:
                lowered[lk] = _lower_cached(t, radius=radius, size_input=size_in_k, size_output=dst_wh)
            except NotLowerable as e:
                LOG.warning("transformer chain is not lowerable (%s): map evaluated by its own NumPy transform()", e)
                lowered[lk] = None
        chain = lowered[lk]
        if chain is None:
            host_mapped.append((k, t, size_in_k))
            continue
        shared, rot = _split_single_rotation(chain)
        key = bytes(shared) + repr(tuple(srcs[k].shape)).encode()  # different source sizes: different plans
        g = acc.setdefault(key, {"chain": shared, "full": chain, "src_hw": tuple(int(v) for v in srcs[k].shape[:2]),
                                 "srcs": [], "dsts": [], "rots": [], "index": []})
        g["srcs"].append(srcs[k])
        g["dsts"].append(dsts[k])
        g["rots"].append(rot)
        g["index"].append(k)
    groups = []
    for g in acc.values():
        # one rotation for the whole group (or none): bake it into the plan, which then computes its
        # tile boxes once and shares coordinates between units; rotations that differ per unit
        # travel with the units and share the rotation-blanked plan
        r0 = g["rots"][0]
        uniform = r0 is None or all(np.array_equal(r, r0) for r in g["rots"])
        groups.append(LaunchGroup(g["full"] if uniform else g["chain"], g["src_hw"], g["srcs"], g["dsts"],
                                  None if uniform else g["rots"], g["index"]))
    return groups, host_mapped


def _remap_with_rotations(transformer, srcs, dsts, rotations, *, radius, interpolation, boarder_mode, boarder_value, size_input):
    from .quat import as_rotation_matrix

    if isinstance(transformer, (list, tuple)) or len(rotations) != len(srcs) or len(dsts) != len(srcs):
        raise ValueError("rotations= needs ONE transformer chain and one rotation / dst per src")
    dev = srcs[0].device
    src_hw = (int(srcs[0].shape[0]), int(srcs[0].shape[1]))
    dst_wh = (int(dsts[0].shape[1]), int(dsts[0].shape[0]))
    chain = lower_for_get_map(transformer, radius=radius, size_input=size_input or src_hw, size_output=dst_wh)
    shared, rot = _split_single_rotation(chain)
    if rot is None:
        raise ValueError("rotations= needs a chain with exactly one Euclidean3DRotator")
    plan = _plan_for(shared, src_hw=src_hw, dst_wh=dst_wh, cn=int(srcs[0].shape[2]), interpolation=interpolation,
                     border_mode=boarder_mode, border_value=boarder_value, device=dev)
    plan.run(srcs, dsts, [as_rotation_matrix(r) for r in rotations])
    _TLS.plans = [plan]
    return [plan.path]


# --------------------------------------------------------------------------------------------
# reference API
# --------------------------------------------------------------------------------------------
def get_map(
    transformer: TransformerBase,
    *,
    radius: float,
    size_input: tuple[int, int],
    size_output: tuple[int, int] = (2048, 2048),
    device: Any = None,
) -> tuple[NDArray[np.float32], NDArray[np.float32]]:
    """Generate the remap map (reference remapper.py:23-59): float32 ``xmap, ymap`` of shape
    ``(size_output[1], size_output[0])``.  Evaluated on the GPU by the same code the fused kernel
    uses; non-lowerable chains are evaluated through their own ``transform`` like the reference."""
    try:
        chain = lower_for_get_map(transformer, radius=radius, size_input=size_input, size_output=size_output)
    except NotLowerable:
        return _host_map(transformer, radius=radius, size_input=size_input, size_output=size_output)
    plan = _plan_for(chain, src_hw=(max(1, size_input[0]), max(1, size_input[1])), dst_wh=size_output, cn=3,
                     interpolation=_abi.INTER_LINEAR, border_mode=BORDER_CONSTANT, border_value=0, device=device)
    xm, ym = plan.get_map()
    return xm.cpu().numpy(), ym.cpu().numpy()


def get_radius_smart(radius: float | Literal["auto", "max"], images: Sequence[Any]) -> float:
    """Reference remapper.py:62-90.  ``images`` may be numpy arrays or device tensors."""
    if isinstance(radius, str) and radius == "auto":
        radius_ = max(_get_radius_any(im) for im in images)
    elif isinstance(radius, str) and radius == "max":
        radius_ = min(images[0].shape[0] / 2, images[0].shape[1] / 2)
    else:
        radius_ = radius
    if LOG.isEnabledFor(logging.INFO):
        LOG.info(f"Radius: {radius_}, strategy: {radius}, image shape: {tuple(images[0].shape)}")
    return radius_


def _get_radius_any(im: Any, threshold: int = 10) -> float:
    if isinstance(im, torch.Tensor) and im.is_cuda:
        _check_image_tensor(im, "image")
        r = C.c_double()
        rc = _native.lib().v1c_get_radius(im.device.index, _stream_ptr(im.device), im.data_ptr(), im.shape[0], im.shape[1],
                                          im.stride(0), im.shape[2], threshold, C.byref(r))
        if rc == _abi.E_INVALID and b"no black border" in _native.lib().v1c_last_error():
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # what the reference raises
        _native.check(rc, "v1c_get_radius")
        return r.value
    return float(get_radius(np.asarray(im), threshold=threshold))


def _to_device(img: Any, dev: torch.device) -> torch.Tensor:
    if isinstance(img, torch.Tensor):
        return img if img.device == dev else img.to(dev)
    a = np.asarray(img)
    if a.dtype != np.uint8:
        raise TypeError("images must be uint8")  # cv2.remap's fixed-point path is the uint8 one
    if a.ndim == 2:
        a = a[..., None]
    # column-sliced views (remapper.py:455-456) are made contiguous on the host before upload; read-only arrays
    # (memory-mapped .npy frames, Pillow buffers) are copied: torch refuses to alias them silently
    a = np.ascontiguousarray(a) if a.flags.writeable else np.array(a, order="C")
    return torch.from_numpy(a).to(dev, non_blocking=False)


def apply(
    transformer: TransformerBase,
    *,
    in_paths: Sequence[Path | str | NDArray] | Path | str | NDArray,
    out_paths: Sequence[Path | str] | None | Path | str = None,
    size_output: tuple[int, int] = (2048, 2048),
    interpolation: int = INTER_LANCZOS4,
    boarder_mode: int = BORDER_CONSTANT,
    boarder_value: int | tuple[int, int, int] = 0,
    radius: float | Literal["auto", "max"] = "auto",
    device: Any = None,
) -> Sequence[NDArray[np.uint8]]:
    """Apply transformer to images (reference remapper.py:324-403).

    Returns one ``(size_output[1], size_output[0], C)`` uint8 array per input (CUDA tensors in ->
    CUDA tensors out).  One map is shared by all images and, like the reference, takes its
    geometry from ``images[0]``."""
    in_paths_ = [in_paths] if isinstance(in_paths, (str, Path, np.ndarray, torch.Tensor)) else in_paths
    out_paths_ = [out_paths] if isinstance(out_paths, (str, Path)) else out_paths
    del in_paths, out_paths

    images = _io.imread_many(list(in_paths_))
    radius_ = get_radius_smart(radius, images)
    on_device = all(isinstance(im, torch.Tensor) and im.is_cuda for im in images)
    dev = images[0].device if on_device else _device(device)

    # host-resident batch: copies in, remap and copies out of different groups overlap (_hostpipe.py)
    host_imgs = None if on_device else [np.asarray(im) if not isinstance(im, torch.Tensor) else None for im in images]
    if (host_imgs is not None and _hostpipe.enabled(len(images), len(images) * size_output[0] * size_output[1] * 4) and all(a is not None and a.dtype == np.uint8 and a.ndim in (2, 3)
                                                                        and a.shape == host_imgs[0].shape for a in host_imgs)
            and boarder_mode != _abi.BORDER_TRANSPARENT):
        imgs3 = [a[..., None] if a.ndim == 2 else a for a in host_imgs]
        size_in = (int(imgs3[0].shape[0]), int(imgs3[0].shape[1]))

        def _group(srcs_g, dsts_g):
            remap_tensors(transformer, srcs_g, dsts_g, radius=radius_, interpolation=interpolation, boarder_mode=boarder_mode,
                          boarder_value=boarder_value, size_input=size_in)

        results = _hostpipe.run(imgs3, dev, (size_output[1], size_output[0], int(imgs3[0].shape[2])), _group)
        results = [r[..., 0] if a.ndim == 2 else r for r, a in zip(results, host_imgs)]
        if out_paths_ is not None:
            _io.imwrite_many(list(out_paths_)[: len(results)], results)
        return results

    srcs = [_to_device(im, dev) for im in images]
    cn = srcs[0].shape[2]
    dsts = [torch.empty((size_output[1], size_output[0], cn), dtype=torch.uint8, device=dev) for _ in srcs]
    if boarder_mode == _abi.BORDER_TRANSPARENT:
        for d in dsts:  # cv2 leaves skipped pixels uninitialised; make them deterministic
            d.zero_()
    remap_tensors(transformer, srcs, dsts, radius=radius_, interpolation=interpolation, boarder_mode=boarder_mode,
                  boarder_value=boarder_value, size_input=(int(srcs[0].shape[0]), int(srcs[0].shape[1])))
    if on_device:
        results: list[Any] = dsts
    else:
        results = [d.cpu().numpy() for d in dsts]
        results = [r[..., 0] if np.asarray(im).ndim == 2 else r for r, im in zip(results, images)]
    if out_paths_ is not None:
        _io.imwrite_many(list(out_paths_)[: len(results)],
                         [image.cpu().numpy() if isinstance(image, torch.Tensor) else image for image in results])
    return results


def auto_radius_tensor(images: Sequence[torch.Tensor], threshold: int = 10) -> torch.Tensor:
    """``get_radius`` (transformer.py:108-140) of every image, left on the device: a float64 ``(n, 2)`` tensor of (radius, status) pairs
    (``v1c_get_radius_async``; status 1.0 / radius NaN where the reference raises IndexError).  Current stream, nothing synchronised."""
    dev = images[0].device
    rad = torch.empty((len(images), 2), dtype=torch.float64, device=dev)
    for k, im in enumerate(images):
        _check_image_tensor(im, "image")
        rc = _native.lib().v1c_get_radius_async(dev.index, _stream_ptr(dev), im.data_ptr(), im.shape[0], im.shape[1], im.stride(0), im.shape[2],
                                                threshold, rad.data_ptr() + 16 * k)
        _native.check(rc, "v1c_get_radius_async")
    return rad


def remap_tensors_auto(transformer: TransformerBase, srcs: Sequence[torch.Tensor], dsts: Sequence[torch.Tensor], *,
                       rad: torch.Tensor | None = None, interpolation: int = INTER_LANCZOS4, boarder_mode: int = BORDER_CONSTANT,
                       boarder_value: Any = 0, size_input: tuple[int, int] | None = None) -> None:
    """``remap_tensors`` with ``radius="auto"`` -- the reference's default, remapper.py:333 -- and the radius never leaving the device:
    estimated from ``srcs`` (or given as ``rad``, ``auto_radius_tensor``), the maximum taken and the Denormalize scale set by ONE small
    launch in front of the remap launch (``v1c_plan_run_auto_images`` / ``v1c_plan_run_auto``).  No stream synchronisation, no
    plan per image (ONE plan serves every radius: it is keyed on the nominal radius "max"), graph-capturable once the plan exists.
    Where the reference raises IndexError (an image without a black border) the output is the border colour.
    Raises NotImplementedError for chains the device-resident form does not serve (see include/vr180_remap.h)."""
    if isinstance(transformer, (list, tuple)):
        raise ValueError("remap_tensors_auto takes ONE transformer (per-eye transformers: one call per eye, as apply_lr does)")
    dev = srcs[0].device
    src_hw = (int(srcs[0].shape[0]), int(srcs[0].shape[1]))
    dst_wh = (int(dsts[0].shape[1]), int(dsts[0].shape[0]))
    size_in = tuple(size_input) if size_input is not None else src_hw
    nominal = min(size_in[0] / 2, size_in[1] / 2)  # (any radius: the launch ignores the plan's own)
    chain = _lower_cached(transformer, radius=nominal, size_input=size_in, size_output=dst_wh)
    plan = _plan_for(chain, src_hw=src_hw, dst_wh=dst_wh, cn=int(srcs[0].shape[2]), interpolation=interpolation,
                     border_mode=boarder_mode, border_value=boarder_value, device=dev)
    plan.run_auto(srcs, dsts, rad)  # (rad None: the call estimates from `srcs` itself -- one small launch in front of the remap)
    _TLS.plans = [plan]


def _remap_host_radius(transformer: TransformerBase, srcs, dsts, r: float, *, interpolation, boarder_mode, boarder_value, size_input) -> bool:
    """``radius="auto"`` with the estimate taken to the host (the exact form: IndexError like the reference), for an image circle that
    moved: the launch that reads the radius from device memory (``v1c_plan_run_auto``: one plan whatever the radius, the kernels
    without plan-time boxes) instead of a plan per radius -- 0.2 ms instead of 0.6 - 1.2 ms of plan creation per new radius.  A
    circle that repeats (the same estimate twice in a row) gets its own plan and the planned kernels.  Same bytes either way.
    False: not served (chains the device-resident launch does not take, a repeated radius) -- the caller runs the planned path."""
    k = _transformer_key(transformer)
    if k is None:
        return False
    dev = srcs[0].device
    key = (k, tuple(size_input), tuple(int(v) for v in dsts[0].shape[:2]), int(srcs[0].shape[2]), int(interpolation), int(boarder_mode),
           border_scalar(boarder_value).tobytes(), dev.index)
    with _PLANS_LOCK:
        last = _AUTO_LAST.get(key)
        _AUTO_LAST[key] = r
        while len(_AUTO_LAST) > 64:
            _AUTO_LAST.pop(next(iter(_AUTO_LAST)))
    if last == r:
        return False
    try:
        rad = torch.tensor([[r, 0.0]], dtype=torch.float64, device=dev)
        remap_tensors_auto(transformer, srcs, dsts, rad=rad, interpolation=interpolation, boarder_mode=boarder_mode, boarder_value=boarder_value,
                           size_input=size_input)
        return True
    except NotImplementedError:
        return False


def apply_lr_tensors(
    transformer: TransformerBase | tuple[TransformerBase, TransformerBase],
    left: torch.Tensor,
    right: torch.Tensor,
    *,
    out: torch.Tensor | None = None,
    size_output: tuple[int, int] = (2048, 2048),
    interpolation: int = INTER_LANCZOS4,
    boarder_mode: int = BORDER_CONSTANT,
    boarder_value: Any = 0,
    radius: float | Literal["auto", "max"] = "auto",
    auto_radius_on_device: bool | None = None,
) -> torch.Tensor:
    """Device-resident ``apply_lr`` (merge=False): both eyes are remapped by ONE launch straight
    into the halves of the ``(H, 2W, C)`` side-by-side tensor (remapper.py:460-484, 517-518).

    ``radius="auto"`` (the reference's default) has two forms.  The exact one brings each estimate to the host (one stream
    synchronisation; raises IndexError like the reference when an image has no black border; a circle that moved is served by the
    launch that reads the radius from device memory -- no plan per radius --, a circle that repeats by a plan of its own).
    ``auto_radius_on_device=True`` keeps it on the device (``remap_tensors_auto``): no synchronisation, one plan for every radius,
    graph-capturable; default (None): taken when the current stream is being captured into a graph, where a synchronisation is illegal."""
    w, h = size_output
    cn = left.shape[2]
    if out is None:
        out = torch.empty((h, 2 * w, cn), dtype=torch.uint8, device=left.device)
        if boarder_mode == _abi.BORDER_TRANSPARENT:
            out.zero_()
    halves = [out[:, :w], out[:, w:]]
    if isinstance(radius, str) and radius == "auto":
        on_dev = auto_radius_on_device
        if on_dev is None:
            on_dev = bool(torch.cuda.is_current_stream_capturing())
        if on_dev:
            try:
                if isinstance(transformer, tuple):  # per-eye transformer AND per-eye radius (remapper.py:460-473)
                    for t, im, d in zip(transformer, (left, right), halves):
                        remap_tensors_auto(t, [im], [d], interpolation=interpolation, boarder_mode=boarder_mode, boarder_value=boarder_value)
                else:
                    remap_tensors_auto(transformer, [left, right], halves, interpolation=interpolation, boarder_mode=boarder_mode,
                                       boarder_value=boarder_value, size_input=(int(left.shape[0]), int(left.shape[1])))
                _TLS.auto_form = "device"
                return out
            except NotImplementedError:
                if auto_radius_on_device is None and torch.cuda.is_current_stream_capturing():
                    raise
                # (this chain is not served by the device-resident form: the exact one below)
        _TLS.auto_form = "exact"
    if isinstance(transformer, tuple):
        # per-eye transformer AND per-eye radius estimate (remapper.py:460-473)
        r = [get_radius_smart(radius, [im]) for im in (left, right)]
        if r[0] == r[1]:
            remap_tensors(list(transformer), [left, right], halves, radius=r[0], interpolation=interpolation,
                          boarder_mode=boarder_mode, boarder_value=boarder_value)
        else:
            for t, im, d, rr in zip(transformer, (left, right), halves, r):
                remap_tensors(t, [im], [d], radius=rr, interpolation=interpolation, boarder_mode=boarder_mode,
                              boarder_value=boarder_value)
    else:
        r_ = get_radius_smart(radius, [left, right])
        size_in = (int(left.shape[0]), int(left.shape[1]))
        if not (isinstance(radius, str) and radius == "auto" and
                _remap_host_radius(transformer, [left, right], halves, float(r_), interpolation=interpolation, boarder_mode=boarder_mode,
                                   boarder_value=boarder_value, size_input=size_in)):
            remap_tensors(transformer, [left, right], halves, radius=r_, interpolation=interpolation,
                          boarder_mode=boarder_mode, boarder_value=boarder_value, size_input=size_in)
    return out


def anaglyph_tensors(left: torch.Tensor, right: torch.Tensor) -> torch.Tensor:
    """``merge=True`` of apply_lr on the device (reference remapper.py:485-497): per-eye channel
    mean times a colour, summed, / 255 -- a float64 ``(H, W, 3)`` tensor like the reference's
    ``combine`` (the "L" / "R" labels of :498-516 are not drawn here).  Current stream, no sync."""
    _check_image_tensor(left, "left")
    _check_image_tensor(right, "right")
    if left.shape != right.shape or left.shape[2] != 3 or left.device != right.device:
        raise ValueError("anaglyph needs two (H, W, 3) uint8 tensors of the same shape on one device")
    h, w = int(left.shape[0]), int(left.shape[1])
    out = torch.empty((h, w, 3), dtype=torch.float64, device=left.device)
    rc = _native.lib().v1c_anaglyph(left.device.index, _stream_ptr(left.device), left.data_ptr(), left.stride(0),
                                    right.data_ptr(), right.stride(0), h, w, out.data_ptr(), out.stride(0) * 8)
    _native.check(rc, "v1c_anaglyph")
    return out


def apply_lr(
    transformer: TransformerBase | tuple[TransformerBase, TransformerBase],
    *,
    left_path: Path | str | NDArray,
    right_path: Path | str | NDArray,
    out_path: Path | str | None,
    size_output: tuple[int, int] = (2048, 2048),
    interpolation: int = INTER_LANCZOS4,
    boarder_mode: int = BORDER_CONSTANT,
    boarder_value: int | tuple[int, int, int] = 0,
    radius: float | Literal["auto", "max"] = "auto",
    merge: bool = False,
    device: Any = None,
) -> None:
    """Apply transformer to a pair of images and save them side by side (reference
    remapper.py:406-520).  ``left_path == right_path`` means one file holding both eyes."""
    if isinstance(left_path, (str, Path)) and isinstance(right_path, (str, Path)) and left_path == right_path:
        image = _io.imread(left_path)
        left_path = image[:, : image.shape[1] // 2]
        right_path = image[:, image.shape[1] // 2 :]
    left, right = _io.imread_many([left_path, right_path])  # (two files: decoded side by side, the codecs release the GIL)
    on_device = isinstance(left, torch.Tensor) and left.is_cuda
    dev = left.device if on_device else _device(device)
    lt, rt = _to_device(left, dev), _to_device(right, dev)
    # "auto" radius: estimated on the host images when they are numpy (one row each)
    sbs = apply_lr_tensors(transformer, lt, rt, size_output=size_output, interpolation=interpolation,
                           boarder_mode=boarder_mode, boarder_value=boarder_value,
                           radius=_radius_for_pair(radius, transformer, left, right))
    if merge:
        # red/cyan anaglyph of the two halves, on the device (remapper.py:485-497); the labels need
        # cv2.putText and are drawn on the host copy when cv2 is importable (:498-516)
        w = size_output[0]
        combine = _io.draw_anaglyph_labels(anaglyph_tensors(sbs[:, :w], sbs[:, w:]).cpu().numpy())
    else:
        combine = sbs.cpu().numpy()
    if out_path is not None:
        _io.imwrite(out_path, combine)
        LOG.info(f"Saved to {Path(out_path).absolute()}")


def _radius_for_pair(radius, transformer, left, right):
    """``radius`` argument to hand to apply_lr_tensors: estimate 'auto' on the host images when
    they are numpy (the reference's estimate, remapper.py:82-84), else let the device path do it."""
    if isinstance(radius, str) and radius == "auto" and not isinstance(left, torch.Tensor):
        if isinstance(transformer, tuple):
            r = [get_radius_smart("auto", [im]) for im in (left, right)]
            return r[0] if r[0] == r[1] else radius
        return get_radius_smart("auto", [left, right])
    return radius


def __getattr__(name: str):
    # rotation_match / rotation_match_robust / match_lr live in vr180_convert.remapper in the
    # reference (remapper.py:93-321); here they are in calibration.py (which imports this module)
    if name in ("rotation_match", "rotation_match_robust", "match_lr", "calibration_rotators"):
        from . import calibration

        return getattr(calibration, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
