"""ctypes binding of the HIP engine (``csrc/libvr180remap.so``, C ABI in include/vr180_remap.h).

There is no CPU fallback: if the library is missing, every image entry point raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C
vr180_convert_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _abi

import os

# V1C_LIB: alternative build of the same ABI (kernel-tuning experiments); default = the in-tree build
LIB_PATH = Path(os.environ.get("V1C_LIB") or (Path(__file__).resolve().parent / "csrc" / "libvr180remap.so"))

SYMBOLS = [
    "v1c_abi_version", "v1c_device_count", "v1c_last_error", "v1c_plan_create", "v1c_plan_destroy",
    "v1c_plan_path", "v1c_plan_run", "v1c_plan_get_map", "v1c_remap_fused", "v1c_remap_lut",
    "v1c_get_radius", "v1c_get_radius_async", "v1c_build_itab", "v1c_anaglyph", "v1c_fused_cache_size", "v1c_plan_last_launch",
    "v1c_plan_release_captures", "v1c_plan_run_auto", "v1c_plan_run_auto_images",
]


class EngineUnavailable(RuntimeError):
    """The HIP library is not built / not loadable."""


class EngineError(RuntimeError):
    """A C-ABI call returned an error code."""


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineUnavailable(
            f"{LIB_PATH} is missing: build the HIP engine first (make -C {LIB_PATH.parent}); "
            "vr180_convert_amd has no CPU fallback"
        )
    try:
        L = C.CDLL(str(LIB_PATH))
    except OSError as e:  # pragma: no cover - depends on the machine
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    L.v1c_abi_version.restype = i32
    L.v1c_device_count.restype = i32
    L.v1c_last_error.restype = C.c_char_p
    L.v1c_plan_create.argtypes = [C.POINTER(vp), i32, C.POINTER(_abi.Chain), i32, i32, i32, i32, i32, i32, i32, vp]
    L.v1c_plan_destroy.argtypes = [vp]
    L.v1c_plan_path.argtypes = [vp]
    L.v1c_plan_last_launch.argtypes = [vp]
    L.v1c_plan_last_launch.restype = i32
    try:
        L.v1c_plan_release_captures.argtypes = [vp]
        L.v1c_plan_run_auto.argtypes = [vp, vp, C.POINTER(_abi.Unit), i32, vp, i32]
        L.v1c_plan_run_auto_images.argtypes = [vp, vp, C.POINTER(_abi.Unit), i32, i32]
    except AttributeError:  # (an older build behind V1C_LIB: A/B runs of tools/ab.sh only)
        pass
    L.v1c_plan_run.argtypes = [vp, vp, C.POINTER(_abi.Unit), i32]
    L.v1c_plan_get_map.argtypes = [vp, vp, vp, vp, i64, vp]
    L.v1c_remap_fused.argtypes = [i32, vp, vp, i32, i32, i64, i32, vp, i32, i32, i64, C.POINTER(_abi.Chain), i32, i32, vp]
    L.v1c_remap_lut.argtypes = [i32, vp, vp, i32, i32, i64, i32, vp, i32, i32, i64, vp, vp, i64, i32, i32, vp]
    L.v1c_get_radius.argtypes = [i32, vp, vp, i32, i32, i64, i32, i32, C.POINTER(C.c_double)]
    L.v1c_get_radius_async.argtypes = [i32, vp, vp, i32, i32, i64, i32, i32, vp]
    L.v1c_fused_cache_size.restype = i32
    L.v1c_build_itab.argtypes = [i32, vp]
    L.v1c_anaglyph.argtypes = [i32, vp, vp, i64, vp, i64, i32, i32, vp, i64]
    if L.v1c_abi_version() != _abi.ABI_VERSION:
        raise EngineUnavailable("libvr180remap.so was built for a different ABI version; rebuild it")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    """Raise the Python exception matching a non-zero return code."""
    if rc == _abi.OK:
        return
    msg = lib().v1c_last_error().decode("utf-8", "replace")
    if rc == _abi.E_INVALID:
        raise ValueError(f"{what}: {msg}")
    if rc == _abi.E_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    raise EngineError(f"{what}: {msg} (code {rc})")
