"""Builds tests/host_emul/libv1c_emul.so: the product's per-pixel code compiled for the HOST."""
from __future__ import annotations

import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
OUT = HERE / "libv1c_emul.so"
DEPS = [HERE / "emul.hip", ROOT / "vr180_convert_amd/csrc/v1c_core.hpp", ROOT / "vr180_convert_amd/csrc/radial_fit.hpp",
        ROOT / "include/vr180_remap.h"]


def build(force: bool = False) -> Path:
    if force or not OUT.exists() or OUT.stat().st_mtime < max(d.stat().st_mtime for d in DEPS):
        subprocess.run(
            ["/opt/rocm/bin/hipcc", "--cuda-host-only", "-O2", "-std=c++17", "-shared", "-fPIC", "-fno-fast-math",
             "-o", str(OUT), str(HERE / "emul.hip")],
            check=True, capture_output=True,
        )
    return OUT
