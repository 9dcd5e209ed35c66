#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage).

    python3 tools/resource_usage.py [kernels_tile.hip kernels_mirror.hip ...] [-DV1C_TUNING ...] [--all] [--json out.json]

(default: the three tile translation units kernels_tile.hip, kernels_mirror.hip, kernels_cn.hip)

Prints the kernels that use scratch or spill SGPRs into vector lanes (`--all`: every kernel) and the totals the round's
verdict asks for: number of kernels, kernels with scratch, largest SGPR spill.  tests/test_resource_budget.py runs it on the
product build so that a regression (a kernel falling into scratch) fails the CPU suite.
"""
from __future__ import annotations

import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "vr180_convert_amd" / "csrc"


def makefile_flags() -> list[str]:
    """CXXFLAGS of csrc/Makefile (the one place the product's compiler flags are written down), with ARCH expanded."""
    text = (CSRC / "Makefile").read_text().replace("\\\n", " ")
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", text, re.M).group(1)
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", text, re.M).group(1)
    return flags.replace("$(ARCH)", arch).split()


FLAGS = makefile_flags()
TILE_SRCS = ["kernels_tile.hip", "kernels_mirror.hip", "kernels_cn.hip"]


def demangle(names: list[str]) -> list[str]:
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return p.stdout.split("\n")[: len(names)]
    except Exception:  # noqa: BLE001
        return names


def resource_usage(src: Path, extra: list[str] | None = None) -> list[dict]:
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *(extra or []), "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", "/dev/null"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=src.parent)
    if p.returncode != 0:
        raise RuntimeError(p.stderr[-4000:])
    blocks = re.split(r"remark: [^\n]*Function Name: ", p.stderr)[1:]
    rows = []
    for b in blocks:
        def g(key: str) -> int:
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append({"name": b.split("\n")[0].strip(), "sgprs": g("SGPRs"), "vgprs": g("VGPRs"), "agprs": g("AGPRs"),
                     "scratch": g(r"ScratchSize \[bytes/lane\]"), "sgpr_spill": g("SGPRs Spill"), "vgpr_spill": g("VGPRs Spill"),
                     "occupancy": g(r"Occupancy \[waves/SIMD\]"), "lds": g(r"LDS Size \[bytes/block\]")})
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = re.sub(r"^void v1c::", "", d)
    return rows


def main(argv: list[str]) -> int:
    show_all = "--all" in argv
    out_json = None
    if "--json" in argv:
        out_json = argv[argv.index("--json") + 1]
    args = [a for a in argv if a not in ("--all", "--json", out_json)]
    srcs = [a for a in args if a.endswith(".hip")] or TILE_SRCS
    extra = [a for a in args if a.startswith("-")]
    total = []
    for s in srcs:
        rows = resource_usage(CSRC / s if not Path(s).is_absolute() else Path(s), extra)
        total += rows
        for r in rows:
            if show_all or r["scratch"] > 0 or r["sgpr_spill"] > 0:
                print(f"{r['kernel'][:100]:100s} sgpr {r['sgprs']:3d} vgpr {r['vgprs']:3d} occ {r['occupancy']} scratch {r['scratch']:3d} "
                      f"sgpr_spill {r['sgpr_spill']:2d} vgpr_spill {r['vgpr_spill']:2d} lds {r['lds']}")
    n_scr = sum(r["scratch"] > 0 for r in total)
    n_sp = sum(r["sgpr_spill"] > 0 for r in total)
    print(f"{len(total)} kernels; {n_scr} with scratch; {n_sp} spill SGPRs (largest {max([r['sgpr_spill'] for r in total] + [0])})")
    if out_json:
        Path(out_json).write_text(json.dumps(total, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
