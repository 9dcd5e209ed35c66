#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE rows collected by tools/profile.sh into HBM bytes per launch.

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
both counters are in KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request, i.e. reports half
of the bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for wide stores.
"""
import csv
import json
import sys
from pathlib import Path

out = Path(sys.argv[1])
workload = "C2"
if "--workload" in sys.argv:
    workload = sys.argv[sys.argv.index("--workload") + 1]


def mean_counter(name):
    f = out / f"pmc_{name}.csv"
    if not f.exists():
        return None
    # the dominant kernel = the one with the most rows
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    if not rows:
        return None
    kern = max({r["Kernel_Name"] for r in rows}, key=lambda k: sum(r["Kernel_Name"] == k for r in rows))
    vals = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"] == kern]
    return kern, sum(vals) / len(vals)


fetch, write = mean_counter("FETCH_SIZE"), mean_counter("WRITE_SIZE")
res = {"workload": workload}
if fetch and write:
    res.update(
        kernel=fetch[0],
        fetch_size_kib_raw=round(fetch[1], 1),
        write_size_kib=round(write[1], 1),
        fetch_bytes_corrected=int(fetch[1] * 1024 * 2),
        write_bytes=int(write[1] * 1024),
        hbm_bytes_per_launch=int(fetch[1] * 1024 * 2 + write[1] * 1024),
        note="FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE as is; separate --pmc passes",
    )
print(json.dumps(res, indent=1))
