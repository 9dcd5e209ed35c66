"""Forty seconds of tools/fuzz.py inside the GPU suite: random chains / sizes / channel counts / interpolations / border modes / views /
per-unit rotations / pairs / mixed source sizes, cv2.remap alone on special-valued maps, get_radius -- product against oracle, byte for
byte (HISTORY.md 2.1; round 5 added the planar / general-mode hot shapes, radius="auto" on the device and v1c_remap_fused by ctypes).  The long runs are the tool's; this keeps a slice of the search in every round's GPU test run."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("seed,extra", [(101, []), (102, ["--lut", "0.6"]), (103, ["--hot", "0.6", "--auto", "0.15", "--fused", "0.15", "--lut", "0.05"])])
def test_a_slice_of_the_differential_fuzz(seed, extra):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz.py"), "--seconds", "20", "--big", "0.05", "--seed", str(seed), *extra],
                       capture_output=True, text=True, timeout=600)
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("fuzz seed")]
    assert last, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and " 0 reported" in last[-1], (r.stdout[-3000:], r.stderr[-1500:])
    assert int(last[-1].split(":")[1].split()[0]) >= 5, last[-1]  # (it did run cases: ~2 per second with the oracle's share)
