"""A short run of the GPU-less differential search (tools/fuzz_cpu.py) inside the CPU suite: random chains through the product's plan
derivation, per-pixel code and the host model of the tile kernels' table slices / entry sharing / m-polynomial lanes, against the oracle."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_fuzz_cpu_short_run():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_cpu.py"), "--seconds", "20", "--seed", "101"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " 0 reported" in last and "fused" in last, last
    fused = int(last.split(" cases, ")[1].split(" fused")[0])
    assert fused >= 20, last
