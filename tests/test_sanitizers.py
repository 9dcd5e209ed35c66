"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU code: the oracle
(oracle/vr180_oracle.c, gcc) and the host build of the product's per-pixel code (tests/host_emul,
hipcc --cuda-host-only).  CPU only -- GPU sanitizers are not available on this pool.  The drivers
(tests/sanitize/) sweep every opcode, interpolation, border mode and channel count with coordinates
far outside the source, NaN and infinities on exact-size heap blocks."""
import os
import subprocess
from pathlib import Path

import pytest

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
BUILD = HERE / ".cache" / "sanitize"
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g", "-O1"]


def _run(exe: Path) -> str:
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    report = p.stdout + p.stderr
    assert p.returncode == 0, report[-4000:]
    assert "AddressSanitizer" not in report and "runtime error" not in report and "LeakSanitizer" not in report, report[-4000:]
    return report


def test_oracle_under_asan_ubsan():
    BUILD.mkdir(parents=True, exist_ok=True)
    exe = BUILD / "san_oracle"
    subprocess.run(["gcc", "-std=gnu11", *SAN, "-ffp-contract=off", "-fopenmp", "-o", str(exe), str(HERE / "sanitize" / "san_oracle.c"),
                    str(ROOT / "oracle" / "vr180_oracle.c"), "-lm"], check=True, capture_output=True)
    assert "san_oracle ok" in _run(exe)


def test_host_emulation_of_the_device_code_under_asan_ubsan():
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("no hipcc")
    BUILD.mkdir(parents=True, exist_ok=True)
    exe = BUILD / "san_emul"
    subprocess.run([str(hipcc), "--cuda-host-only", "-std=c++17", "-fno-fast-math", *SAN, "-o", str(exe),
                    str(HERE / "sanitize" / "san_emul.hip")], check=True, capture_output=True)
    assert "san_emul ok" in _run(exe)
