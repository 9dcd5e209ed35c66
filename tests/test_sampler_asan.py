"""The product's per-pixel code (v1c_core.hpp, radial_fit.hpp), compiled for the host with AddressSanitizer: the sampler on degenerate
and random small sources (1 x 1, one pixel wide, one row high, pitched) and maps full of special values, against the oracle, then
random chains through the interpreter and the ray path with its fitted tables -- in a subprocess with the sanitizer runtime preloaded.  GPU AddressSanitizer is not available on the pool; the sampler is __host__ __device__ code, so the host
build finds what a GPU run would answer with a memory fault (round 4: a source one pixel wide, found by tools/fuzz.py)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HERE = ROOT / "tests" / "host_emul"
RT = sorted(Path("/opt/rocm/lib/llvm/lib/clang").glob("*/lib/linux/libclang_rt.asan-x86_64.so"))


def test_sampler_under_address_sanitizer(product_lib, oracle_mod):
    if not RT:
        pytest.skip("clang's ASan runtime is not installed")
    out = HERE / "libv1c_emul_asan.so"
    deps = [HERE / "emul.hip", ROOT / "vr180_convert_amd/csrc/v1c_core.hpp", ROOT / "vr180_convert_amd/csrc/radial_fit.hpp"]
    if not out.exists() or out.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-O1", "-g", "-std=c++17", "-shared", "-fPIC", "-fno-fast-math",
                        "-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan", "-o", str(out), str(HERE / "emul.hip")],
                       check=True, capture_output=True)
    from vr180_convert_amd import _native

    env = dict(os.environ, LD_PRELOAD=str(RT[-1]), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=97")
    r = subprocess.run([sys.executable, str(HERE / "sampler_fuzz.py"), str(out), str(_native.LIB_PATH), "15"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout
