import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _log_launch_kinds(request):
    """V1C_LOG_KINDS=<file>: append, per test, which kernel families its plan runs used (Plan.last_launch) -- the audit behind
    profiles/*/test_kernel_coverage.log: a parity test only counts for a tiled kernel if the case reached it."""
    path = os.environ.get("V1C_LOG_KINDS")
    if not path or request.node.get_closest_marker("gpu") is None:
        yield
        return
    import collections

    from vr180_convert_amd import remapper

    seen: collections.Counter = collections.Counter()
    orig = remapper.Plan.run

    def run(self, *a, **k):
        r = orig(self, *a, **k)
        seen[(self.last_launch(), f"cn{self.cn}")] += 1
        return r

    remapper.Plan.run = run
    try:
        yield
    finally:
        remapper.Plan.run = orig
        with open(path, "a") as f:
            f.write(f"{request.node.nodeid}: " + ", ".join(f"{k[0]}/{k[1]} x{n}" for k, n in sorted(seen.items())) + "\n")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O

    O.build()
    O.set_threads(min(8, os.cpu_count() or 1))
    return O


@pytest.fixture(scope="session")
def product_lib():
    """libvr180remap.so (built by __graft_entry__.build(); hipcc cross-compiles without a GPU)."""
    from vr180_convert_amd import _native

    if not _native.LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_native.LIB_PATH.parent), "-j2"], check=True, capture_output=True)
    return _native.lib()


@pytest.fixture(scope="session")
def emul_lib(product_lib):
    """Host build of the product's __host__ __device__ per-pixel code (tests/host_emul)."""
    import ctypes

    from host_emul.build import build

    return ctypes.CDLL(str(build()))
