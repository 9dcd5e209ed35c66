"""The C-ABI library loads on a GPU-less box and exports every symbol include/*.h declares; the
argument-validation paths that need no device behave (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "vr180_remap.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v1c_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(product_lib):
    from vr180_convert_amd import _native

    names = declared_symbols()
    assert set(names) == set(_native.SYMBOLS), "keep _native.SYMBOLS in sync with the header"
    for n in names:
        assert getattr(product_lib, n) is not None


def test_abi_version_and_structs(product_lib):
    from vr180_convert_amd import _abi

    assert product_lib.v1c_abi_version() == _abi.ABI_VERSION
    assert C.sizeof(_abi.Op) == 16 + 8 * 16 and C.sizeof(_abi.Chain) == 8 + 16 * C.sizeof(_abi.Op)
    assert C.sizeof(_abi.Unit) == 8 * 4 + 72 + 8


def test_argument_validation_without_device(product_lib):
    from vr180_convert_amd import _abi

    h = C.c_void_p()
    bad = _abi.chain([_abi.op(99)])
    rc = product_lib.v1c_plan_create(C.byref(h), 0, C.byref(bad), 8, 8, 8, 8, 3, 1, 0, None)
    assert rc == _abi.E_INVALID and b"opcode" in product_lib.v1c_last_error()
    ok = _abi.chain([_abi.op(_abi.OP_NORMALIZE, 0, [4, 4, 8]), _abi.op(_abi.OP_DENORMALIZE, 0, [4, 4, 4, 4])])
    rc = product_lib.v1c_plan_create(C.byref(h), 0, C.byref(ok), 8, 40000, 8, 8, 3, 1, 0, None)
    assert rc == _abi.E_INVALID and b"32768" in product_lib.v1c_last_error()
    rc = product_lib.v1c_plan_create(C.byref(h), 0, C.byref(ok), 8, 8, 8, 8, 2, 1, 0, None)
    assert rc == _abi.E_INVALID and b"cn" in product_lib.v1c_last_error()
    rc = product_lib.v1c_plan_create(C.byref(h), 0, C.byref(ok), 8, 8, 8, 8, 3, 9, 0, None)
    assert rc == _abi.E_INVALID
    assert product_lib.v1c_plan_destroy(None) == 0
    assert product_lib.v1c_build_itab(1, None) == _abi.E_INVALID


def test_product_itab_equals_oracle(product_lib, oracle_mod):
    for interp, k in ((2, 4), (4, 8)):
        buf = np.zeros(1024 * k * k, np.int16)
        assert product_lib.v1c_build_itab(interp, buf.ctypes.data) == 0
        assert np.array_equal(buf.reshape(1024, k, k), oracle_mod.build_itab(interp))
