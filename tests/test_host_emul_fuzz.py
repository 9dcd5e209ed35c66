"""Random chains (the grammar of tools/fuzz.py: every lowered stage, stacked, rotated, zoomed) through the product's coordinate code
compiled for the host -- the fp64 interpreter and, where the plan takes it, the fused ray path with its fitted radial table -- against
the oracle's map: the same 1/32-pixel bucket for every pixel that is not ill-conditioned, and a plan that says "no fix-up pass needed"
must not have needed one.  The GPU fuzz checks the same through pixels; this one needs no GPU and runs in every CPU test run."""
import ctypes as C
import importlib.util
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]


def _bucket(m):
    with np.errstate(invalid="ignore", over="ignore"):
        v = m.astype(np.float32) * np.float32(32)
        ok = np.abs(v) < 2147483648.0
        out = np.full(m.shape, -2 ** 31, np.int64)  # cvRound's answer out of range / for NaN
        out[ok] = np.rint(v[ok]).astype(np.int64)
    return out


def test_random_chains_interpreter_and_ray_path_land_in_the_oracles_buckets(emul_lib, oracle_mod, product_lib):
    O = oracle_mod
    sp = importlib.util.spec_from_file_location("v1c_fuzz_tool", ROOT / "tools" / "fuzz.py")
    F = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(F)
    t0 = time.time()
    n = n_ray = case = 0
    while time.time() - t0 < 20 or n_ray < 30:
        rng = np.random.default_rng([20241004, case])
        case += 1
        spec, rot_at = F.rand_spec(rng)
        rot = None
        if rot_at is not None and rng.random() < 0.5:  # a unit that overrides the chain's rotate stage (per-frame calibration)
            rot = np.ascontiguousarray(F.rand_rot(rng, rng.random() < 0.3), np.float64)
        wo, ho = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        hs, ws = int(rng.integers(2, 800)), int(rng.integers(3, 800))
        radius = float(rng.uniform(0.2, 1.5) * min(hs, ws) / 2)
        ch = O.chain_from_spec(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
        if rot is not None:  # the oracle evaluates the chain with the matrix in place
            spec = list(spec)
            spec[rot_at] = ("rot", rot.tolist())
        xo, yo = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
        ill = F.ill_conditioned(spec, radius, (hs, ws), (wo, ho))
        bo = (_bucket(xo), _bucket(yo))
        for mode in (0, 1):
            xm, ym = np.empty((ho, wo), np.float32), np.empty((ho, wo), np.float32)
            st = (C.c_longlong * 5)()
            rc = emul_lib.emul_get_map(C.byref(ch), C.c_void_p(None if rot is None else rot.ctypes.data), wo, ho, mode, C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), st)
            if rc:
                assert mode == 1  # (not of the ray shape)
                continue
            d = ((_bucket(xm) != bo[0]) | (_bucket(ym) != bo[1])) & ~ill
            assert not d.any(), (mode, spec, (wo, ho), (hs, ws), radius, int(d.sum()), np.argwhere(d)[0].tolist())
            if mode == 1:
                n_ray += 1
                assert not (st[4] and st[1]), ("the plan claimed that no fix-up pass is needed", spec, list(st))
        n += 1
    assert n >= 50 and n_ray >= 30, (n, n_ray)
