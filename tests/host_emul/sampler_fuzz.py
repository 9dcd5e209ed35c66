"""Random small / degenerate sources and maps through the product's sampler compiled for the HOST with AddressSanitizer, against the
oracle.  Run as a script in a process that has the ASan runtime preloaded (tests/test_sampler_asan.py does that): an access outside
the source or destination arrays aborts the process with ASan's report.  Found in round 4 by the GPU fuzz (a memory fault), kept out
by this: a source ONE pixel wide made the bilinear fast path's unsigned bound wrap (v1c_core.hpp, sample_linear_t).

    python tests/host_emul/sampler_fuzz.py <libv1c_emul_asan.so> <libvr180remap.so> [seconds]
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402


def main() -> int:
    E = C.CDLL(sys.argv[1])
    P = C.CDLL(sys.argv[2])
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
    tabs = {}
    for interp, k in ((2, 4), (4, 8)):
        tabs[interp] = np.zeros(1024 * k * k, np.int16)
        assert P.v1c_build_itab(interp, C.c_void_p(tabs[interp].ctypes.data)) == 0
    rng = np.random.default_rng(12345)
    t0 = time.time()
    n = 0
    special = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 2.0 ** 26, 32767.0, 32767.5, 32768.0, -32768.0, -32768.5, -0.5, -1.0,
                        0.0, 1e-30], np.float32)
    while time.time() - t0 < budget:
        cn = int(rng.choice([1, 3, 4]))
        hs = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 9, int(rng.integers(1, 40))]))
        ws = int(rng.choice([1, 1, 2, 3, 4, 7, 8, 9, int(rng.integers(1, 40))]))
        ho, wo = int(rng.integers(1, 24)), int(rng.integers(1, 40))
        # exactly sized allocations (ASan puts red zones around them); a pitched source now and then
        pitch_pad = int(rng.choice([0, 0, 1, 5]))
        srcw = rng.integers(0, 256, (hs, ws * cn + pitch_pad), dtype=np.uint8)
        src = np.lib.stride_tricks.as_strided(srcw, (hs, ws, cn), (srcw.strides[0], cn, 1))
        xm = rng.uniform(-12, ws + 12, (ho, wo)).astype(np.float32)
        ym = rng.uniform(-12, hs + 12, (ho, wo)).astype(np.float32)
        for m in (xm, ym):
            k = int(rng.integers(0, max(2, m.size // 8)))
            m.reshape(-1)[rng.integers(0, m.size, k)] = rng.choice(special, k)
        bv = tuple(int(v) for v in rng.integers(0, 256, 4))
        cv = O.border_scalar(bv)
        for interp in (0, 1, 2, 3, 4):
            for border in range(6):
                fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
                ref = fill.copy()
                O.remap(np.ascontiguousarray(src), xm, ym, interp, border, bv, dst=ref)
                out = fill.copy()
                it = tabs.get(interp)
                rc = E.emul_remap(C.c_void_p(srcw.ctypes.data), hs, ws, C.c_int64(srcw.strides[0]), cn, C.c_void_p(out.ctypes.data), ho, wo,
                                  C.c_int64(out.strides[0]), C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), interp, border,
                                  C.c_void_p(cv.ctypes.data), C.c_void_p(None if it is None else it.ctypes.data))
                if rc != 0 or not np.array_equal(ref, out):
                    print("MISMATCH", dict(cn=cn, hs=hs, ws=ws, ho=ho, wo=wo, interp=interp, border=border, pitch_pad=pitch_pad, rc=rc,
                                           bad=int((ref != out).sum())))
                    return 1
                n += 1
    # the coordinate code under the sanitizer too: random chains through the interpreter and, where the plan takes it, the ray path
    # with its fitted tables (radial_fit.hpp: table slices, m-polynomial tables, row / column tables) -- reads outside any of them abort
    models = ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]
    t1 = time.time()
    m = 0
    while time.time() - t1 < budget / 2:
        spec = [("equirect_enc", bool(rng.random() < 0.8))] if rng.random() < 0.7 else [("fisheye_enc", models[int(rng.integers(5))])]
        for _ in range(int(rng.integers(0, 3))):
            k = rng.random()
            if k < 0.4:
                a = rng.normal(0, 0.5, 3)
                th = float(np.linalg.norm(a)) or 1.0
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
                spec.append(("rot", (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K).tolist()))
            elif k < 0.7:
                spec.append(("poly", [0.0, 1.0] + [float(v) for v in rng.normal(0, 0.08, int(rng.integers(0, 3)))]))
            else:
                spec.append(("zoom", float(rng.uniform(0.4, 2.5))))
        spec.append(("fisheye_dec", models[int(rng.integers(5))]) if rng.random() < 0.85 else ("rectilinear_dec", 12.0, 17.0))
        wo, ho = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        hs, ws = int(rng.integers(2, 3000)), int(rng.integers(3, 3000))
        ch = O.chain_from_spec(spec, radius=float(rng.uniform(0.2, 1.5) * min(hs, ws) / 2), size_input=(hs, ws), size_output=(wo, ho))
        xm, ym = np.empty((ho, wo), np.float32), np.empty((ho, wo), np.float32)
        st = (C.c_longlong * 5)()
        for mode in (0, 1):
            E.emul_get_map(C.byref(ch), C.c_void_p(None), wo, ho, mode, C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), st)
        m += 1
    print(f"sampler fuzz: {n} remaps, all equal to the oracle; {m} random chains through the coordinate code; no sanitizer report")
    return 0


if __name__ == "__main__":
    sys.exit(main())
