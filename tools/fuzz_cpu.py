#!/usr/bin/env python3
"""Differential fuzz WITHOUT a GPU: random chains and output sizes through the product's plan derivation and per-pixel code compiled
for the host (tests/host_emul: radial_fit.hpp's analysis, table fits and proofs, v1c_core.hpp's ray path with the interpreter as its
per-pixel fallback -- what the kernels and plan.hip are made of) against the oracle's float64 chain, bucket by bucket.

    python3 tools/fuzz_cpu.py [--seconds 300] [--seed 1] [--gen2 0.3] [--hot 0.3]

What it covers: which chains the plan accepts (base 0 / 1 / 2, general modes 1 / 2), the S / Cm and G tables with their flagged
intervals, the row / column tables, the fix-up rule (a pixel the ray path declines takes the interpreter).  The tile kernels' table
slices and entry sharing are covered through a host MODEL (emul_lane_model_all: the sharing rule itself is the kernels' function); what
it cannot cover are the kernels themselves -- boxes, staging, the sampler's LDS paths: that is tools/fuzz.py on a GPU box.  Chains come from
tools/fuzz.py's grammar (rand_spec, the hot shapes, --gen2); pixels where the chain amplifies a perturbation of the output position by
1e6 or more are left out and counted (tools/fuzz.py: ill_conditioned).  Exit code 1 if a bucket differed elsewhere."""
from __future__ import annotations

import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "host_emul"))
sys.path.insert(0, str(ROOT / "tools"))

import build as emul_build  # noqa: E402
import fuzz as F  # noqa: E402  (the grammar; importing it needs no GPU)
from oracle import oracle as O  # noqa: E402


def emul_map(E, ch, W, H, mode):
    xm = np.empty((H, W), np.float32)
    ym = np.empty((H, W), np.float32)
    st = (C.c_longlong * 5)()
    rc = E.emul_get_map(C.byref(ch), C.c_void_p(None), W, H, mode, C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), st)
    return rc, xm, ym, list(st)


def buckets(x, y):
    ok = np.isfinite(x) & np.isfinite(y)
    bx = np.where(ok, np.rint(np.where(ok, x, 0).astype(np.float64) * 32), -2.0 ** 40)
    by = np.where(ok, np.rint(np.where(ok, y, 0).astype(np.float64) * 32), -2.0 ** 40)
    return bx, by


def draw(rng, gen2: float, hot: float):
    spec, _ = F.rand_spec(rng)
    r = rng.random()
    if r < gen2:
        enc = ("equirect_enc", bool(rng.random() < 0.8)) if rng.random() < 0.7 else ("fisheye_enc", F.MODELS[int(rng.integers(5))])
        pre = []
        for _ in range(int(rng.choice([1, 1, 2]))):
            k = rng.random()
            pre.append(("zoom", float(rng.uniform(0.5, 2.6))) if k < 0.5 else ("poly", [0.0, 1.0, float(rng.uniform(-0.2, 0.2))]) if k < 0.8
                       else ("inverse", ("zoom", float(rng.uniform(0.5, 2.0)))))
        rots = [("rot", F.rand_rot(rng, rng.random() < 0.5).tolist()) for _ in range(int(rng.choice([1, 1, 2])))]
        post = [("poly", [0.0, 1.0, float(rng.uniform(-0.12, 0.06))])] if rng.random() < 0.3 else []
        dec = ("fisheye_dec", "equidistant") if rng.random() < 0.7 else ("fisheye_dec", F.MODELS[int(rng.integers(5))])
        spec = [enc] + pre + rots + post + [dec]
    elif r < gen2 + hot:
        fam = rng.random()
        enc = ("equirect_enc", fam >= 0.3) if fam < 0.6 else ("fisheye_enc", F.MODELS[int(rng.integers(5))])
        mid = []
        if rng.random() < 0.3:
            mid.append(("poly", [0.0, 1.0, float(rng.uniform(-0.15, 0.08))]))
        if rng.random() < 0.3:
            mid.insert(int(rng.integers(0, len(mid) + 1)), ("rot", F.rand_rot(rng, rng.random() < 0.5).tolist()))
        dec = ("fisheye_dec", "equidistant") if rng.random() < 0.7 else ("fisheye_dec", F.MODELS[int(rng.integers(5))])
        spec = [enc] + mid + [dec]
    wo, ho = int(rng.integers(16, 700)), int(rng.integers(16, 700))
    if rng.random() < 0.5:
        ho = wo
    hs = int(rng.integers(40, 2000))
    ws = hs if rng.random() < 0.5 else max(8, hs + int(rng.integers(-hs // 3, hs // 2)))
    rsel = rng.random()
    radius = min(ws, hs) / 2 if rsel < 0.5 else float(rng.uniform(0.2, 1.6) * min(ws, hs) / 2) if rsel < 0.92 else -float(rng.uniform(5, 100))
    return spec, (wo, ho), (hs, ws), float(radius)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gen2", type=float, default=0.3)
    ap.add_argument("--hot", type=float, default=0.3)
    a = ap.parse_args()
    E = C.CDLL(str(emul_build.build()))
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    n = fused = reported = 0
    left_out = 0
    ties = 0  # differing pixels at a float32 rounding tie
    modes: dict = {}
    lanes = [0.0, 0.0]  # pixels served by a shared entry, lanes whose pixel 1 points outside the slice
    proven = 0  # cases whose plan proves one entry per lane
    mpoly_px = 0.0  # pixel checks of the m-polynomial model
    nofix = 0  # ... no fix-up pass
    units = [0, 0.0]  # unit-rotation cases, claims made in them
    while time.time() - t0 < a.seconds:
        if rng.random() < 0.15:
            # a unit that overrides the rotation of a classic chain (per-frame calibration; v1c_plan_run_auto): the host's closed-form claims
            # (covered / one entry per lane / coordinates bounded: plan.hip decide_launch, restated by emul_unit_rotation_check) against the pixels
            mid = [("poly", [0.0, 1.0, float(rng.uniform(-0.15, 0.08))])] if rng.random() < 0.4 else []
            dec = ("fisheye_dec", "equidistant") if rng.random() < 0.7 else ("fisheye_dec", F.MODELS[int(rng.integers(5))])
            spec = [("equirect_enc", True), ("rot", np.eye(3).tolist())] + mid + [dec]
            size = int(rng.integers(64, 1300))
            wo = size if rng.random() < 0.7 else max(16, size + int(rng.integers(-100, 100)))
            rad = float(rng.uniform(0.3, 0.62) * size)
            R = np.ascontiguousarray(F.rand_rot(rng, rng.random() < 0.4).reshape(9))
            try:
                ch = O.chain_from_spec(spec, radius=rad, size_input=(size, size), size_output=(wo, size))
            except Exception:  # noqa: BLE001
                continue
            ub = (C.c_double * 11)()
            if E.emul_unit_rotation_check(C.byref(ch), C.c_void_p(R.ctypes.data), wo, size, ub) == 0:
                units[0] += 1
                units[1] += ub[1] + ub[2] + ub[3]
                if (ub[1] and ub[4] > 0) or (ub[2] and (ub[5] > 0 or ub[7] > 4e-15)) or (ub[3] and not ub[6] < 2097152.0) or ub[9] > 4e-15 or ub[10] > 0:
                    reported += 1
                    print(f"[unit rotation] claims {list(ub)[:4]} contradicted by the pixels {list(ub)[4:]}: spec={spec!r} out=({wo},{size}) radius={rad!r} R={R.tolist()!r}",
                          flush=True)
            continue
        spec, out, inp, radius = draw(rng, a.gen2, a.hot)
        n += 1
        try:
            ch = O.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
        except Exception:  # noqa: BLE001  (a chain the oracle's lowering rejects: not a case)
            continue
        info = (C.c_longlong * 12)()
        E.emul_plan_info(C.byref(ch), out[0], out[1], info)
        if not (info[0] and info[1]):
            continue
        rc, xm, ym, st = emul_map(E, ch, out[0], out[1], 1)
        if rc != 0:
            continue
        fused += 1
        key = f"base{info[2]}/gen{info[3]}"
        modes[key] = modes.get(key, 0) + 1
        ox, oy = O.get_map(spec, radius=radius, size_input=inp, size_output=out)
        bx, by = buckets(xm, ym)
        rx, ry = buckets(ox, oy)
        d = (bx != rx) | (by != ry)
        if d.any():
            sing = F.ill_conditioned(spec, radius, inp, out)
            left_out += int((d & sing).sum())
            d &= ~sing
        if d.any():
            # a coordinate whose exact value is a tie of the float64 -> float32 rounding (to within 1e-14 relative) may land either side:
            # both the oracle's libm and the product's tables are a few ulps of float64 off the truth (seed 10 case 49)
            fx, fy = O.get_map(ch, radius=radius, size_input=inp, size_output=out, f64=True)
            tie = np.zeros_like(d)
            for v in (fx, fy):
                f = v.astype(np.float32)
                up, dn = np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))
                with np.errstate(invalid="ignore", over="ignore"):
                    m1, m2 = (f.astype(np.float64) + up.astype(np.float64)) / 2, (f.astype(np.float64) + dn.astype(np.float64)) / 2
                    tie |= np.minimum(np.abs(v - m1), np.abs(v - m2)) <= 1e-14 * np.abs(v)
            ties += int((d & tie).sum())
            d &= ~tie
        # the plan says no fix-up pass is needed (the kernels then carry no flag words): no pixel may have declined the ray path
        if info[11] and st[1] > 0:
            reported += 1
            print(f"[case {n}] the plan proves 'no fix-up pass', {st[1]} pixels took the interpreter: spec={spec!r} out={out} src={inp} radius={radius!r}", flush=True)
        nofix += bool(info[11])
        # the tile kernels' table slices and entry sharing, modelled on the host over every tile (tests/host_emul: emul_lane_model_all)
        lm = (C.c_double * 9)()
        if E.emul_lane_model_all(C.byref(ch), out[0], out[1], 0, lm) == 0:
            lanes[0] += lm[2]
            lanes[1] += lm[3]
            if lm[0] > 4e-15:
                reported += 1
                print(f"[case {n}] lane model: a shared entry is {lm[0]:.3e} off its pixel's own: spec={spec!r} out={out} src={inp} radius={radius!r}", flush=True)
            if info[10] and lm[5] > 0:  # the plan proves one entry per lane (OWN = 0 kernels): no pixel may need its own
                reported += 1
                print(f"[case {n}] lane model: the plan says one entry per lane, {lm[5]:.0f} pixels need their own: spec={spec!r} out={out} src={inp} radius={radius!r}", flush=True)
            if lm[7] > 4e-15 or lm[8] > 0:  # the m-polynomial twin: either candidate entry of an fp32 index, inside the slice and the level
                reported += 1
                print(f"[case {n}] lane model, m-polynomials: error {lm[7]:.3e}, {lm[8]:.0f} candidate entries not covered: spec={spec!r} out={out} src={inp} radius={radius!r}", flush=True)
            mpoly_px += lm[6]
            proven += bool(info[10])
        if d.any():
            reported += 1
            j, i = np.argwhere(d)[0]
            print(f"[case {n}] {int(d.sum())} buckets differ, first at ({j}, {i}): emul ({xm[j, i]!r}, {ym[j, i]!r}) oracle ({ox[j, i]!r}, {oy[j, i]!r}) "
                  f"spec={spec!r} out={out} src={inp} radius={radius!r} fixup_pixels={st[1]}", flush=True)
    print(f"fuzz_cpu seed {a.seed}: {n} cases, {fused} fused ({modes}), {reported} reported; {left_out} differing ill-conditioned pixels and {ties} at float32 ties left out; lane model: {lanes[0]:.0f} pixels on a shared entry, "
          f"{lanes[1]:.0f} lanes with pixel 1 outside the slice, {proven} plans proving one entry per lane ({mpoly_px:.0f} m-polynomial pixel checks), {nofix} no fix-up pass; {units[0]} unit-rotation cases with {units[1]:.0f} claims; "
          f"{time.time() - t0:.0f} s")
    return 1 if reported else 0


if __name__ == "__main__":
    sys.exit(main())
