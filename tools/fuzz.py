#!/usr/bin/env python3
"""Differential fuzz on the GPU: random chains / sizes / flags through the product (C ABI, HIP kernels) against the CPU oracle,
byte for byte, for a bounded time.  Wider than the seeded sweeps of tests/test_gpu_parity.py (those are the regression net; this
is the search): chains drawn from a grammar over every lowered stage, sizes from 1 px to a few thousand (several tile rows, XCD
strips, rest lists, the unit ring), cn 1 / 3 / 4, every interpolation and border mode incl. BORDER_TRANSPARENT, pitched source
and destination views (dword-aligned or not), per-unit rotations, apply_lr pairs; a share of the cases (--lut) runs cv2.remap alone
(v1c_remap_lut) on random float32 maps sprinkled with NaN, infinities, 2^15 / 2^26 / 2^31-scale values and ties of the 1/32 grid.

    python3 tools/fuzz.py [--seconds 300] [--seed 1] [--big 0.15] [--lut 0.15] [--hot 0.3] [--gen2 0] [--api 0.1] [--auto 0.06] [--fused 0.06] [--log gpurun_out/fuzz.log]

Round 5 added to the grammar: hot shapes of the chains that left the interpreter (planar fisheye -> fisheye, is_latitude_y=False, a
rotation behind radial stages), outputs of 64 ... 416 px, launches recorded into a graph and replayed, radius='auto' with the radius on
the device (--auto) and v1c_remap_fused through raw ctypes (--fused); --gen2: chains forced into general mode 2 (radial stages / zooms in front
of a rotation, lanes split between the tile kernel and the fix-up pass: what case 1974 of seed 34 was).  tools/fuzz_cpu.py is the GPU-less half.

Ill-conditioned pixels are left out and counted: where the chain amplifies a perturbation of the output position by 1e6 or more
(measured on the oracle's fp64 map, `ill_conditioned`), the last bits of every intermediate -- they differ between glibc and the
device's libm, and with fused multiply-adds -- decide the pixel.  Two kinds turned up: poles of a rectilinear projection (tan(theta)
at or beside 90 degrees: a coordinate of 1e6 ... 1e19 px, the other one the ratio of two rounding residues of pi / 2; seed 1, many
cases; seed 2 case 1423) and stacked polynomial stages that take the angle to 1e12 rad before a decoder takes its tangent (seed 5
case 1534: 5.6 % of the pixels).  The product's per-pixel code compiled for the HOST equals the oracle on them up to a handful of
pixels (tests/test_host_emul.py's emulation): it is the arithmetic environment, not the algorithm.  Under the border modes that read
source pixels far outside (REPLICATE, REFLECT, WRAP, REFLECT_101) pixels with a map coordinate of magnitude >= 2^20 are left out as
well (cv2 itself saturates integer coordinates at 2^15).  A unit that still differs is checked for float32 rounding ties (float32_ties:
measure zero, counted apart).

Every mismatch is printed as a self-contained case description (seed + case number reproduce it: `--seed S --only N`); exit code 1
if there was one.  The oracle is test infrastructure: this tool is not part of the product.
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import chainspecs as CS  # noqa: E402
import vr180_convert_amd as V  # noqa: E402
from oracle import oracle as O  # noqa: E402

MODELS = ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]


def rand_rot(rng, big: bool):
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector

    v = rng.normal(0, 1.0 if big else 0.03, 3)
    return np.asarray(as_rotation_matrix(from_rotation_vector(v)), float)


def rand_spec(rng):
    """(spec, index of the chain's only rotate stage or None)"""
    r = rng.random()
    enc = ("equirect_enc", bool(rng.random() < 0.85)) if r < 0.7 else ("fisheye_enc", MODELS[int(rng.integers(5))])
    mid = []
    n_rot = 0
    for _ in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
        k = rng.random()
        if k < 0.35:
            mid.append(("rot", rand_rot(rng, rng.random() < 0.4).tolist()))
            n_rot += 1
        elif k < 0.45:
            q = rng.normal(0, 1, 4)
            q[0] = abs(q[0]) + 0.5  # non-unit on purpose (cli.py:308-319 builds such quaternions)
            mid.append(("rot_quat", tuple(float(x) for x in q)))
            n_rot += 1
        elif k < 0.7:
            n = int(rng.integers(2, 6))
            co = [0.0, 1.0] + [float(x) for x in rng.normal(0, 0.08, n - 2)]
            if rng.random() < 0.3:
                co[0] = float(rng.normal(0, 0.02))
            mid.append(("poly", co))
        elif k < 0.9:
            mid.append(("zoom", float(rng.uniform(0.4, 2.5))))
        else:
            mid.append(("inverse", ("zoom", float(rng.uniform(0.5, 2.0)))))
    d = rng.random()
    if d < 0.75:
        dec = ("fisheye_dec", "equidistant")
    elif d < 0.93:
        dec = ("fisheye_dec", MODELS[int(rng.integers(5))])
    else:
        dec = ("rectilinear_dec", float(rng.uniform(8, 30)), float(rng.uniform(10, 40)))
    spec = [enc] + mid + [dec]
    rot_at = None
    if n_rot == 1:
        rot_at = [i for i, it in enumerate(spec) if it[0] in ("rot", "rot_quat")][0]
    return spec, rot_at


def rand_size(rng, big: float, lo: int = 1):
    r = rng.random()
    if r < big:
        return int(rng.integers(1200, 2700))
    if r < big + 0.35:
        return int(rng.integers(300, 1200))
    return int(rng.integers(lo, 300))


def make_view(rng, arr: np.ndarray, dev, allow_unaligned: bool):
    """the array as a device tensor: contiguous, or a column slice of a wider buffer (pitched; offset dword-aligned or not)"""
    h, w, cn = arr.shape
    if rng.random() < 0.5:
        return torch.from_numpy(arr).to(dev)
    pad_l = int(rng.integers(0, 9)) if allow_unaligned and rng.random() < 0.3 else 4 * int(rng.integers(0, 5))
    pad_r = int(rng.integers(0, 9)) if allow_unaligned and rng.random() < 0.3 else 4 * int(rng.integers(0, 5))
    wide = rng.integers(0, 256, (h, pad_l + w + pad_r, cn), dtype=np.uint8)
    wide[:, pad_l:pad_l + w] = arr
    return torch.from_numpy(wide).to(dev)[:, pad_l:pad_l + w]


DUMP = [False]
HOT = [0.3]  # share of the chain cases drawn from the shapes the tuned kernels serve (--hot)
GEN2 = [0.0]  # share of the chain cases forced into general mode 2 (--gen2)
KINDS: dict = {}  # kernel family -> launch groups it served (remapper.last_launch_kinds): which kernels the run reached
SINGULAR = [0]  # differing pixels among the ill-conditioned ones that are left out (module docstring)
TIES = [0]  # ... among those at a float32 rounding tie (float32_ties)


def dump_diff(k, got, want, maps, pmaps=None, fill=None) -> None:
    """where unit k differs: counts, bounding box, the first pixels with their map coordinates (the oracle's and the product's)"""
    d = np.argwhere((got != want).any(axis=2))
    print(f"  unit {k}: {len(d)} pixels differ, rows {d[:, 0].min()}..{d[:, 0].max()}, cols {d[:, 1].min()}..{d[:, 1].max()}; "
          f"rows mod 16 {sorted(set((d[:, 0] % 16).tolist()))[:16]}, cols//4 mod 16 {sorted(set(((d[:, 1] // 4) % 16).tolist()))[:16]}")
    for (j, i) in d[:12]:
        pm = "" if pmaps is None else f" product map=({pmaps[0][j, i]!r}, {pmaps[1][j, i]!r})"
        fl = "" if fill is None else f" prefill={fill[j, i].tolist()}"
        print(f"    (row {j}, col {i}) map=({maps[0][j, i]!r}, {maps[1][j, i]!r}){pm} got={got[j, i].tolist()} want={want[j, i].tolist()}{fl}")


def ill_conditioned(spec, radius, size_in, size_out) -> np.ndarray:
    """Pixels where the chain amplifies a perturbation of the output position by 1e6 or more (the fp64 map at positions shifted by
    1e-7 px against the map itself): the last bits of every intermediate -- which differ between libms and with fused multiply-adds
    -- decide the 1/32-pixel bucket there.  Poles of a projection, and stacked polynomial stages that take an angle to 1e12 rad."""
    W, H = size_out
    ch = O.chain_from_spec(spec, radius=radius, size_input=size_in, size_output=size_out)
    x0, y0 = O.get_map(ch, radius=radius, size_input=size_in, size_output=size_out, f64=True)
    ch.ops[0].p[0] -= 1e-7  # Normalize's centre: the same as every pixel 1e-7 further right / down
    ch.ops[0].p[1] -= 1e-7
    x1, y1 = O.get_map(ch, radius=radius, size_input=size_in, size_output=size_out, f64=True)
    with np.errstate(invalid="ignore", over="ignore"):
        amp = np.maximum(np.abs(x1 - x0), np.abs(y1 - y0)) / 1e-7
    return ~(amp < 1e6)



def float32_ties(spec, radius, size_in, size_out) -> np.ndarray:
    """Pixels whose float64 coordinate is a tie of the float64 -> float32 rounding to within 1e-14 (relative): the oracle's libm and the
    product's tables are each a few float64 ulps off the exact value, so the cast may go either way and the 1/32-pixel bucket with it.
    Measure zero; tools/fuzz_cpu.py met one such pixel (and its three mirror images) in 30 000 chains.  Only evaluated for a unit that
    still differs behind the ill-conditioned mask."""
    ch = O.chain_from_spec(spec, radius=radius, size_input=size_in, size_output=size_out)
    fx, fy = O.get_map(ch, radius=radius, size_input=size_in, size_output=size_out, f64=True)
    tie = np.zeros(fx.shape, bool)
    for v in (fx, fy):
        f = v.astype(np.float32)
        up, dn = np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))
        with np.errstate(invalid="ignore", over="ignore"):
            m1, m2 = (f.astype(np.float64) + up.astype(np.float64)) / 2, (f.astype(np.float64) + dn.astype(np.float64)) / 2
            tie |= np.minimum(np.abs(v - m1), np.abs(v - m2)) <= 1e-14 * np.abs(v)
    return tie

def one_case(rng, dev, big: float) -> tuple[str, int]:
    """runs one random case; returns (description, number of differing bytes)"""
    spec, rot_at = rand_spec(rng)
    cn = int(rng.choice([3, 3, 3, 1, 4]))
    interp = int(rng.choice([1, 1, 1, 0, 2, 4, 4]))
    border = int(rng.choice([0, 0, 0, 0, 1, 2, 3, 4, 5]))
    bval = tuple(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 5)))) if rng.random() < 0.7 else int(rng.integers(0, 256))
    pair = rng.random() < 0.3
    wo, ho = rand_size(rng, big), rand_size(rng, big)
    if rng.random() < 0.3:
        ho = wo
    ws, hs = rand_size(rng, big, 1), rand_size(rng, big, 1)
    if rng.random() < 0.4:
        ws = hs = max(ws, 1)
    if interp in (2, 4) and max(wo * ho, ws * hs) > 1500 * 1500:  # keep the oracle's K x K loops in seconds
        wo, ho = min(wo, 1400), min(ho, 1400)
    rsel = rng.random()
    radius = min(ws, hs) / 2 if rsel < 0.5 else float(rng.uniform(0.2, 1.6) * min(ws, hs) / 2) if rsel < 0.92 else -float(rng.uniform(5, 100))
    radius = float(max(radius, 1.0)) if radius > 0 else radius
    if pair:
        n = 2
    else:
        n = int(rng.choice([1, 1, 2, 3, 5, 8, 17, 33]))
        if wo * ho * n > 6e6:
            n = max(1, int(6e6 // (wo * ho)))
    use_rot = (rot_at is not None) and (not pair) and rng.random() < 0.5
    unaligned_views = True
    if rng.random() < HOT[0]:
        # the shapes the tuned kernels are selected for (the free grammar above reaches them once in a hundred cases): an unrotated
        # equirectangular -> equidistant chain (now and then with one polynomial, a zoom, or a rotate stage whose matrix the units
        # override), rays that stay in the front hemisphere (output about square, rows pairing up about the equator), outputs of 416
        # px and more (one table entry per lane), mostly bilinear + BORDER_CONSTANT, pairs / single images / batches / per-unit rotations
        mid = []
        rot_units = rng.random() < 0.3
        if rot_units:
            mid.append(("rot", np.eye(3).tolist()))
        pk = rng.random()
        if pk < 0.25:
            mid.append(("poly", [0.0, 1.0, float(rng.uniform(-0.15, 0.08))]))
        elif pk < 0.35:
            mid.append(("zoom", float(rng.uniform(0.8, 1.3))))
        spec = [("equirect_enc", True)] + mid + [("fisheye_dec", "equidistant")]
        rot_at = 1 if rot_units else None
        if (not rot_units) and rng.random() < 0.4:
            # round 5: the chains that left the interpreter -- planar (fisheye -> fisheye: the reference's own test chains,
            # tests/test_remapper.py:42-91), is_latitude_y=False, a rotation behind radial stages (baked into the plan)
            fam = rng.random()
            enc = ("equirect_enc", False) if fam < 0.25 else ("fisheye_enc", MODELS[int(rng.integers(5))])
            mid2 = list(mid)
            if rng.random() < 0.35:
                mid2.insert(int(rng.integers(0, len(mid2) + 1)), ("rot", rand_rot(rng, rng.random() < 0.5).tolist()))
            if fam >= 0.25 and rng.random() < 0.15:
                enc = ("equirect_enc", True)  # (equirect, radial stage, rotation: the S / Cm tables behind an EquirectangularEncoder)
                mid2 = [("poly", [0.0, 1.0, float(rng.uniform(-0.1, 0.05))]), ("rot", rand_rot(rng, False).tolist())]
            dec = ("fisheye_dec", "equidistant") if rng.random() < 0.7 else ("fisheye_dec", MODELS[int(rng.integers(5))])
            spec = [enc] + mid2 + [dec]
        cn = int(rng.choice([3, 3, 3, 3, 1, 4]))
        interp = int(rng.choice([1, 1, 1, 1, 1, 0, 2, 4]))
        border = int(rng.choice([0, 0, 0, 0, 0, 1, 4, 5]))
        ho = int(rng.integers(13, 66)) * 32 if rng.random() < 0.85 else int(rng.integers(416, 2100))
        if rng.random() < 0.15:
            ho = int(rng.integers(64, 417))  # small outputs (the reference's tests: 256 x 256): per-pixel table entries, tiny grids
        wo = max(1, ho + int(rng.choice([0, 0, 0, -64, -4, 4, 60, 64])) + (0 if rng.random() < 0.7 else int(rng.integers(-100, 30))))
        if interp in (2, 4):
            wo, ho = min(wo, 1100), min(ho, 1088)
        hs = int(rng.integers(60, 2600))
        ws = hs + (0 if rng.random() < 0.5 else int(rng.integers(-hs // 3, hs // 2)))
        if rng.random() < 0.7:
            ws = (ws + 3) & ~3  # (a width whose rows stay dword-aligned: the LDS-DMA kernels)
        radius = float(rng.uniform(0.3, 0.62) * min(ws, hs))
        shape = rng.random()
        pair = (not rot_units) and shape < 0.4
        n = 2 if pair else (1 if shape < 0.55 else int(rng.choice([2, 3, 4, 5, 8, 16, 17, 40])))
        if not pair and wo * ho * n > 2.5e7:
            n = max(1, int(2.5e7 // (wo * ho)))
        use_rot = rot_units
        unaligned_views = rng.random() < 0.15
    if GEN2[0] > 0 and rng.random() < GEN2[0]:  # (no draw when the option is off: earlier seeds replay as they ran)
        # general mode 2 on purpose (--gen2): radial stages / zooms IN FRONT of a rotation, so that the point enters 3-D through the S / Cm
        # tables -- with outputs far from square and zooms that push the base variable out of those tables for part of the image (lanes
        # whose pixels are split between the tile kernel and the fix-up pass: case 1974 of seed 34 was such a lane)
        enc = ("equirect_enc", bool(rng.random() < 0.8)) if rng.random() < 0.7 else ("fisheye_enc", MODELS[int(rng.integers(5))])
        pre = []
        for _ in range(int(rng.choice([1, 1, 2]))):
            k = rng.random()
            if k < 0.5:
                pre.append(("zoom", float(rng.uniform(0.5, 2.6))))
            elif k < 0.8:
                pre.append(("poly", [0.0, 1.0, float(rng.uniform(-0.2, 0.2))]))
            else:
                pre.append(("inverse", ("zoom", float(rng.uniform(0.5, 2.0)))))
        rots = [("rot", rand_rot(rng, rng.random() < 0.5).tolist()) for _ in range(int(rng.choice([1, 1, 2])))]
        post = [("poly", [0.0, 1.0, float(rng.uniform(-0.12, 0.06))])] if rng.random() < 0.3 else []
        dec = ("fisheye_dec", "equidistant") if rng.random() < 0.7 else ("fisheye_dec", MODELS[int(rng.integers(5))])
        spec = [enc] + pre + rots + post + [dec]
        rot_at, use_rot = None, False
        cn = int(rng.choice([3, 3, 3, 1, 4]))
        interp = int(rng.choice([1, 1, 1, 0, 2, 4]))
        border = int(rng.choice([0, 0, 0, 1, 4, 5]))
        wo, ho = int(rng.integers(64, 1300)), int(rng.integers(64, 1500))
        if interp in (2, 4):
            wo, ho = min(wo, 1000), min(ho, 1000)
        hs = int(rng.integers(60, 1200))
        ws = hs + (0 if rng.random() < 0.5 else int(rng.integers(-hs // 3, hs // 2)))
        radius = float(rng.uniform(0.3, 0.7) * min(ws, hs))
        pair = rng.random() < 0.4
        n = 2 if pair else int(rng.choice([1, 1, 2, 3]))
        unaligned_views = rng.random() < 0.15
    rots = [rand_rot(rng, False) for _ in range(n)] if use_rot else None
    # units of different source sizes behind one transformer: the map is for images[0] (remapper.py:385), every image is
    # sampled within its own bounds; a pair of per-eye transformers (remapper.py:460-473): every eye its own chain and geometry
    mixed = (not pair) and (not use_rot) and n > 1 and rng.random() < 0.2
    tuple_t = pair and rng.random() < 0.25
    sizes = [(hs, ws)] * n
    if mixed or (tuple_t and rng.random() < 0.5):
        sizes = [(hs, ws)] + [(max(1, hs + int(rng.integers(-40, 41))), max(1, ws + int(rng.integers(-40, 41)))) for _ in range(n - 1)]
    spec2 = rand_spec(rng)[0] if tuple_t else None
    imgs = [rng.integers(0, 256, (h_, w_, cn), dtype=np.uint8) for (h_, w_) in sizes]
    if rng.random() < 0.3:  # a fisheye disc with a black surround, like the real inputs
        for im in imgs:
            yy, xx = np.mgrid[:im.shape[0], :im.shape[1]]
            im[((xx - im.shape[1] // 2) ** 2 + (yy - im.shape[0] // 2) ** 2) > (min(im.shape[:2]) / 2) ** 2] = 0
    fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
    desc = (f"spec={spec!r} cn={cn} interp={interp} border={border} bval={bval!r} out=({wo},{ho}) src=({ws},{hs}) radius={radius!r} n={n} pair={pair} "
            f"rots={use_rot}" + (f" sizes={sizes!r}" if sizes[1:] != sizes[:-1] else "") + (f" right_eye_spec={spec2!r}" if tuple_t else ""))
    t = CS.to_product(spec)
    srcs = [make_view(rng, im, dev, allow_unaligned=unaligned_views) for im in imgs]
    if pair:
        sbs = torch.from_numpy(np.concatenate([fill, fill], axis=1)).to(dev)
        tt = (t, CS.to_product(spec2)) if tuple_t else t
        V.apply_lr_tensors(tt, srcs[0], srcs[1], out=sbs, size_output=(wo, ho), interpolation=interp, boarder_mode=border, boarder_value=bval,
                           radius=radius)
        got = [sbs[:, :wo].cpu().numpy(), sbs[:, wo:].cpu().numpy()]
    else:
        dsts = [make_view(rng, fill.copy(), dev, allow_unaligned=unaligned_views) for _ in range(n)]
        kw = {}
        if use_rot:
            kw["rotations"] = rots
        V.remap_tensors(t, srcs, dsts, radius=radius, interpolation=interp, boarder_mode=border, boarder_value=bval, **kw)
        from vr180_convert_amd import remapper as _rm

        if rng.random() < 0.12 and n <= 16 and len(_rm.last_launch_kinds()) == 1:
            # the same launch recorded into a graph and replayed on restored destinations (plan_run is launch-only; a recorded launch
            # with a fix-up pass neither waits for nor records the plan's flag event)
            torch.cuda.synchronize()
            for d in dsts:
                d.copy_(torch.from_numpy(fill).to(dev))
            st = torch.cuda.Stream(device=dev)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                V.remap_tensors(t, srcs, dsts, radius=radius, interpolation=interp, boarder_mode=border, boarder_value=bval, **kw)
            for d in dsts:
                d.copy_(torch.from_numpy(fill).to(dev))
            torch.cuda.synchronize()
            gr.replay()
            torch.cuda.synchronize()
            desc += " graph-replayed"
        got = [d.cpu().numpy() for d in dsts]
    from vr180_convert_amd import remapper

    for kind in remapper.last_launch_kinds():
        KINDS[kind] = KINDS.get(kind, 0) + 1
    bad = 0
    maps = None
    sing = None
    for k in range(n):
        if use_rot:
            sp = list(spec)
            sp[rot_at] = ("rot", rots[k].tolist())
            maps = O.get_map(sp, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
        elif tuple_t:  # every eye: its own chain, its own source geometry
            maps = O.get_map(spec if k == 0 else spec2, radius=radius, size_input=sizes[k], size_output=(wo, ho))
            sing = None
        elif maps is None:
            maps = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
        want = O.remap(imgs[k], maps[0], maps[1], interp, border, bval, dst=fill.copy())
        if use_rot or tuple_t or sing is None:
            sp_now = sp if use_rot else (spec2 if (tuple_t and k == 1) else spec)
            sing = ill_conditioned(sp_now, radius, sizes[k] if tuple_t else (hs, ws), (wo, ho))
            if border in (1, 2, 3, 4):
                sing |= ~((np.abs(maps[0]) < 2.0 ** 20) & (np.abs(maps[1]) < 2.0 ** 20))  # (NaN counts as singular)
        if sing is not None and sing.any():
            diff = (got[k] != want).any(axis=2)
            SINGULAR[0] += int((diff & sing).sum())
            want = want.copy()
            want[sing] = got[k][sing]
        if (got[k] != want).any() and not use_rot:
            tie = float32_ties(spec2 if (tuple_t and k == 1) else spec, radius, sizes[k] if tuple_t else (hs, ws), (wo, ho))
            if tie.any():
                TIES[0] += int(((got[k] != want).any(axis=2) & tie).sum())
                want = want.copy()
                want[tie] = got[k][tie]
        bad += int((got[k] != want).sum())
        if DUMP[0] and (got[k] != want).any():
            pm = None
            try:
                sp_k = (spec2 if (tuple_t and k == 1) else spec) if not use_rot else sp
                pm = V.get_map(CS.to_product(sp_k), radius=radius, size_input=sizes[k] if tuple_t else (hs, ws), size_output=(wo, ho))
            except Exception as e:  # noqa: BLE001
                print("  (product map unavailable:", e, ")")
            dump_diff(k, got[k], want, maps, pm, fill)
    return desc, bad


def lut_case(rng, dev) -> tuple[str, int]:
    """cv2.remap alone (v1c_remap_lut, what chains with user-defined stages use) on random float32 maps: smooth, noisy, and
    sprinkled with the values the fixed-point conversion treats specially"""
    import ctypes as C

    from vr180_convert_amd import _native
    from vr180_convert_amd.remapper import _stream_ptr, border_scalar

    cn = int(rng.choice([1, 3, 4]))
    interp = int(rng.choice([0, 1, 2, 3, 4]))
    border = int(rng.integers(0, 6))
    bval = tuple(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 5))))
    hs, ws = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    ho, wo = int(rng.integers(1, 500)), int(rng.integers(1, 500))
    src = rng.integers(0, 256, (hs, ws, cn), dtype=np.uint8)
    jj, ii = np.mgrid[:ho, :wo].astype(np.float64)
    kind = int(rng.integers(0, 3))
    if kind == 0:  # affine + noise
        a = rng.normal(0, 1, 6)
        xm = a[0] * ii + a[1] * jj + rng.uniform(-ws, 2 * ws) + rng.normal(0, 0.3, (ho, wo))
        ym = a[2] * ii + a[3] * jj + rng.uniform(-hs, 2 * hs) + rng.normal(0, 0.3, (ho, wo))
    elif kind == 1:  # anywhere around the source
        xm = rng.uniform(-40, ws + 40, (ho, wo))
        ym = rng.uniform(-40, hs + 40, (ho, wo))
    else:  # on the 1/32 grid and half-way between its points (ties of cvRound)
        xm = rng.integers(-64, 32 * ws + 64, (ho, wo)) / 32.0 + rng.choice([0.0, 1 / 64, -1 / 64, 1e-7], (ho, wo))
        ym = rng.integers(-64, 32 * hs + 64, (ho, wo)) / 32.0 + rng.choice([0.0, 1 / 64, -1 / 64, 1e-7], (ho, wo))
    xm, ym = xm.astype(np.float32), ym.astype(np.float32)
    special = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 2.0 ** 26, -(2.0 ** 26), 67108863.0, 32767.0, 32767.5, 32768.0, -32768.0,
                        -32768.5, -32769.0, -0.5, -1.0, 0.0, -0.0, ws - 1.0, ws - 0.5, float(ws), hs - 1.0, float(hs), 1e-30, -1e-30], np.float32)
    for m in (xm, ym):
        k = int(rng.integers(0, max(2, m.size // 20)))
        m.reshape(-1)[rng.integers(0, m.size, k)] = rng.choice(special, k)
    pad = int(rng.integers(0, 3)) * 4
    xw = np.zeros((ho, wo + pad), np.float32)
    yw = np.zeros((ho, wo + pad), np.float32)
    xw[:, :wo], yw[:, :wo] = xm, ym
    fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
    s_d = make_view(rng, src, dev, allow_unaligned=True)
    d_d = make_view(rng, fill.copy(), dev, allow_unaligned=True)
    x_d, y_d = torch.from_numpy(xw).to(dev), torch.from_numpy(yw).to(dev)
    bv = border_scalar(bval)
    rc = _native.lib().v1c_remap_lut(dev.index, _stream_ptr(dev), s_d.data_ptr(), hs, ws, s_d.stride(0), cn, d_d.data_ptr(), ho, wo, d_d.stride(0),
                                     x_d.data_ptr(), y_d.data_ptr(), x_d.stride(0) * 4, interp, border, bv.ctypes.data)
    _native.check(rc, "v1c_remap_lut")
    got = d_d.cpu().numpy()
    want = O.remap(src, xm, ym, interp, border, bval, dst=fill.copy())
    bad = int((got != want).sum())
    desc = f"LUT cn={cn} interp={interp} border={border} bval={bval!r} out=({wo},{ho}) src=({ws},{hs}) maps={kind} map_pad={pad}"
    if bad and DUMP[0]:
        dump_diff(0, got, want, (xm, ym))
    return desc, bad


def api_case(rng, dev) -> tuple[str, int]:
    """the reference's own entry points on host arrays: apply() (remapper.py:324-403: a list of images, 2-D grayscale among them,
    radius 'auto' / 'max' / a number, the host pipeline behind it) and apply_lr() on arrays (merge=False)"""
    spec, _ = rand_spec(rng)
    t = CS.to_product(spec)
    interp = int(rng.choice([1, 1, 0, 2, 4]))
    border = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
    bval = int(rng.integers(0, 256)) if rng.random() < 0.5 else tuple(int(x) for x in rng.integers(0, 256, 3))
    wo, ho = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    hs, ws = int(rng.integers(8, 500)), int(rng.integers(8, 500))
    gray2d = rng.random() < 0.2
    cn = 1 if gray2d else int(rng.choice([3, 3, 1, 4]))
    n = int(rng.choice([1, 2, 3, 7, 12]))
    lr = rng.random() < 0.35

    def disc():
        im = rng.integers(40, 256, (hs, ws) if gray2d else (hs, ws, cn), dtype=np.uint8)
        yy, xx = np.mgrid[:hs, :ws]
        r = min(hs, ws) * float(rng.uniform(0.3, 0.49))
        im[((xx - ws // 2) ** 2 + (yy - hs // 2) ** 2) > r * r] = 0  # black surround: radius='auto' finds an edge
        return im

    rsel = rng.random()
    # (a 2-D image with radius='auto' raises IndexError in the reference -- get_radius indexes three axes, transformer.py:128-131 -- and here)
    radius = "auto" if rsel < 0.35 and not gray2d else "max" if rsel < 0.7 else float(rng.uniform(0.3, 1.2) * min(hs, ws) / 2)

    def masked_diff(g, w, r_used):
        """differing bytes outside the ill-conditioned pixels (module docstring)"""
        sing = ill_conditioned(spec, r_used, (hs, ws), (wo, ho))
        if border in (1, 2, 3, 4):
            xm, ym = O.get_map(spec, radius=r_used, size_input=(hs, ws), size_output=(wo, ho))
            sing |= ~((np.abs(xm) < 2.0 ** 20) & (np.abs(ym) < 2.0 ** 20))
        d = (g != w)
        d = d.any(axis=2) if d.ndim == 3 else d
        SINGULAR[0] += int((d & sing).sum())
        return int((g != w)[~sing].sum())

    desc = f"API {'apply_lr' if lr else 'apply'} spec={spec!r} cn={'2-D' if gray2d else cn} interp={interp} border={border} bval={bval!r} out=({wo},{ho}) src=({ws},{hs}) radius={radius!r} n={2 if lr else n}"
    if lr and not gray2d and cn == 3:
        import tempfile

        from vr180_convert_amd import _io

        left, right = disc(), disc()
        with tempfile.TemporaryDirectory() as td:  # apply_lr saves (a PNG here: lossless) and returns nothing, like the reference
            out_p = Path(td) / "sbs.png"
            V.apply_lr(t, left_path=left, right_path=right, out_path=out_p, size_output=(wo, ho), interpolation=interp, boarder_mode=border,
                       boarder_value=bval, radius=radius, device=dev)
            got = _io.imread(out_p)
        want = O.apply_lr(spec, left, right, size_output=(wo, ho), interpolation=interp, radius=radius, border_mode=border, border_value=bval)
        got = np.asarray(got)
        if got.shape != want.shape:
            return desc, want.size
        r_used = O.get_radius_smart(radius, [left, right])
        return desc, masked_diff(got[:, :wo], want[:, :wo], r_used) + masked_diff(got[:, wo:], want[:, wo:], r_used)
    if (not gray2d) and cn == 3 and rng.random() < 0.25:
        # remap_sharded: apply_lr over a batch of frames by worker threads (one per listed device -- the one card several times)
        frames = [(disc(), disc()) if rng.random() < 0.5 else np.concatenate([disc(), disc()], axis=1) for _ in range(n)]
        ndev = int(rng.choice([1, 2, 3, 5]))
        rad = radius if not isinstance(radius, str) or radius == "max" else float(min(hs, ws) * 0.45)
        got = V.remap_sharded(t, frames, size_output=(wo, ho), interpolation=interp, boarder_mode=border, boarder_value=bval, radius=rad,
                              devices=[dev.index] * ndev)
        bad = 0
        for f, g in zip(frames, got):
            l_, r_ = (f if isinstance(f, tuple) else (f[:, :ws], f[:, ws:]))
            r_used = O.get_radius_smart(rad, [l_, r_])
            want = O.apply_lr(spec, np.ascontiguousarray(l_), np.ascontiguousarray(r_), size_output=(wo, ho), interpolation=interp, radius=rad,
                              border_mode=border, border_value=bval)
            g = np.asarray(g)
            bad += (masked_diff(g[:, :wo], want[:, :wo], r_used) + masked_diff(g[:, wo:], want[:, wo:], r_used)) if g.shape == want.shape else want.size
        return desc.replace("API apply", f"API remap_sharded x{ndev}"), bad
    imgs = [disc() for _ in range(n)]
    got = V.apply(t, in_paths=imgs, size_output=(wo, ho), interpolation=interp, boarder_mode=border, boarder_value=bval, radius=radius, device=dev)
    want = O.apply(spec, [im[..., None] if gray2d else im for im in imgs], size_output=(wo, ho), interpolation=interp, border_mode=border,
                   border_value=bval, radius=radius)
    bad = 0
    r_used = O.get_radius_smart(radius, [im[..., None] if gray2d else im for im in imgs])
    for g, w in zip(got, want):
        g = np.asarray(g)
        bad += masked_diff(g.reshape(w.shape), w, r_used) if g.size == w.size else w.size
        if gray2d and g.ndim != 2:
            bad += 1  # a 2-D image comes back 2-D (cv2.remap keeps the rank)
    return desc, bad


def auto_case(rng, dev) -> tuple[str, int]:
    """radius='auto' with the radius never leaving the device (v1c_plan_run_auto; remapper.py:62-90 + :51-57): synthetic image
    circles of random radius and offset, square-ish outputs, every interpolation, one transformer or one per eye -- against the
    oracle's apply_lr(radius='auto'); where the device-resident form declines (fix-up pass needed) the exact one is compared."""
    from vr180_convert_amd import remapper

    mid = []
    if rng.random() < 0.3:
        mid.append(("poly", [0.0, 1.0, float(rng.uniform(-0.15, 0.08))]))
    if rng.random() < 0.3:
        mid.insert(0, ("rot", rand_rot(rng, False).tolist()))
    spec = [("equirect_enc", True)] + mid + [("fisheye_dec", "equidistant" if rng.random() < 0.8 else MODELS[int(rng.integers(5))])]
    cn = int(rng.choice([3, 3, 3, 1, 4]))
    interp = int(rng.choice([4, 4, 1, 1, 0, 2]))
    border = int(rng.choice([0, 0, 0, 1, 4]))
    hs, ws = int(rng.integers(64, 900)), int(rng.integers(64, 900))
    if rng.random() < 0.6:
        ws = (ws + 3) & ~3
    wo = int(rng.integers(64, 1100))
    ho = wo if rng.random() < 0.7 else max(16, wo + int(rng.integers(-80, 80)))
    tuple_t = rng.random() < 0.25

    def disc():
        im = rng.integers(30, 256, (hs, ws, cn), dtype=np.uint8)
        yy, xx = np.mgrid[:hs, :ws]
        r = min(hs, ws) * float(rng.uniform(0.25, 0.52))
        cx, cy = ws / 2 + float(rng.uniform(-4, 4)), hs / 2 + float(rng.uniform(-4, 4))
        im[((xx - cx) ** 2 + (yy - cy) ** 2) > r * r] = 0
        return im

    a, b = disc(), disc()
    desc = f"AUTO spec={spec!r} cn={cn} interp={interp} border={border} out=({wo},{ho}) src=({ws},{hs}) tuple={tuple_t}"
    try:
        ra, rb = O.get_radius(a), O.get_radius(b)
    except IndexError:
        return desc + " (no black border: skipped)", 0
    t = CS.to_product(spec)
    tt = (t, t) if tuple_t else t
    sbs = V.apply_lr_tensors(tt, torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), size_output=(wo, ho), interpolation=interp,
                             boarder_mode=border, radius="auto", auto_radius_on_device=True)
    form = remapper.last_auto_radius_form()
    KINDS["auto:" + form] = KINDS.get("auto:" + form, 0) + 1
    got = sbs.cpu().numpy()
    bad = 0
    for k, (im, r_used) in enumerate(((a, ra if tuple_t else max(ra, rb)), (b, rb if tuple_t else max(ra, rb)))):
        xm, ym = O.get_map(spec, radius=r_used, size_input=(hs, ws), size_output=(wo, ho))
        want = O.remap(im, xm, ym, interp, border, 0)
        g = got[:, k * wo:(k + 1) * wo]
        sing = ill_conditioned(spec, r_used, (hs, ws), (wo, ho))
        if border in (1, 2, 3, 4):
            sing |= ~((np.abs(xm) < 2.0 ** 20) & (np.abs(ym) < 2.0 ** 20))
        d = (g != want).any(axis=2)
        SINGULAR[0] += int((d & sing).sum())
        bad += int((g != want)[~sing].sum())
    return desc + f" form={form}", bad


def fused_case(rng, dev) -> tuple[str, int]:
    """v1c_remap_fused through raw ctypes (the one-shot export a foreign host binds: INTEGRATION.md): a lowered chain, pointers and
    pitches, nothing of the Python plan cache in between."""
    from vr180_convert_amd import _native
    from vr180_convert_amd.chain import lower_for_get_map
    from vr180_convert_amd.remapper import _stream_ptr, border_scalar

    spec, _ = rand_spec(rng)
    cn = int(rng.choice([3, 3, 1, 4]))
    interp = int(rng.choice([1, 1, 0, 2, 4]))
    border = int(rng.choice([0, 0, 0, 1, 2, 3, 4, 5]))
    bval = tuple(int(x) for x in rng.integers(0, 256, 3))
    wo, ho = rand_size(rng, 0.05), rand_size(rng, 0.05)
    hs, ws = rand_size(rng, 0.05, 2), rand_size(rng, 0.05, 2)
    if interp in (2, 4):
        wo, ho = min(wo, 900), min(ho, 900)
    radius = float(rng.uniform(0.3, 1.3) * min(ws, hs) / 2 + 1.0)
    src = rng.integers(0, 256, (hs, ws, cn), dtype=np.uint8)
    fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
    s_d = make_view(rng, src, dev, allow_unaligned=True)
    d_d = make_view(rng, fill.copy(), dev, allow_unaligned=True)
    chain = lower_for_get_map(CS.to_product(spec), radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    bv = border_scalar(bval)
    import ctypes as C

    rc = _native.lib().v1c_remap_fused(dev.index, _stream_ptr(dev), s_d.data_ptr(), hs, ws, s_d.stride(0), cn, d_d.data_ptr(), ho, wo, d_d.stride(0),
                                       C.byref(chain), interp, border, bv.ctypes.data)
    _native.check(rc, "v1c_remap_fused")
    got = d_d.cpu().numpy()
    xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    want = O.remap(src, xm, ym, interp, border, bval, dst=fill.copy())
    sing = ill_conditioned(spec, radius, (hs, ws), (wo, ho))
    if border in (1, 2, 3, 4):
        sing |= ~((np.abs(xm) < 2.0 ** 20) & (np.abs(ym) < 2.0 ** 20))
    d = (got != want).any(axis=2)
    SINGULAR[0] += int((d & sing).sum())
    return f"FUSED spec={spec!r} cn={cn} interp={interp} border={border} bval={bval!r} out=({wo},{ho}) src=({ws},{hs}) radius={radius!r}", int((got != want)[~sing].sum())


def radius_case(rng, dev) -> tuple[str, int]:
    """get_radius (transformer.py:108-140) as the device kernel against the oracle: random images with black margins, noise around the
    threshold, every channel count, pitched views, images wider than high and the other way round, none / several rises and falls"""
    from vr180_convert_amd.remapper import _get_radius_any

    cn = int(rng.choice([1, 3, 4]))
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    thr = int(rng.choice([10, 10, 1, 25, 200, 0, 255]))
    img = rng.integers(0, 30 if rng.random() < 0.5 else 256, (h, w, cn), dtype=np.uint8)
    if rng.random() < 0.8:  # black margins of random widths (possibly none, possibly everything)
        a, b = sorted(int(v) for v in rng.integers(0, w + 1, 2))
        c, d = sorted(int(v) for v in rng.integers(0, h + 1, 2))
        img[:, :a] = 0
        img[:, b:] = 0
        img[:c] = 0
        img[d:] = 0
    t = make_view(rng, img, dev, allow_unaligned=True)
    try:
        want = ("value", O.get_radius(img, thr))
    except IndexError:
        want = ("IndexError", None)
    try:
        got = ("value", _get_radius_any(t, thr))
    except IndexError:
        got = ("IndexError", None)
    return f"RADIUS cn={cn} size=({w},{h}) threshold={thr} want={want} got={got}", 0 if got == want else 1


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", type=float, default=0.15, help="share of sizes drawn from 1200 - 2700")
    ap.add_argument("--lut", type=float, default=0.15, help="share of cases that fuzz cv2.remap alone (v1c_remap_lut) on random maps")
    ap.add_argument("--hot", type=float, default=0.3, help="share of the chain cases drawn from the shapes the tuned kernels are selected for")
    ap.add_argument("--gen2", type=float, default=0.0, help="share of the chain cases with radial stages / zooms in front of a rotation (general mode 2)")
    ap.add_argument("--api", type=float, default=0.1, help="share of cases through apply() / apply_lr() on host arrays")
    ap.add_argument("--auto", type=float, default=0.06, help="share of cases through apply_lr_tensors(radius='auto') with the radius on the device")
    ap.add_argument("--fused", type=float, default=0.06, help="share of cases through v1c_remap_fused by raw ctypes")
    ap.add_argument("--only", type=int, default=None, help="run only this case number (reproduce)")
    ap.add_argument("--log", default=None)
    ap.add_argument("--trace", default=None, help="file that always holds the number of the case being run")
    ap.add_argument("--dump", action="store_true", help="print where a mismatching unit differs")
    a = ap.parse_args()
    DUMP[0] = a.dump
    HOT[0] = a.hot
    GEN2[0] = a.gen2
    dev = torch.device("cuda", 0)
    log = open(a.log, "a") if a.log else None

    def say(s: str) -> None:
        print(s, flush=True)
        if log:
            log.write(s + "\n")
            log.flush()

    t0 = time.time()
    n_cases = n_bad = 0
    last = t0
    case = 0
    while time.time() - t0 < a.seconds:
        rng = np.random.default_rng([a.seed, case])  # every case reproducible by itself
        if a.only is not None:
            rng = np.random.default_rng([a.seed, a.only])
        if a.trace:  # (a crash of the process -- a GPU memory fault aborts it -- leaves the case that was running on record)
            with open(a.trace, "w") as tf:
                tf.write(f"seed {a.seed} case {a.only if a.only is not None else case}\n")
        try:
            r_kind = rng.random()
            if r_kind < a.lut:
                desc, bad = lut_case(rng, dev)
            elif r_kind < a.lut + 0.05:
                desc, bad = radius_case(rng, dev)
            elif r_kind < a.lut + 0.05 + a.api:
                desc, bad = api_case(rng, dev)
            elif r_kind < a.lut + 0.05 + a.api + a.auto:
                desc, bad = auto_case(rng, dev)
            elif r_kind < a.lut + 0.05 + a.api + a.auto + a.fused:
                desc, bad = fused_case(rng, dev)
            else:
                desc, bad = one_case(rng, dev, a.big)
        except Exception as e:  # noqa: BLE001 -- a refusal of the product (documented limits) is reported, not fatal
            desc, bad = f"EXCEPTION {type(e).__name__}: {e}", -1
        n_cases += 1
        if bad != 0:
            n_bad += 1
            say(f"[case {a.only if a.only is not None else case}] {'MISMATCH ' + str(bad) + ' bytes' if bad > 0 else ''} {desc}")
        if a.only is not None:
            say(f"case {a.only}: {bad} differing bytes; {desc}")
            break
        case += 1
        if time.time() - last > 30:
            last = time.time()
            say(f"... {n_cases} cases, {n_bad} reported, {time.time() - t0:.0f} s")
    say(f"fuzz seed {a.seed}: {n_cases} cases in {time.time() - t0:.0f} s, {n_bad} reported; {SINGULAR[0]} differing ill-conditioned pixels left out"
        + (f", {TIES[0]} at float32 rounding ties" if TIES[0] else ""))
    say("kernel families of the chain cases' launch groups: " + ", ".join(f"{k} x{v}" for k, v in sorted(KINDS.items())))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
