"""Runs the product's per-pixel device code (v1c_core.hpp: interpreter, fused ray path, sampler)
compiled for the HOST (tests/host_emul) against the oracle / the reference goldens.  This is how
the kernels' arithmetic is validated in the GPU-less build container; the -m gpu tests repeat the
comparisons through the real kernels."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import chainspecs as CS
from test_oracle_golden import assert_maps_match


def emul_map(E, ch, W, H, mode, rot=None):
    xm = np.empty((H, W), np.float32)
    ym = np.empty((H, W), np.float32)
    st = (C.c_longlong * 5)()
    r = None if rot is None else np.ascontiguousarray(rot, np.float64)
    rc = E.emul_get_map(C.byref(ch), C.c_void_p(None if r is None else r.ctypes.data), W, H, mode,
                        C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), st)
    return rc, xm, ym, list(st)


@pytest.mark.parametrize("name", list(CS.SMALL_CASES))
def test_interpreter_and_ray_path_vs_reference(emul_lib, oracle_mod, golden_dir, name):
    g = np.load(golden_dir / "maps_small.npz")
    spec, out, inp, radius = CS.SMALL_CASES[name]
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    rc, xm, ym, _ = emul_map(emul_lib, ch, out[0], out[1], 0)
    assert rc == 0
    assert_maps_match(xm, ym, g[f"{name}__x"], g[f"{name}__y"], name + " literal")
    rc, xm, ym, st = emul_map(emul_lib, ch, out[0], out[1], 1)
    if rc == 0:  # chain has the ray shape
        assert_maps_match(xm, ym, g[f"{name}__x"], g[f"{name}__y"], name + " ray")
        if st[4]:  # plan says "no fix-up launch needed": then no pixel may have needed it
            assert st[1] == 0


def test_ray_path_expected_coverage(emul_lib, oracle_mod):
    """Which chains take the fused path, with which table variable, and how much fix-up."""
    expect = {  # name: (ray?, var_is_w, needs fix-up pixels?)
        "apply_equirectangular": (True, 0, False), "c2_poly": (True, 1, False), "c4_rot_poly": (True, 1, False),
        "c5_calib_left": (True, 0, False), "zoom": (True, 0, False), "nonsquare": (True, 0, False),
        "poly_c0": (True, 0, True), "poly_signchange": (True, 0, True), "back_hemisphere": (True, 0, True),
        # round 5: chains that do not start with EquirectangularEncoder(is_latitude_y=True) (SURVEY.md 8a "planar mode")
        "apply_rectilinear": (True, 0, False), "apply_orthographic": (True, 1, True), "transformer_poly": (True, 1, False),
        "transformer_rotator": (True, 0, None), "rot_after_radial": (True, 0, False), "equirect_lat_x": (True, 0, False),
        "equirect_decoder": (False, 0, False),
    }
    for name, (ray, var_w, fix) in expect.items():
        spec, out, inp, radius = CS.SMALL_CASES[name]
        ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
        rc, _, _, st = emul_map(emul_lib, ch, out[0], out[1], 1)
        assert (rc == 0) == ray, name
        if ray:
            assert st[2] == var_w, name
            assert fix is None or (st[1] > 0) == fix, (name, st)


def plan_info(E, ch, W, H):
    out = (C.c_longlong * 12)()
    assert E.emul_plan_info(C.byref(ch), W, H, out) == 0
    keys = ["ok", "usable", "base", "gen_mode", "fn", "var_is_w", "n_int", "below_lv1", "below_lv2", "first_invalid", "shared_entry", "no_fixup"]
    return dict(zip(keys, list(out)))


@pytest.mark.parametrize("name", list(CS.PLANAR_CASES))
def test_planar_and_general_modes_full_size_vs_reference(emul_lib, oracle_mod, golden_dir, name):
    """The chains the fused path serves since round 5 -- planar (fisheye -> fisheye), is_latitude_y=False, a rotation behind radial
    stages -- through the product's per-pixel code compiled for the host, at 1024 x 1024 and a non-square size: the bucket planes are
    SHA-equal to the REFERENCE's get_map (tests/golden/make_golden.py planar), NaN pattern included (orthographic: the pixels the
    reference makes NaN take the interpreter, which makes them NaN too)."""
    g = np.load(golden_dir / "maps_planar.npz")
    spec, out, inp, radius = CS.PLANAR_CASES[name]
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    info = plan_info(emul_lib, ch, out[0], out[1])
    assert info["ok"] and info["usable"], info
    rc, xm, ym, st = emul_map(emul_lib, ch, out[0], out[1], 1)
    assert rc == 0 and st[0] == 1
    s = CS.FULL_STRIDE
    assert_maps_match(xm[::s], ym[::s], g[f"{name}__rows_x"], g[f"{name}__rows_y"], name + " rows")
    assert_maps_match(xm[:, ::s], ym[:, ::s], g[f"{name}__cols_x"], g[f"{name}__cols_y"], name + " cols")
    assert int(np.isnan(xm).sum()) == int(g[f"{name}__nan"])
    assert hashlib.sha256(CS.buckets(xm).tobytes()).digest() == g[f"{name}__sha_bx"].tobytes()
    assert hashlib.sha256(CS.buckets(ym).tobytes()).digest() == g[f"{name}__sha_by"].tobytes()
    if info["no_fixup"]:
        assert st[1] == 0
    # which form of the kernels the plan may use: unrotated planar chains whose composite is defined everywhere prove "one table entry
    # per lane, no fix-up pass" like BASELINE's configurations do (mirror / batch kernels)
    expect_fast = {"apply_rectilinear", "apply_stereographic", "apply_equidistant", "transformer_poly", "equirect_lat_x", "planar_nonsquare"}
    if name in expect_fast:
        assert info["shared_entry"] and info["no_fixup"], info
    assert info["base"] == (0 if name == "rot_after_radial" else 2 if name.startswith("equirect_lat_x") else 1)
    assert info["gen_mode"] == {"transformer_rotator": 2, "planar_rot_small_angle": 2, "rot_after_radial": 2, "equirect_lat_x": 1,
                                "equirect_lat_x_rot": 1}.get(name, 0)


@pytest.mark.parametrize("name", ["C1", "C2", "C3"])
def test_ray_path_full_size_sha(emul_lib, oracle_mod, golden_dir, name):
    g = np.load(golden_dir / "maps_full.npz")
    spec, out, inp, radius = CS.FULL_CASES[name]
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    rc, xm, ym, st = emul_map(emul_lib, ch, out[0], out[1], 1)
    assert rc == 0 and st[0] == 1 and st[1] == 0 and st[4] == 1
    assert hashlib.sha256(CS.buckets(xm).tobytes()).digest() == g[f"{name}__sha_bx"].tobytes()
    assert hashlib.sha256(CS.buckets(ym).tobytes()).digest() == g[f"{name}__sha_by"].tobytes()


@pytest.mark.parametrize("name,mode", [("C1", 1), ("C2", 1), ("C2", 0)])
def test_rows_mirror_about_the_equator(emul_lib, oracle_mod, name, mode):
    """The premise of k_ray_lin3_pair_mirror (kernels_mirror.hip): for an unrotated equirectangular chain output rows j and
    H - j differ only in the sign of sin(lat) -- the x coordinates are the SAME float32 bits and y mirrors about the source
    centre (y' - c_y == -(y - c_y) up to the rounding of one fma).  Checked on the product's per-pixel code compiled for
    the host, at the full BASELINE sizes, fused ray path (and the literal interpreter for C2)."""
    spec, out, inp, radius = CS.FULL_CASES[name]
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    W, H = out
    rc, xm, ym, _ = emul_map(emul_lib, ch, W, H, mode)
    assert rc == 0
    top, bottom = slice(1, H // 2), slice(H - 1, H // 2, -1)  # rows 1 .. H/2 - 1 and their mirror images H - 1 .. H/2 + 1
    if mode == 1:
        assert np.array_equal(xm[top].view(np.uint32), xm[bottom].view(np.uint32))
    else:  # the literal chain goes through sqrt / atan2 / cos / sin per stage: equal to the last float32 bit or two
        assert np.max(np.abs(xm[top] - xm[bottom])) <= 2 * np.spacing(np.float32(W))
    cy = inp[0] // 2
    assert np.max(np.abs((ym[bottom].astype(np.float64) - cy) + (ym[top].astype(np.float64) - cy))) <= np.spacing(np.float32(H))


def test_per_unit_rotation_override(emul_lib, oracle_mod, golden_dir):
    """BASELINE config 5: one plan, the rotation arrives per unit."""
    g = np.load(golden_dir / "maps_c5.npz")
    base = oracle_mod.chain_from_spec([("equirect_enc", True), ("rot", np.eye(3)), CS.EQUI], radius=96.0,
                                      size_input=(192, 192), size_output=(192, 192))
    for frame in (0, 1, 7):
        for eye in (0, 1):
            rot = oracle_mod.quat_to_matrix(CS.c5_spec(frame, eye)[1][1])
            for mode in (0, 1):
                rc, xm, ym, _ = emul_map(emul_lib, base, 192, 192, mode, rot)
                assert rc == 0
                assert_maps_match(xm, ym, g[f"f{frame}_e{eye}__x"], g[f"f{frame}_e{eye}__y"], f"c5 {frame} {eye} mode {mode}")


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_sampler_equals_oracle(emul_lib, product_lib, oracle_mod, cn):
    O = oracle_mod
    rng = np.random.default_rng(1)
    tabs = {}
    for interp, k in ((2, 4), (4, 8)):
        tabs[interp] = np.zeros(1024 * k * k, np.int16)
        assert product_lib.v1c_build_itab(interp, tabs[interp].ctypes.data) == 0
    Hs, Ws, H, W = 61, 83, 70, 90
    src = rng.integers(0, 256, (Hs, Ws, cn), dtype=np.uint8)
    xm = (rng.random((H, W)) * (Ws + 24) - 12).astype(np.float32)
    ym = (rng.random((H, W)) * (Hs + 24) - 12).astype(np.float32)
    xm[0, :5] = [np.nan, np.inf, -np.inf, 1e30, -1e30]
    ym[1, :3] = [np.nan, 3e9, -3e9]
    xm[2, :8] = np.arange(8)
    ym[2, :8] = np.arange(8)
    xm[3, :4] = [0.5 / 32, 1.5 / 32, 2.5 / 32, -0.5 / 32]  # round-half-even ties
    for interp in (0, 1, 2, 3, 4):
        for border in range(6):
            for bv in (0, (10, 200, 30, 77)):
                ref = np.full((H, W, cn), 123, np.uint8)
                out = ref.copy()
                O.remap(src, xm, ym, interp, border, bv, dst=ref)
                cv = O.border_scalar(bv)
                it = tabs.get(interp)
                rc = emul_lib.emul_remap(
                    C.c_void_p(src.ctypes.data), Hs, Ws, C.c_int64(src.strides[0]), cn, C.c_void_p(out.ctypes.data), H, W,
                    C.c_int64(out.strides[0]), C.c_void_p(xm.ctypes.data), C.c_void_p(ym.ctypes.data), interp, border,
                    C.c_void_p(cv.ctypes.data), C.c_void_p(None if it is None else it.ctypes.data))
                assert rc == 0
                assert np.array_equal(ref, out), (cn, interp, border, bv)


@pytest.mark.parametrize("name", ["c2_poly", "c4_rot_poly", "apply_equirectangular", "transformer_rotator"])
def test_an_entrys_level_covers_the_lanes_it_is_shared_by(emul_lib, oracle_mod, name):
    """The tile kernels evaluate a lane's 4 pixels with pixel 1's table entry wherever the entry's level (the two low mantissa bits of
    c7: valid on |z| <= 0.5 + level) says so.  emul_lane_probe applies that rule on the host: G by the shared rule against G by the
    pixel's own entry -- both within 1.5e-15 of the function, so within 4e-15 of each other -- for lanes all over the image."""
    spec, out, inp, radius = CS.SMALL_CASES[name]
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    buf = (C.c_double * 32)()
    shared = 0
    for j in range(0, out[1], max(1, out[1] // 23)):
        for i4 in range(0, out[0] - 3, 4 * max(1, out[0] // 92)):
            if emul_lib.emul_lane_probe(C.byref(ch), out[0], out[1], j, i4, buf) != 0:
                pytest.skip("not a fused chain")
            for k in range(4):
                t, idx, z1, level, gs, go = buf[8 * k:8 * k + 6]
                if np.isfinite(gs) and np.isfinite(go):
                    assert abs(gs - go) <= 4e-15 * max(abs(go), 1.0), (name, j, i4 + k, z1, level, gs, go)
                    shared += abs(z1) > 0.5 and abs(z1) <= 0.5 + level
    assert shared > 0 or name == "transformer_rotator"


FUZZ_1974 = [("equirect_enc", True), ("zoom", 1.7738941798002221),
             ("rot", [[0.9982728060815481, 0.058690132718145105, 0.002621633002223754], [-0.05872470490000005, 0.9981458957992685, 0.01600561568587129],
                      [-0.0016774005126221434, -0.016131925508209265, 0.9998684650027312]]),
             ("rot", [[0.9999105236968615, -0.00792577709040616, 0.010776207949987819], [0.007884260865708426, 0.999961353887725, 0.003889622299068912],
                      [-0.010806619770753778, -0.003804311835424161, 0.999934369936642]]), ("fisheye_dec", "equidistant")]


def lane_model(E, ch, w, h, tx, ty, ignore_read=0):
    buf = (C.c_double * 6)()
    rc = E.emul_tile_lane_model(C.byref(ch), w, h, tx, ty, ignore_read, buf)
    return rc, dict(zip(["err", "in_table", "shared", "pixel1_outside_slice", "slice", "not_shared"], list(buf)))


def test_lane_model_reproduces_and_clears_the_shared_entry_bug(emul_lib, oracle_mod):
    """tools/fuzz.py seed 34 case 1974 on the HOST: the model of a tile's table slice and of lane_coords' entry sharing
    (emul_tile_lane_model; the sharing rule itself is the kernels' own function, v1c_core.hpp: shared_entry_serves).  Tile (1, 54) of that
    case has 44 lanes whose pixel 1 points outside the tile's slice; without the index test of round 5's fix two good pixels take the
    clamped neighbour's polynomial -- 0.47 % off in G, 0.4 px --, with it every shared pixel agrees with its own entry."""
    ch = oracle_mod.chain_from_spec(FUZZ_1974, radius=96.0, size_input=(192, 192), size_output=(154, 1427))
    rc, old = lane_model(emul_lib, ch, 154, 1427, 1, 54, ignore_read=1)
    assert rc == 0 and old["pixel1_outside_slice"] > 0 and old["err"] > 1e-3, old
    rc, new = lane_model(emul_lib, ch, 154, 1427, 1, 54)
    assert rc == 0 and new["shared"] > 0 and new["err"] <= 4e-15, new


def test_lane_model_on_general_mode_chains(emul_lib, oracle_mod):
    """The same model over tiles of random chains with radial stages / zooms in front of a rotation (lanes split between the table and the
    fix-up pass) and of the classic shapes: whatever entry a lane shares, it gives what the pixel's own entry gives."""
    rng = np.random.default_rng(11)
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector

    tiles = shared = outside = 0
    for case in range(14):
        pre = [("zoom", float(rng.uniform(0.5, 2.6)))] if rng.random() < 0.6 else [("poly", [0.0, 1.0, float(rng.uniform(-0.2, 0.2))])]
        rot = ("rot", np.asarray(as_rotation_matrix(from_rotation_vector(rng.normal(0, 0.3, 3))), float).tolist())
        enc = ("equirect_enc", True) if rng.random() < 0.7 else ("fisheye_enc", "equidistant")
        spec = [enc] + (pre if case % 4 else []) + [rot, ("fisheye_dec", "equidistant")]
        w, h = int(rng.integers(100, 900)), int(rng.integers(100, 1500))
        src = int(rng.integers(100, 800))
        ch = oracle_mod.chain_from_spec(spec, radius=float(rng.uniform(0.3, 0.7) * src), size_input=(src, src), size_output=(w, h))
        for _ in range(6):
            tx, ty = int(rng.integers(0, (w + 63) // 64)), int(rng.integers(0, (h + 15) // 16))
            rc, m = lane_model(emul_lib, ch, w, h, tx, ty)
            if rc != 0:
                break
            tiles += 1
            shared += m["shared"]
            outside += m["pixel1_outside_slice"]
            assert m["err"] <= 4e-15, (spec, (w, h), src, (tx, ty), m)
    assert tiles > 40 and shared > 1000, (tiles, shared, outside)


@pytest.mark.parametrize("name,size", [("c2_poly", 1024), ("lr_rotator", 2048), ("c5_calib_left", 1024), ("apply_equirectangular", 832), ("apply_rectilinear", 1024)])
def test_one_entry_per_lane_where_the_plan_says_so(emul_lib, oracle_mod, name, size):
    """Where the plan proves "one table entry serves a lane's 4 pixels" (ray_entry_is_shared: what selects the OWN = 0 kernels, which use
    pixel 1's entry WITHOUT a per-pixel test -- every BASELINE configuration) the host model of the tiles must find no in-table pixel
    the sharing rule would refuse, and every shared value within 4e-15 of the pixel's own entry."""
    spec = CS.SMALL_CASES[name][0]
    ch = oracle_mod.chain_from_spec(spec, radius=size / 2, size_input=(size, size), size_output=(size, size))
    info = plan_info(emul_lib, ch, size, size)
    assert info["ok"] and info["usable"] and info["shared_entry"], info
    buf = (C.c_double * 9)()
    assert emul_lib.emul_lane_model_all(C.byref(ch), size, size, 0, buf) == 0
    err, in_table, shared, outside, slices, not_shared, mp_pixels, mp_err, mp_uncovered = list(buf)
    assert in_table > 0.5 * size * size and not_shared == 0 and err <= 4e-15, list(buf)
    # the m-polynomial twin (lane_coords<..., MPOLY>: the interval of pixel 1 from an fp32 root, off by one near a boundary): both candidate
    # entries inside the tile's slice and of the plan's level, their polynomial in m within 4e-15 of the pixel's own entry
    assert mp_pixels > 0.5 * size * size and mp_uncovered == 0 and mp_err <= 4e-15, list(buf)


def test_the_float32_tie_case_differs_only_at_ties(emul_lib, oracle_mod):
    """The one measure-zero exception to bucket equality (tools/fuzz_cpu.py seed 10 case 49): FisheyeEncoder("stereographic")
    * FisheyeDecoder("rectilinear"), 658 x 658 from 1626 x 1626 -- one pixel and its three mirror images have a float64 coordinate 3 ulps
    from a float32 midpoint, and the table path rounds them the other way than libm.  Nothing else differs."""
    spec = [("fisheye_enc", "stereographic"), ("fisheye_dec", "rectilinear")]
    out, inp, radius = (658, 658), (1626, 1626), 813.0
    ch = oracle_mod.chain_from_spec(spec, radius=radius, size_input=inp, size_output=out)
    rc, xm, ym, _ = emul_map(emul_lib, ch, out[0], out[1], 1)
    assert rc == 0
    ox, oy = oracle_mod.get_map(spec, radius=radius, size_input=inp, size_output=out)
    fx, fy = oracle_mod.get_map(ch, radius=radius, size_input=inp, size_output=out, f64=True)

    def bucket(v):
        ok = np.isfinite(v)
        return np.where(ok, np.rint(np.where(ok, v, 0).astype(np.float64) * 32), -2.0 ** 40)

    d = (bucket(xm) != bucket(ox)) | (bucket(ym) != bucket(oy))
    tie = np.zeros_like(d)
    for v in (fx, fy):
        f = v.astype(np.float32)
        up, dn = np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))
        with np.errstate(invalid="ignore", over="ignore"):
            m1, m2 = (f.astype(np.float64) + up.astype(np.float64)) / 2, (f.astype(np.float64) + dn.astype(np.float64)) / 2
            tie |= np.minimum(np.abs(v - m1), np.abs(v - m2)) <= 1e-14 * np.abs(v)
    assert int(d.sum()) <= 8 and not (d & ~tie).any(), (int(d.sum()), np.argwhere(d & ~tie)[:4].tolist())


def test_claims_about_units_that_override_the_rotation(emul_lib, oracle_mod):
    """Per-frame calibration rotations (BASELINE config 5) and v1c_plan_run_auto run kernels that trust three closed-form claims of the
    host (plan.hip: decide_launch) without testing a pixel: the rotated reach stays in valid table intervals, one table entry serves a
    lane, every |32 x|, |32 y| < 2^21.  emul_unit_rotation_check restates the claims and counts what the pixels do, for small and large
    random rotations of equidistant / polynomial / stereographic chains."""
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector

    rng = np.random.default_rng(21)
    claimed = [0, 0, 0]
    mp_total = 0
    for case in range(18):
        mid = [("poly", [0.0, 1.0, float(rng.uniform(-0.15, 0.05))])] if case % 3 == 1 else []
        dec = ("fisheye_dec", "stereographic") if case % 3 == 2 else CS.EQUI
        spec = [("equirect_enc", True), ("rot", np.eye(3).tolist())] + mid + [dec]
        size = int(rng.integers(300, 1500))
        ch = oracle_mod.chain_from_spec(spec, radius=float(rng.uniform(0.35, 0.6) * size), size_input=(size, size), size_output=(size, size))
        R = np.ascontiguousarray(np.asarray(as_rotation_matrix(from_rotation_vector(rng.normal(0, 0.04 if case % 2 else 0.6, 3))), float).reshape(9))
        buf = (C.c_double * 11)()
        assert emul_lib.emul_unit_rotation_check(C.byref(ch), C.c_void_p(R.ctypes.data), size, size, buf) == 0
        front, covered, shared, bounded, declined, refused, cmax, err, mp_px, mp_err, mp_uncovered = list(buf)
        assert mp_uncovered == 0 and mp_err <= 4e-15, (spec, size, R.tolist(), list(buf))
        mp_total += mp_px
        assert front == 1
        assert not covered or declined == 0, (spec, size, R.tolist(), list(buf))
        assert not shared or (refused == 0 and err <= 4e-15), (spec, size, R.tolist(), list(buf))
        assert not bounded or cmax < 2097152.0, (spec, size, R.tolist(), list(buf))
        claimed = [claimed[0] + covered, claimed[1] + shared, claimed[2] + bounded]
    assert min(claimed) >= 4 and mp_total > 1e6, (claimed, mp_total)
