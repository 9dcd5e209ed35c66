"""Round-5 parity tests on a real MI355X, through the C ABI, bit-exact against the oracle:

* bicubic / Lanczos4 footprints that cross the edge of the source served from the LDS box (BORDER_CONSTANT: the border colour staged
  around the image) instead of the per-pixel patch path -- the reference's all-defaults call (remapper.py:330-333: Lanczos4, radius from
  a full-frame circle) puts whole tiles there;
* the unit ring under many threads (advisor finding of round 4).
"""
import numpy as np
import pytest
import torch

import chainspecs as CS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import vr180_convert_amd as V
    from vr180_convert_amd import _native

    _native.lib()
    assert torch.cuda.is_available()
    return V


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def _noise(h, w, seed, cn=3):
    return np.random.default_rng(seed).integers(1, 256, (h, w, cn), dtype=np.uint8)


KXK_EDGE_CASES = {
    # name: (spec, source (H, W), output (W, H), radius, border value)
    "defaults_circle_touches_frame": ([("equirect_enc", True), CS.EQUI], (640, 640), (640, 640), 320.0, 0),
    "circle_beyond_the_frame": ([("equirect_enc", True), CS.EQUI], (600, 600), (640, 576), 330.0, (9, 200, 77)),
    "width_not_a_multiple_of_4": ([("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI], (499, 613), (704, 512), 320.0, (255, 1, 128)),
    "rotated": ([("equirect_enc", True), ("rot", CS.ry(0.5)), CS.EQUI], (512, 512), (512, 512), 262.0, 37),
    "zoomed_out_wide_rim": ([("equirect_enc", True), ("zoom", 0.6), CS.EQUI], (300, 300), (512, 480), 150.0, (3, 2, 1)),
    "tiny_source": ([("equirect_enc", True), CS.EQUI], (9, 11), (448, 432), 5.0, (100, 150, 200)),
}


@pytest.mark.parametrize("interp", [4, 2])
@pytest.mark.parametrize("name", list(KXK_EDGE_CASES))
def test_kxk_footprints_crossing_the_edge_are_served_from_the_box(V, oracle_mod, dev, name, interp):
    """Noise all the way to the edge of the source (no black rim: every tap beyond the edge matters), BORDER_CONSTANT with a non-zero
    colour, a pair, a single image and a batch of three; every byte against the oracle, on the tile kernel."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec, (hs, ws), (wo, ho), radius, bv = KXK_EDGE_CASES[name]
    imgs = [_noise(hs, ws, 500 + k) for k in range(3)]
    xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    want = [O.remap(im, xm, ym, interp, 0, bv) for im in imgs]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    t = CS.to_product(spec)
    for group in (srcs[:2], srcs[:1], srcs):
        dsts = [torch.full((ho, wo, 3), 99, dtype=torch.uint8, device=dev) for _ in group]
        assert V.remap_tensors(t, group, dsts, radius=radius, interpolation=interp, boarder_value=bv) == ["ray"]
        kind = remapper.last_launch_kinds()[0].split("+")[0]
        assert kind == "tile", (name, interp, kind)
        for k, d in enumerate(dsts):
            got = d.cpu().numpy()
            assert np.array_equal(got, want[k]), (name, interp, len(group), k, int((got != want[k]).sum()))


@pytest.mark.parametrize("border", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("interp", [4, 2])
def test_kxk_footprints_crossing_the_edge_other_border_modes(V, oracle_mod, dev, interp, border):
    """REPLICATE / REFLECT / WRAP / REFLECT_101 stage the pixel borderInterpolate maps every cell of the box to (TRANSPARENT keeps the
    per-pixel sampler): noise to the edge, circle beyond the frame, odd width; pairs and a single image, every byte against the oracle."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec, (hs, ws), (wo, ho), radius = [("equirect_enc", True), ("poly", [0, 1, -0.05]), CS.EQUI], (333, 411), (576, 448), 215.0
    imgs = [_noise(hs, ws, 600 + k) for k in range(2)]
    xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    fill = np.full((ho, wo, 3), 55, np.uint8)
    want = [O.remap(im, xm, ym, interp, border, 0, dst=fill.copy()) for im in imgs]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    for group in (srcs, srcs[:1]):
        dsts = [torch.from_numpy(fill.copy()).to(dev) for _ in group]
        assert V.remap_tensors(CS.to_product(spec), group, dsts, radius=radius, interpolation=interp, boarder_mode=border) == ["ray"]
        assert remapper.last_launch_kinds()[0].split("+")[0] == "tile"
        for k, d in enumerate(dsts):
            got = d.cpu().numpy()
            assert np.array_equal(got, want[k]), (interp, border, len(group), k, int((got != want[k]).sum()))


@pytest.mark.parametrize("interp", [4, 2])
def test_kxk_edge_footprints_with_a_rotation_per_unit(V, oracle_mod, dev, interp):
    """Units that override the rotation reduce their boxes in the kernel: the same border-colour staging there."""
    from vr180_convert_amd import transformer as T

    O = oracle_mod
    n, size = 4, 384
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    imgs = [_noise(size, size, 700 + f) for f in range(n)]
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(n)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.zeros_like(s) for s in srcs]
    V.remap_tensors(base, srcs, dsts, radius=size / 2 + 3, interpolation=interp, rotations=quats, boarder_value=(7, 8, 9))
    for f in range(n):
        want = O.apply(CS.c5_spec(f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=interp, radius=size / 2 + 3,
                       border_value=(7, 8, 9))[0]
        got = dsts[f].cpu().numpy()
        assert np.array_equal(got, want), (interp, f, int((got != want).sum()))


# ---------------------------------------------------------------------------- planar / general modes (SURVEY.md 8a "planar mode")
@pytest.mark.parametrize("name", list(CS.PLANAR_CASES))
def test_planar_and_general_modes_device_maps_vs_reference(V, golden_dir, name):
    """v1c_plan_get_map of the chains the fused path serves since round 5 against the REFERENCE's get_map at full size: every 256th row
    and column in the same 1/32-pixel buckets (NaN <=> NaN), SHA-256 of the whole bucket planes equal."""
    import hashlib

    from test_oracle_golden import assert_maps_match

    g = np.load(golden_dir / "maps_planar.npz")
    spec, out, inp, radius = CS.PLANAR_CASES[name]
    xm, ym = V.get_map(CS.to_product(spec), radius=radius, size_input=inp, size_output=out)
    s = CS.FULL_STRIDE
    assert_maps_match(xm[::s], ym[::s], g[f"{name}__rows_x"], g[f"{name}__rows_y"], name + " rows")
    assert_maps_match(xm[:, ::s], ym[:, ::s], g[f"{name}__cols_x"], g[f"{name}__cols_y"], name + " cols")
    assert int(np.isnan(xm).sum()) == int(g[f"{name}__nan"])
    assert hashlib.sha256(CS.buckets(xm).tobytes()).digest() == g[f"{name}__sha_bx"].tobytes()
    assert hashlib.sha256(CS.buckets(ym).tobytes()).digest() == g[f"{name}__sha_by"].tobytes()


PLANAR_KINDS = {
    # the kernel family a bilinear PAIR of the chain must reach (a single image / a batch of three: mirror / batch likewise, or tile)
    "apply_rectilinear": "mirror", "apply_stereographic": "mirror", "apply_equidistant": "mirror", "transformer_poly": "mirror",
    "planar_nonsquare": "mirror",
}


@pytest.mark.parametrize("interp", [1, 4, 0, 2])
@pytest.mark.parametrize("name", list(CS.PLANAR_CASES))
def test_planar_and_general_modes_pixels_on_the_tile_kernels(V, oracle_mod, dev, name, interp):
    """The reference's own test chains (tests/test_remapper.py:42-91) and their relatives at 1024 x 1024 on the LDS-tiled kernels: a
    pair, a single image and a batch of three, every byte against the oracle (fp64 chain stage by stage + cv2.remap restated) --
    including the NaN rim of FisheyeEncoder("orthographic") (border colour) -- and the kernel family asserted: never the generic kernel."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec, (wo, ho), (hs, ws), radius = CS.PLANAR_CASES[name]
    imgs = [_noise(hs, ws, 900 + k) for k in range(3)]
    for im in imgs:
        im[:3], im[-3:], im[:, :3], im[:, -3:] = 0, 0, 0, 0
    xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    want = [O.remap(im, xm, ym, interp, 0, (4, 5, 6)) for im in imgs]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    t = CS.to_product(spec)
    for group in (srcs[:2], srcs[:1], srcs):
        dsts = [torch.full((ho, wo, 3), 99, dtype=torch.uint8, device=dev) for _ in group]
        paths = V.remap_tensors(t, group, dsts, radius=radius, interpolation=interp, boarder_value=(4, 5, 6))
        assert paths in (["planar"], ["ray"]), paths
        kind = remapper.last_launch_kinds()[0].split("+")[0]
        assert kind in ("tile", "mirror", "batch"), (name, interp, len(group), kind)
        if interp == 1 and len(group) == 2 and name in PLANAR_KINDS:
            assert kind == PLANAR_KINDS[name], (name, kind)
        for k, d in enumerate(dsts):
            got = d.cpu().numpy()
            assert np.array_equal(got, want[k]), (name, interp, len(group), k, int((got != want[k]).sum()))


def test_the_references_ten_test_calls_as_written(V, oracle_mod, tmp_path):
    """tests/test_remapper.py:42-109 of the reference, call for call -- apply() of one file with six encoders * FisheyeDecoder and with the
    rotator / polynomial between equidistant encoder and decoder, apply_lr() with left_path == right_path (the split-in-halves branch)
    -- with their arguments (256 x 256 outputs, the default Lanczos4 and BORDER_CONSTANT, radius="max"), which the reference only
    smoke-tests: here every output byte is compared with the oracle's and the launch must have been a tile kernel."""
    from vr180_convert_amd import _io, remapper
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.quat import from_euler_angles
    from vr180_convert_amd.synth import pattern

    O = oracle_mod
    img = pattern(256, 256)
    path = tmp_path / "test.png"
    _io.imwrite(path, img)
    img = _io.imread(path)
    rot = ("rot", CS.ry(np.pi / 4))
    calls = []
    for fmt in ["rectilinear", "stereographic", "equidistant", "equisolid", "orthographic"]:
        calls.append((T.FisheyeEncoder(fmt) * T.FisheyeDecoder("equidistant"), [("fisheye_enc", fmt), CS.EQUI]))
    calls.append((T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant"), [("equirect_enc", True), CS.EQUI]))
    for tr, sp in ((T.Euclidean3DRotator(from_euler_angles(0.0, np.pi / 4, 0.0)), rot), (T.PolynomialScaler([0, 1, -0.1]), ("poly", [0, 1, -0.1]))):
        calls.append((T.FisheyeEncoder("equidistant") * tr * T.FisheyeDecoder("equidistant"), [("fisheye_enc", "equidistant"), sp, CS.EQUI]))
    for k, (t, spec) in enumerate(calls):
        out = tmp_path / f"test.{k}.png"
        got = V.apply(t, in_paths=path, out_paths=out, radius="max", size_output=(256, 256))[0]
        kinds = [x.split("+")[0] for x in remapper.last_launch_kinds()]
        assert kinds and all(x in ("tile", "mirror", "batch") for x in kinds), (spec, kinds)
        want = O.apply(spec, [img], size_output=(256, 256), interpolation=4, radius="max")[0]
        assert np.array_equal(got, want), (spec, int((got != want).sum()))
        assert np.array_equal(_io.imread(out), want)
    for tr, sp in ((T.Euclidean3DRotator(from_euler_angles(0.0, np.pi / 4, 0.0)), rot), (T.PolynomialScaler(), ("poly", [0, 1]))):
        t = T.EquirectangularEncoder() * tr * T.FisheyeDecoder("equidistant")
        spec = [("equirect_enc", True), sp, CS.EQUI]
        out = tmp_path / f"test.lr.{tr.__class__.__name__}.png"
        V.apply_lr(t, left_path=path, right_path=path, out_path=out, radius="max", size_output=(256, 256))
        kinds = [x.split("+")[0] for x in remapper.last_launch_kinds()]
        assert kinds and all(x in ("tile", "mirror", "batch") for x in kinds), (spec, kinds)
        halves = [np.ascontiguousarray(img[:, :128]), np.ascontiguousarray(img[:, 128:])]
        want = O.apply_lr(spec, halves[0], halves[1], size_output=(256, 256), interpolation=4, radius="max")
        assert np.array_equal(_io.imread(out), want), spec


def test_row_bands_of_planar_and_general_mode_chains(V, oracle_mod):
    """A single pair on 4 GPUs = bands of output rows, for the chains that are fused since round 5: the band's plan
    learns the whole grid's rows from its Normalize stage (a planar chain's radial table covers m up to the corners of the WHOLE
    output), so the assembled rows equal the oracle's byte for byte -- four workers on the one card of the test box."""
    O = oracle_mod
    left, right = _noise(300, 300, 41), _noise(300, 300, 42)
    for spec in ([("fisheye_enc", "stereographic"), ("poly", [0, 1, -0.1]), CS.EQUI], [("fisheye_enc", "equidistant"), ("rot", CS.ry(0.3)), CS.EQUI],
                 [("equirect_enc", False), CS.EQUI]):
        for interp in (1, 4):
            want = O.apply_lr(spec, left, right, size_output=(512, 448), interpolation=interp, radius="max")
            got = V.remap_sharded(CS.to_product(spec), [(left, right)], size_output=(512, 448), interpolation=interp, radius="max", devices=[0] * 4)
            assert np.array_equal(got[0], want), (spec, interp, int((got[0] != want).sum()))


# ---------------------------------------------------------------------------- radius="auto" on the device
def _disc(h, w, r, seed, cx=None, cy=None):
    """A noisy image circle of radius r on black (what get_radius looks for: transformer.py:125-140)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    cx, cy = (w / 2 if cx is None else cx), (h / 2 if cy is None else cy)
    img = rng.integers(40, 256, (h, w, 3), dtype=np.uint8)
    img[(xx - cx) ** 2 + (yy - cy) ** 2 > r * r] = 0
    return img


@pytest.mark.parametrize("interp", [4, 1])
def test_auto_radius_on_the_device_equals_the_host_radius_path(V, oracle_mod, golden_dir, dev, interp):
    """apply_lr_tensors(radius="auto", auto_radius_on_device=True): the estimate of every image stays on the device (v1c_get_radius_async),
    a one-thread kernel takes the maximum and sets the Denormalize scale, the launch reduces its boxes itself -- no synchronisation, ONE
    plan whatever the radius.  Bytes equal to the exact path (radius brought to the host: a plan per radius) and to the oracle, for the
    reference's own get_radius fixtures (tests/golden/radius.npz: the NEGATIVE radius a clean disc yields -- the 180-degree flip
    quirk --, speckle around the threshold) and discs of several radii through one and the same plan."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    g = np.load(golden_dir / "radius.npz")
    t = CS.to_product([("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI])
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    pairs = [(g["landscape_img"], g["noisy_img"]), (g["noisy_img"], g["landscape_img"])]
    pairs += [(_disc(480, 640, r0, 3), _disc(480, 640, r1, 4, cx=300)) for r0, r1 in ((200, 231), (150.5, 120), (239, 238))]
    n_plans = None
    for k, (a, b) in enumerate(pairs):
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        r_ref = max(O.get_radius(a), O.get_radius(b))  # get_radius_smart("auto"), remapper.py:83-84
        la, lb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        got = V.apply_lr_tensors(t, la, lb, size_output=(512, 512), interpolation=interp, radius="auto", auto_radius_on_device=True)
        kinds = remapper.last_launch_kinds()
        assert remapper.last_auto_radius_form() == "device" and kinds and kinds[0] in ("tile", "rot_pair"), (remapper.last_auto_radius_form(), kinds)
        exact = V.apply_lr_tensors(t, la, lb, size_output=(512, 512), interpolation=interp, radius="auto", auto_radius_on_device=False)
        want = O.apply_lr(spec, a, b, size_output=(512, 512), interpolation=interp, radius=r_ref)
        assert np.array_equal(exact.cpu().numpy(), want), (k, "exact path")
        assert np.array_equal(got.cpu().numpy(), want), (k, r_ref, int((got.cpu().numpy() != want).sum()))
    # the reference raises IndexError for an image without a black border (radius.npz: noborder_raises); the exact path does too,
    # the device-resident one cannot -- the radius becomes NaN and the output the border colour
    full = torch.full((480, 640, 3), 90, dtype=torch.uint8, device=dev)
    disc = torch.from_numpy(_disc(480, 640, 200, 9)).to(dev)
    with pytest.raises(IndexError):
        V.apply_lr_tensors(t, full, disc, size_output=(512, 512), interpolation=interp, radius="auto", auto_radius_on_device=False)
    out = V.apply_lr_tensors(t, full, disc, size_output=(512, 512), interpolation=interp, radius="auto", auto_radius_on_device=True,
                             boarder_value=(1, 2, 3))
    assert remapper.last_auto_radius_form() == "device"
    assert torch.equal(out, torch.tensor([1, 2, 3], dtype=torch.uint8, device=dev).expand_as(out))
    # per-eye transformers: per-eye radius (remapper.py:460-473)
    tl = CS.to_product([("equirect_enc", True), ("rot", CS.ry(0.05)), CS.EQUI])
    a, b = _disc(480, 640, 180, 5), _disc(480, 640, 222, 6)
    got = V.apply_lr_tensors((tl, t), torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), size_output=(512, 512), interpolation=interp,
                             radius="auto", auto_radius_on_device=True)
    assert remapper.last_auto_radius_form() == "device"
    # a non-square output looks beyond the front hemisphere: its table needs a fix-up pass, the exact form serves it (silently)
    V.apply_lr_tensors(t, torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), size_output=(640, 512), interpolation=interp, radius="auto",
                       auto_radius_on_device=True)
    assert remapper.last_auto_radius_form() == "exact"
    want = O.apply_lr(([("equirect_enc", True), ("rot", CS.ry(0.05)), CS.EQUI], spec), a, b, size_output=(512, 512), interpolation=interp, radius="auto")
    assert np.array_equal(got.cpu().numpy(), want)


def test_auto_radius_estimated_by_the_launch_itself_equals_the_two_step_form(V, oracle_mod, dev):
    """v1c_plan_run_auto_images (one workgroup scans every source's centre line and sets the scale: two launches per call) against
    v1c_get_radius_async per image + v1c_plan_run_auto, and the oracle: centre COLUMN of square sources, centre ROW of landscape ones,
    the two eyes as halves of one side-by-side tensor (a pitch of two rows) and -- a pitch each -- one half beside a contiguous image."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    t = CS.to_product(spec)
    for (h, w), (r0, r1) in (((512, 512), (250, 231.5)), ((480, 640), (200, 236)), ((640, 480), (225, 239))):
        a, b = _disc(h, w, r0, 11), _disc(h, w, r1, 12, cx=w / 2 - 7)
        sbs = torch.from_numpy(np.concatenate([a, b], axis=1)).to(dev)
        want = O.apply_lr(spec, a, b, size_output=(384, 384), interpolation=1, radius="auto")
        for srcs in ([sbs[:, :w], sbs[:, w:]], [sbs[:, :w], torch.from_numpy(b).to(dev)]):
            outs = []
            for two_step in (False, True):
                out = torch.zeros((384, 768, 3), dtype=torch.uint8, device=dev)
                rad = remapper.auto_radius_tensor(srcs) if two_step else None
                remapper.remap_tensors_auto(t, srcs, [out[:, :384], out[:, 384:]], rad=rad, interpolation=1, size_input=(h, w))
                outs.append(out.cpu().numpy())
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], want), ((h, w), srcs[1].stride(0))


def test_exact_auto_radius_takes_no_plan_per_radius(V, oracle_mod, dev):
    """The exact form of radius="auto" (estimates to the host: IndexError like the reference) on a stream whose image circle moves: every
    new radius is served by the launch that reads it from device memory (ONE plan), bytes equal to the oracle's apply_lr; a circle that
    repeats gets a plan of its own -- the planned kernels -- on its second call."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    t = CS.to_product(spec)
    remapper.clear_caches()
    n_plans = []
    pairs = [(_disc(512, 512, r0, 20 + k), _disc(512, 512, r1, 30 + k)) for k, (r0, r1) in enumerate(((250, 240), (231.5, 244), (199, 180.5)))]
    for a, b in pairs + pairs[-1:]:
        got = V.apply_lr_tensors(t, torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), size_output=(512, 512), interpolation=1,
                                 radius="auto", auto_radius_on_device=False)
        assert remapper.last_auto_radius_form() == "exact"
        n_plans.append((len(remapper._PLANS), remapper.last_launch_kinds()[0]))
        want = O.apply_lr(spec, a, b, size_output=(512, 512), interpolation=1, radius="auto")
        assert np.array_equal(got.cpu().numpy(), want), n_plans
    assert [n for n, _ in n_plans] == [1, 1, 1, 2], n_plans
    assert all(k in ("rot_pair", "tile") for _, k in n_plans[:3]) and n_plans[3][1] in ("mirror", "tile"), n_plans
    with pytest.raises(IndexError):
        V.apply_lr_tensors(t, torch.full((512, 512, 3), 90, dtype=torch.uint8, device=dev), torch.from_numpy(pairs[0][0]).to(dev),
                           size_output=(512, 512), interpolation=1, radius="auto", auto_radius_on_device=False)


@pytest.mark.parametrize("interp", [1, 4])
def test_auto_radius_of_one_three_and_four_images(V, oracle_mod, dev, interp):
    """remap_tensors_auto with 1, 3 and 4 images of one call (apply() of a list, remapper.py:379-398: ONE radius = the maximum of the
    images' estimates, one map): the scan launch walks every image's centre line, odd counts leave the pair kernels."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.05]), CS.EQUI]
    t = CS.to_product(spec)
    imgs = [_disc(400, 400, r, 40 + k) for k, r in enumerate((180.5, 190, 150, 192))]
    for n in (1, 3, 4):
        srcs = [torch.from_numpy(im).to(dev) for im in imgs[:n]]
        dsts = [torch.zeros((448, 448, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
        remapper.remap_tensors_auto(t, srcs, dsts, interpolation=interp)
        r_ref = max(O.get_radius(im) for im in imgs[:n])
        xm, ym = O.get_map(spec, radius=r_ref, size_input=(400, 400), size_output=(448, 448))
        for k in range(n):
            want = O.remap(imgs[k], xm, ym, interp, 0, 0)
            assert np.array_equal(dsts[k].cpu().numpy(), want), (n, k, r_ref)


def test_apply_lr_auto_radius_is_graph_capturable_end_to_end(V, oracle_mod, dev):
    """radius="auto" -- the reference's default -- recorded into a graph: estimate, maximum, scale and remap are four launches and no
    synchronisation.  The graph is replayed on NEW pixels with ANOTHER image circle in the same buffers: the radius follows the image."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec = [("equirect_enc", True), CS.EQUI]
    t = CS.to_product(spec)
    imgs = [(_disc(512, 512, 250, 1), _disc(512, 512, 240, 2)), (_disc(512, 512, 199.5, 3), _disc(512, 512, 221, 4))]
    left, right = (torch.from_numpy(x).to(dev) for x in imgs[0])
    out = torch.zeros((512, 1024, 3), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    # (the plan is created, and its first launch made, on ANOTHER stream than the one that records: the recorded launch must not wait
    #  for an event of that stream)
    V.apply_lr_tensors(t, left, right, out=out, size_output=(512, 512), radius="auto", auto_radius_on_device=True)
    torch.cuda.synchronize()
    n_plans = len(remapper._PLANS)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        V.apply_lr_tensors(t, left, right, out=out, size_output=(512, 512), radius="auto")  # (capturing: the device-resident form by itself)
    for a, b in imgs[::-1] + imgs:
        left.copy_(torch.from_numpy(a).to(dev)), right.copy_(torch.from_numpy(b).to(dev))
        out.zero_()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = O.apply_lr(spec, a, b, size_output=(512, 512), interpolation=4, radius="auto")
        assert np.array_equal(out.cpu().numpy(), want)
    assert len(remapper._PLANS) == n_plans  # one plan served every radius


def test_unit_ring_from_six_threads_without_syncs(V, oracle_mod, dev):
    """One plan, six threads with a stream each, eight back-to-back launches of 20 units per thread with nothing synchronised in
    between: up to 48 launches queue up against the ring's four slots.  A slot's bookkeeping is only complete once the event behind the
    launch that reads it is recorded; the ring mutex is held until then (plan.hip: ring_put), so no launch can see a slot another
    thread has picked but not yet published.  Every output against the oracle."""
    import threading

    from vr180_convert_amd import transformer as T

    O = oracle_mod
    n, size, n_threads, n_launches = 20, 128, 6, 8
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    imgs = [_noise(size, size, 40 + f) for f in range(n)]
    for im in imgs:  # a black rim like a fisheye frame: keeps the chain's border pixels simple
        im[:2], im[-2:], im[:, :2], im[:, -2:] = 0, 0, 0, 0
    quat_sets = [[CS.c5_spec(11 * k + f // 2, f % 2)[1][1] for f in range(n)] for k in range(n_threads)]
    wants = [[O.apply(CS.c5_spec(11 * k + f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=1, radius=size / 2)[0]
              for f in range(n)] for k in range(n_threads)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    torch.cuda.synchronize()
    errs: list = []
    start = threading.Barrier(n_threads)

    def work(k: int) -> None:
        try:
            torch.cuda.set_device(dev)
            s = torch.cuda.Stream(device=dev)
            outs = []
            with torch.cuda.stream(s):
                all_dsts = [[torch.zeros_like(x) for x in srcs] for _ in range(n_launches)]
                s.synchronize()
                start.wait()
                for it in range(n_launches):
                    V.remap_tensors(base, srcs, all_dsts[it], radius=size / 2, interpolation=1, rotations=quat_sets[k])
                s.synchronize()
                outs = [[d.cpu().numpy() for d in dsts] for dsts in all_dsts]
            for it in range(n_launches):
                for f in range(n):
                    if not np.array_equal(outs[it][f], wants[k][f]):
                        errs.append((k, it, f))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:8]


def test_capture_slots_can_be_released(V, dev):
    """A graph-captured launch of more than 16 units keeps one of the plan's four capture-owned unit buffers; v1c_plan_release_captures
    hands them out again once the graphs are gone (advisor finding of round 4: a process that re-captures over and over)."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd import transformer as T

    n, size = 18, 96
    t = T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant")
    srcs = [torch.from_numpy(_noise(size, size, f)).to(dev) for f in range(n)]
    dsts = [torch.zeros_like(s) for s in srcs]
    V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)  # (plan creation outside any capture)
    torch.cuda.synchronize()
    ref = [d.clone() for d in dsts]
    plan = remapper._TLS.plans[0]

    def capture_once():
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)
        return g

    for round_ in range(3):
        graphs = [capture_once() for _ in range(4)]  # (12 captures over the three rounds: the 5th would be refused without the release)
        for d in dsts:
            d.zero_()
        graphs[round_].replay()
        torch.cuda.synchronize()
        for d, r in zip(dsts, ref):
            assert torch.equal(d, r)
        del graphs
        plan.release_captures()


@pytest.mark.parametrize("workload,kinds", [("P1", {"mirror"}), ("P2", {"tile+fixup", "tile"}), ("P3L", {"tile"}), ("C0", {"tile"})])
def test_bench_lines_of_the_round5_workloads(workload, kinds):
    """bench.py's workloads for the chains of the reference's tests (P*) and its test size (C0): the line names a tiled kernel family,
    its own parity check against the oracle finds no differing byte, and a Lanczos4 line measures its VALU ceiling live or says it did not."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "1", "--traffic", "none",
                        "--no-cold-extra", "--no-sustained"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert set(line["config"]["kernels"]) <= kinds, line["config"]["kernels"]
    assert line["parity_vs_oracle"]["bytes_differing"] == 0, line["parity_vs_oracle"]
    assert line["world_size_seen"] == 1
    if workload in ("P3L", "C0"):
        assert line["roofline"]["bound"] == "valu-int" and "source" in line["roofline"]["valu_int"]


def test_lane_whose_second_pixel_is_outside_the_pre_tables(V, oracle_mod, dev):
    """tools/fuzz.py --seed 34, case 1974 (round 5): a Zoom in front of two rotations (general mode 2), 154 x 1427 from 192 x 192.  In row
    872 the base variable of columns <= 93 lies outside the S / Cm tables (fix-up pass), columns 94 and 95 are good pixels of the same
    lane: pixel 1's arbitrary table index pointed one entry beside the tile's LDS slice, the clamped read returned the neighbouring
    entry, and the two good pixels evaluated it about the wrong centre -- 0.4 px off.  Bytes against the oracle, every interpolation."""
    from vr180_convert_amd import remapper

    O = oracle_mod
    spec = [("equirect_enc", True), ("zoom", 1.7738941798002221),
            ("rot", [[0.9982728060815481, 0.058690132718145105, 0.002621633002223754], [-0.05872470490000005, 0.9981458957992685, 0.01600561568587129],
                     [-0.0016774005126221434, -0.016131925508209265, 0.9998684650027312]]),
            ("rot", [[0.9999105236968615, -0.00792577709040616, 0.010776207949987819], [0.007884260865708426, 0.999961353887725, 0.003889622299068912],
                     [-0.010806619770753778, -0.003804311835424161, 0.999934369936642]]), ("fisheye_dec", "equidistant")]
    wo, ho, ws, hs, radius = 154, 1427, 192, 192, 96.0
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
    fill = rng.integers(0, 256, (ho, wo, 3), dtype=np.uint8)
    xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
    t = CS.to_product(spec)
    for interp, border in ((1, 0), (2, 5), (4, 0), (0, 0)):
        want = O.remap(img, xm, ym, interp, border, (1, 2, 3), dst=fill.copy())
        dst = torch.from_numpy(fill.copy()).to(dev)
        V.remap_tensors(t, [torch.from_numpy(img).to(dev)], [dst], radius=radius, interpolation=interp, boarder_mode=border, boarder_value=(1, 2, 3))
        assert remapper.last_launch_kinds()[0].startswith("tile"), remapper.last_launch_kinds()
        d = np.argwhere((dst.cpu().numpy() != want).any(axis=2))
        assert len(d) == 0, (interp, border, d[:4].tolist())
