"""Parity tests proper: the HIP path (through the C ABI) against the oracle and the reference
goldens on a real MI355X.  Bar: coordinates land in identical cv2 1/32-pixel buckets as the
reference's float32 maps; pixels are BIT-EXACT against the oracle (stricter than the north star's
+-1 per uint8 channel)."""
import hashlib
import math

import numpy as np
import pytest
import torch

import chainspecs as CS
from test_oracle_golden import assert_maps_match

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


def dmap(V, spec, out, inp, radius):
    return V.get_map(CS.to_product(spec), radius=radius, size_input=inp, size_output=out)


# ---------------------------------------------------------------------------- coordinates
@pytest.mark.parametrize("name", list(CS.SMALL_CASES))
def test_device_maps_vs_reference_goldens(V, golden_dir, name):
    g = np.load(golden_dir / "maps_small.npz")
    spec, out, inp, radius = CS.SMALL_CASES[name]
    xm, ym = dmap(V, spec, out, inp, radius)
    assert_maps_match(xm, ym, g[f"{name}__x"], g[f"{name}__y"], name)


@pytest.mark.parametrize("name", list(CS.FULL_CASES))
def test_device_maps_full_size_sha(V, golden_dir, name):
    g = np.load(golden_dir / "maps_full.npz")
    spec, out, inp, radius = CS.FULL_CASES[name]
    xm, ym = dmap(V, spec, out, inp, radius)
    s = CS.FULL_STRIDE
    assert_maps_match(xm[::s], ym[::s], g[f"{name}__rows_x"], g[f"{name}__rows_y"], name + " rows")
    assert hashlib.sha256(CS.buckets(xm).tobytes()).digest() == g[f"{name}__sha_bx"].tobytes()
    assert hashlib.sha256(CS.buckets(ym).tobytes()).digest() == g[f"{name}__sha_by"].tobytes()


def test_device_maps_c5_per_unit_rotation(V, golden_dir, dev):
    from vr180_convert_amd.chain import lower_for_get_map
    from vr180_convert_amd.remapper import _plan_for, _split_single_rotation

    g = np.load(golden_dir / "maps_c5.npz")
    plans = set()
    for frame in (0, 1, 7):
        for eye in (0, 1):
            ch = lower_for_get_map(CS.to_product(CS.c5_spec(frame, eye)), radius=96.0, size_input=(192, 192), size_output=(192, 192))
            shared, rot = _split_single_rotation(ch)
            plan = _plan_for(shared, src_hw=(192, 192), dst_wh=(192, 192), cn=3, interpolation=1, border_mode=0,
                             border_value=0, device=dev)
            plans.add(id(plan))
            assert plan.path == "ray"
            xm, ym = plan.get_map(rot)
            assert_maps_match(xm.cpu().numpy(), ym.cpu().numpy(), g[f"f{frame}_e{eye}__x"], g[f"f{frame}_e{eye}__y"], f"c5 {frame} {eye}")
    assert len(plans) == 1  # one plan serves every calibration rotation


# ---------------------------------------------------------------------------- sampler (LUT entry)
@pytest.mark.parametrize("cn", [1, 3, 4])
def test_remap_lut_all_modes_vs_oracle(V, oracle_mod, dev, cn):
    import ctypes as C

    from vr180_convert_amd import _native
    from vr180_convert_amd.remapper import border_scalar

    O = oracle_mod
    rng = np.random.default_rng(1)
    Hs, Ws, H, W = 61, 83, 70, 90
    src = rng.integers(0, 256, (Hs, Ws, cn), dtype=np.uint8)
    xm = (rng.random((H, W)) * (Ws + 24) - 12).astype(np.float32)
    ym = (rng.random((H, W)) * (Hs + 24) - 12).astype(np.float32)
    xm[0, :5] = [np.nan, np.inf, -np.inf, 1e30, -1e30]
    ym[1, :3] = [np.nan, 3e9, -3e9]
    xm[2, :8] = np.arange(8)
    ym[2, :8] = np.arange(8)
    xm[3, :4] = [0.5 / 32, 1.5 / 32, 2.5 / 32, -0.5 / 32]
    s_d, x_d, y_d = (torch.from_numpy(a).to(dev) for a in (src, xm, ym))
    for interp in (0, 1, 2, 3, 4):
        for border in range(6):
            for bv in (0, (10, 200, 30, 77)):
                ref = np.full((H, W, cn), 123, np.uint8)
                O.remap(src, xm, ym, interp, border, bv, dst=ref)
                out = torch.full((H, W, cn), 123, dtype=torch.uint8, device=dev)
                cv = border_scalar(bv)
                rc = _native.lib().v1c_remap_lut(0, None, s_d.data_ptr(), Hs, Ws, s_d.stride(0), cn, out.data_ptr(), H, W,
                                                 out.stride(0), x_d.data_ptr(), y_d.data_ptr(), W * 4, interp, border, cv.ctypes.data)
                assert rc == 0, _native.lib().v1c_last_error()
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), ref), (cn, interp, border, bv)


@pytest.mark.parametrize("hw", [(101, 1), (1, 57), (1, 1), (2, 2), (3, 1)])
def test_sources_one_pixel_wide_or_high(V, oracle_mod, dev, hw):
    """Degenerate sources through the fused path and through cv2.remap alone: a source ONE pixel wide made the bilinear fast path's
    unsigned bound wrap (reads far outside the image: a GPU memory fault, found by tools/fuzz.py in round 4; the host build of the
    sampler runs under AddressSanitizer in tests/test_sampler_asan.py)."""
    import ctypes as C

    from vr180_convert_amd import _native
    from vr180_convert_amd.remapper import border_scalar

    O = oracle_mod
    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    hs, ws = hw
    spec = [("equirect_enc", True), CS.EQUI]
    for cn in (3, 1, 4):
        src = rng.integers(0, 256, (hs, ws, cn), dtype=np.uint8)
        s_d = torch.from_numpy(src).to(dev)
        H, W = 40, 72
        xm = rng.uniform(-6, ws + 6, (H, W)).astype(np.float32)
        ym = rng.uniform(-6, hs + 6, (H, W)).astype(np.float32)
        x_d, y_d = torch.from_numpy(xm).to(dev), torch.from_numpy(ym).to(dev)
        for interp in (0, 1, 2, 3, 4):
            for border in range(6):
                fill = rng.integers(0, 256, (H, W, cn), dtype=np.uint8)
                want = O.remap(src, xm, ym, interp, border, (9, 8, 7, 6), dst=fill.copy())
                out = torch.from_numpy(fill.copy()).to(dev)
                rc = _native.lib().v1c_remap_lut(0, None, s_d.data_ptr(), hs, ws, s_d.stride(0), cn, out.data_ptr(), H, W, out.stride(0),
                                                 x_d.data_ptr(), y_d.data_ptr(), W * 4, interp, border, border_scalar((9, 8, 7, 6)).ctypes.data)
                assert rc == 0, _native.lib().v1c_last_error()
                assert np.array_equal(out.cpu().numpy(), want), ("lut", hw, cn, interp, border)
                if interp == 3:
                    continue
                dst = torch.from_numpy(fill.copy()).to(dev)
                V.remap_tensors(CS.to_product(spec), [s_d], [dst], radius=max(hs, ws) / 2, interpolation=interp, boarder_mode=border,
                                boarder_value=(9, 8, 7, 6))
                xo, yo = O.get_map(spec, radius=max(hs, ws) / 2, size_input=(hs, ws), size_output=(W, H))
                want = O.remap(src, xo, yo, interp, border, (9, 8, 7, 6), dst=fill.copy())
                assert np.array_equal(dst.cpu().numpy(), want), ("chain", hw, cn, interp, border)


def test_views_of_buffers_beyond_4_gib_take_the_64_bit_kernels(V, oracle_mod, dev):
    """The tile kernels address source and destination with 32-bit offsets; the host routes units whose rows end beyond 2^32 - 256 bytes
    (or whose pitch is 2^24 or more) to the generic kernel with 64-bit offsets (decide_launch, csrc/plan.hip).  Sources and destinations
    that are column slices of 4.6 GB / 5.1 GB buffers -- a pitch of 2.3 MB over 2 000 rows, a pitch of 17 MB over 300 -- against the
    oracle, pairs and batches, and the same call on small buffers takes the tile kernels."""
    from vr180_convert_amd import remapper

    if torch.cuda.mem_get_info(0)[0] < 24 << 30:
        pytest.skip("needs 24 GB of free device memory")
    O = oracle_mod
    rng = np.random.default_rng(99)
    spec = [("equirect_enc", True), CS.EQUI]
    t = CS.to_product(spec)
    for (rows, pitch, ws, out) in ((2000, 2_300_000, 1504, 640), (300, 17_000_000, 400, 512)):
        big_src = torch.empty((rows, pitch), dtype=torch.uint8, device=dev)
        big_dst = torch.empty((out, 8_000_000 if pitch < (1 << 24) else pitch), dtype=torch.uint8, device=dev)
        assert big_src.numel() > 1 << 32 and big_dst.numel() > 1 << 32
        imgs = [rng.integers(0, 256, (rows, ws, 3), dtype=np.uint8) for _ in range(2)]
        off = big_src.shape[1] - 2 * 3 * ws - 64  # the views sit at the END of the rows: byte offsets of the last rows exceed 2^32
        srcs = []
        for k, im in enumerate(imgs):
            v = big_src[:, off + k * 3 * ws: off + (k + 1) * 3 * ws].unflatten(1, (ws, 3))
            v.copy_(torch.from_numpy(im).to(dev))
            srcs.append(v)
        doff = big_dst.shape[1] - 2 * 3 * out - 128
        dsts = [big_dst[:, doff + k * 3 * out: doff + (k + 1) * 3 * out].unflatten(1, (out, 3)) for k in range(2)]
        assert srcs[1].data_ptr() + (rows - 1) * srcs[1].stride(0) - big_src.data_ptr() > 1 << 32
        xm, ym = O.get_map(spec, radius=min(rows, ws) / 2, size_input=(rows, ws), size_output=(out, out))
        for interp in (1, 4):
            for d in dsts:
                d.zero_()
            assert V.remap_tensors(t, srcs, dsts, radius=min(rows, ws) / 2, interpolation=interp) == ["ray"]
            assert remapper.last_launch_kinds() == ["generic"], remapper.last_launch_kinds()
            for k in range(2):
                want = O.remap(imgs[k], xm, ym, interp, 0, 0)
                assert np.array_equal(dsts[k].cpu().numpy(), want), (rows, pitch, interp, k)
        # the same units on buffers of their own: the tile kernels
        small = [torch.from_numpy(im).to(dev) for im in imgs]
        outs = [torch.empty((out, out, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        V.remap_tensors(t, small, outs, radius=min(rows, ws) / 2, interpolation=1)
        assert remapper.last_launch_kinds()[0] in ("mirror", "tile"), remapper.last_launch_kinds()
        for k in range(2):
            assert np.array_equal(outs[k].cpu().numpy(), O.remap(imgs[k], xm, ym, 1, 0, 0)), k
        del big_src, big_dst
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------- fused path vs oracle
CUTS = {
    # 1024^2 cuts of every BASELINE config (SURVEY.md 8d "Parity gate") + C1 in full
    "C1_full": (CS.FULL_CASES["C1"][0], 2048, 1, "ray"),
    "C2_cut": (CS.FULL_CASES["C2"][0], 1024, 1, "ray"),
    "C3_cut": (CS.FULL_CASES["C3"][0], 1024, 1, "ray"),
    "C4_cut_lanczos": (CS.FULL_CASES["C4"][0], 1024, 4, "ray"),
    "C4_cut_cubic": (CS.FULL_CASES["C4"][0], 512, 2, "ray"),
    "C5_cut": (CS.c5_spec(3, 1), 1024, 1, "ray"),
    "nearest": (CS.FULL_CASES["C2"][0], 512, 0, "ray"),
    "fixup_back_hemisphere": (CS.SMALL_CASES["back_hemisphere"][0], 512, 1, "ray"),
    "fixup_poly_c0": (CS.SMALL_CASES["poly_c0"][0], 512, 1, "ray"),
    # (fused since round 5: planar chains, a rotation behind radial stages -- tests/test_gpu_round5.py has them at full size)
    "planar_fisheye_to_fisheye_rotated": (CS.SMALL_CASES["transformer_rotator"][0], 512, 4, "planar"),
    "rot_after_radial": (CS.SMALL_CASES["rot_after_radial"][0], 512, 1, "ray"),
    "planar_orthographic_nan": (CS.SMALL_CASES["apply_orthographic"][0], 512, 1, "planar"),
    # (still the fp64 interpreter: decoders, i.e. chains that do not end in a radial composite)
    "literal_equirect_decoder": (CS.SMALL_CASES["equirect_decoder"][0], 512, 1, "literal"),
}


@pytest.mark.parametrize("name", list(CUTS))
def test_fused_apply_bit_exact_vs_oracle(V, oracle_mod, dev, name):
    from vr180_convert_amd.synth import noise_disc

    spec, size, interp, want_path = CUTS[name]
    img = noise_disc(size, size, frame=7)
    want = oracle_mod.apply(spec, [img], size_output=(size, size), interpolation=interp, radius="max")[0]
    src = torch.from_numpy(img).to(dev)
    dst = torch.empty_like(src)
    paths = V.remap_tensors(CS.to_product(spec), [src], [dst], radius=size / 2, interpolation=interp)
    torch.cuda.synchronize()
    assert paths == [want_path]
    got = dst.cpu().numpy()
    nd = int((got != want).sum())
    assert nd == 0, f"{name}: {nd} bytes differ, max |d| = {int(np.abs(got.astype(int) - want).max())}"


@pytest.mark.parametrize("border", [1, 2, 3, 4])
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_tile_kernels_other_border_modes(V, oracle_mod, dev, interp, border):
    """REPLICATE / REFLECT / WRAP / REFLECT_101 run the tile kernels too (the border only matters to
    pixels whose footprint leaves the source): a zoomed-out chain puts a third of the output outside.
    INTER_NEAREST (0) rides the bilinear tile kernels with coordinates 32 * cvRound(x) (lane_coords<..., NN = 1>)."""
    from vr180_convert_amd.synth import noise_disc

    size = 200
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), ("zoom", 0.6), CS.EQUI]
    imgs = [noise_disc(size, size, 30 + f) for f in range(3)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.empty((231, 277, 3), dtype=torch.uint8, device=dev) for _ in imgs]
    paths = V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=size / 2, interpolation=interp, boarder_mode=border)
    assert paths == ["ray"]
    want = oracle_mod.apply(spec, imgs, size_output=(277, 231), interpolation=interp, border_mode=border, radius=size / 2)
    for d, w in zip(dsts, want):
        assert np.array_equal(d.cpu().numpy(), w), (interp, border, int((d.cpu().numpy() != w).sum()))


def test_border_transparent_bilinear_through_the_tile_kernels(V, oracle_mod, dev):
    """BORDER_TRANSPARENT runs the tile kernels (INTER_LINEAR since round 3, bicubic / Lanczos4 since round 4): remapBilinear leaves
    every pixel whose 2 x 2 footprint is not fully inside the source untouched, remapBicubic / remapLanczos4 every pixel whose centre
    tap is outside (the others reflect their missing taps) -- exactly the pixels the tile kernels' patch path handles one by one, with
    a per-pixel store mask.  Destinations start from a pattern (not zeros): skipped pixels must keep it.  Pairs, a batch, per-unit
    rotations, zoomed-out chains with a wide skipped rim; NEAREST (remapNearest: the pixel itself outside) through the NN form's patch path."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n = 300
    imgs = [noise_disc(n, n, 80 + f) for f in range(4)]
    for im in imgs:
        im[:, :] = np.maximum(im, 1)
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    fill = np.full((288, 352, 3), (11, 22, 33), np.uint8)
    for spec in ([("equirect_enc", True), ("zoom", 0.55), CS.EQUI], [("equirect_enc", True), ("poly", [0, 1, -0.1]), ("zoom", 0.7), CS.EQUI],
                 [("equirect_enc", True), ("rot", CS.ry(0.4)), ("zoom", 0.8), CS.EQUI]):
        for interp in (1, 0, 4, 2):
            xm, ym = O.get_map(spec, radius=n / 2, size_input=(n, n), size_output=(352, 288))
            want = [O.remap(im, xm, ym, interp, 5, 0, dst=fill.copy()) for im in imgs]
            assert any((w == fill).all(axis=-1).any() for w in want) and any((w != fill).any() for w in want)
            for group in (srcs[:2], srcs):  # a pair, a batch
                dsts = [torch.from_numpy(fill.copy()).to(dev) for _ in group]
                assert V.remap_tensors(CS.to_product(spec), group, dsts, radius=n / 2, interpolation=interp, boarder_mode=5) == ["ray"]
                kind = remapper.last_launch_kinds()[0].split("+")[0]
                assert kind in ("tile", "batch", "mirror"), (kind, interp)
                for k, d in enumerate(dsts):
                    got = d.cpu().numpy()
                    assert np.array_equal(got, want[k]), (spec, interp, len(group), k, int((got != want[k]).sum()))
    # per-unit rotations
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.ZoomTransformer(0.7) * T.FisheyeDecoder("equidistant")
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(4)]
    dsts = [torch.from_numpy(fill.copy()).to(dev) for _ in range(4)]
    V.remap_tensors(base, srcs, dsts, radius=n / 2, interpolation=1, boarder_mode=5, rotations=quats)
    for f in range(4):
        sp = [("equirect_enc", True), CS.c5_spec(f // 2, f % 2)[1], ("zoom", 0.7), CS.EQUI]
        xm, ym = O.get_map(sp, radius=n / 2, size_input=(n, n), size_output=(352, 288))
        want = O.remap(imgs[f], xm, ym, 1, 5, 0, dst=fill.copy())
        assert np.array_equal(dsts[f].cpu().numpy(), want), f


def test_nearest_through_the_tile_kernels(V, oracle_mod, dev):
    """INTER_NEAREST on the fused path: pairs (apply_lr), a batch sharing one map, per-unit rotations, a pitched view and a
    half-pixel-exact identity map (ties go to even) -- every byte against the oracle's remapNearest."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n = 448
    left, right = noise_disc(n, n, 51), noise_disc(n, n, 52)
    for spec in ([("equirect_enc", True), CS.EQUI], [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI],
                 [("equirect_enc", True), ("rot", CS.ry(0.3)), ("poly", [0, 1, -0.1]), CS.EQUI]):
        for border, bval in ((0, (9, 8, 7)), (1, 0)):
            want = O.apply_lr(spec, left, right, size_output=(512, 384), interpolation=0, radius="max", border_mode=border, border_value=bval)
            got = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev), size_output=(512, 384),
                                     interpolation=0, radius="max", boarder_mode=border, boarder_value=bval).cpu().numpy()
            assert np.array_equal(got, want), (spec, border, int((got != want).sum()))
    # a batch of 5 units sharing the map (general batch loop), sources = halves of SBS frames (pitched views)
    frames = [noise_disc(200, 400, 60 + f) for f in range(3)]
    imgs = [fr[:, e * 200:(e + 1) * 200] for fr in frames for e in (0, 1)][:5]
    spec = [("equirect_enc", True), CS.EQUI]
    dev_frames = [torch.from_numpy(fr).to(dev) for fr in frames]
    srcs = [dev_frames[k // 2][:, (k % 2) * 200:(k % 2 + 1) * 200] for k in range(5)]
    dsts = [torch.empty((192, 256, 3), dtype=torch.uint8, device=dev) for _ in range(5)]
    assert V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=100.0, interpolation=0) == ["ray"]
    want = O.apply(spec, [np.ascontiguousarray(i) for i in imgs], size_output=(256, 192), interpolation=0, radius=100.0)
    for k in range(5):
        assert np.array_equal(dsts[k].cpu().numpy(), want[k]), k
    # per-unit calibration rotations (units that override the rotation)
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(4)]
    dsts = [torch.empty((200, 200, 3), dtype=torch.uint8, device=dev) for _ in range(4)]
    V.remap_tensors(base, srcs[:4], dsts, radius=100.0, interpolation=0, rotations=quats)
    for f in range(4):
        want = O.apply(CS.c5_spec(f // 2, f % 2), [np.ascontiguousarray(imgs[f])], size_output=(200, 200), interpolation=0, radius=100.0)[0]
        assert np.array_equal(dsts[f].cpu().numpy(), want), f


@pytest.mark.parametrize("interp", [1, 0, 2, 4])
def test_gray_and_bgra_bilinear_through_the_cn_tile_kernel(V, oracle_mod, dev, interp):
    """Grayscale and BGRA sources (cv2.remap takes whatever array the caller passes, remapper.py:388-398) run k_ray_lin_cn: INTER_LINEAR
    and (round 4) INTER_NEAREST, INTER_CUBIC, INTER_LANCZOS4: plain and rotated chains, every border mode (TRANSPARENT over a pre-filled
    destination, each interpolation's own skip rule), batches of 1 - 5 units sharing the map, units with a
    rotation of their own (boxes reduced in the kernel),
    odd output sizes, sources that are pitched views, a radius larger than the source (rays leaving it) -- every byte against the oracle."""
    from vr180_convert_amd.synth import noise_disc

    from vr180_convert_amd import remapper

    O = oracle_mod
    rng = np.random.default_rng(777 + interp)
    seen = set()
    specs = ([("equirect_enc", True), CS.EQUI], [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI],
             [("equirect_enc", True), ("rot", CS.ry(0.3)), ("poly", [0, 1, -0.1]), CS.EQUI])
    for cn in (1, 4):
        for si, spec in enumerate(specs):
            for border, bval in ((0, 77), (1, 0), (2, 0), (4, 0), (5, 0)):
                hs, ws = 300 + 4 * si, 320
                # outputs of >= 512 rows and columns reach k_ray_lin_cn (one table entry per lane needs rays that close together);
                # the small one stays: the generic kernel serves it, as it does TRANSPARENT with anything but bilinear
                # (the polynomial chains' w-table: from ~1000 px on)
                big = 512 if si == 0 else 1024
                wo, ho = ({0: (613, 587), 1: (601, 587), 2: (333, 250), 4: (589, 613), 5: (613, 587)} if si == 0 else
                          {0: (1040, 1024), 1: (333, 250), 2: (512, 384), 4: (1027, 1040), 5: (333, 250)})[border]
                n = 1 + (si + border) % 5
                wide = [rng.integers(0, 256, (hs, ws + 8, cn), dtype=np.uint8) for _ in range(n)]
                imgs = [np.ascontiguousarray(w[:, 4:4 + ws]) for w in wide]
                dev_wide = [torch.from_numpy(w).to(dev) for w in wide]
                srcs = [w[:, 4:4 + ws] for w in dev_wide]  # pitched views (dword-aligned)
                fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
                dsts = [torch.from_numpy(fill.copy()).to(dev) for _ in range(n)]
                radius = 140.0 if border != 4 else 190.0
                modes = V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=radius, interpolation=interp, boarder_mode=border, boarder_value=bval)
                assert modes == ["ray"], modes
                kinds = remapper.last_launch_kinds()
                if min(wo, ho) >= big and si < 2:  # (si = 2: the rotation takes rays into the back hemisphere -> fix-up pass -> generic)
                    assert kinds == ["cn"], (kinds, cn, si, border)
                seen.update(kinds)
                xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
                for k in range(n):
                    want = O.remap(imgs[k], xm, ym, interp, border, bval, dst=fill.copy())
                    got = dsts[k].cpu().numpy()
                    assert np.array_equal(got, want), (cn, si, border, k, int((got != want).sum()))
    assert {"cn", "generic"} <= seen, seen
    # seeded random geometries: odd / tiny / non-square sizes, radii beyond the source, negative radii, 1 - 6 units
    menus = [[("equirect_enc", True), CS.EQUI], [("equirect_enc", True), ("poly", [0.02, 0.9, 0.05]), ("zoom", 1.1), CS.EQUI],
             [("equirect_enc", True), ("rot", CS.ry(-0.2)), CS.EQUI], [("equirect_enc", True), ("fisheye_dec", "stereographic")]]
    for case in range(32):
        spec = menus[int(rng.integers(len(menus)))]
        cn = int(rng.choice([1, 4]))
        wo, ho = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        ws, hs = int(rng.integers(1, 80)) * 4, int(rng.integers(2, 300))
        border = int(rng.choice([0, 0, 1, 2, 3, 4, 5]))
        bval = int(rng.integers(0, 256))
        radius = float(rng.choice([min(ws, hs) / 2, rng.uniform(5, 250), -rng.uniform(5, 100)]))
        n = int(rng.integers(1, 7))
        imgs = [rng.integers(0, 256, (hs, ws, cn), dtype=np.uint8) for _ in range(n)]
        fill = rng.integers(0, 256, (ho, wo, cn), dtype=np.uint8)
        dsts = [torch.from_numpy(fill.copy()).to(dev) for _ in range(n)]
        V.remap_tensors(CS.to_product(spec), [torch.from_numpy(i).to(dev) for i in imgs], dsts, radius=radius, interpolation=interp,
                        boarder_mode=border, boarder_value=bval)
        xm, ym = O.get_map(spec, radius=radius, size_input=(hs, ws), size_output=(wo, ho))
        for k in range(n):
            want = O.remap(imgs[k], xm, ym, interp, border, bval, dst=fill.copy())
            got = dsts[k].cpu().numpy()
            assert np.array_equal(got, want), (case, cn, spec, (wo, ho), (ws, hs), border, radius, k, int((got != want).sum()))
    # units that override the rotation (per-frame calibration on gray / BGRA frames): k_ray_lin_cn<..., BOXES = 0>
    from vr180_convert_amd import transformer as T

    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    for cn in (1, 4):
        for (hs, ws, wo, ho, radius, border) in ((256, 256, 256, 256, 128.0, 0), (200, 320, 333, 129, 170.0, 1), (96, 64, 70, 50, 30.0, 4),
                                                 (400, 400, 608, 608, 200.0, 0), (300, 360, 577, 577, 190.0, 2)):  # (square: rays stay in the front hemisphere)
            imgs = [rng.integers(0, 256, (hs, ws, cn), dtype=np.uint8) for _ in range(5)]
            quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(5)]
            dsts = [torch.zeros((ho, wo, cn), dtype=torch.uint8, device=dev) for _ in range(5)]
            assert V.remap_tensors(base, [torch.from_numpy(i).to(dev) for i in imgs], dsts, radius=radius, interpolation=interp,
                                   boarder_mode=border, rotations=quats) == ["ray"]
            if min(wo, ho) >= 512:  # (the small ones: the generic kernel)
                assert remapper.last_launch_kinds() == ["cn_rot"], (remapper.last_launch_kinds(), cn, wo, ho)
            for f in range(5):
                xm, ym = O.get_map(CS.c5_spec(f // 2, f % 2), radius=radius, size_input=(hs, ws), size_output=(wo, ho))
                want = O.remap(imgs[f], xm, ym, interp, border, 0)
                got = dsts[f].cpu().numpy()
                assert np.array_equal(got, want), ("rot", cn, (hs, ws), f, int((got != want).sum()))
    # a full-size pair of gray halves, box buffers too small for most tiles (V1C_CN_KB is a tuning-build switch; here: huge magnification)
    img = rng.integers(0, 256, (1024, 1024, 1), dtype=np.uint8)
    d = torch.empty((96, 96, 1), dtype=torch.uint8, device=dev)
    spec = [("equirect_enc", True), ("zoom", 0.08), CS.EQUI]
    V.remap_tensors(CS.to_product(spec), [torch.from_numpy(img).to(dev)], [d], radius=512.0, interpolation=interp)
    xm, ym = O.get_map(spec, radius=512.0, size_input=(1024, 1024), size_output=(96, 96))
    assert np.array_equal(d.cpu().numpy(), O.remap(img, xm, ym, interp, 0, 0))


def test_seeded_random_cases_bit_exact(V, oracle_mod, dev):
    """Differential sweep: seeded random output / source sizes (odd, tiny, non-square), radii,
    chains of the ray and literal kinds, interpolations, border modes / values and unit counts --
    product (through the C ABI) == oracle, byte for byte."""
    rng = np.random.default_rng(424242)
    menus = [
        [("equirect_enc", True), CS.EQUI],
        [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI],
        [("equirect_enc", True), ("rot", CS.ry(0.3)), CS.EQUI],
        [("equirect_enc", True), ("poly", [0.02, 0.9, 0.05]), ("zoom", 1.1), CS.EQUI],
        [("equirect_enc", True), ("fisheye_dec", "stereographic")],
        [("fisheye_enc", "rectilinear"), CS.EQUI],
        [("fisheye_enc", "equisolid"), ("rot", CS.ry(-0.2)), CS.EQUI],
    ]
    for case in range(40):
        spec = menus[int(rng.integers(len(menus)))]
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 300))
        ws, hs = int(rng.integers(3, 260)), int(rng.integers(2, 260))
        interp = int(rng.choice([0, 1, 2, 4]))
        border = int(rng.choice([0, 0, 0, 1, 2, 3, 4]))
        bval = int(rng.integers(0, 256))
        radius = float(rng.choice([min(ws, hs) / 2, rng.uniform(5, 200), -rng.uniform(5, 100)]))
        n = int(rng.integers(1, 6))
        imgs = [rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8) for _ in range(n)]
        want = oracle_mod.apply(spec, imgs, size_output=(w, h), interpolation=interp, border_mode=border, border_value=bval, radius=radius)
        srcs = [torch.from_numpy(i).to(dev) for i in imgs]
        dsts = [torch.empty((h, w, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
        V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=radius, interpolation=interp, boarder_mode=border, boarder_value=bval)
        torch.cuda.synchronize()
        for k in range(n):
            got = dsts[k].cpu().numpy()
            assert np.array_equal(got, want[k]), (case, spec, (w, h), (ws, hs), interp, border, bval, radius, k, int((got != want[k]).sum()))


def test_c2_full_size_apply_lr_vs_oracle(V, oracle_mod, dev):
    """BASELINE config C2 end to end: L+R 4096^2 -> 8192x4096 SBS, one launch."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    spec, out, inp, radius = CS.FULL_CASES["C2"]
    left, right = noise_disc(4096, 4096, 0), noise_disc(4096, 4096, 1)
    want = oracle_mod.apply_lr(spec, left, right, size_output=out, interpolation=1, radius="max")
    sbs = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev),
                             size_output=out, interpolation=1, radius="max")
    assert remapper.last_launch_kinds() == ["mirror"]
    got = sbs.cpu().numpy()
    assert got.shape == (4096, 8192, 3)
    assert np.array_equal(got, want)


def test_full_size_properties(V, dev):
    """Size-independent properties at BASELINE's largest single-GPU size (C4: 8192^2, Lanczos4)."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc_torch

    n = 8192
    img = noise_disc_torch(n, n, 0, dev)
    # (1) the identity chain reproduces the input for every interpolation: FisheyeEncoder("equidistant")
    #     * FisheyeDecoder("equidistant") with radius = n/2 maps pixel i to i (|err| ~1e-13 px)
    ident = T.FisheyeEncoder("equidistant") * T.FisheyeDecoder("equidistant")
    for interp in (4, 1):
        out = torch.empty_like(img)
        V.remap_tensors(ident, [img], [out], radius=n / 2, interpolation=interp)
        assert torch.equal(out, img), interp
    # (2) identical eyes give identical halves; (3) an all-outside map gives pure border colour
    t = T.EquirectangularEncoder() * T.PolynomialScaler([0, 1, -0.1]) * T.FisheyeDecoder("equidistant")
    sbs = V.apply_lr_tensors(t, img, img, size_output=(n, n), interpolation=4, radius="max")
    assert torch.equal(sbs[:, :n], sbs[:, n:])
    far = T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant") * T.ZoomTransformer(1e-3)
    out = torch.empty((64, 64, 3), dtype=torch.uint8, device=dev)
    V.remap_tensors(far, [img], [out], radius=n / 2, interpolation=1, boarder_value=(9, 8, 7))
    torch.cuda.synchronize()
    inner = out.cpu().numpy()
    inner[32, 32] = (9, 8, 7)  # the centre pixel maps to the image centre
    assert np.all(inner == np.array([9, 8, 7], np.uint8))


def _lut_remap(V, src, xm, ym, out_wh, interp=1):
    """cv.remap alone on the device (v1c_remap_lut) with maps that are already there."""
    import ctypes as C

    from vr180_convert_amd import _native
    from vr180_convert_amd.remapper import _stream_ptr, border_scalar

    dst = torch.empty((out_wh[1], out_wh[0], 3), dtype=torch.uint8, device=src.device)
    bv = border_scalar(0)
    rc = _native.lib().v1c_remap_lut(src.device.index, _stream_ptr(src.device), src.data_ptr(), src.shape[0], src.shape[1], src.stride(0), 3,
                                     dst.data_ptr(), out_wh[1], out_wh[0], dst.stride(0), xm.data_ptr(), ym.data_ptr(),
                                     xm.stride(0) * 4, interp, 0, bv.ctypes.data)
    _native.check(rc, "v1c_remap_lut")
    return dst


def test_full_size_batch_and_rotation_paths_agree(V, dev):
    """BASELINE sizes of the two batch configs, through size-independent properties:
    C3 (16 units of 2880^2 sharing one map): the lean batch kernel + the remaining-tile launch give, unit
    for unit, the bytes the pair kernel gives for that unit alone;
    C5 (3840^2, a calibration rotation per unit): the in-kernel-box fast path equals cv.remap alone
    (v1c_remap_lut) fed with the map v1c_plan_get_map produces for that rotation -- i.e. the shared
    table entry / m-polynomial coordinates land in the same 1/32 buckets as the per-pixel evaluation
    on all 14.7 M pixels."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.chain import lower_for_get_map
    from vr180_convert_amd.quat import as_rotation_matrix
    from vr180_convert_amd.remapper import _plan_for
    from vr180_convert_amd.synth import noise_disc_torch

    n = 2880
    t = T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant")
    frames = [noise_disc_torch(n, 2 * n, 100 + f, dev) for f in range(8)]
    srcs = [v for fr in frames for v in (fr[:, :n], fr[:, n:])]  # pitched views, 16 units
    dsts = [torch.empty((n, n, 3), dtype=torch.uint8, device=dev) for _ in srcs]
    assert V.remap_tensors(t, srcs, dsts, radius=n / 2, interpolation=1) == ["ray"]
    for k in (0, 5, 15):
        alone = torch.empty((n, n, 3), dtype=torch.uint8, device=dev)
        V.remap_tensors(t, [srcs[k]], [alone], radius=n / 2, interpolation=1)
        assert torch.equal(alone, dsts[k]), k
    del frames, srcs, dsts

    n = 3840
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    quats = [CS.c5_spec(3, eye)[1][1] for eye in (0, 1)]
    imgs = [noise_disc_torch(n, n, 200 + e, dev) for e in (0, 1)]
    outs = [torch.empty((n, n, 3), dtype=torch.uint8, device=dev) for _ in imgs]
    assert V.remap_tensors(base, imgs, outs, radius=n / 2, interpolation=1, rotations=quats) == ["ray"]
    chain = lower_for_get_map(base, radius=n / 2, size_input=(n, n), size_output=(n, n))
    plan = _plan_for(chain, src_hw=(n, n), dst_wh=(n, n), cn=3, interpolation=1, border_mode=0, border_value=0, device=dev)
    for e in (0, 1):
        xm, ym = plan.get_map(as_rotation_matrix(quats[e]))
        want = _lut_remap(V, imgs[e], xm, ym, (n, n))
        torch.cuda.synchronize()
        assert torch.equal(want, outs[e]), e


# ---------------------------------------------------------------------------- API behaviour
def test_apply_numpy_inputs_views_and_borders(V, oracle_mod):
    from vr180_convert_amd.synth import noise_disc, pattern

    O = oracle_mod
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    t = CS.to_product(spec)
    sbs = np.concatenate([noise_disc(300, 260, 1), pattern(300, 260)], axis=1)
    halves = [sbs[:, :260], sbs[:, 260:]]  # non-contiguous column views (remapper.py:455-456)
    for border, bv in ((0, 0), (0, 77), (0, (1, 2, 3)), (1, 0), (2, 0), (3, 0), (4, 0)):
        for interp in (1, 4):
            got = V.apply(t, in_paths=halves, size_output=(200, 240), interpolation=interp, boarder_mode=border,
                          boarder_value=bv, radius="max")
            want = O.apply(spec, halves, size_output=(200, 240), interpolation=interp, border_mode=border,
                           border_value=bv, radius="max")
            assert len(got) == 2 and got[0].shape == (240, 200, 3) and got[0].dtype == np.uint8
            for a, b in zip(got, want):
                assert np.array_equal(a, b), (border, bv, interp)
    # single ndarray input, numeric and negative radius, grayscale
    one = V.apply(t, in_paths=halves[0], size_output=(96, 96), interpolation=1, radius=-120.5)
    assert np.array_equal(one[0], O.apply(spec, [halves[0]], size_output=(96, 96), interpolation=1, radius=-120.5)[0])
    gray = np.ascontiguousarray(halves[0][..., 1])
    g1 = V.apply(t, in_paths=gray, size_output=(96, 96), interpolation=2, radius="max")[0]
    assert g1.shape == (96, 96) and np.array_equal(g1, O.apply(spec, [gray], size_output=(96, 96), interpolation=2, radius="max")[0])
    # transparent border: untouched pixels stay as allocated (zeros here)
    tr = V.apply(t * CS.to_product([("zoom", 0.5)]), in_paths=halves[0], size_output=(96, 96), interpolation=1, boarder_mode=5, radius="max")[0]
    ref = np.zeros((96, 96, 3), np.uint8)
    xm, ym = O.get_map(spec + [("zoom", 0.5)], radius=130.0, size_input=(300, 260), size_output=(96, 96))
    O.remap(np.ascontiguousarray(halves[0]), xm, ym, 1, 5, 0, dst=ref)
    assert np.array_equal(tr, ref)


def test_apply_host_batch_is_pipelined_and_exact(V, oracle_mod, monkeypatch):
    """apply() on a list of host arrays goes through the copy / remap / copy pipeline
    (_hostpipe.py: groups of 4, ring of 3 slots, staging copies on a thread pool): contiguous arrays,
    a read-only one and a column-sliced view, results equal to the oracle and to the unpipelined
    path; grayscale batches too."""
    from vr180_convert_amd import _hostpipe
    from vr180_convert_amd.synth import noise_disc

    calls = []
    orig = _hostpipe.run
    monkeypatch.setattr(_hostpipe, "run", lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1])
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    t = CS.to_product(spec)
    imgs = [noise_disc(200, 240, 80 + f) for f in range(14)]  # 4 groups: the ring of 3 slots is reused
    wide = noise_disc(200, 480, 99)
    imgs[3] = wide[:, 120:360]  # non-contiguous view
    imgs[5] = imgs[5].copy()
    imgs[5].setflags(write=False)
    want = oracle_mod.apply(spec, [np.ascontiguousarray(i) for i in imgs], size_output=(224, 192), interpolation=1, radius=100.0)
    got = V.apply(t, in_paths=imgs, size_output=(224, 192), interpolation=1, radius=100.0)
    assert calls == [14] and len(got) == 14
    for f in range(14):
        assert isinstance(got[f], np.ndarray) and np.array_equal(got[f], want[f]), f
    monkeypatch.setenv("V1C_HOST_PIPELINE", "0")
    plain = V.apply(t, in_paths=imgs, size_output=(224, 192), interpolation=1, radius=100.0)
    assert calls == [14] and all(np.array_equal(a, b) for a, b in zip(plain, got))
    monkeypatch.delenv("V1C_HOST_PIPELINE")
    gray = [np.ascontiguousarray(i[..., 1]) for i in imgs[:5]]
    want_g = oracle_mod.apply(spec, [g[..., None] for g in gray], size_output=(96, 80), interpolation=4, radius=100.0)
    got_g = V.apply(t, in_paths=gray, size_output=(96, 80), interpolation=4, radius=100.0)
    assert calls == [14, 5]
    for f in range(5):
        assert got_g[f].shape == (80, 96) and np.array_equal(got_g[f], want_g[f][..., 0]), f


def test_repeated_calls_take_the_memoised_path_and_stay_exact(V, oracle_mod, dev):
    """remap_tensors remembers the plan of the previous shared-transformer call and Plan.run the
    marshalled units of identical tensors (a steady stream of frames costs ~half the Python time):
    new buffers, a different unit count, changed parameters and changed geometry must all still give
    the oracle's bytes."""
    from vr180_convert_amd.synth import noise_disc

    def check(spec, imgs, out, radius):
        srcs = [torch.from_numpy(i).to(dev) for i in imgs]
        dsts = [torch.empty((out[1], out[0], 3), dtype=torch.uint8, device=dev) for _ in imgs]
        t = CS.to_product(spec)
        for _ in range(3):  # 2nd / 3rd call: memoised plan + reused unit array
            for d in dsts:
                d.zero_()
            V.remap_tensors(t, srcs, dsts, radius=radius, interpolation=1)
        want = oracle_mod.apply(spec, imgs, size_output=out, interpolation=1, radius=radius)
        for k in range(len(imgs)):
            assert np.array_equal(dsts[k].cpu().numpy(), want[k]), (spec, k)
        return t, srcs, dsts

    a = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    b = [("equirect_enc", True), ("poly", [0, 1, -0.2]), CS.EQUI]
    imgs = [noise_disc(160, 160, 90 + k) for k in range(3)]
    t, srcs, dsts = check(a, imgs[:2], (160, 160), 80.0)
    check(a, imgs, (160, 160), 80.0)             # same key, other buffers, 3 units
    check(b, imgs[:2], (160, 160), 80.0)         # other parameters
    check(a, imgs[:2], (192, 128), 80.0)         # other output geometry
    check(a, imgs[:2], (160, 160), 70.0)         # other radius
    # same transformer object and buffers again after all that: still its own plan
    for d in dsts:
        d.zero_()
    V.remap_tensors(t, srcs, dsts, radius=80.0, interpolation=1)
    want = oracle_mod.apply(a, imgs[:2], size_output=(160, 160), interpolation=1, radius=80.0)
    assert all(np.array_equal(dsts[k].cpu().numpy(), want[k]) for k in range(2))


def test_apply_lr_files_auto_radius_and_tuple(V, oracle_mod, tmp_path):
    from PIL import Image

    from vr180_convert_amd import _io
    from vr180_convert_amd.synth import pattern

    O = oracle_mod
    img = pattern(256, 256)
    img[:, :20] = 0
    img[:, -20:] = 0
    img[:24] = 0
    img[-24:] = 0  # black frame: radius="auto" scans the centre COLUMN of a square image
    sbs_in = np.concatenate([img, img[:, ::-1]], axis=1)
    p = tmp_path / "in.png"
    _io.imwrite(p, sbs_in)
    assert np.array_equal(_io.imread(p), sbs_in)
    spec = [("equirect_enc", True), CS.EQUI]
    out_p = tmp_path / "out.png"
    # same path for both eyes -> split in halves (remapper.py:448-456); auto radius = max over eyes
    assert V.apply_lr(CS.to_product(spec), left_path=p, right_path=p, out_path=out_p, size_output=(128, 128),
                      interpolation=1, radius="auto") is None
    want = O.apply_lr(spec, sbs_in[:, :256], sbs_in[:, 256:], size_output=(128, 128), interpolation=1, radius="auto")
    assert np.array_equal(_io.imread(out_p), want)
    # per-eye transformer tuple: per-eye map AND per-eye radius estimate (remapper.py:460-473)
    specs = ([("equirect_enc", True), ("rot", CS.ry(0.05)), CS.EQUI], [("equirect_enc", True), ("rot", CS.ry(-0.05)), CS.EQUI])
    left, right = sbs_in[:, :256], np.ascontiguousarray(sbs_in[:, 256:])
    right[:, :30] = 0
    V.apply_lr(tuple(CS.to_product(s) for s in specs), left_path=left, right_path=right, out_path=out_p,
               size_output=(128, 128), interpolation=4, radius="auto")
    want = O.apply_lr(specs, left, right, size_output=(128, 128), interpolation=4, radius="auto")
    assert np.array_equal(_io.imread(out_p), want)
    # device-side auto radius == host estimate, including the IndexError
    d = torch.from_numpy(np.ascontiguousarray(left)).cuda()
    from vr180_convert_amd.remapper import get_radius_smart

    assert get_radius_smart("auto", [d]) == O.get_radius(np.ascontiguousarray(left))
    with pytest.raises(IndexError):
        get_radius_smart("auto", [torch.full((64, 80, 3), 90, dtype=torch.uint8, device="cuda")])


def test_anaglyph_merge_bit_exact(V, oracle_mod, dev, tmp_path):
    """apply_lr(merge=True), remapper.py:485-497: the device anaglyph equals the reference's NumPy
    float64 expression bit for bit (same operations, no contraction), also on pitched SBS halves."""
    rng = np.random.default_rng(7)
    for h, w in ((1, 1), (37, 53), (128, 300)):
        sbs = torch.from_numpy(rng.integers(0, 256, (h, 2 * w, 3), dtype=np.uint8)).to(dev)
        got = V.anaglyph_tensors(sbs[:, :w], sbs[:, w:])
        assert got.dtype == torch.float64 and tuple(got.shape) == (h, w, 3)
        host = sbs.cpu().numpy()
        want = oracle_mod.anaglyph(host[:, :w], host[:, w:])
        assert np.array_equal(got.cpu().numpy(), want)
    # extremes: all-white eyes give exactly (255, 256, 255) / ... as NumPy does
    white = torch.full((4, 4, 3), 255, dtype=torch.uint8, device=dev)
    assert np.array_equal(V.anaglyph_tensors(white, white).cpu().numpy(), oracle_mod.anaglyph(white.cpu().numpy(), white.cpu().numpy()))
    with pytest.raises(ValueError):
        V.anaglyph_tensors(white, white[:2])
    # through apply_lr: the merged picture written to disk is the anaglyph of the remapped halves
    from vr180_convert_amd import _io
    from vr180_convert_amd.synth import pattern

    img = pattern(160, 160)
    spec = [("equirect_enc", True), CS.EQUI]
    out_p = tmp_path / "merged.png"
    V.apply_lr(CS.to_product(spec), left_path=img, right_path=np.ascontiguousarray(img[:, ::-1]), out_path=out_p,
               size_output=(96, 96), interpolation=1, radius="max", merge=True)
    halves = oracle_mod.apply_lr(spec, img, np.ascontiguousarray(img[:, ::-1]), size_output=(96, 96), interpolation=1, radius="max")
    want = oracle_mod.anaglyph(halves[:, :96], halves[:, 96:])
    if _io._cv is None:  # no labels without cv2: the file holds the rounded anaglyph itself
        assert np.array_equal(_io.imread(out_p), np.clip(np.rint(want), 0, 255).astype(np.uint8))


def test_user_defined_transformer_takes_lut_path(V, oracle_mod, dev):
    """README.md:204-219: any TransformerBase subclass must work; its map comes from its own
    transform(), the gather runs on the GPU (v1c_remap_lut)."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    class Swirl(T.TransformerBase):
        def transform(self, x, y, **kw):
            r = np.sqrt(x**2 + y**2)
            a = 0.3 * r
            return x * np.cos(a) - y * np.sin(a), x * np.sin(a) + y * np.cos(a)

        def inverse_transform(self, x, y, **kw):
            raise NotImplementedError

    img = noise_disc(200, 200, 3)
    t = T.EquirectangularEncoder() * Swirl() * T.FisheyeDecoder("equidistant")
    src = torch.from_numpy(img).to(dev)
    dst = torch.empty((150, 160, 3), dtype=torch.uint8, device=dev)
    assert V.remap_tensors(t, [src], [dst], radius=100.0, interpolation=4) == ["lut"]
    xm, ym = V.get_map(t, radius=100.0, size_input=(200, 200), size_output=(160, 150))
    want = oracle_mod.remap(img, xm, ym, 4)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), want)


def test_batch_per_unit_rotation_shares_one_plan(V, oracle_mod, dev):
    """BASELINE config 5 in small: frames x eyes, each with its own calibration rotation, one plan."""
    from vr180_convert_amd.remapper import _PLANS
    from vr180_convert_amd.synth import noise_disc

    n_frames, size = 10, 256  # 20 units: exercises the 16-units-per-launch chunking
    sbs_in = [noise_disc(size, 2 * size, f) for f in range(n_frames)]
    srcs_d, dsts_d, ts, want = [], [], [], []
    outs = [torch.empty((size, 2 * size, 3), dtype=torch.uint8, device=dev) for _ in range(n_frames)]
    for f in range(n_frames):
        frame_d = torch.from_numpy(sbs_in[f]).to(dev)
        for eye in (0, 1):
            srcs_d.append(frame_d[:, eye * size:(eye + 1) * size])  # pitched views of the SBS frame
            dsts_d.append(outs[f][:, eye * size:(eye + 1) * size])
            ts.append(CS.to_product(CS.c5_spec(f, eye)))
    _PLANS.clear()
    paths = V.remap_tensors(ts, srcs_d, dsts_d, radius=size / 2, interpolation=1)
    torch.cuda.synchronize()
    assert paths == ["ray"] and len(_PLANS) == 1
    for f in range(n_frames):
        specs = (CS.c5_spec(f, 0), CS.c5_spec(f, 1))
        w = oracle_mod.apply_lr(specs, sbs_in[f][:, :size], sbs_in[f][:, size:], size_output=(size, size), interpolation=1, radius=size / 2)
        assert np.array_equal(outs[f].cpu().numpy(), w), f


def test_errors(V, dev):
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.remapper import Plan
    from vr180_convert_amd.chain import lower_for_get_map

    t = T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant")
    a = torch.zeros((32, 32, 3), dtype=torch.uint8, device=dev)
    with pytest.raises(TypeError):
        V.remap_tensors(t, [a.float()], [a], radius=16.0)
    with pytest.raises(ValueError):
        V.remap_tensors(t, [a[:, ::2]], [a[:, :16]], radius=16.0)  # pixel stride != channels
    ch = lower_for_get_map(t, radius=16.0, size_input=(32, 32), size_output=(32, 32))
    plan = Plan(ch, src_hw=(32, 32), dst_wh=(32, 32), cn=3, interpolation=1, border_mode=0, border_value=0, device=dev)
    with pytest.raises(ValueError):
        plan.run([a], [torch.zeros((16, 16, 3), dtype=torch.uint8, device=dev)])
    with pytest.raises(ValueError, match="rotate"):
        plan.run([a], [torch.empty_like(a)], [np.eye(3)])
    with pytest.raises(ValueError, match="Unknown mapping type"):
        V.apply(T.FisheyeEncoder("nope") * T.FisheyeDecoder("equidistant"), in_paths=np.zeros((8, 8, 3), np.uint8), radius="max")
    with pytest.raises(IndexError):
        V.apply(t, in_paths=np.full((16, 20, 3), 200, np.uint8), radius="auto")  # no black border


def test_known_answer_reference_docs_pair_gpu(V, golden_dir):
    from PIL import Image

    from vr180_convert_amd import transformer as T

    src = np.asarray(Image.open(golden_dir / "ref_docs" / "test.jpg").convert("RGB"))[..., ::-1].copy()
    ref = np.asarray(Image.open(golden_dir / "ref_docs" / "test.lr.PolynomialScaler.jpg").convert("RGB"))[..., ::-1]
    t = T.EquirectangularEncoder() * T.PolynomialScaler() * T.FisheyeDecoder("equidistant")
    out = V.apply(t, in_paths=[src, src], size_output=(2048, 2048), radius="max")  # library defaults: Lanczos4
    sbs = np.concatenate(out, axis=1)
    d = sbs.astype(np.float64) - ref
    assert 10 * np.log10(255.0**2 / np.mean(d * d)) >= 30.0


# ---------------------------------------------------------------------------- tile-kernel edge paths
TILE_EDGE = {
    # name: (src (H, W), out (W, H), spec, radius)
    "ragged_sizes": ((275, 301), (333, 217), [("equirect_enc", True), CS.EQUI], 130.0),
    "tiny": ((9, 7), (5, 3), [("equirect_enc", True), CS.EQUI], 3.0),
    "one_tile_minus_one": ((64, 64), (63, 15), [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI], 32.0),
    "minification_box_too_big": ((1500, 1500), (96, 96), [("equirect_enc", True), CS.EQUI], 750.0),
    "magnification": ((40, 40), (640, 480), [("equirect_enc", True), ("zoom", 3.0), CS.EQUI], 20.0),
    "rotated_45_boxes": ((700, 700), (512, 512), [("equirect_enc", True), ("rot", CS.ry(0.0)), ("rot_quat", CS.rotvec_quat([0, 0, 0.8])), CS.EQUI], 350.0),
    "circle_edge_outside": ((300, 300), (256, 256), [("equirect_enc", True), ("zoom", 0.7), CS.EQUI], 150.0),
}


@pytest.mark.parametrize("name", list(TILE_EDGE))
def test_tile_kernel_edge_paths(V, oracle_mod, dev, name):
    from vr180_convert_amd.synth import noise_disc

    (sh, sw), out, spec, radius = TILE_EDGE[name]
    img = noise_disc(sh, sw, 11)
    img[::7, ::5] = 255  # make the black outside non-uniform so border handling shows
    want = oracle_mod.apply(spec, [img], size_output=out, interpolation=1, radius=radius, border_value=(3, 250, 77))[0]
    src = torch.from_numpy(img).to(dev)
    dst = torch.full((out[1], out[0], 3), 9, dtype=torch.uint8, device=dev)
    paths = V.remap_tensors(CS.to_product(spec), [src], [dst], radius=radius, interpolation=1, boarder_value=(3, 250, 77))
    torch.cuda.synchronize()
    assert paths == ["ray"]
    assert np.array_equal(dst.cpu().numpy(), want), name


def test_tile_kernel_unaligned_source_and_pitched_views(V, oracle_mod, dev):
    """Source pointers / pitches that are not dword aligned (tile kernel gathers from global
    memory instead of staging) and destinations that are column views of a wider buffer."""
    from vr180_convert_amd.synth import noise_disc

    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    t = CS.to_product(spec)
    img = noise_disc(200, 203, 5)  # row bytes 609: odd pitch
    want = oracle_mod.apply(spec, [img], size_output=(180, 170), interpolation=1, radius="max")[0]
    flat = torch.zeros(200 * 609 + 64, dtype=torch.uint8, device=dev)
    for shift in (0, 1, 3):
        src = flat[shift:shift + 200 * 609].view(200, 203, 3)
        src.copy_(torch.from_numpy(img))
        wide = torch.zeros((170, 400, 3), dtype=torch.uint8, device=dev)
        dst = wide[:, 101:281]  # byte offset 303: unaligned destination rows
        V.remap_tensors(t, [src], [dst], radius=100.0, interpolation=1)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), want), shift
        assert int(wide[:, :101].max()) == 0 and int(wide[:, 281:].max()) == 0


def test_rotations_argument_equals_per_unit_transformers(V, oracle_mod, dev):
    """remap_tensors(..., rotations=[...]) == one Euclidean3DRotator chain per unit."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    size = 192
    imgs = [noise_disc(size, size, f) for f in range(6)]
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(6)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    for interp in (1, 4):
        dsts = [torch.empty_like(s) for s in srcs]
        assert V.remap_tensors(base, srcs, dsts, radius=size / 2, interpolation=interp, rotations=quats) == ["ray"]
        torch.cuda.synchronize()
        for f in range(6):
            want = oracle_mod.apply(CS.c5_spec(f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=interp, radius=size / 2)[0]
            assert np.array_equal(dsts[f].cpu().numpy(), want), (interp, f)
    with pytest.raises(ValueError):
        V.remap_tensors(T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant"), srcs, dsts, radius=96.0, rotations=quats)


@pytest.mark.parametrize("interp", [1, 4])
def test_shared_map_unit_loop_mixed_alignment(V, oracle_mod, dev, interp):
    """One launch group, 11 units sharing the map (a workgroup serves up to 8 units, reusing its
    coordinates): units alternate between dword-aligned sources (LDS-staged) and byte-shifted ones
    (gathered from global memory), with different row pitches."""
    from vr180_convert_amd.synth import noise_disc

    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    size = 160
    imgs = [noise_disc(size, size, f) for f in range(11)]
    srcs, keep = [], []
    for f, im in enumerate(imgs):
        shift = (0, 1, 0, 2, 0, 0, 3, 0, 0, 1, 0)[f]
        pitch = size * 3 + (0, 4, 8)[f % 3]
        flat = torch.zeros(size * pitch + 16, dtype=torch.uint8, device=dev)
        v = flat[shift:shift + size * pitch].view(size, pitch)[:, : size * 3].view(size, size, 3)
        v.copy_(torch.from_numpy(im))
        srcs.append(v)
        keep.append(flat)
    dsts = [torch.empty((size, size, 3), dtype=torch.uint8, device=dev) for _ in imgs]
    assert V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=size / 2, interpolation=interp) == ["ray"]
    torch.cuda.synchronize()
    want = oracle_mod.apply(spec, imgs, size_output=(size, size), interpolation=interp, radius=size / 2)
    for f in range(11):
        assert np.array_equal(dsts[f].cpu().numpy(), want[f]), (interp, f)


LEAN_BATCH = {
    # name: (src (H, W), out (W, H), spec, radius): aligned sources, 11 units = groups of 8 + 3
    "interior_mostly": ((512, 512), (512, 400), [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI], 256.0),
    "m_table": ((384, 384), (448, 384), [("equirect_enc", True), CS.EQUI], 192.0),
    "diagonal_big_boxes": ((700, 700), (576, 512), [("equirect_enc", True), ("rot_quat", CS.rotvec_quat([0, 0, 0.8])), CS.EQUI], 350.0),
    "footprints_leave_source": ((300, 300), (320, 256), [("equirect_enc", True), ("zoom", 0.7), CS.EQUI], 150.0),
    "minified_boxes_do_not_fit": ((1500, 1500), (128, 96), [("equirect_enc", True), CS.EQUI], 750.0),
}


@pytest.mark.parametrize("name", list(LEAN_BATCH))
def test_lean_batch_kernel_and_rest_tiles(V, oracle_mod, dev, name):
    """Shared-map batches of more than two units per workgroup: interior tiles run in the lean batch
    kernel, the tiles it leaves out (edges of the destination, footprints leaving the source, boxes
    beyond its LDS buffers) in the pair kernel on the plan's tile list -- every byte equals the oracle's."""
    from vr180_convert_amd.synth import noise_disc

    (sh, sw), out, spec, radius = LEAN_BATCH[name]
    imgs = [noise_disc(sh, sw, 30 + f) for f in range(11)]
    for im in imgs:
        im[::9, ::7] = 200
    want = oracle_mod.apply(spec, imgs, size_output=out, interpolation=1, radius=radius, border_value=(5, 6, 7))
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.full((out[1], out[0], 3), 9, dtype=torch.uint8, device=dev) for _ in imgs]
    assert V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=radius, interpolation=1, boarder_value=(5, 6, 7)) == ["ray"]
    torch.cuda.synchronize()
    for f in range(11):
        assert np.array_equal(dsts[f].cpu().numpy(), want[f]), (name, f)


ROT_UNITS = {
    # name: (src (H, W), out (W, H), chain before / after the rotation, radius, rotation vectors)
    "small_calibration": ((320, 320), (320, 320), [], [CS.EQUI], 160.0, [[0.02, -0.01, 0.03], [-0.03, 0.02, 0.0], [0.0, 0.0, -0.04]]),
    "partial_tiles": ((300, 280), (333, 217), [], [CS.EQUI], 140.0, [[0.05, 0.0, 0.0], [0.0, -0.06, 0.02]]),
    "w_table": ((320, 320), (384, 320), [], [("poly", [0, 1, -0.1]), CS.EQUI], 160.0, [[0.03, 0.01, 0.0], [0.0, 0.02, -0.05]]),
    "rays_leave_source": ((300, 300), (320, 256), [], [("zoom", 0.7), CS.EQUI], 150.0, [[0.1, 0.0, 0.0], [0.0, -0.2, 0.1]]),
    "large_rotation": ((400, 400), (384, 384), [], [CS.EQUI], 200.0, [[0.0, 0.9, 0.0], [0.5, -0.4, 0.8], [0.0, 0.0, 1.5]]),
}


@pytest.mark.parametrize("name", list(ROT_UNITS))
def test_per_unit_rotation_tile_paths(V, oracle_mod, dev, name):
    """Units overriding the rotation (one unit per workgroup, box reduced in the kernel): tiles whose
    bounding box of all pixels lies inside the source take the unpredicated path, the others the
    general one; with and without the m-polynomial table; rotations from calibration-sized to large."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    (sh, sw), out, pre, post, radius, rvecs = ROT_UNITS[name]
    quats = [CS.rotvec_quat(r) for r in rvecs]
    imgs = [noise_disc(sh, sw, 50 + f) for f in range(len(quats))]
    for im in imgs:
        im[::6, ::11] = 180
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.full((out[1], out[0], 3), 9, dtype=torch.uint8, device=dev) for _ in imgs]
    base = CS.to_product([("equirect_enc", True), *pre, ("rot_quat", (1.0, 0.0, 0.0, 0.0)), *post])
    assert V.remap_tensors(base, srcs, dsts, radius=radius, interpolation=1, boarder_value=(1, 2, 3), rotations=quats) == ["ray"]
    torch.cuda.synchronize()
    for f, q in enumerate(quats):
        spec = [("equirect_enc", True), *pre, ("rot_quat", q), *post]
        want = oracle_mod.apply(spec, [imgs[f]], size_output=out, interpolation=1, radius=radius, border_value=(1, 2, 3))[0]
        assert np.array_equal(dsts[f].cpu().numpy(), want), (name, f)


def test_plan_run_is_graph_capturable(V, dev):
    """v1c_plan_run neither allocates nor synchronises -- a warmed-up call can be
    captured into a HIP graph on a side stream and replayed."""
    from vr180_convert_amd.synth import noise_disc

    size = 256
    t = CS.to_product([("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI])
    l, r = (torch.from_numpy(noise_disc(size, size, k)).to(dev) for k in (0, 1))
    ref = V.apply_lr_tensors(t, l, r, size_output=(size, size), interpolation=1, radius="max").clone()
    out = torch.empty_like(ref)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):  # warm-up on the capture stream: the plan exists before capture starts
        V.apply_lr_tensors(t, l, r, out=out, size_output=(size, size), interpolation=1, radius="max")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        V.apply_lr_tensors(t, l, r, out=out, size_output=(size, size), interpolation=1, radius="max")
    for _ in range(2):
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_batch_plan_run_is_graph_capturable(V, oracle_mod, dev):
    """The same for a shared-map batch (the lean batch kernel with the remaining tiles as its first grid slice)."""
    from vr180_convert_amd.synth import noise_disc

    size = 320
    spec = [("equirect_enc", True), ("zoom", 0.8), CS.EQUI]
    t = CS.to_product(spec)
    imgs = [noise_disc(size, size, 70 + k) for k in range(5)]
    want = oracle_mod.apply(spec, imgs, size_output=(size, size), interpolation=1, radius=size / 2)
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.empty_like(x) for x in srcs]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)
    for _ in range(2):
        for d in dsts:
            d.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for k in range(5):
            assert np.array_equal(dsts[k].cpu().numpy(), want[k]), k


def test_launches_of_more_than_16_units_go_through_the_unit_ring(V, oracle_mod, dev):
    """Up to 16 units travel in the kernel arguments; longer batches are copied into a slot of the plan's device ring by
    launches of their own (plan.hip: ring_put) and run as ONE launch: 37 frames sharing a map (the lean batch kernel, 5 groups),
    21 units with a rotation each (k_ray_lin3_rot_pair_raw: 11 pairs, the last one half empty), repeated so that the ring wraps,
    and the same batch recorded into a graph (capture slots)."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    size = 256
    spec = [("equirect_enc", True), CS.EQUI]
    t = CS.to_product(spec)
    imgs = [noise_disc(size, size, 300 + k) for k in range(37)]
    want = oracle_mod.apply(spec, imgs, size_output=(size, size), interpolation=1, radius=size / 2)
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    for rep in range(6):  # (4 ring slots)
        dsts = [torch.zeros_like(x) for x in srcs]
        assert V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1) == ["ray"]
        for k in range(37):
            assert np.array_equal(dsts[k].cpu().numpy(), want[k]), (rep, k)
    # a rotation per unit
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(21)]
    dsts = [torch.zeros_like(x) for x in srcs[:21]]
    V.remap_tensors(base, srcs[:21], dsts, radius=size / 2, interpolation=1, rotations=quats)
    for f in range(21):
        w = oracle_mod.apply(CS.c5_spec(f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=1, radius=size / 2)[0]
        assert np.array_equal(dsts[f].cpu().numpy(), w), f
    # recorded into a graph: the launch owns a capture slot of the ring
    dsts = [torch.zeros_like(x) for x in srcs]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        V.remap_tensors(t, srcs, dsts, radius=size / 2, interpolation=1)
    for _ in range(2):
        for d in dsts:
            d.zero_()
        V.remap_tensors(t, srcs[:20], [torch.empty_like(x) for x in srcs[:20]], radius=size / 2, interpolation=1)  # eager traffic in between
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for k in range(37):
            assert np.array_equal(dsts[k].cpu().numpy(), want[k]), k


def test_more_units_than_a_ring_slot_holds(V, oracle_mod, dev):
    """v1c_plan_run takes any number of units: a ring slot holds 256, longer arrays go out 256 at a time (csrc/plan.hip).  300 frames
    sharing one map (the batch kernel) and 270 units with a rotation each (the rotation-pair kernel): every unit against the oracle."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd import transformer as T

    O = oracle_mod
    rng = np.random.default_rng(4242)
    hs = ws = 160
    wo = ho = 448
    spec = [("equirect_enc", True), CS.EQUI]
    imgs = [rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8) for _ in range(300)]
    srcs = [torch.from_numpy(i).to(dev) for i in imgs]
    dsts = [torch.empty((ho, wo, 3), dtype=torch.uint8, device=dev) for _ in imgs]
    assert V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=hs / 2, interpolation=1) == ["ray"]
    assert remapper.last_launch_kinds() == ["batch"], remapper.last_launch_kinds()
    xm, ym = O.get_map(spec, radius=hs / 2, size_input=(hs, ws), size_output=(wo, ho))
    for k in range(300):
        assert np.array_equal(dsts[k].cpu().numpy(), O.remap(imgs[k], xm, ym, 1, 0, 0)), k
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    n = 270
    quats = [CS.c5_spec(f // 2, f % 2)[1][1] for f in range(n)]
    for d in dsts[:n]:
        d.zero_()
    assert V.remap_tensors(base, srcs[:n], dsts[:n], radius=hs / 2, interpolation=1, rotations=quats) == ["ray"]
    assert remapper.last_launch_kinds() == ["rot_pair"], remapper.last_launch_kinds()
    for f in range(n):
        xm, ym = O.get_map(CS.c5_spec(f // 2, f % 2), radius=hs / 2, size_input=(hs, ws), size_output=(wo, ho))
        assert np.array_equal(dsts[f].cpu().numpy(), O.remap(imgs[f], xm, ym, 1, 0, 0)), f


@pytest.mark.parametrize("env", [{"V1C_DISABLE_SHARED_ENTRY": "1"}, {"V1C_DISABLE_MPOLY": "1"}, {"V1C_UPB": "1"}, {"V1C_UPB": "3"},
                                 {"V1C_DISABLE_FAST": "1"}, {"V1C_DISABLE_COORDS_BOUNDED": "1"}, {"V1C_DISABLE_LEAN": "1"}, {"V1C_DISABLE_MERGE": "1"}, {"V1C_XCD_STRIPS": "2"},
                                 {"V1C_DISABLE_MIRROR": "1"}, {"V1C_MIRROR_RAW": "0"}, {"V1C_MIRROR_RAW": "4"}, {"V1C_MIRROR_RAW": "7"},
                                 {"V1C_MIRROR_SEQ": "0"}, {"V1C_SEQ_NOREST": "0"}, {"V1C_MIRROR_SEQ_KB": "5"}, {"V1C_MIRROR_SEQ_KB": "16"}, {"V1C_MIRROR_SEQ": "0", "V1C_MIRROR_RAW": "4"}, {"V1C_MIRROR_SEQ": "0", "V1C_MIRROR_RAW": "7"},
                                 {"V1C_LEAN_RAW": "0"}, {"V1C_LEAN_RAW": "4"}, {}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_kernel_variants_bit_exact(env):
    """The instantiations the default configuration does not reach (per-pixel table fallback,
    one / three units per workgroup, generic kernels) give the same bytes: tests/variant_probe.py
    in a subprocess, because the engine reads these switches once per process."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    from vr180_convert_amd import _native

    # the switches exist only in the -DV1C_TUNING build of the same sources (csrc/Makefile `tuning`)
    tuning = _native.LIB_PATH.with_name("libvr180remap_tuning.so")
    if not tuning.exists():  # normally built by __graft_entry__.build() and shipped with the snapshot
        subprocess.run(["make", "-C", str(tuning.parent), "-j4", "tuning"], check=True, capture_output=True, timeout=900)
    probe = Path(__file__).with_name("variant_probe.py")
    r = subprocess.run([sys.executable, str(probe)], env={**os.environ, **env, "V1C_LIB": str(tuning)}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------- full size vs the oracle
def _all_cores(O):
    import os

    O.set_threads(os.cpu_count() or 1)


def test_c4_full_size_eye_vs_oracle(V, oracle_mod, dev):
    """BASELINE config 4 at its real size and with its real chain: one 8192 x 8192 eye,
    EquirectangularEncoder * Euclidean3DRotator(from_euler_angles(0, pi/4, 0)) * PolynomialScaler([0,1,-0.1])
    * FisheyeDecoder("equidistant"), INTER_LANCZOS4 -- all 201 326 592 output bytes against the oracle
    (literal fp64 chain + restated cv2.remap, every host core of the GPU box)."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    spec, out, inp, radius = CS.FULL_CASES["C4"]
    O = oracle_mod
    _all_cores(O)
    try:
        img = noise_disc(8192, 8192, 40)
        want = O.apply(spec, [img], size_output=out, interpolation=4, radius="max")[0]
        dst = torch.empty((8192, 8192, 3), dtype=torch.uint8, device=dev)
        assert V.remap_tensors(CS.to_product(spec), [torch.from_numpy(img).to(dev)], [dst], radius=4096.0, interpolation=4) == ["ray"]
        assert remapper.last_launch_kinds() == ["tile"]
        got = dst.cpu().numpy()
        assert got.shape == want.shape == (8192, 8192, 3)
        assert np.array_equal(got, want), int((got != want).sum())
        # and the pair kernel (both eyes of the apply_lr launch) writes the same bytes for this eye
        sbs = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(img).to(dev), torch.from_numpy(img).to(dev),
                                 size_output=out, interpolation=4, radius="max")
        assert torch.equal(sbs[:, :8192], dst) and torch.equal(sbs[:, 8192:], dst)
    finally:
        O.set_threads(min(8, __import__("os").cpu_count() or 1))


def test_c5_full_size_units_vs_oracle(V, oracle_mod, dev):
    """BASELINE config 5 at its real size: FOUR 7680 x 3840 SBS frames (8 units of 3840^2 in one launch), every eye with
    its own calibration rotation (cli.py:308-319 pseudo-half quaternions), bilinear -- through the
    rotations= path that shares one plan -- against the oracle evaluating each unit's own chain."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    _all_cores(O)
    try:
        n = 3840
        fids = (7, 8, 100, 255)
        host = [noise_disc(n, 2 * n, 50 + f) for f in fids]
        base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
        quats = [CS.c5_spec(f, eye)[1][1] for f in fids for eye in (0, 1)]
        frs = [torch.from_numpy(h).to(dev) for h in host]
        outs = [torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev) for _ in frs]
        srcs = [v for fr in frs for v in (fr[:, :n], fr[:, n:])]
        dsts = [v for o in outs for v in (o[:, :n], o[:, n:])]
        assert V.remap_tensors(base, srcs, dsts, radius=n / 2, interpolation=1, rotations=quats) == ["ray"]
        assert remapper.last_launch_kinds() == ["rot_pair"]
        for k, f in enumerate(fids):
            got = outs[k].cpu().numpy()
            for eye in (0, 1):
                want = O.apply(CS.c5_spec(f, eye), [host[k][:, eye * n:(eye + 1) * n]], size_output=(n, n), interpolation=1, radius="max")[0]
                assert np.array_equal(got[:, eye * n:(eye + 1) * n], want), (f, eye, int((got[:, eye * n:(eye + 1) * n] != want).sum()))
    finally:
        O.set_threads(min(8, __import__("os").cpu_count() or 1))


def test_c3_full_size_frames_vs_oracle(V, oracle_mod, dev):
    """BASELINE config 3 at its real size: SBS frames of 5760 x 2880 split into halves (remapper.py:448-456),
    one shared equidistant map, bilinear, through the lean batch kernel (8 frames = 16 units in one launch: one rank's
    share of the 64 frames); ALL 16 units against the oracle."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    _all_cores(O)
    try:
        n = 2880
        t = T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant")
        host = [noise_disc(n, 2 * n, 300 + f) for f in range(8)]
        frames = [torch.from_numpy(h).to(dev) for h in host]
        outs = [torch.empty((n, 2 * n, 3), dtype=torch.uint8, device=dev) for _ in frames]
        srcs = [v for fr in frames for v in (fr[:, :n], fr[:, n:])]
        dsts = [v for fr in outs for v in (fr[:, :n], fr[:, n:])]
        assert V.remap_tensors(t, srcs, dsts, radius=n / 2, interpolation=1) == ["ray"]
        assert remapper.last_launch_kinds() == ["batch"]
        spec = [("equirect_enc", True), ("fisheye_dec", "equidistant")]
        xm, ym = O.get_map(spec, radius=n / 2, size_input=(n, n), size_output=(n, n))
        for f, eye in [(f, e) for f in range(8) for e in (0, 1)]:
            want = O.remap(host[f][:, eye * n:(eye + 1) * n], xm, ym, 1)
            got = outs[f][:, eye * n:(eye + 1) * n].cpu().numpy()
            assert np.array_equal(got, want), (f, eye)
    finally:
        O.set_threads(min(8, __import__("os").cpu_count() or 1))


def test_sizes_beyond_the_baseline_configs(V, oracle_mod, dev):
    """Four times BASELINE's largest output: 16 384 x 16 384 per eye (a 32 768 x 16 384 side-by-side buffer of 1.6 GB: 32-bit offsets at
    their far end, 65 536 tiles per eye) through the mirror launch, and the widest output the ABI takes, 32 767 columns -- every byte
    against the oracle."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    if torch.cuda.mem_get_info(0)[0] < 12 << 30:
        pytest.skip("needs 12 GB of free device memory")
    O = oracle_mod
    _all_cores(O)
    try:
        spec = CS.FULL_CASES["C2"][0]
        n_in, n = 6000, 16384
        left, right = noise_disc(n_in, n_in, 1), noise_disc(n_in, n_in, 2)
        sbs = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev), size_output=(n, n),
                                 interpolation=1, radius="max")
        assert remapper.last_launch_kinds() == ["mirror"]
        got = sbs.cpu().numpy()
        del sbs
        xm, ym = O.get_map(spec, radius=n_in / 2, size_input=(n_in, n_in), size_output=(n, n))
        for e, im in enumerate((left, right)):
            want = O.remap(im, xm, ym, 1, 0, 0)
            assert np.array_equal(got[:, e * n:(e + 1) * n], want), e
        del got, want, xm, ym
        w2, h2 = 32767, 2048
        d = torch.empty((h2, w2, 3), dtype=torch.uint8, device=dev)
        V.remap_tensors(CS.to_product(spec), [torch.from_numpy(left).to(dev)], [d], radius=n_in / 2, interpolation=1)
        xm, ym = O.get_map(spec, radius=n_in / 2, size_input=(n_in, n_in), size_output=(w2, h2))
        assert np.array_equal(d.cpu().numpy(), O.remap(left, xm, ym, 1, 0, 0))
    finally:
        O.set_threads(min(8, __import__("os").cpu_count() or 1))
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------- live third-party hooks
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_device_remap_equals_live_cv2_when_present(V, oracle_mod, dev, interp):
    """where opencv-python is importable ON THE GPU BOX it is the live third-party oracle
    for cv2.remap (Appendix A): the device sampler (v1c_remap_lut through the C ABI) and the C restatement
    must both equal it bit for bit, every border mode.  Skipped (and recorded as such by smoke()) when
    the image has no cv2."""
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(11)
    src = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    xm = rng.uniform(-12, 95, (70, 90)).astype(np.float32)
    ym = rng.uniform(-12, 73, (70, 90)).astype(np.float32)
    xm[3, 4] = np.nan
    ym[5, 6] = np.inf
    from vr180_convert_amd import _native
    from vr180_convert_amd.remapper import _stream_ptr, border_scalar

    s_d, x_d, y_d = (torch.from_numpy(a).to(dev) for a in (src, xm, ym))
    for border in (0, 1, 2, 3, 4):
        want = cv2.remap(src, xm, ym, interpolation=interp, borderMode=border, borderValue=(7, 0, 0))
        dst = torch.zeros((70, 90, 3), dtype=torch.uint8, device=dev)
        bv = border_scalar(7)
        rc = _native.lib().v1c_remap_lut(0, _stream_ptr(dev), s_d.data_ptr(), 61, 83, s_d.stride(0), 3, dst.data_ptr(), 70, 90,
                                         dst.stride(0), x_d.data_ptr(), y_d.data_ptr(), x_d.stride(0) * 4, interp, border, bv.ctypes.data)
        _native.check(rc, "v1c_remap_lut")
        got = dst.cpu().numpy()
        assert np.array_equal(got, want), (cv2.__version__, interp, border, int((got != want).sum()))
        assert np.array_equal(oracle_mod.remap(src, xm, ym, interp, border, 7), want), (cv2.__version__, interp, border)


def test_quat_helpers_equal_live_numpy_quaternion_when_present():
    """Where numpy-quaternion is importable on the GPU box: quat.py's as_rotation_matrix / rotate_vectors /
    from_euler_angles / from_rotation_vector against the package the reference uses (transformer.py:10,676)."""
    quaternion = pytest.importorskip("quaternion")
    from vr180_convert_amd import quat as Q

    rng = np.random.default_rng(5)
    for _ in range(20):
        w, x, y, z = rng.normal(size=4)
        ref = quaternion.as_rotation_matrix(quaternion.quaternion(w, x, y, z))
        np.testing.assert_allclose(Q.as_rotation_matrix((w, x, y, z)), ref, atol=1e-14)
        v = rng.normal(size=(7, 3))
        np.testing.assert_allclose(Q.rotate_vectors((w, x, y, z), v), quaternion.rotate_vectors(quaternion.quaternion(w, x, y, z), v), atol=1e-13)
        a, b, c = rng.uniform(-3, 3, 3)
        e = quaternion.from_euler_angles(a, b, c)
        np.testing.assert_allclose(Q.from_euler_angles(a, b, c).components(), [e.w, e.x, e.y, e.z], atol=1e-14)
        r = quaternion.from_rotation_vector([a, b, c])
        np.testing.assert_allclose(Q.from_rotation_vector([a, b, c]).components(), [r.w, r.x, r.y, r.z], atol=1e-14)


# ---------------------------------------------------------------------------- round-2 regressions
def test_apply_lr_tuple_transformers_unequal_eye_widths(V, oracle_mod, tmp_path):
    """An odd-width SBS file with left_path == right_path splits into W // 2 and W - W // 2 columns
    (remapper.py:455-456); with per-eye transformers every eye is its own apply() call and takes the
    Denormalize centre from ITS shape (remapper.py:460-473, :385) -- also when the two radii are equal."""
    from vr180_convert_amd import _io
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    sbs_in = noise_disc(120, 241, 9)  # halves: 120 and 121 columns
    p = tmp_path / "odd.png"
    _io.imwrite(p, sbs_in)
    specs = ([("equirect_enc", True), ("rot", CS.ry(0.04)), CS.EQUI], [("equirect_enc", True), ("rot", CS.ry(-0.04)), CS.EQUI])
    out_p = tmp_path / "odd_out.png"
    V.apply_lr(tuple(CS.to_product(s) for s in specs), left_path=p, right_path=p, out_path=out_p, size_output=(96, 96),
               interpolation=1, radius=57.0)
    want = O.apply_lr(specs, sbs_in[:, :120], sbs_in[:, 120:], size_output=(96, 96), interpolation=1, radius=57.0)
    assert np.array_equal(_io.imread(out_p), want)


@pytest.mark.parametrize("robust", [False, True])
def test_calibrated_pair_pixels_vs_oracle(V, oracle_mod, dev, robust):
    """SURVEY 8f-3 on the device path (cli.py:286-319): matched points -> match_lr -> rotation_match(_robust) ->
    calibration_rotators -> a (tL, tR) pair of chains through the HIP kernels; the oracle renders the same two
    matrices.  The points are the pixel positions at which the two eyes see the same rays when the right camera is
    rotated by a known small rotation (plus two gross mismatches for the robust fit), so the fit is also checked
    against that rotation."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.calibration import calibration_rotators, match_lr, rotate_vectors, rotation_match, rotation_match_robust
    from vr180_convert_amd.chain import DenormalizeTransformer, equidistant_from_3d
    from vr180_convert_amd.quat import as_rotation_matrix, from_rotation_vector
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n, radius = 400, 200.0
    left, right = noise_disc(n, n, 41), noise_disc(n, n, 42)
    dec = T.FisheyeDecoder("equidistant")
    q_true = from_rotation_vector([0.015, -0.03, 0.02])
    rng = np.random.default_rng(5)
    rays = rng.normal(size=(60, 3)) * [0.5, 0.5, 0.1] + [0, 0, 1]
    rays /= np.linalg.norm(rays, axis=-1, keepdims=True)
    to_px = dec * DenormalizeTransformer(scale=(radius, radius), center=(n // 2, n // 2))
    pl = np.stack(to_px.transform(*equidistant_from_3d(rays)), -1)
    pr = np.stack(to_px.transform(*equidistant_from_3d(rotate_vectors(q_true, rays))), -1)
    if robust:
        pr[:2] += [[40.0, -35.0], [-50.0, 30.0]]
    vl, vr = match_lr(dec, pl, pr, [left, right], radius=radius)
    if robust:
        q, dropped = rotation_match_robust(vl, vr)
        assert dropped[:2].all()
    else:
        q = rotation_match(vl, vr)
    np.testing.assert_allclose(as_rotation_matrix(q), as_rotation_matrix(q_true), atol=2e-6)  # (float32 points: match_lr)
    ql, qr = calibration_rotators(q)
    head = T.EquirectangularEncoder()
    pair = (head * T.Euclidean3DRotator(ql) * dec, head * T.Euclidean3DRotator(qr) * dec)
    specs = tuple([("equirect_enc", True), ("rot", as_rotation_matrix(h)), CS.EQUI] for h in (ql, qr))
    for interp in (1, 4):
        got = V.apply_lr_tensors(pair, torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev), size_output=(320, 320),
                                 interpolation=interp, radius=radius).cpu().numpy()
        want = O.apply_lr(specs, left, right, size_output=(320, 320), interpolation=interp, radius=radius)
        assert np.array_equal(got, want), (interp, int((got != want).sum()))


def test_memo_keys_are_exact_not_printed(V, oracle_mod, dev):
    """Two chains whose parameters differ below NumPy's print precision (and under a coarse
    np.set_printoptions) must not share a lowered chain / plan: each call equals ITS oracle result."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    img = noise_disc(256, 256, 3)
    src = torch.from_numpy(img).to(dev)
    old = np.get_printoptions()
    np.set_printoptions(precision=2)
    try:
        outs = []
        for c2 in (-0.1, -0.1000000001, -0.1004):
            t = T.EquirectangularEncoder() * T.PolynomialScaler(np.array([0, 1, c2])) * T.FisheyeDecoder("equidistant")
            dst = torch.empty((256, 256, 3), dtype=torch.uint8, device=dev)
            V.remap_tensors(t, [src], [dst], radius=128.0, interpolation=1)
            want = O.apply([("equirect_enc", True), ("poly", [0, 1, c2]), CS.EQUI], [img], size_output=(256, 256), interpolation=1, radius=128.0)[0]
            assert np.array_equal(dst.cpu().numpy(), want), c2
            outs.append(dst.cpu().numpy())
        assert not np.array_equal(outs[0], outs[2])
        # rotations 2e-9 apart, given as matrices (ndarray fields)
        for ang in (0.3, 0.3 + 2e-9, 0.31):
            t = T.EquirectangularEncoder() * T.Euclidean3DRotator(np.array(CS.ry(ang))) * T.FisheyeDecoder("equidistant")
            dst = torch.empty((256, 256, 3), dtype=torch.uint8, device=dev)
            V.remap_tensors(t, [src], [dst], radius=128.0, interpolation=1)
            want = O.apply([("equirect_enc", True), ("rot", CS.ry(ang)), CS.EQUI], [img], size_output=(256, 256), interpolation=1, radius=128.0)[0]
            assert np.array_equal(dst.cpu().numpy(), want), ang
    finally:
        np.set_printoptions(**old)


def test_one_plan_from_two_threads_and_streams(V, oracle_mod, dev):
    """A plan is shared by every thread of the process (remapper._PLANS): two threads remap the same geometry
    on their own streams, 24 calls each, with a chain whose radial table has flagged intervals (rectilinear
    decoder: pole at 90 degrees) so that every call runs the ray pass AND the fix-up pass that consumes the
    plan's tile-flag words.  Every result must equal the oracle's."""
    import threading

    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    spec = [("equirect_enc", True), ("zoom", 1.4), ("fisheye_dec", "rectilinear")]
    t = CS.to_product(spec)
    imgs = [noise_disc(300, 300, 60 + k) for k in range(2)]
    wants = [O.apply(spec, [im], size_output=(320, 256), interpolation=1, radius=40.0)[0] for im in imgs]
    errs = []

    def work(k):
        try:
            torch.cuda.set_device(dev)
            st = torch.cuda.Stream(dev)
            src = torch.from_numpy(imgs[k]).to(dev)
            with torch.cuda.stream(st):
                for it in range(24):
                    dst = torch.full((256, 320, 3), 77, dtype=torch.uint8, device=dev)
                    V.remap_tensors(t, [src], [dst], radius=40.0, interpolation=1)
                    st.synchronize()
                    if not np.array_equal(dst.cpu().numpy(), wants[k]):
                        errs.append((k, it))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:5]


def test_unit_ring_from_two_threads_and_streams(V, oracle_mod, dev):
    """One plan, two threads, a stream each, launches of 20 units with a rotation per unit: both go through the plan's unit ring
    (4 slots: each thread's launches reuse slots the other thread's launches have just used -- the slot events order them across
    streams), every output against the oracle."""
    import threading

    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n, size = 20, 192
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    sets = []
    for k in range(2):
        imgs = [noise_disc(size, size, 900 + 50 * k + f) for f in range(n)]
        quats = [CS.c5_spec(7 * k + f // 2, f % 2)[1][1] for f in range(n)]
        want = [O.apply(CS.c5_spec(7 * k + f // 2, f % 2), [imgs[f]], size_output=(size, size), interpolation=1, radius=size / 2)[0] for f in range(n)]
        sets.append((imgs, quats, want))
    errs: list = []

    def work(k: int) -> None:
        try:
            imgs, quats, want = sets[k]
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                srcs = [torch.from_numpy(i).to(dev) for i in imgs]
                for it in range(12):
                    dsts = [torch.zeros_like(x) for x in srcs]
                    V.remap_tensors(base, srcs, dsts, radius=size / 2, interpolation=1, rotations=quats)
                    s.synchronize()
                    for f in range(n):
                        if not np.array_equal(dsts[f].cpu().numpy(), want[f]):
                            errs.append((k, it, f))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs[:5]


def test_remap_sharded_on_the_devices_there_are(V, oracle_mod):
    """sharding.remap_sharded: the in-process multi-device dispatcher (worker thread + stream + staging ring
    per device).  One GPU on the test box: devices=[0] and devices=[0, 0] (two workers sharing the card)
    exercise the same code as 8 devices; SBS frames and (L, R) pairs, shared transformer, per-frame rotations."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n = 96
    frames = [noise_disc(n, 2 * n, 70 + f) for f in range(5)]
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    want = [O.apply_lr(spec, fr[:, :n], fr[:, n:], size_output=(80, 64), interpolation=1, radius="max") for fr in frames]
    for devices in ([0], [0, 0]):
        got = V.remap_sharded(CS.to_product(spec), frames, size_output=(80, 64), interpolation=1, radius="max", devices=devices)
        assert len(got) == 5 and all(np.array_equal(g, w) for g, w in zip(got, want)), devices
    # a single pair on two workers: one eye each
    got = V.remap_sharded(CS.to_product(spec), [(frames[0][:, :n], frames[0][:, n:])], size_output=(80, 64), interpolation=1,
                          radius="max", devices=[0, 0])
    assert np.array_equal(got[0], want[0])
    # per-frame, per-eye calibration rotations (BASELINE config 5)
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    rots = [tuple(CS.c5_spec(f, eye)[1][1] for eye in (0, 1)) for f in range(5)]
    got = V.remap_sharded(base, frames, size_output=(64, 64), interpolation=1, radius=n / 2, rotations=rots, devices=[0, 0])
    for f in range(5):
        w = np.concatenate([O.apply(CS.c5_spec(f, eye), [frames[f][:, eye * n:(eye + 1) * n]], size_output=(64, 64), interpolation=1,
                                    radius=n / 2)[0] for eye in (0, 1)], axis=1)
        assert np.array_equal(got[f], w), f


def test_remap_sharded_32_frames_with_rotations_two_workers(V, oracle_mod):
    """One rank's share of BASELINE config 5 -- 32 frames, a calibration rotation per eye -- through the dispatch an 8-GPU node
    runs (remap_sharded: frames over workers, host ring, rotations= batches; here two workers share the one card), every
    frame against the oracle."""
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n, w, h = 192, 160, 144
    frames = [noise_disc(n, 2 * n, 500 + f) for f in range(32)]
    base = T.EquirectangularEncoder() * T.Euclidean3DRotator((1, 0, 0, 0)) * T.FisheyeDecoder("equidistant")
    rots = [tuple(CS.c5_spec(f, eye)[1][1] for eye in (0, 1)) for f in range(32)]
    got = V.remap_sharded(base, frames, size_output=(w, h), interpolation=1, radius=n / 2, rotations=rots, devices=[0, 0])
    assert len(got) == 32
    for f in range(32):
        want = np.concatenate([O.apply(CS.c5_spec(f, eye), [frames[f][:, eye * n:(eye + 1) * n]], size_output=(w, h), interpolation=1,
                                       radius=n / 2)[0] for eye in (0, 1)], axis=1)
        assert np.array_equal(got[f], want), f


@pytest.mark.parametrize("split,workload", [("eyes", "C1"), ("bands", "C1"), ("frames", "C1S")])
def test_bench_split_modes_rehearsal(split, workload):
    """bench.py --split eyes / bands (ONE pair dealt to the ranks: SURVEY.md 8e) and the single-image workload: 2 ranks
    sharing the one card of the test box (V1C_BENCH_REHEARSAL=1: gloo for the barrier / MAX, never a measurement) print one
    JSON line with the right pixel count; at 1 rank the same modes are checked against the oracle by the bench itself."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = {**os.environ, "V1C_BENCH_REHEARSAL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--workload", workload, "--split", split, "--steps", "3", "--warmup", "1", "--traffic", "none", "--no-cold-extra"]
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", *common], env=env, capture_output=True, text=True, timeout=600)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["parity_vs_oracle"]["bytes_differing"] == 0, line["parity_vs_oracle"]
    assert line["config"]["split"] == split and line["scaling"] == ("weak" if split == "frames" else "strong")
    # what the process group reported travels in the line: a single process says so ...
    assert line["world_size_seen"] == 1 and line["backend"].startswith("none") and len(line["per_rank_kernel_ms"]) == 1
    if split == "frames":
        return
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", *common, "--no-cpu-baseline"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line2 = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line2["scaling"] == "strong" and "REHEARSAL" in line2["data"]
    # ... and N ranks show the backend and the world size torch.distributed saw (the rehearsal: gloo; the driver's run: nccl = RCCL)
    assert line2["world_size_seen"] == 2 and line2["backend"] == "gloo" and len(line2["per_rank_kernel_ms"]) == 2
    # strong scaling: the job is ONE pair whatever the rank count
    px = 2 * 2048 * 2048
    assert abs(line2["value"] * line2["ms_per_step"] * 1e3 - px) / px < 0.02


def test_remap_sharded_border_transparent_is_deterministic(V, oracle_mod):
    """BORDER_TRANSPARENT skips pixels: the staging ring of remap_sharded reuses its destination slots across groups, so
    they are zeroed per group like apply()'s destinations (the reference's cv2 leaves such pixels undefined; the engine and
    the oracle define them as 0).  More frames than ring slots, two workers, banded and unit-sharded paths agree."""
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    n = 96
    frames = [noise_disc(n, 2 * n, 170 + f) for f in range(9)]
    for f in frames:
        f[:, :] = np.maximum(f, 1)  # no zero pixel anywhere: a stale pixel of another frame cannot pass for a zero
    spec = [("equirect_enc", True), ("zoom", 0.6), CS.EQUI]  # zoomed out: a wide rim of skipped pixels
    want = [O.apply_lr(spec, fr[:, :n], fr[:, n:], size_output=(80, 64), interpolation=1, radius="max", border_mode=5) for fr in frames]
    assert any((w == 0).any() for w in want)
    for devices in ([0], [0, 0]):
        got = V.remap_sharded(CS.to_product(spec), frames, size_output=(80, 64), interpolation=1, radius="max", boarder_mode=5,
                              devices=devices)
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), devices
    got = V.remap_sharded(CS.to_product(spec), [(frames[0][:, :n], frames[0][:, n:])], size_output=(80, 64), interpolation=1,
                          radius="max", boarder_mode=5, devices=[0] * 8)
    assert np.array_equal(got[0], want[0])


# (source size, output size, radius, kernel family for the equidistant chain, for the polynomial chain): the mirror kernels need an
# output whose rows pair up about the equator (H a multiple of 32), rays close enough for one table entry per lane (>= ~416 px for an
# m-table, ~1000 for the w-table of the polynomial chain) and boxes the DMA buffers hold; everything else is the pair kernel's
MIRROR_GEOMETRIES = [
    ((300, 300), (448, 448), 150.0, "mirror", "tile"),      # smallest mirror grid: 28 tile rows
    ((512, 640), (544, 576), 250.0, "mirror", "tile"),      # non-square both ways, source wider than high
    ((1000, 1000), (1028, 1024), 470.0, "mirror", "mirror"),  # width not a multiple of the tile (64): ragged last tile column
    ((1000, 1000), (1028, 1056), 470.0, "mirror", "mirror"),  # 1056 = 33 * 32: odd number of tile-row pairs
    ((257, 263), (128, 100), 120.0, "tile", "tile"),        # 100 rows: no mirror launch, plain pair kernel
    ((1800, 1800), (512, 512), 900.0, "tile", "tile"),      # 3.5 x minification: boxes beyond the DMA buffers, plain pair kernel
    ((120, 120), (1024, 1024), 60.0, "mirror", "mirror"),   # 8.5 x magnification: boxes of a few rows
    ((900, 1200), (640, 640), 450.0, "mirror", "tile"),     # 2 x minification in x: boxes around the buffer size, rest tiles
    ((640, 640), (640, 640), 200.0, "mirror", "tile"),      # radius well inside the source: most rays leave the image circle
]


@pytest.mark.parametrize("src_hw,out_wh,radius,kind_m,kind_w", MIRROR_GEOMETRIES)
def test_mirror_pair_launch_geometries(V, oracle_mod, dev, src_hw, out_wh, radius, kind_m, kind_w):
    """apply_lr pairs of unrotated bilinear chains take k_ray_lin3_pair_mirror_seq (a tile and its mirror image about the equator
    from one set of coordinates, boxes by LDS-DMA; ineligible tiles through the pair code in the same launch): every output byte
    against the oracle, m-table and w-table chains -- and the launch IS the mirror launch where the table says so (round 4: the
    geometries of round 3 turned out to be served by the pair kernel, v1c_plan_last_launch tells).
    (The register-staged k_ray_lin3_pair_mirror and other box-buffer sizes: test_kernel_variants_bit_exact.)"""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    left, right = noise_disc(*src_hw, 21), noise_disc(*src_hw, 22)
    left[::9, ::7] = 255
    for spec, kind in (([("equirect_enc", True), CS.EQUI], kind_m), ([("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI], kind_w)):
        want = O.apply_lr(spec, left, right, size_output=out_wh, interpolation=1, radius=radius, border_value=(5, 6, 7))
        got = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev),
                                 size_output=out_wh, interpolation=1, radius=radius, boarder_value=(5, 6, 7)).cpu().numpy()
        assert remapper.last_launch_kinds() == [kind], (remapper.last_launch_kinds(), spec)
        assert np.array_equal(got, want), (spec, int((got != want).sum()))


@pytest.mark.parametrize("src_hw,out_wh,radius,kind_m,kind_w", MIRROR_GEOMETRIES)
def test_mirror_single_image_launch_geometries(V, oracle_mod, dev, src_hw, out_wh, radius, kind_m, kind_w):
    """apply() of ONE image of an unrotated bilinear chain (BASELINE config 1) takes the one-eye instantiation of
    k_ray_lin3_pair_mirror_raw (tile + mirror image from one set of coordinates, two boxes by LDS-DMA): every byte vs the
    oracle on the pair launch's geometries, m-table and w-table chains, a pitched source view."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    wide = noise_disc(src_hw[0], src_hw[1] + 8, 23)
    img = wide[:, 4:4 + src_hw[1]]  # a column slice: pitch != 3 * width, rows dword-aligned (12-byte shift)
    for spec, kind in (([("equirect_enc", True), CS.EQUI], kind_m), ([("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI], kind_w)):
        want = O.apply(spec, [np.ascontiguousarray(img)], size_output=out_wh, interpolation=1, radius=radius, border_value=(5, 6, 7))[0]
        src = torch.from_numpy(wide).to(dev)[:, 4:4 + src_hw[1]]
        dst = torch.empty((out_wh[1], out_wh[0], 3), dtype=torch.uint8, device=dev)
        V.remap_tensors(CS.to_product(spec), [src], [dst], radius=radius, interpolation=1, boarder_value=(5, 6, 7))
        assert remapper.last_launch_kinds() == [kind], (remapper.last_launch_kinds(), spec)
        got = dst.cpu().numpy()
        assert np.array_equal(got, want), (spec, int((got != want).sum()))


def test_remap_sharded_splits_rows_when_devices_outnumber_eyes(V, oracle_mod):
    """a single pair on 4 / 8 GPUs = bands of output rows per GPU (each needs the whole source eye, no
    exchange).  Eight workers on the one card of the test box: 2 eyes x 4 bands, assembled rows equal the oracle's."""
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    left, right = noise_disc(300, 300, 31), noise_disc(300, 300, 32)
    spec = [("equirect_enc", True), ("rot", CS.ry(0.2)), ("poly", [0, 1, -0.1]), CS.EQUI]
    want = O.apply_lr(spec, left, right, size_output=(256, 208), interpolation=4, radius="max")
    for ndev in (4, 8):
        got = V.remap_sharded(CS.to_product(spec), [(left, right)], size_output=(256, 208), interpolation=4, radius="max", devices=[0] * ndev)
        assert np.array_equal(got[0], want), ndev


_DMA_KINDS: list = []  # (what, kernel family) of the pair / single-image launches of the test below


@pytest.mark.parametrize("seed", range(10))
def test_dma_kernels_random_geometries(V, oracle_mod, dev, seed):
    """The LDS-DMA kernels (k_ray_lin3_pair_mirror_raw for pairs, k_ray_lin3_batch_lean_raw for batches) on seeded random
    geometries: sources that are column halves of one SBS frame (pitched views whose rows end inside the frame / at its
    last byte -- the 16-byte units of a box row may read up to 12 bytes past the box, never past the allocation),
    magnification and minification, radius, polynomial.  Every output byte against the oracle."""
    from vr180_convert_amd import remapper
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    rng = np.random.default_rng(900 + seed)
    h_in = int(rng.integers(6, 40)) * 16 + int(rng.integers(0, 16))
    w_in = int(rng.integers(6, 40)) * 16 + int(rng.integers(0, 4)) * 4  # (views of an SBS frame stay dword-aligned: w_in % 4 == 0)
    # (outputs the mirror launch takes: rows pair up about the equator, at least ~416 px either way, not wider than high by more than
    #  a tile -- rays beyond 90 degrees of longitude need the fix-up pass, which rules the mirror kernels out)
    out_h = int(rng.integers(14, 23)) * 32
    out_w = (out_h // 64 - int(rng.integers(0, 2))) * 64 + int(rng.choice([0, 0, 4, 60]))
    radius = float(rng.uniform(0.35, 0.75 if seed % 4 == 2 else 0.55)) * min(h_in, w_in)  # (beyond 0.5: rays leave the source)
    spec = [("equirect_enc", True)]
    poly = seed % 4 == 3  # (a w-table: one entry per lane only from ~1000 px on -> the pair kernel here)
    if poly:
        spec.append(("poly", [0, 1, float(rng.uniform(-0.2, 0.1))]))
    spec.append(CS.EQUI)
    frame = noise_disc(h_in, 2 * w_in, 40 + seed)
    frame[::7, ::5] = 255
    frame[-1, -16:] = 200  # the frame's last bytes are not black: an over-read would show
    left, right = frame[:, :w_in], frame[:, w_in:]
    want = O.apply_lr(spec, left, right, size_output=(out_w, out_h), interpolation=1, radius=radius, border_value=(9, 8, 7))
    fr = torch.from_numpy(frame).to(dev)
    got = V.apply_lr_tensors(CS.to_product(spec), fr[:, :w_in], fr[:, w_in:], size_output=(out_w, out_h), interpolation=1,
                             radius=radius, boarder_value=(9, 8, 7)).cpu().numpy()
    kinds = remapper.last_launch_kinds()
    assert kinds in (["mirror"], ["tile"]), (kinds, spec, out_w, out_h)  # (boxes beyond the DMA buffers: the pair kernel)
    _DMA_KINDS.append(("pair", kinds[0]))
    assert np.array_equal(got, want), ("pair", h_in, w_in, out_w, out_h, int((got != want).sum()))
    # the same map over a batch of 5 views (3 + 2 units per workgroup): the batch kernel
    frames = [noise_disc(h_in, 2 * w_in, 60 + 5 * seed + f) for f in range(3)]
    srcs_np = [f[:, :w_in] for f in frames] + [f[:, w_in:] for f in frames[:2]]
    dev_frames = [torch.from_numpy(f).to(dev) for f in frames]
    srcs = [d[:, :w_in] for d in dev_frames] + [d[:, w_in:] for d in dev_frames[:2]]
    dsts = [torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=dev) for _ in srcs]
    V.remap_tensors(CS.to_product(spec), srcs, dsts, radius=radius, interpolation=1)
    assert remapper.last_launch_kinds() == ["batch"], remapper.last_launch_kinds()
    wants = O.apply(spec, srcs_np, size_output=(out_w, out_h), interpolation=1, radius=radius)
    for k in range(5):
        assert np.array_equal(dsts[k].cpu().numpy(), wants[k]), ("batch", k, h_in, w_in, out_w, out_h)
    # ONE image of the same map: the one-eye instantiation of the mirror kernel (its boxes have a pair's four buffers)
    dst = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=dev)
    V.remap_tensors(CS.to_product(spec), [fr[:, w_in:]], [dst], radius=radius, interpolation=1, boarder_value=(9, 8, 7))
    _DMA_KINDS.append(("single", remapper.last_launch_kinds()[0]))
    want1 = O.apply(spec, [np.ascontiguousarray(right)], size_output=(out_w, out_h), interpolation=1, radius=radius, border_value=(9, 8, 7))[0]
    assert np.array_equal(dst.cpu().numpy(), want1), ("single", h_in, w_in, out_w, out_h)
    # the pair with INTER_NEAREST (bilinear tile kernels, coordinates 32 * cvRound(x)) and with bilinear BORDER_TRANSPARENT (store mask
    # in the patch path; the destination starts from a pattern)
    want = O.apply_lr(spec, left, right, size_output=(out_w, out_h), interpolation=0, radius=radius, border_mode=int(seed % 5), border_value=(9, 8, 7))
    got = V.apply_lr_tensors(CS.to_product(spec), fr[:, :w_in], fr[:, w_in:], size_output=(out_w, out_h), interpolation=0, radius=radius,
                             boarder_mode=int(seed % 5), boarder_value=(9, 8, 7)).cpu().numpy()
    assert np.array_equal(got, want), ("nearest", seed % 5, h_in, w_in, out_w, out_h, int((got != want).sum()))
    xm, ym = O.get_map(spec, radius=radius, size_input=(h_in, w_in), size_output=(out_w, out_h))
    fill = np.full((out_h, out_w, 3), (1, 2, 3), np.uint8)
    wt = [O.remap(np.ascontiguousarray(e), xm, ym, 1, 5, 0, dst=fill.copy()) for e in (left, right)]
    dts = [torch.from_numpy(fill.copy()).to(dev) for _ in range(2)]
    V.remap_tensors(CS.to_product(spec), [fr[:, :w_in], fr[:, w_in:]], dts, radius=radius, interpolation=1, boarder_mode=5)
    for e in range(2):
        assert np.array_equal(dts[e].cpu().numpy(), wt[e]), ("transparent", e, h_in, w_in, out_w, out_h)


def test_dma_kernels_random_geometries_reached_the_mirror_kernels():
    """at least four of the ten seeds above ran their pair AND their single image through the mirror kernels"""
    if len(_DMA_KINDS) < 20:
        pytest.skip("runs behind all ten seeds of test_dma_kernels_random_geometries")
    assert sum(k == ("pair", "mirror") for k in _DMA_KINDS) >= 4 and sum(k == ("single", "mirror") for k in _DMA_KINDS) >= 4, _DMA_KINDS


@pytest.mark.parametrize("seed", range(8))
def test_pairs_random_interpolation_border_rotation(V, oracle_mod, dev, seed):
    """apply_lr pairs over interpolation x border mode x rotation x size, seeded: the lane-remapped K x K pair gather (bicubic,
    Lanczos4), the pair kernels with and without a rotation, footprints leaving the source under every border mode."""
    from vr180_convert_amd import _abi
    from vr180_convert_amd.synth import noise_disc

    O = oracle_mod
    rng = np.random.default_rng(1700 + seed)
    interp = int(rng.choice([0, 1, 2, 4]))
    border = int(rng.choice([_abi.BORDER_CONSTANT, _abi.BORDER_REPLICATE, _abi.BORDER_REFLECT, _abi.BORDER_WRAP, _abi.BORDER_REFLECT_101]))
    h_in, w_in = int(rng.integers(40, 300)), int(rng.integers(40, 300))
    out_w, out_h = int(rng.integers(1, 6)) * 64 + int(rng.choice([0, 4, 33])), int(rng.integers(2, 10)) * 16 + int(rng.choice([0, 0, 5]))
    radius = float(rng.uniform(0.3, 1.2)) * min(h_in, w_in) / 2
    spec = [("equirect_enc", True)]
    if rng.random() < 0.5:
        spec.append(("rot", CS.ry(float(rng.uniform(-0.8, 0.8)))))
    if rng.random() < 0.5:
        spec.append(("poly", [0, 1, float(rng.uniform(-0.2, 0.1))]))
    spec.append(CS.EQUI)
    left, right = noise_disc(h_in, w_in, 300 + seed), noise_disc(h_in, w_in, 400 + seed)
    left[::5, ::3] = 255
    want = O.apply_lr(spec, left, right, size_output=(out_w, out_h), interpolation=interp, radius=radius, border_mode=border,
                      border_value=(3, 200, 77))
    got = V.apply_lr_tensors(CS.to_product(spec), torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev),
                             size_output=(out_w, out_h), interpolation=interp, radius=radius, boarder_mode=border,
                             boarder_value=(3, 200, 77)).cpu().numpy()
    assert np.array_equal(got, want), (interp, border, h_in, w_in, out_w, out_h, int((got != want).sum()))
