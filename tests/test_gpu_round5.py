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
    "rotated": ([("equirect_enc", True), ("rot", CS.ry(0.5)), CS.EQUI], (512, 512), (640, 512), 262.0, 37),
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
