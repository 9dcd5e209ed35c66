"""The boundary a maintainer of the reference would bind, exercised as documented -- on the GPU, through raw ctypes:

* the ctypes stub of INTEGRATION.md section 1 is extracted from the document and executed verbatim (only the library path is
  filled in): ``v1c_remap_fused`` on a C2 cut (bilinear), a C4 cut (Lanczos4 with the Euler rotation) and into the pitched halves of
  one side-by-side buffer, byte for byte against the oracle;
* the one-shot entry point's plan cache is bounded (a sweep over 100 radii keeps at most 32 plans and gives the device memory back);
* ``v1c_get_radius`` / ``v1c_get_radius_async`` -- the reference's get_radius (transformer.py:108-140) as a device kernel -- against
  the values the REFERENCE returned for the fixtures of tests/golden/radius.npz, sign quirk and IndexError included.
"""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import chainspecs as CS

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def stub():
    """namespace of the code block INTEGRATION.md tells a maintainer to add as vr180_convert/_hip.py"""
    from vr180_convert_amd import _native

    _native.lib()
    text = (ROOT / "INTEGRATION.md").read_text()
    m = re.search(r"```python\n(# vr180_convert/_hip\.py.*?)```", text, flags=re.S)
    assert m, "INTEGRATION.md section 1 lost its stub"
    code = m.group(1).replace('C.CDLL("libvr180remap.so")', f'C.CDLL("{_native.LIB_PATH}")')
    ns: dict = {}
    exec(compile(code, "INTEGRATION.md:_hip.py", "exec"), ns)  # noqa: S102 - the repository's own documentation
    return ns


def _stub_chain(stub, spec, radius, size_input, size_output):
    """the lowered chain in the STUB's own ctypes types, op by op from the product's lowering (same POD layout)"""
    from vr180_convert_amd.chain import lower_for_get_map

    mine = lower_for_get_map(CS.to_product(spec), radius=radius, size_input=size_input, size_output=size_output)
    ch = stub["Chain"]()
    C.memmove(C.byref(ch), C.byref(mine), C.sizeof(ch))
    assert C.sizeof(ch) == C.sizeof(mine) and ch.n_ops == mine.n_ops
    return ch


@pytest.mark.parametrize("case,interp", [("C2", 1), ("C4", 4)])
def test_integration_stub_remap_fused_vs_oracle(stub, oracle_mod, case, interp):
    from vr180_convert_amd.synth import noise_disc

    dev = torch.device("cuda", 0)
    spec = CS.FULL_CASES[case][0]
    n = 640
    img = noise_disc(n, n, 11)
    src = torch.from_numpy(img).to(dev)
    dst = torch.zeros((n, n, 3), dtype=torch.uint8, device=dev)
    ch = _stub_chain(stub, spec, n / 2, (n, n), (n, n))
    stub["remap_fused"](0, torch.cuda.current_stream(dev).cuda_stream, src.data_ptr(), (n, n), src.stride(0), dst.data_ptr(), (n, n),
                        dst.stride(0), ch, interp, 0, (0, 0, 0, 0))
    torch.cuda.synchronize()
    want = oracle_mod.apply(spec, [img], size_output=(n, n), interpolation=interp, radius=n / 2)[0]
    assert np.array_equal(dst.cpu().numpy(), want), int((dst.cpu().numpy() != want).sum())


def test_integration_stub_writes_the_pitched_halves_of_one_sbs_buffer(stub, oracle_mod):
    """apply_lr's np.concatenate (remapper.py:517-518) through the stub: each eye into its half of one (H, 2W, 3) buffer, a non-zero
    border value, non-square output"""
    from vr180_convert_amd.synth import noise_disc

    dev = torch.device("cuda", 0)
    spec = [("equirect_enc", True), ("poly", [0, 1, -0.1]), CS.EQUI]
    h_in, w_in, w, h = 480, 560, 352, 288
    eyes = [noise_disc(h_in, w_in, 21), noise_disc(h_in, w_in, 22)]
    sbs = torch.zeros((h, 2 * w, 3), dtype=torch.uint8, device=dev)
    ch = _stub_chain(stub, spec, 230.5, (h_in, w_in), (w, h))
    for e, img in enumerate(eyes):
        src = torch.from_numpy(img).to(dev)
        half = sbs[:, e * w:(e + 1) * w]
        stub["remap_fused"](0, torch.cuda.current_stream(dev).cuda_stream, src.data_ptr(), (h_in, w_in), src.stride(0), half.data_ptr(),
                            (w, h), sbs.stride(0), ch, 1, 0, (7, 0, 0, 0))
    torch.cuda.synchronize()
    want = oracle_mod.apply(spec, eyes, size_output=(w, h), interpolation=1, radius=230.5, border_value=7)
    got = sbs.cpu().numpy()
    for e in range(2):
        assert np.array_equal(got[:, e * w:(e + 1) * w], want[e]), e


def test_fused_plan_cache_is_bounded(stub, oracle_mod):
    """100 radii through v1c_remap_fused: at most 32 plans stay cached, the device memory of the evicted ones comes back, and an
    evicted radius is simply planned again (same bytes)."""
    from vr180_convert_amd import _native
    from vr180_convert_amd.synth import noise_disc

    L = _native.lib()
    dev = torch.device("cuda", 0)
    spec = [("equirect_enc", True), CS.EQUI]
    n = 512
    img = noise_disc(n, n, 5)
    src = torch.from_numpy(img).to(dev)
    dst = torch.zeros((n, n, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def run(radius):
        ch = _stub_chain(stub, spec, radius, (n, n), (n, n))
        stub["remap_fused"](0, stream, src.data_ptr(), (n, n), src.stride(0), dst.data_ptr(), (n, n), dst.stride(0), ch, 1, 0, (0, 0, 0, 0))

    for k in range(40):  # fill the cache
        run(200.0 + k)
    torch.cuda.synchronize()
    assert L.v1c_fused_cache_size() <= 32
    free0 = torch.cuda.mem_get_info(0)[0]
    for k in range(100):
        run(100.0 + 0.5 * k)
    torch.cuda.synchronize()
    assert L.v1c_fused_cache_size() <= 32
    free1 = torch.cuda.mem_get_info(0)[0]
    # a plan of this size holds < 1 MB; 100 leaked plans would be tens of MB
    assert free0 - free1 < 8 << 20, (free0, free1)
    first = dst.clone()
    run(200.0)  # evicted long ago: planned again
    torch.cuda.synchronize()
    want = oracle_mod.apply(spec, [img], size_output=(n, n), interpolation=1, radius=200.0)[0]
    assert np.array_equal(dst.cpu().numpy(), want)
    assert not torch.equal(first, dst)


def _dev_radius(L, im, threshold=10):
    r = C.c_double()
    rc = L.v1c_get_radius(0, torch.cuda.current_stream(im.device).cuda_stream, im.data_ptr(), im.shape[0], im.shape[1], im.stride(0), im.shape[2],
                          threshold, C.byref(r))
    return rc, r.value


def test_device_get_radius_equals_the_references_values(golden_dir):
    from vr180_convert_amd import _abi, _native

    L = _native.lib()
    dev = torch.device("cuda", 0)
    g = np.load(golden_dir / "radius.npz")
    for name, thr, key in (("landscape_img", 10, "landscape_radius"), ("portrait_img", 10, "portrait_radius"), ("noisy_img", 10, "noisy_radius"),
                           ("noisy_img", 25, "thr_radius")):
        im = torch.from_numpy(np.ascontiguousarray(g[name])).to(dev)
        rc, r = _dev_radius(L, im, thr)
        assert rc == 0 and r == float(g[key]), (name, thr, r, float(g[key]))
        # the asynchronous form: the value stays on the device
        out = torch.full((2,), -1.0, dtype=torch.float64, device=dev)
        assert L.v1c_get_radius_async(0, torch.cuda.current_stream(dev).cuda_stream, im.data_ptr(), im.shape[0], im.shape[1], im.stride(0),
                                      im.shape[2], thr, out.data_ptr()) == 0
        assert out.cpu().tolist() == [float(g[key]), 0.0]
    assert float(g["landscape_radius"]) < 0  # the sign quirk travels
    # column-sliced view (apply_lr's split halves), grayscale and BGRA
    wide = torch.from_numpy(np.ascontiguousarray(np.concatenate([g["landscape_img"], g["noisy_img"]], axis=1))).to(dev)
    w = g["landscape_img"].shape[1]
    assert _dev_radius(L, wide[:, :w])[1] == float(g["landscape_radius"])
    assert _dev_radius(L, wide[:, w:])[1] == float(g["noisy_radius"])
    from vr180_convert_amd.chain import get_radius as host_radius

    for cn in (1, 4):
        a = np.ascontiguousarray(np.repeat(g["noisy_img"][:, :, :1], cn, axis=2))
        assert _dev_radius(L, torch.from_numpy(a).to(dev))[1] == host_radius(a)
    # no black border: the reference raises IndexError
    full = torch.full((64, 80, 3), 90, dtype=torch.uint8, device=dev)
    rc, _ = _dev_radius(L, full)
    assert rc == _abi.E_INVALID and b"no black border" in L.v1c_last_error()
    out = torch.zeros((2,), dtype=torch.float64, device=dev)
    assert L.v1c_get_radius_async(0, torch.cuda.current_stream(dev).cuda_stream, full.data_ptr(), 64, 80, full.stride(0), 3, 10, out.data_ptr()) == 0
    o = out.cpu().tolist()
    assert np.isnan(o[0]) and o[1] == 1.0
    # recorded into a graph
    im = torch.from_numpy(np.ascontiguousarray(g["landscape_img"])).to(dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        L.v1c_get_radius_async(0, s.cuda_stream, im.data_ptr(), im.shape[0], im.shape[1], im.stride(0), 3, 10, out.data_ptr())
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        L.v1c_get_radius_async(0, s.cuda_stream, im.data_ptr(), im.shape[0], im.shape[1], im.stride(0), 3, 10, out.data_ptr())
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert out.cpu().tolist() == [float(g["landscape_radius"]), 0.0]
