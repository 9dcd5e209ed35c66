"""Command line (reference cli.py:116-559, tests/test_cli.py, tests/test_dunder_main.py): the parts that need no
GPU -- help texts, option parsing, file-name and time-match rules, the swap command -- and, gpu-marked, ``lr`` / ``s``
end to end against the oracle."""
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np
import pytest
from typer.testing import CliRunner

import chainspecs as CS
from vr180_convert_amd import _io, cli

runner = CliRunner()
ROOT = Path(__file__).resolve().parents[1]


def test_help_of_every_command():
    assert runner.invoke(cli.app, ["--help"]).exit_code == 0
    for cmd in ("lr", "s", "swap", "xmp"):
        r = runner.invoke(cli.app, [cmd, "--help"])
        assert r.exit_code == 0, r.stdout


def test_dunder_main_prints_the_prog_name():
    r = subprocess.run([sys.executable, "-m", "vr180_convert_amd", "--help"], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and "vr180-convert" in r.stdout


def test_transformer_expressions_evaluate_in_the_cli_namespace():
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.chain import MultiTransformer

    t = cli.parse_transformer("")
    assert isinstance(t, MultiTransformer) and [type(s).__name__ for s in t.transformers] == ["EquirectangularEncoder", "InverseTransformer"]
    # the expressions of the reference's CLI tests (tests/test_cli.py:32-34, 53-55)
    t = cli.parse_transformer('FisheyeEncoder("equidistant") * Euclidean3DRotator(from_rotation_vector([0, np.pi / 4, 0])) * FisheyeDecoder("equidistant")')
    assert isinstance(t.transformers[1], T.Euclidean3DRotator)
    np.testing.assert_allclose(t.transformers[1].matrix, [[np.cos(np.pi / 4), 0, np.sin(np.pi / 4)], [0, 1, 0], [-np.sin(np.pi / 4), 0, np.cos(np.pi / 4)]], atol=1e-15)
    t = cli.parse_transformer("EquirectangularEncoder() * PolynomialScaler([0, 1, -0.1]) * FisheyeDecoder('equidistant')")
    assert list(t.transformers[1].coefs_reverse) == [0, 1, -0.1]


def test_option_parsing():
    assert cli.parse_size("4096x2048") == (4096, 2048)
    assert cli.parse_radius("auto") == "auto" and cli.parse_radius("max") == "max" and cli.parse_radius("512.5") == 512.5
    for spelling in ("INTER_LANCZOS4", "inter_lanczos4", "lanczos4"):
        assert cli._flag(spelling, cli._INTERPOLATIONS, "inter_", "interpolation") == 4
    assert cli._flag("border_reflect_101", cli._BORDERS, "border_", "border mode") == 4
    with pytest.raises(Exception):
        cli._flag("inter_bogus", cli._INTERPOLATIONS, "inter_", "interpolation")


def test_output_names(tmp_path):
    left, right = tmp_path / "a" / "L001.jpg", tmp_path / "b" / "R001.jpg"
    assert cli.output_path(Path(""), left, right, "") == tmp_path / "a" / "L001-R001.png"
    (tmp_path / "out").mkdir()
    assert cli.output_path(tmp_path / "out", left, right, "-12345678") == tmp_path / "out" / "L001-R001-12345678.png"
    assert cli.output_path(tmp_path / "x.jpg", left, right, "") == tmp_path / "x.jpg"
    tag = cli.unique_suffix("T", "4096x4096", "inter_lanczos4", "border_constant", 0, "auto", False, 0.0, False)
    assert len(tag) == 9 and tag[0] == "-" and tag != cli.unique_suffix("T", "2048x2048", "inter_lanczos4", "border_constant", 0, "auto", False, 0.0, False)


def test_time_matched_search(tmp_path):
    """One argument may be a directory: the image whose mtime is closest to the other eye's, after the clock
    offset, is taken (cli.py:178-216)."""
    d = tmp_path / "right_cam"
    d.mkdir()
    left = tmp_path / "left.jpg"
    left.write_bytes(b"x")
    now = time.time()
    os.utime(left, (now, now))
    for name, dt in (("r0.jpg", -30.0), ("r1.jpg", 2.0), ("r2.jpg", 11.0), ("r1.png", 0.0)):
        (d / name).write_bytes(b"x")
        os.utime(d / name, (now + dt, now + dt))
    assert cli.resolve_pair(left, d, 0.0) == (left, d / "r1.jpg")        # same suffix only
    assert cli.resolve_pair(left, d, 10.0) == (left, d / "r2.jpg")       # right clock 10 s ahead
    assert cli.resolve_pair(d, left, 0.0) == (d / "r1.jpg", left)
    with pytest.raises(ValueError):
        cli.resolve_pair(d, d, 0.0)
    with pytest.raises(ValueError):
        cli.closest_in_time(tmp_path / "right_cam", tmp_path / "left.tif" if (tmp_path / "left.tif").write_bytes(b"x") else left, 0.0)


def test_swap_command(tmp_path):
    img = np.arange(6 * 8 * 3, dtype=np.uint8).reshape(6, 8, 3)
    p = tmp_path / "sbs.png"
    _io.imwrite(p, img)
    r = runner.invoke(cli.app, ["swap", str(p), "--no-overwrite"])
    assert r.exit_code == 0, r.stdout
    assert np.array_equal(_io.imread(tmp_path / "sbs.swap.png"), np.hstack([img[:, 4:], img[:, :4]]))
    assert np.array_equal(_io.imread(p), img)
    assert runner.invoke(cli.app, ["swap", str(p)]).exit_code == 0
    assert np.array_equal(_io.imread(p), np.hstack([img[:, 4:], img[:, :4]]))


def test_automatch_split_and_missing_cv2():
    from vr180_convert_amd import transformer as T

    t = T.EquirectangularEncoder() * T.PolynomialScaler([0, 1, -0.1]) * T.FisheyeDecoder("equidistant")
    head, tail = cli.split_at_first_encoder(t)
    assert [type(s).__name__ for s in head.transformers] == ["EquirectangularEncoder"]
    assert [type(s).__name__ for s in tail.transformers] == ["PolynomialScaler", "InverseTransformer"]
    with pytest.raises(ValueError):
        cli.split_at_first_encoder(T.ZoomTransformer(2.0))


class _FakeCv2:
    """Records what the CLI's OpenCV front ends (calibration_cv.py) hand to cv2: a stand-in module, no arithmetic."""
    WND_PROP_FULLSCREEN, WINDOW_FULLSCREEN, EVENT_LBUTTONDOWN = 0, 1, 1

    def __init__(self, clicks=()):
        self.calls, self.clicks, self.callback, self.shown = [], list(clicks), None, 0

    # --- AKAZE + BFMatcher
    def resize(self, image, size):
        self.calls.append(("resize", size))
        return np.zeros((size[1], size[0], 3), np.uint8)

    def AKAZE_create(self):
        fake = self

        class Det:
            def detectAndCompute(self, image, mask):
                fake.calls.append(("detect", image.shape))
                n = 12
                kps = [type("KP", (), {"pt": (10.0 + 3 * i, 20.0 + 2 * i)})() for i in range(n)]
                return kps, np.zeros((n, 61), np.uint8)
        return Det()

    def BFMatcher(self):
        class BF:
            def match(self, d1, d2):
                return [type("M", (), {"queryIdx": i, "trainIdx": (i + 1) % len(d2)})() for i in range(len(d1))]
        return BF()

    def drawMatches(self, img_l, kp_l, img_r, kp_r, matches, out):
        self.calls.append(("drawMatches", len(matches)))
        return np.full((8, 16, 3), 7, np.uint8)

    # --- window
    def imread(self, p):
        return np.zeros((4, 4, 3), np.uint8)

    def namedWindow(self, *a):
        self.calls.append(("namedWindow",) + a)

    def setWindowProperty(self, *a):
        pass

    def setMouseCallback(self, title, cb):
        self.callback = cb

    def imshow(self, title, im):
        self.shown += 1

    def waitKey(self, ms):
        x, y = self.clicks.pop(0)
        self.callback(self.EVENT_LBUTTONDOWN, x, y, 0, None)

    def destroyAllWindows(self):
        self.calls.append(("destroyAllWindows",))


def test_automatch_front_ends_report_a_missing_opencv(tmp_path, monkeypatch):
    """Without cv2 (the GPU image): --automatch fm / gui say which package is missing; --savematch without fm is ignored like in the
    reference (cli.py:365), with a warning -- never a usage error: scripts that always pass the flag keep working."""
    import builtins

    from vr180_convert_amd.synth import pattern

    real_import = builtins.__import__
    monkeypatch.setattr(builtins, "__import__", lambda name, *a, **k: (_ for _ in ()).throw(ImportError("no cv2")) if name == "cv2" else real_import(name, *a, **k))
    monkeypatch.delitem(sys.modules, "cv2", raising=False)
    img = tmp_path / "a.png"
    _io.imwrite(img, pattern(64, 64))
    for opt in ("fm0.5", "gui"):
        r = runner.invoke(cli.app, ["lr", str(img), str(img), "--radius", "max", "--size", "32x32", "--automatch", opt])
        assert r.exit_code != 0 and "OpenCV" in (r.stdout + str(r.exception) + getattr(r, "stderr", "")), (opt, r.stdout, r.exception)
    seen = []
    monkeypatch.setattr(cli.LOG, "warning", lambda msg, *a, **k: seen.append(str(msg)))
    from vr180_convert_amd import remapper

    monkeypatch.setattr(remapper, "apply_lr", lambda *a, **k: seen.append("apply_lr ran"))
    r = runner.invoke(cli.app, ["lr", str(img), str(img), "--radius", "max", "--size", "32x32", "--savematch"])
    assert r.exit_code == 0, (r.stdout, r.exception)
    assert any("--savematch ignored" in m for m in seen) and "apply_lr ran" in seen


def test_automatch_fm_gui_and_savematch_glue_with_a_stand_in_cv2(tmp_path, monkeypatch):
    """With cv2 importable the front ends work as in the reference (cli.py:255-304, 362-365): `fm<scale>` resizes, matches, un-scales
    the points, fits robustly and -- with --savematch -- writes <out>.match<ext>; `gui<n>` collects 2 n clicks, even ones for the left
    eye.  cv2 is a recording stand-in: the glue is what is under test."""
    import vr180_convert_amd.remapper as R
    from vr180_convert_amd import calibration_cv as CV
    from vr180_convert_amd.synth import pattern

    fake = _FakeCv2()
    monkeypatch.setitem(sys.modules, "cv2", fake)
    # match_points: scaled detection, points back in original pixels
    a = np.zeros((100, 200, 3), np.uint8)
    p1, p2, kp1, kp2, matches, s1, s2 = CV.match_points(a, a, scale=0.5)
    assert fake.calls[:2] == [("resize", (100, 50)), ("resize", (100, 50))] and s1.shape == (50, 100, 3)
    assert np.allclose(p1[0], (10.0 / 0.5, 20.0 / 0.5)) and np.allclose(p2[0], (13.0 / 0.5, 22.0 / 0.5)) and len(matches) == 12
    # the window: one click per image, in order
    fake.clicks = [(1, 2), (3, 4), (5, 6), (7, 8)]
    assert CV.pick_points_gui([a, a, a, a]) == [(1, 2), (3, 4), (5, 6), (7, 8)] and fake.shown == 4
    # the CLI around them
    calls = []
    monkeypatch.setattr(R, "apply_lr", lambda *args, **k: calls.append(k))
    l, r_ = tmp_path / "L.png", tmp_path / "R.png"
    _io.imwrite(l, pattern(128, 128)), _io.imwrite(r_, pattern(128, 128))
    out = tmp_path / "o.png"
    fake.calls.clear()
    res = runner.invoke(cli.app, ["lr", str(l), str(r_), "--radius", "max", "--size", "64x64", "--automatch", "fm0.25", "--savematch", "--out-path", str(out)])
    assert res.exit_code == 0, (res.stdout, res.exception)
    assert ("resize", (32, 32)) in fake.calls and any(c[0] == "drawMatches" and c[1] <= 100 for c in fake.calls)
    assert (tmp_path / "o.match.png").exists() and isinstance(calls[0]["transformer"] if "transformer" in calls[0] else None, (tuple, type(None)))
    fake.clicks = [(60, 60), (62, 61), (70, 50), (73, 52), (40, 80), (42, 83)]
    fake.shown = 0
    res = runner.invoke(cli.app, ["lr", str(l), str(r_), "--radius", "max", "--size", "64x64", "--automatch", "gui3", "--out-path", str(out)])
    assert res.exit_code == 0 and fake.shown == 6, (res.stdout, res.exception)


def test_xmp_command_glue_with_a_stand_in_libxmp(tmp_path, monkeypatch):
    """The xmp command (cli.py:439-540): left half written as <name>.xmp<ext>, the right half embedded base64 in GImage:Data, the
    GPano block of a (height, width) side-by-side frame.  libxmp is a recording stand-in; without it the command names the package."""
    import base64
    import types

    from vr180_convert_amd import calibration_cv as CV
    from vr180_convert_amd.synth import pattern

    sbs = np.concatenate([pattern(64, 96), pattern(64, 96)[:, ::-1]], axis=1)
    p = tmp_path / "pair.png"
    _io.imwrite(p, sbs)
    res = runner.invoke(cli.app, ["xmp", str(p)])
    if "libxmp" not in sys.modules:
        assert res.exit_code != 0 and "python-xmp-toolkit" in (res.stdout + str(res.exception) + getattr(res, "stderr", ""))
    props: dict = {}
    state = {"files": [], "put": 0, "closed": 0, "ns": []}

    class XMPMeta:
        @staticmethod
        def register_namespace(uri, prefix):
            state["ns"].append((uri, prefix))

        def set_property(self, ns, name, value):
            props[(ns, name)] = value

        def set_property_int(self, ns, name, value):
            props[(ns, name)] = ("int", value)

    class XMPFiles:
        def __init__(self, file_path, open_forupdate):
            state["files"].append((file_path, open_forupdate))

        def can_put_xmp(self, meta):
            return True

        def put_xmp(self, meta):
            state["put"] += 1

        def close_file(self):
            state["closed"] += 1

    monkeypatch.setitem(sys.modules, "libxmp", types.SimpleNamespace(XMPFiles=XMPFiles, XMPMeta=XMPMeta))
    res = runner.invoke(cli.app, ["xmp", str(p)])
    assert res.exit_code == 0, (res.stdout, res.exception)
    left = tmp_path / "pair.xmp.png"
    assert state["files"] == [(left.as_posix(), True)] and state["put"] == 1 and state["closed"] == 1
    assert np.array_equal(_io.imread(left), sbs[:, :96])
    assert props[(CV.XMP_GPANO, "ProjectionType")] == "equirectangular" and props[(CV.XMP_GPANO, "FullPanoWidthPixels")] == ("int", 192)
    assert props[(CV.XMP_GPANO, "CroppedAreaImageWidthPixels")] == ("int", 96.0) and props[(CV.XMP_GPANO, "CroppedAreaLeftPixels")] == ("int", 48.0)
    assert props[(CV.XMP_GPANO, "FullPanoHeightPixels")] == ("int", 64) and props[(CV.XMP_GPANO, "InitialViewHeadingDegrees")] == ("int", 180)
    assert props[(CV.XMP_GIMAGE, "Mime")] == "image/jpeg" and props[(CV.XMP_NOTE, "HasExtendedXMP")] == "06A56CB0A1A7FAFDA459CA3FAA14B474"
    # the embedded right half decodes back to the right half
    right_file = tmp_path / "right.png"
    right_file.write_bytes(base64.b64decode(props[(CV.XMP_GIMAGE, "Data")]))
    assert np.array_equal(_io.imread(right_file), sbs[:, 96:])
    assert {pfx for _, pfx in state["ns"]} == {"GImage", "GPano", "xmpNote"}


@pytest.mark.gpu
def test_lr_and_s_commands_match_the_oracle(tmp_path, oracle_mod):
    """The reference's CLI tests (tests/test_cli.py:24-64) with assertions: outputs equal the oracle's."""
    from vr180_convert_amd.synth import pattern

    O = oracle_mod
    img = pattern(256, 256)
    p = tmp_path / "test.png"
    _io.imwrite(p, img)
    out = tmp_path / "test.cli.lr.png"
    expr = 'FisheyeEncoder("equidistant") * Euclidean3DRotator(from_rotation_vector([0, np.pi / 4, 0])) * FisheyeDecoder("equidistant")'
    r = runner.invoke(cli.app, ["lr", str(p), str(p), "--transformer", expr, "--radius", "max", "--out-path", str(out), "--size", "256x256"])
    assert r.exit_code == 0, (r.stdout, r.exception)
    spec = [("fisheye_enc", "equidistant"), ("rot", CS.ry(np.pi / 4)), ("fisheye_dec", "equidistant")]
    # same path for both eyes: the file is split into halves (remapper.py:448-456)
    want = O.apply_lr(spec, img[:, :128], img[:, 128:], size_output=(256, 256), interpolation=4, radius="max")
    assert np.array_equal(_io.imread(out), want)
    out_s = tmp_path / "test.cli.s.png"
    expr = 'FisheyeEncoder("equidistant") * Euclidean3DRotator(from_rotation_vector([np.pi / 4, 0, 0])) * FisheyeDecoder("equidistant")'
    r = runner.invoke(cli.app, ["s", str(p), "--transformer", expr, "--radius", "max", "--out-path", str(out_s), "--size", "256x256",
                                "--interpolation", "inter_linear"])
    assert r.exit_code == 0, (r.stdout, r.exception)
    c, s_ = np.cos(np.pi / 4), np.sin(np.pi / 4)
    spec = [("fisheye_enc", "equidistant"), ("rot", [[1, 0, 0], [0, c, -s_], [0, s_, c]]), ("fisheye_dec", "equidistant")]
    assert np.array_equal(_io.imread(out_s), O.apply(spec, [img], size_output=(256, 256), interpolation=1, radius="max")[0])
    # explicit calibration points, default output name next to the left image, unique suffix
    l, r_ = tmp_path / "L.png", tmp_path / "R.png"
    _io.imwrite(l, img)
    _io.imwrite(r_, img)
    r = runner.invoke(cli.app, ["lr", str(l), str(r_), "--radius", "max", "--size", "128x128", "--interpolation", "inter_linear",
                                "--automatch", "100,100;104,101;160,90;163,92;80,170;82,173", "--name-unique"])
    assert r.exit_code == 0, (r.stdout, r.exception)
    made = list(tmp_path.glob("L-R-*.png"))
    assert len(made) == 1 and _io.imread(made[0]).shape == (128, 256, 3)
    # ... and its pixels: the same points through the (reference-pinned, tests/test_calibration.py) calibration maths give
    # the two per-eye matrices the CLI inserted behind the encoder (cli.py:286-319); the oracle renders them
    from vr180_convert_amd import transformer as T
    from vr180_convert_amd.calibration import calibration_rotators, match_lr, rotation_match
    from vr180_convert_amd.quat import as_rotation_matrix

    pts = [(100, 100), (104, 101), (160, 90), (163, 92), (80, 170), (82, 173)]
    vl, vr = match_lr(T.FisheyeDecoder("equidistant"), pts[0::2], pts[1::2], [img, img], radius="max")
    ql, qr = calibration_rotators(rotation_match(vl, vr))
    specs = tuple([("equirect_enc", True), ("rot", as_rotation_matrix(q)), CS.EQUI] for q in (ql, qr))
    want = O.apply_lr(specs, img, img, size_output=(128, 128), interpolation=1, radius="max")
    assert np.array_equal(_io.imread(made[0]), want)


def test_name_unique_hashes_the_swapped_clock_offset(tmp_path, monkeypatch):
    """cli.py:174-176, 334-352: --swap negates the clock offset BEFORE the name hash is taken, so the default 0.0 enters
    the hash as "-0.0"; the same options give the file name the reference's CLI gives."""
    import hashlib

    import vr180_convert_amd.remapper as R

    calls = []
    monkeypatch.setattr(R, "apply_lr", lambda *a, **k: calls.append(k))
    l, r_ = tmp_path / "L.png", tmp_path / "R.png"
    l.write_bytes(b"x"), r_.write_bytes(b"x")
    for swap, off in ((True, "-0.0"), (False, "0.0")):
        calls.clear()
        args = ["lr", str(l), str(r_), "--radius", "max", "--size", "64x64", "--name-unique"] + (["--swap"] if swap else [])
        res = runner.invoke(cli.app, args)
        assert res.exit_code == 0, (res.stdout, res.exception)
        tag = hashlib.sha256("".join(["", "64x64", "inter_lanczos4", "border_constant", "0", "max", "False", off, str(swap)]).encode()).hexdigest()[:8]
        first, second = ("R", "L") if swap else ("L", "R")
        assert calls[0]["out_path"] == tmp_path / f"{first}-{second}-{tag}.png", calls[0]["out_path"]
