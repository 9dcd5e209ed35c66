"""Command line of the engine: ``lr`` / ``s`` / ``swap`` / ``xmp`` with the options, defaults and file-name
rules of the reference's typer app (reference cli.py:116-559; entry points ``vr180-convert`` / ``v1c``,
pyproject.toml:25-27), so that ``v1c lr left.jpg right.jpg --transformer "..."`` runs unchanged -- the remap
itself happens on the MI355X through ``apply`` / ``apply_lr``.

What is OpenCV or libxmp from end to end -- AKAZE feature matching (``--automatch fm[scale]``, cli.py:255-262, with ``--savematch``),
the point-picking window (``--automatch gui[n]``, cli.py:82-113) and the ``xmp`` command (cli.py:439-540) -- lives in
``calibration_cv.py`` behind lazy imports: with the library installed the options work as in the reference, without it they report
which package is missing (the GPU image of this engine ships neither); explicit points (``--automatch "x,y;x,y;..."``) need nothing.
"""
from __future__ import annotations

import hashlib
import logging
import re
from datetime import datetime, timezone
from pathlib import Path
from typing import Any, List, Optional, Sequence

import numpy as np
import typer
from typing_extensions import Annotated

from . import _abi, _io
from . import quat as _quat
from . import transformer as _T
from .calibration import calibration_rotators, match_lr, rotation_match, rotation_match_robust
from .chain import MultiTransformer

LOG = logging.getLogger(__name__)
DEFAULT_EXTENSION = "png"  # cli.py:39

app = typer.Typer(help="Fisheye -> VR180 side-by-side equirectangular conversion on MI355X.")

_INTERPOLATIONS = {"nearest": _abi.INTER_NEAREST, "linear": _abi.INTER_LINEAR, "cubic": _abi.INTER_CUBIC,
                   "area": _abi.INTER_AREA, "lanczos4": _abi.INTER_LANCZOS4}
_BORDERS = {"constant": _abi.BORDER_CONSTANT, "replicate": _abi.BORDER_REPLICATE, "reflect": _abi.BORDER_REFLECT,
            "wrap": _abi.BORDER_WRAP, "reflect_101": _abi.BORDER_REFLECT_101, "transparent": _abi.BORDER_TRANSPARENT}


def _flag(value: str, table: dict, prefix: str, what: str) -> int:
    """``INTER_LANCZOS4`` / ``inter_lanczos4`` / ``lanczos4`` -> the cv2 enum value (cli.py:57-79, :376-377)."""
    key = value.strip().lower()
    key = key[len(prefix):] if key.startswith(prefix) else key
    if key not in table:
        raise typer.BadParameter(f"unknown {what} {value!r}; one of " + ", ".join(prefix.upper() + k.upper() for k in table))
    return table[key]


def _namespace() -> dict:
    """Names a ``--transformer`` expression may use: the reference evaluates it inside cli.py, i.e. with numpy as
    ``np``, everything numpy-quaternion exports and every transformer name in scope (cli.py:12,15,20,233)."""
    ns: dict[str, Any] = {"np": np}
    ns.update({k: getattr(_T, k) for k in dir(_T) if not k.startswith("_")})
    ns.update({k: getattr(_quat, k) for k in ("quaternion", "from_euler_angles", "from_rotation_vector", "as_rotation_matrix",
                                              "rotate_vectors", "as_quat_array")})
    return ns


def parse_transformer(code: str) -> Any:
    """The transformer of a command: the default chain for ``""`` (cli.py:231), else the evaluated expression."""
    if code == "":
        return _T.EquirectangularEncoder() * _T.FisheyeDecoder("equidistant")
    return eval(code, _namespace())  # noqa: S307 - the CLI contract: the option is Python source (cli.py:233)


def parse_size(size: str) -> tuple[int, int]:
    w, h = (int(v) for v in size.lower().split("x"))
    return w, h


def parse_radius(radius: str) -> Any:
    return radius if radius in ("auto", "max") else float(radius)


def closest_in_time(directory: Path, anchor: Path, offset_s: float) -> Path:
    """The file under ``directory`` (recursively) with ``anchor``'s suffix whose modification time is closest to
    ``anchor``'s, shifted by ``offset_s`` seconds (cli.py:178-216)."""
    target = anchor.stat().st_mtime + offset_s
    found = [p for p in directory.rglob("*") if p != anchor and p.suffix == anchor.suffix and p.is_file()]
    if not found:
        raise ValueError(f"No time-matched image found under {directory}")
    return min(found, key=lambda p: abs(p.stat().st_mtime - target))


def resolve_pair(left: Path, right: Path, r_earlier_l: float) -> tuple[Path, Path]:
    """One of the two paths may be a directory: it is searched for the image taken at the same time as the other
    one, the right camera's clock being ``r_earlier_l`` seconds ahead."""
    if left.is_dir() and right.is_dir():
        raise ValueError("Both left and right paths must not be directories")
    if left.is_dir():
        return closest_in_time(left, right, -r_earlier_l), right
    if right.is_dir():
        return left, closest_in_time(right, left, r_earlier_l)
    return left, right


def unique_suffix(*options: Any) -> str:
    """``-xxxxxxxx``: eight hex digits of the SHA-256 of the options that shape the output (cli.py:334-352)."""
    return "-" + hashlib.sha256("".join(str(o) for o in options).encode("utf-8")).hexdigest()[:8]


def output_path(out_path: Path, left: Path, right: Path, tag: str) -> Path:
    name = f"{left.stem}-{right.stem}{tag}.{DEFAULT_EXTENSION}"
    if out_path == Path(""):
        return left.parent / name
    return out_path / name if out_path.is_dir() else out_path


def split_at_first_encoder(t: Any) -> tuple[MultiTransformer, MultiTransformer]:
    """Stages up to and including the first ``*Encoder`` and the stages after it: the calibration rotation goes
    between them (cli.py:237-250)."""
    if not isinstance(t, MultiTransformer):
        raise ValueError("Automatch requires MultiTransformer")
    names = [type(s).__name__ for s in t.transformers]
    cut = next((i for i, n in enumerate(names) if n.endswith("Encoder")), None)
    if cut is None:
        raise ValueError("Automatch requires a chain with an Encoder stage")
    return MultiTransformer(t.transformers[: cut + 1]), MultiTransformer(t.transformers[cut + 1:])


def _option_number(automatch: str, prefix: str, pattern: str, default: float) -> float:
    """``fm0.5`` -> 0.5, ``gui3`` -> 3, the bare keyword -> ``default`` (cli.py:258-262, 268-274)."""
    m = re.match(prefix + pattern, automatch)
    return float(m.group(1)) if m and m.group(1) else default


def calibrated_pair(transformer: Any, automatch: str, left: Path, right: Path, radius: Any, match_image_path: Optional[Path] = None) -> tuple[Any, Any]:
    """``--automatch``: estimate the rotation between the eyes from matched points and give each eye half of it
    (cli.py:234-319): (left chain, right chain).  The points come from the option itself (``"xl,yl;xr,yr;..."``: even entries the
    left eye, odd entries the right; plain least-squares fit ``rotation_match``), from clicks (``gui[n]``: n pairs, default 2) or from
    AKAZE feature matching (``fm[scale]``: outlier-ridden, hence ``rotation_match_robust``; ``match_image_path`` = where
    ``--savematch`` wants 100 of the surviving matches drawn)."""
    head, tail = split_at_first_encoder(transformer)
    matched = None
    if automatch.startswith("fm") or automatch.startswith("gui"):
        from . import calibration_cv as _cvx

        try:
            img_l, img_r = _io.imread(left), _io.imread(right)
            if automatch.startswith("fm"):
                matched = _cvx.match_points(img_l, img_r, scale=_option_number(automatch, "fm", r"([\d\.]+)", 1))
                points_l, points_r = matched[0], matched[1]
            else:
                n_pairs = int(_option_number(automatch, "gui", r"(\d+)", 2))
                clicks = _cvx.pick_points_gui([img_l, img_r] * n_pairs)
                LOG.info("Automatched position: " + ";".join(",".join(map(str, p)) for p in clicks))
                points_l, points_r = clicks[::2], clicks[1::2]
        except _cvx.OptionalDependencyMissing as e:
            raise typer.BadParameter(f'--automatch {automatch}: {e}; or pass the matched points explicitly, e.g. '
                                     '--automatch "xl,yl;xr,yr;xl,yl;xr,yr"') from e
    else:
        flat = [(int(c.split(",")[0]), int(c.split(",")[1])) for c in automatch.split(";")]
        points_l, points_r = flat[::2], flat[1::2]  # even entries: left eye, odd entries: right eye
    vl, vr = match_lr(tail, points_l, points_r, in_paths=[left, right], radius=radius)
    if matched is not None:
        q, discarded = rotation_match_robust(vl, vr)
        if match_image_path is not None:
            from . import calibration_cv as _cvx

            _, _, kp_l, kp_r, matches, small_l, small_r = matched
            _io.imwrite(match_image_path, _cvx.draw_match_image(small_l, kp_l, small_r, kp_r, matches, discarded))
    else:
        q = rotation_match(vl, vr)
    LOG.info(f"Automatched quaternion: {q}")
    q_left, q_right = calibration_rotators(q)
    return head * _T.Euclidean3DRotator(q_left) * tail, head * _T.Euclidean3DRotator(q_right) * tail


@app.callback()
def _main(verbose: bool = typer.Option(False, "--verbose", "-v")) -> None:
    handlers = None
    try:
        from rich.logging import RichHandler

        handlers = [RichHandler(rich_tracebacks=True)]
    except Exception:  # noqa: BLE001
        pass
    logging.basicConfig(level=logging.DEBUG if verbose else logging.INFO, format="%(message)s", datefmt="[%X]", handlers=handlers)


@app.command()
def lr(
    left_path: Annotated[Path, typer.Argument(help="Left image path (or a directory to search for the time-matched image)")],
    right_path: Annotated[Path, typer.Argument(help="Right image path (or a directory)")],
    transformer: Annotated[str, typer.Option(help="Transformer Python code (to be `eval()`ed)")] = "",
    out_path: Annotated[Path, typer.Option(help="Output image path or directory, defaults to <left>-<right>.png next to the left image")] = Path(""),
    size: Annotated[str, typer.Option(help="Output image size per eye")] = "4096x4096",
    interpolation: Annotated[str, typer.Option(help="Interpolation method (cv2 name)")] = "inter_lanczos4",
    border_mode: Annotated[str, typer.Option(help="Border mode (cv2 name)")] = "border_constant",
    border_value: int = 0,
    radius: Annotated[str, typer.Option(help="Radius of the fisheye image: a number, 'auto' or 'max'")] = "auto",
    merge: Annotated[bool, typer.Option("-m", "--merge", "--anaglyph", help="Export as an anaglyph")] = False,
    autosearch_timestamp_calib_r_earlier_l: Annotated[float, typer.Option(
        "--autosearch-timestamp-calib-r-earlier-l", "-ac",
        help="Autosearch timestamp calibration (right timestamp -= this) (in seconds)")] = 0.0,
    swap: Annotated[bool, typer.Option(help="Swap left and right images as well as transformer, etc.")] = False,
    name_unique: Annotated[bool, typer.Option(help="Make output name unique")] = False,
    automatch: Annotated[str, typer.Option(help='Calibrate rotation. e.g. "0,0;0,0;1,1;1,1", "gui[n]" (click n point pairs) or '
                                                 '"fm[scale]" (AKAZE feature matching); the last two need OpenCV')] = "",
    savematch: Annotated[bool, typer.Option(help="Save the match image <out>.match<ext> (only with automatch=fm)")] = False,
) -> None:
    """Remap a pair of fisheye images to a pair of SBS equirectangular images."""
    from .remapper import apply_lr

    r_earlier_l = autosearch_timestamp_calib_r_earlier_l
    if swap:
        left_path, right_path, r_earlier_l = right_path, left_path, -r_earlier_l
    left_path, right_path = resolve_pair(left_path, right_path, r_earlier_l)
    LOG.info("L: %s@%s, R: %s@%s", left_path, datetime.fromtimestamp(left_path.stat().st_mtime, timezone.utc),
             right_path, datetime.fromtimestamp(right_path.stat().st_mtime, timezone.utc))
    interp = _flag(interpolation, _INTERPOLATIONS, "inter_", "interpolation")
    border = _flag(border_mode, _BORDERS, "border_", "border mode")
    radius_ = parse_radius(radius)
    chain: Any = parse_transformer(transformer)
    # (the reference hashes the option AFTER --swap negated it, cli.py:174-176, 347: str(-0.0) == "-0.0" even for the default)
    tag = unique_suffix(transformer, size, interpolation, border_mode, border_value, radius, merge,
                        r_earlier_l, swap) if name_unique else ""
    out = output_path(out_path, left_path, right_path, tag)
    if savematch and not automatch.startswith("fm"):
        # (the reference ignores the flag silently, cli.py:365: scripts that always pass it keep working)
        LOG.warning("--savematch ignored: it draws the feature matches of --automatch fm, and there are none without it")
    if automatch != "":
        match_image = out.with_suffix(f".match{out.suffix}") if savematch else None  # cli.py:362-365
        chain = calibrated_pair(chain, automatch, left_path, right_path, radius_, match_image)
        LOG.info(f"Automatched transformer: {chain}")
    apply_lr(chain, left_path=left_path, right_path=right_path, out_path=out, radius=radius_, size_output=parse_size(size),
             interpolation=interp, boarder_mode=border, boarder_value=border_value, merge=merge)


@app.command()
def s(
    in_paths: Annotated[List[Path], typer.Argument(help="Image paths")],
    transformer: Annotated[str, typer.Option(help="Transformer Python code (to be `eval()`ed)")] = "",
    out_path: Annotated[Path, typer.Option(help="Output image path or directory, defaults to <in>.out.png")] = Path(""),
    size: Annotated[str, typer.Option(help="Output image size")] = "4096x4096",
    interpolation: Annotated[str, typer.Option(help="Interpolation method (cv2 name)")] = "inter_lanczos4",
    boarder_mode: Annotated[str, typer.Option(help="Border mode (cv2 name)")] = "border_constant",
    boarder_value: int = 0,
    radius: Annotated[str, typer.Option(help="Radius of the fisheye image: a number, 'auto' or 'max'")] = "auto",
) -> None:
    """Remap fisheye images to equirectangular images (one shared map for all of them)."""
    from .remapper import apply

    if out_path == Path(""):
        out_paths = [p.with_suffix(f".out.{DEFAULT_EXTENSION}") for p in in_paths]
    elif out_path.is_dir():
        out_paths = [out_path / p.name for p in in_paths]
    elif len(in_paths) > 1:
        raise ValueError("Output path must be a directory when multiple input paths are provided")
    else:
        out_paths = [out_path]
    apply(parse_transformer(transformer), in_paths=list(in_paths), out_paths=out_paths, radius=parse_radius(radius),
          size_output=parse_size(size), interpolation=_flag(interpolation, _INTERPOLATIONS, "inter_", "interpolation"),
          boarder_mode=_flag(boarder_mode, _BORDERS, "border_", "border mode"), boarder_value=boarder_value)


@app.command()
def swap(
    in_paths: Annotated[List[Path], typer.Argument(help="Image paths")],
    overwrite: Annotated[bool, typer.Option(help="Overwrite the original images")] = True,
) -> None:
    """Swap the left and right halves of side-by-side images."""
    for p in in_paths:
        image = _io.imread(p)
        if image is None:
            raise ValueError(f"cannot read {p}")
        half = image.shape[1] // 2
        _io.imwrite(p if overwrite else p.with_suffix(f".swap{p.suffix}"), np.hstack([image[:, half:], image[:, :half]]))


@app.command()
def xmp(
    in_paths: Annotated[List[Path], typer.Argument(help="Side-by-side image paths")],
    wslpath: Annotated[bool, typer.Option("-wsl", "--wslpath", help="Convert Windows paths to WSL paths (runs `wslpath`)")] = False,
) -> None:
    """Write the left half as <name>.xmp<ext> with the right half embedded as Google VR180 photo metadata
    (GPano / GImage XMP, cli.py:439-540).  Needs python-xmp-toolkit (libxmp / exempi)."""
    import subprocess as sp

    from . import calibration_cv as _cvx

    for in_path in in_paths:
        if wslpath:  # cli.py:477-482
            in_path = Path(sp.run(["wslpath", "-u", "-a", str(in_path)], capture_output=True).stdout.decode().strip())  # noqa: S603, S607
        try:
            written = _cvx.write_vr180_xmp(in_path)
        except _cvx.OptionalDependencyMissing as e:
            raise typer.BadParameter(str(e)) from e
        LOG.info(f"Saved {written}")


def main(argv: Optional[Sequence[str]] = None) -> None:
    app(args=None if argv is None else list(argv), prog_name="vr180-convert")
