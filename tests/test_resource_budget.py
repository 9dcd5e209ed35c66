"""Register / scratch budget and kernel-argument layout of the product's gfx950 code objects, read from their metadata notes.

Round 3 ended with two instantiations of the per-unit-rotation kernel in scratch and 32 kernels spilling scalar registers into vector
lanes (the by-value KernelCtx + UnitArgs arguments pinned 98 of the 102 SGPRs).  Since round 4 the tile kernels read one TileArgs block
through constant-address-space references (csrc/kernels.hpp); this test keeps it that way:

* no kernel of the shipped library uses scratch or spills a register (the -DV1C_TUNING twin may: its general batch loop is an A/B form);
* the tile kernels' one by-value argument sits at byte 0 of the kernel-argument segment (byte 48 behind the preloaded head of the mirror
  kernels) and k_put_units' records at byte 8 -- the kernels address them through ``__builtin_amdgcn_kernarg_segment_ptr()`` at exactly
  those offsets.
"""
import shutil
import subprocess
from pathlib import Path

import pytest
import yaml

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "vr180_convert_amd" / "csrc"
LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_metadata(tmp: Path, obj: Path) -> list[dict]:
    if not (LLVM / "llvm-readelf").exists():
        pytest.fail("llvm-readelf not available: the image ships it under /opt/rocm/lib/llvm/bin -- the check must run, here and on the GPU box")
    shutil.copy(obj, tmp / "o.o")
    subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", "o.o"], cwd=tmp, check=True, capture_output=True, timeout=300)
    code = [p for p in tmp.iterdir() if "gfx950" in p.name]
    assert len(code) == 1
    r = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", code[0].name], cwd=tmp, check=True, capture_output=True, text=True, timeout=300)
    text = r.stdout
    doc = text[text.index("---"):]
    doc = doc[: doc.index("\n...")] if "\n..." in doc else doc
    return yaml.safe_load(doc)["amdhsa.kernels"]


@pytest.fixture(scope="module")
def product_kernels(tmp_path_factory, product_lib):
    out = []
    for name in ("kernels_tile.o", "kernels_mirror.o", "kernels_cn.o", "kernels.o"):
        obj = CSRC / name
        assert obj.exists(), f"{name} is built by __graft_entry__.build() / make"
        out += kernel_metadata(tmp_path_factory.mktemp("meta"), obj)
    return out


def test_no_kernel_of_the_product_uses_scratch_or_spills(product_kernels):
    assert len(product_kernels) > 100
    bad = [(k[".name"], k[".private_segment_fixed_size"], k[".sgpr_spill_count"], k[".vgpr_spill_count"]) for k in product_kernels
           if k[".private_segment_fixed_size"] or k[".sgpr_spill_count"] or k[".vgpr_spill_count"]]
    # the literal interpreter (generic k_remap / k_get_map in MODE_LITERAL / MODE_FIXUP) keeps the lowered chain's op loop and libm calls:
    # a slow path by construction, allowed its spills
    bad = [b for b in bad if "k_remap" not in b[0] and "k_get_map" not in b[0]]
    assert not bad, bad


def test_tile_kernels_take_one_argument_block_at_a_known_offset(product_kernels):
    tile = [k for k in product_kernels if "8TileArgsE" in k[".name"]]
    assert len(tile) > 100
    sizes = set()
    n_head = 0
    for k in tile:
        explicit = [a for a in k[".args"] if not a[".value_kind"].startswith("hidden_")]
        block = explicit[-1]
        assert block[".value_kind"] == "by_value", k[".name"]
        sizes.add(block[".size"])
        if "pair_mirror_raw" in k[".name"] or "pair_mirror_seq" in k[".name"]:
            # V1C_MIRROR_HEAD: eight scalar parameters (11 dwords, preloaded into SGPRs at wave start) in front of the block at byte 48
            assert len(explicit) == 9 and block[".offset"] == 48, (k[".name"], block[".offset"])
            assert [a[".offset"] for a in explicit[:8]] == [0, 8, 12, 16, 20, 24, 32, 40], k[".name"]
            n_head += 1
        else:
            assert len(explicit) == 1 and block[".offset"] == 0, k[".name"]
    assert len(sizes) == 1 and n_head >= 6
    put = [k for k in product_kernels if "k_put_units" in k[".name"]]
    assert len(put) == 1
    assert put[0][".args"][1][".value_kind"] == "by_value" and put[0][".args"][1][".offset"] == 8 and put[0][".args"][1][".size"] == 32 * 112


def test_hot_kernels_keep_their_occupancy(product_kernels):
    """Waves per SIMD of the kernels the BASELINE configurations run are set by their VGPR count (512 / count, in steps of 8 registers).
    An innocent-looking edit moves it: sharing one lambda between the two unit loops of k_ray_lin_cn cost its bilinear forms 14 VGPRs and
    a wave per SIMD (+6 % on gray pairs, round 4).  Ceilings = the counts the round's measurements were made with."""
    import re

    ceilings = {
        r"k_ray_lin3_pair_mirror_seqILi[01]ELi0EE": 72,     # C1 / C2 pairs: 7 waves per SIMD
        r"k_ray_lin3_pair_mirror_rawILi0ELi1EE": 72,        # C1S single image
        r"k_ray_lin3_batch_lean_rawILi0ELi0ELi0EE": 80,     # C3 batches: 6
        r"k_ray_lin3_rot_pair_rawILi0ELi1ELi1EE": 96,       # C5 per-unit rotations: 5
        r"k_ray_lin3_tileILi1ELi1ELi1ELi8ELi0ELi1ELi0ELi0EE": 128,  # C4 Lanczos4 pair: 4
        r"k_ray_lin3_tileILi[01]ELi[01]ELi1ELi4ELi0ELi1ELi0ELi0EE": 80,  # bicubic pairs: 6 (the two-round staging of round 5 had cost them 27 VGPRs)
        r"k_ray_lin3_tileILi[01]ELi[01]ELi1ELi2ELi0ELi1ELi0ELi0EE": 96,  # bilinear pairs through the general tile kernel (rotated; rest tiles): 5
        r"k_ray_lin_cnILi1ELi0ELi[14]ELi0ELi2ELi1EE": 64,   # gray / BGRA bilinear pairs (w-table): 8
        r"k_ray_lin_cnILi0ELi0ELi[14]ELi0ELi2ELi1EE": 72,   # (m-table): 7
    }
    for pat, ceil in ceilings.items():
        hits = [k for k in product_kernels if re.search(pat, k[".name"])]
        assert hits, pat
        for k in hits:
            assert k[".vgpr_count"] <= ceil, (k[".name"], k[".vgpr_count"], ceil)
