"""Build-time guard of the fence-less barrier protocol of the LDS-DMA kernels (csrc/kernels_tile.hip:
k_ray_lin3_pair_mirror_raw, k_ray_lin3_pair_mirror_pipe, k_ray_lin3_batch_lean_raw).

Those kernels publish LDS-DMA data with a bare ``s_barrier`` behind a hand-counted ``s_waitcnt vmcnt(n)``: the count
is the number of vector-memory requests the wave has issued BEHIND the data it waits for (vmcnt retires in issue
order).  The source therefore assumes things about the instruction stream the compiler emits:

* between the DMA requests and the last wait of the path there is no compiler-visible vector load (its own
  ``s_waitcnt vmcnt`` does not count LDS-DMA and would wait for every box) -- the row / column table loads are forced in
  front of the requests;
* requests, stores and waits appear in the order of the source (``asm volatile(... ::: "memory")``);
* a ``store_pair_row`` is ``kStoresPerPairRow`` = 2 requests (one ``global_store_dwordx3`` per eye), whatever the alignment
  path -- an over-count in the source would under-wait.

This test disassembles the gfx950 code object that ``make`` built and checks exactly that, so that a compiler upgrade
which sinks a load past an ``asm volatile`` or splits a store fails HERE instead of as a rare race on the GPU
(tools/soak.py is the dynamic check; its log of this round: profiles/r03*/soak.log)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "vr180_convert_amd" / "csrc"
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory, product_lib):
    if not OBJDUMP.exists():
        pytest.skip("llvm-objdump not available")
    obj = CSRC / "kernels_tile.o"
    assert obj.exists(), "kernels_tile.o is built by __graft_entry__.build() / make"
    d = tmp_path_factory.mktemp("dis")
    shutil.copy(obj, d / "kt.o")
    subprocess.run([str(OBJDUMP), "--offloading", "kt.o"], cwd=d, check=True, capture_output=True, timeout=300)
    code = [p for p in d.iterdir() if "gfx950" in p.name]
    assert len(code) == 1, [p.name for p in d.iterdir()]
    r = subprocess.run([str(OBJDUMP), "-d", "--no-show-raw-insn", code[0].name], cwd=d, check=True, capture_output=True, text=True,
                       timeout=600)
    bodies: dict[str, str] = {}
    cur = None
    for line in r.stdout.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(line)
    return {k: v for k, v in bodies.items()}


def events(lines):
    """The vector-memory skeleton of a kernel in program (address) order:
    D = LDS-DMA request, L = other vector load, S<form> = vector store, W<n>[g] = hand-written ``s_waitcnt vmcnt(n) [lgkmcnt(0)]``
    + ``s_barrier``, w<n> = any other vmcnt wait (compiler-inserted), B = other barrier."""
    ins = []
    for line in lines:
        m = re.match(r"^\s*(\S+)\s*(.*?)\s*//", line)
        if m:
            ins.append((m.group(1), m.group(2)))
    ev = []
    i = 0
    while i < len(ins):
        op, args = ins[i]
        if op.startswith("global_load_lds"):
            ev.append("D")
        elif re.match(r"(global|buffer|scratch|flat)_load", op):
            ev.append("L")
        elif re.match(r"(global|buffer|scratch|flat)_(store|atomic)", op):
            ev.append("S:" + op + (":nt" if re.search(r"\bnt\b", args) else ""))
        elif op == "s_waitcnt" and "vmcnt" in args:
            n = re.search(r"vmcnt\((\d+)\)", args).group(1)
            if i + 1 < len(ins) and ins[i + 1][0] == "s_barrier":
                ev.append(f"W{n}" + ("g" if "lgkmcnt(0)" in args else ""))
                i += 1
            else:
                ev.append(f"w{n}")
        elif op == "s_barrier":
            ev.append("B")
        i += 1
    return ev


def fast_path(ev):
    """From the first DMA request to the last event before the general pair code of the same kernel (which starts with
    ordinary loads and compiler-made waits)."""
    a = ev.index("D")
    b = a
    while b < len(ev) and not (ev[b] == "L" or ev[b].startswith("w")):
        b += 1
    return ev[a:b]


def shape(ev):
    """Events as one string: requests D, waits W, barrier B, stores S."""
    return "".join("S" if e.startswith("S:") else e[0] for e in ev)


def kernels(dis, name):
    out = {k: v for k, v in dis.items() if name in k}
    assert out, f"no {name} in the code object"
    return out


def check_stores(fp, per_group):
    stores = [e for e in fp if e.startswith("S:")]
    # one 12-byte store per eye and output row in either alignment path: the dword-aligned one non-temporal
    assert all(s.startswith("S:global_store_dwordx3") for s in stores), stores
    assert sum(s.endswith(":nt") for s in stores) * 2 == len(stores), stores
    assert len(stores) % per_group == 0


def test_mirror_raw_stream(disassembly):
    for name, lines in kernels(disassembly, "k_ray_lin3_pair_mirror_raw").items():
        ev = events(lines)
        fp = fast_path(ev)
        # requests (table slice + 4 boxes) | waits: table, the tile's boxes, everything | the two rows' stores (2 eyes x 2
        # alignment paths each) -- and nothing else: no load, no compiler-made vmcnt wait, no store in front of the last wait
        # (the one-eye instantiation <VAR_W, 1>: table slice + 2 boxes, 2 x 2 stores)
        eyes = 1 if re.search(r"mirror_rawILi\dELi1E", name) else 2
        assert re.fullmatch(r"D{%d,}W+S{%d}" % (1 + 2 * eyes, 4 * eyes), shape(fp)), (name, shape(fp))
        waits = [e for e in fp if e.startswith("W")]
        assert waits[-1] == "W0" and not any(w.endswith("g") for w in waits), waits
        check_stores(fp, 2 * eyes)
        # in front of the requests: the row / column table loads and the compiler's wait for them
        head = ev[: ev.index("D")]
        assert head and head[-1] == "w0" and set(head[:-1]) == {"L"}, head


def test_mirror_pipe_stream(disassembly):
    for name, lines in kernels(disassembly, "k_ray_lin3_pair_mirror_pipe").items():
        fp = fast_path(events(lines))
        s = shape(fp)
        # requests of pair 0 | waits (tables, tile, band + LDS reads) | requests b | store a (+ the early exit's store a') |
        # barrier | requests b' | store a' | waits (b, then b' behind kStoresPerPairRow requests) | stores of pair 1
        m = re.fullmatch(r"(D{7,})(W+)(D+)(S{8})B(D+)(S{4})(W+)(S{8})", s)
        assert m, (name, s)
        waits = [e for e in fp if e.startswith("W")]
        first = waits[: len(m.group(2))]
        assert first[-1] == "W0g", first  # the wait that frees the tile's buffers also waits for this wave's LDS reads
        assert waits[-1] == "W2", waits  # = kStoresPerPairRow: the band of pair 1 is followed by store a' only
        check_stores(fp, 4)


def test_batch_lean_raw_stream(disassembly):
    ks = kernels(disassembly, "k_ray_lin3_batch_lean_raw")
    assert len(ks) >= 4
    for name, lines in ks.items():
        if re.search(r"lean_rawILi\dELi\dELi1E", name):
            # OWN = 1: the per-pixel table fallback reads the radial table from global memory inside the coordinates; the compiler's
            # wait for those loads is a vmcnt(0) in front of the unit loop, and the loop's own counts (requests behind a box) only get
            # stricter by loads they do not count -- over-waiting, never under-waiting
            continue
        fp = fast_path(events(lines))
        s = shape(fp)
        # requests (table slice + the first boxes) | per unit: wait, next request, store -- no load, no compiler-made wait
        assert re.fullmatch(r"D{3,}[WDS]+", s) and "S" in s and re.search(r"WD", s), (name, s)
        assert not any(e.endswith("g") for e in fp if e.startswith("W"))
        # the unit's store: `issued += 1` in the source is a lower bound of these
        assert all(e.startswith("S:global_store_dwordx3") for e in fp if e.startswith("S:"))
