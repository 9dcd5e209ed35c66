"""Build-time guard of the fence-less barrier protocol of the LDS-DMA kernels (csrc/tile_device.hpp, kernels_tile.hip, kernels_mirror.hip, kernels_cn.hip:
k_ray_lin3_pair_mirror_seq, k_ray_lin3_pair_mirror_raw, k_ray_lin3_batch_lean_raw, k_ray_lin3_rot_pair_raw, k_ray_lin_cn).

Those kernels publish LDS-DMA data with a bare ``s_barrier`` behind a hand-counted ``s_waitcnt vmcnt(n)``: n is the
number of vector-memory requests the wave has issued BEHIND the data it waits for (vmcnt retires in issue order).
What the source counts are its own requests (loop trip counts, exact) and LOWER bounds of its stores (one per eye and
row), so an extra instruction the compiler emits can only make a wait longer.  What the source relies on beyond that:

* requests, stores and waits stay in source order (``asm volatile(... ::: "memory")``);
* in the kernels that promise it, NO compiler-visible vector load executes behind an LDS-DMA request -- the compiler's own
  wait for such a load does not count LDS-DMA and becomes a vmcnt(0) that serialises the boxes (the row / column table
  loads are forced in front of the requests);
* a wait is one ``s_waitcnt vmcnt(n) [lgkmcnt(0)]`` + ``s_barrier``, selected by a computed jump into a table of 21
  sixteen-byte blocks (``V1C_WAIT_JUMP``): block i must wait for vmcnt(i) and sit at byte 16 i behind the jump.

This test disassembles the gfx950 code object ``make`` built, verifies every wait table block by block and checks them, so that a compiler upgrade which sinks a load past an ``asm volatile``
or re-lays-out the jump table fails here instead of as a rare race on the GPU (tools/soak.py is the dynamic check; its
log of this round: profiles/r03b_final/soak.log)."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "vr180_convert_amd" / "csrc"
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")


class Ins:
    __slots__ = ("addr", "op", "args", "kind")

    def __init__(self, addr, op, args):
        self.addr, self.op, self.args = addr, op, args
        if op.startswith("global_load_lds"):
            self.kind = "D"
        elif re.match(r"(global|buffer|scratch|flat)_load", op):
            self.kind = "L"
        elif re.match(r"(global|buffer|scratch|flat)_(store|atomic)", op):
            self.kind = "S"
        elif op == "s_waitcnt" and "vmcnt" in args:
            self.kind = "w"  # (hand-written waits are reclassified in analyse())
        else:
            self.kind = ""


TILE_OBJECTS = ("kernels_tile", "kernels_mirror", "kernels_cn")  # the translation units of the tile kernels (csrc/Makefile)


def _disassemble_all(tmp_path_factory, suffix):
    out: dict[str, list[Ins]] = {}
    for stem in TILE_OBJECTS:
        one = _disassemble(tmp_path_factory, stem + suffix)
        assert not (set(one) & set(out)), sorted(set(one) & set(out))[:4]
        out.update(one)
    return out


def _disassemble(tmp_path_factory, objname):
    if not OBJDUMP.exists():
        pytest.fail("llvm-objdump not available: the image ships it under /opt/rocm/lib/llvm/bin -- the check must run, here and on the GPU box")
    obj = CSRC / objname
    assert obj.exists(), f"{objname} is built by __graft_entry__.build() / make"
    d = tmp_path_factory.mktemp("dis")
    shutil.copy(obj, d / "kt.o")
    subprocess.run([str(OBJDUMP), "--offloading", "kt.o"], cwd=d, check=True, capture_output=True, timeout=300)
    code = [p for p in d.iterdir() if "gfx950" in p.name]
    assert len(code) == 1, [p.name for p in d.iterdir()]
    r = subprocess.run([str(OBJDUMP), "-d", "--no-show-raw-insn", code[0].name], cwd=d, check=True, capture_output=True, text=True,
                       timeout=600)
    out: dict[str, list[Ins]] = {}
    cur = None
    for line in r.stdout.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        m = re.match(r"^\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append(Ins(int(m.group(3), 16), m.group(1), m.group(2)))
    return out


@pytest.fixture(scope="module")
def kernels_dis(tmp_path_factory, product_lib):
    """the product's code object"""
    return _disassemble_all(tmp_path_factory, ".o")


@pytest.fixture(scope="module")
def tuning_dis(tmp_path_factory, product_lib):
    """the -DV1C_TUNING twin: the product's kernels plus the A/B partners (four-buffer pair kernel, register-staged forms)"""
    return _disassemble_all(tmp_path_factory, ".tuning.o")


def analyse(ins):
    """Successor lists of a kernel's instructions (index based) with the hand-written waits recognised:
    * ``s_waitcnt vmcnt(n) [lgkmcnt(0)]`` directly followed by ``s_barrier`` -> kind "W";
    * ``s_setpc_b64 vcc`` + 21 blocks of (s_waitcnt vmcnt(i) | s_barrier | s_branch end | s_nop) -> the jump goes to every
      block (the table itself is verified here)."""
    at = {x.addr: i for i, x in enumerate(ins)}
    succ = [[] for _ in ins]
    tables = 0
    for i, x in enumerate(ins):
        if x.kind == "w" and i + 1 < len(ins) and ins[i + 1].op == "s_barrier":
            x.kind = "W"
        nxt = [i + 1] if i + 1 < len(ins) else []
        if x.op == "s_endpgm":
            nxt = []
        elif x.op == "s_branch" or x.op.startswith("s_cbranch"):
            simm = int(x.args.split()[0])
            simm -= 65536 if simm >= 32768 else 0
            tgt = at[x.addr + 4 + 4 * simm]
            nxt = [tgt] if x.op == "s_branch" else [i + 1, tgt]
        elif x.op == "s_setpc_b64" and x.args.strip() == "vcc":
            # V1C_WAIT_JUMP: s_getpc_b64 vcc five instructions back; block k at (address behind s_getpc) + 20 + 16 k
            assert ins[i - 5].op == "s_getpc_b64" and ins[i - 5].args.strip() == "vcc", hex(x.addr)
            base = ins[i - 4].addr + 20
            assert base == x.addr + 4
            nxt = []
            for k in range(21):
                j = at[base + 16 * k]
                blk = ins[j: j + 4]
                assert blk[0].op == "s_waitcnt" and re.search(r"vmcnt\(%d\)" % k, blk[0].args), (hex(blk[0].addr), blk[0].args)
                assert blk[1].op == "s_barrier" and (blk[2].op == "s_branch" or k == 20), hex(blk[0].addr)
                nxt.append(j)
            tables += 1
        elif x.op == "s_setpc_b64":
            nxt = []  # (return of a device function; not inside kernels)
        succ[i] = nxt
    return succ, tables


def pick(dis, name):
    out = {k: v for k, v in dis.items() if name in k}
    assert out, f"no {name} in the code object"
    return out


@pytest.mark.parametrize("which", ["product", "tuning"])
def test_no_vector_load_between_a_request_and_its_wait(which, request):
    """mirror_seq (pairs: the default) and mirror_raw (single images; pairs as an A/B form of the tuning build): from every LDS-DMA request, in address
    order up to the next hand-written wait, there is no compiler-visible vector load and no vmcnt wait the compiler made
    (the compiler lays the request loops, the wait tables and the store blocks of a phase out together; its structurised
    control flow -- flag registers -- makes a path-exact check meaningless, so this is the layout-local form of "nothing loads
    through registers while boxes are in flight").  Every kernel has its wait tables, each verified block by block."""
    kernels_dis = request.getfixturevalue("kernels_dis" if which == "product" else "tuning_dis")
    checked = 0
    for fam in ("k_ray_lin3_pair_mirror_seq", "k_ray_lin3_pair_mirror_raw", "k_ray_lin3_batch_lean_raw"):
        for name, ins in pick(kernels_dis, fam).items():
            succ, tables = analyse(ins)  # (verifies the jump tables)
            assert tables >= 2, (name, tables)
            if "batch_lean_raw" in name:
                # (its requests sit in the unit loop, whose wait is at the loop head -- BEHIND them in address order; OWN = 1 reads the
                # radial table inside the coordinates and ROT = 1 keeps a wait of its own: all of that only over-waits.  The wait
                # tables are verified above.)
                continue
            for i, x in enumerate(ins):
                if x.kind != "D":
                    continue
                j = i + 1
                while j < len(ins) and ins[j].kind != "W" and ins[j].op != "s_endpgm":
                    assert ins[j].kind not in ("L", "w"), (name, hex(x.addr), hex(ins[j].addr), ins[j].op)
                    j += 1
                assert j < len(ins) and ins[j].kind == "W", (name, hex(x.addr), "no hand-written wait behind the request")
            checked += 1
    assert checked >= (6 if which == "product" else 8)


def test_rot_pair_has_its_tables(kernels_dis):
    for name, ins in pick(kernels_dis, "k_ray_lin3_rot_pair_raw").items():
        _, tables = analyse(ins)
        assert tables == 2, (name, tables)  # the waits for unit A's and unit B's box
        assert sum(x.kind == "D" for x in ins) == 2, name


def test_mirror_raw_has_its_requests_and_waits(kernels_dis, tuning_dis):
    # (the product holds the one-eye instantiation only; the two-eye one is an A/B partner of the seq kernel)
    assert all(re.search(r"mirror_rawILi\dELi1E", n) for n in pick(kernels_dis, "k_ray_lin3_pair_mirror_raw")), "product: one-eye form only"
    for name, ins in pick(tuning_dis, "k_ray_lin3_pair_mirror_raw").items():
        analyse(ins)
        eyes = 1 if re.search(r"mirror_rawILi\dELi1E", name) else 2
        n_req = sum(x.kind == "D" for x in ins)
        # table slice + the boxes (tile and band, per eye)
        assert n_req == 1 + 2 * eyes, (name, n_req)
        # in front of the requests: the row / column table loads and the compiler's wait for them
        first = next(i for i, x in enumerate(ins) if x.kind == "D")
        head = [x.kind for x in ins[:first] if x.kind]
        assert "L" in head and head[-1] == "w", (name, head[-6:])
        # the stores of the fast path: one 12-byte store per eye and row in either alignment path
        assert sum(x.op == "global_store_dwordx3" for x in ins) >= 4 * eyes


def test_cn_kernel_waits_for_every_request_before_the_barrier(kernels_dis):
    """k_ray_lin_cn publishes a box with ``s_waitcnt vmcnt(0)`` + ``s_barrier`` (no counting): every instantiation has its LDS-DMA
    requests (the first unit's in front of the coordinates, the next unit's inside the loop) and at least one such wait, and no
    LDS-DMA request is followed by a plain ``s_barrier`` without the wait in front of it."""
    for name, ins in pick(kernels_dis, "k_ray_lin_cn").items():
        analyse(ins)
        # (template arguments VAR_W, ROT, CN, NN, K, BOXES: without plan-time boxes a workgroup serves ONE unit -- one request site)
        one_unit = re.search(r"k_ray_lin_cnILi\dELi\dELi\dELi\dELi\dELi0EE", name) is not None
        assert sum(x.kind == "D" for x in ins) >= (1 if one_unit else 2), name
        full_waits = [i for i, x in enumerate(ins) if x.kind == "W" and "vmcnt(0)" in x.args]
        assert full_waits, name
        for i, x in enumerate(ins):
            if x.op == "s_barrier" and ins[i - 1].kind != "W":
                # a bare barrier (the table slice's __syncthreads) must sit in front of the first request
                assert not any(y.kind == "D" for y in ins[:i]), (name, hex(x.addr))


def test_hand_written_memory_instructions_keep_their_wait_states(kernels_dis):
    """The two memory instructions issued from asm statements are invisible to the compiler's hazard recogniser, so their wait
    states are part of the statements (csrc/tile_device.hpp raw_box_dma, store4<1>) -- checked here in the code object:
    * every SGPR-base LDS-DMA request of a box (``global_load_lds_dwordx4 v, s[..]`` behind an ``s_mov_b32 m0``) has ``s_nop 2``
      directly in front of it (M0 write -> use, VALU-written SGPR -> VMEM read) and puts M0 back directly behind it;
    * every system-scope streaming store (``sc0 sc1 nt``) is directly followed by ``s_nop 1`` (store data of more than 8 bytes ->
      VALU write of those registers)."""
    n_dma = n_st = 0
    for name, ins in kernels_dis.items():
        if not name.startswith("_ZN3v1c"):
            continue
        for i, x in enumerate(ins):
            # (the statement: s_mov_b32 sA, m0 | s_mov_b32 m0, sB | s_nop 2 | request | s_mov_b32 m0, sA; requests the compiler
            # emits for the builtin have no save in front and get their wait states from its own hazard recogniser)
            if (x.op == "global_load_lds_dwordx4" and i >= 3 and ins[i - 2].op == "s_mov_b32" and ins[i - 2].args.startswith("m0")
                    and ins[i - 3].op == "s_mov_b32" and ins[i - 3].args.replace(" ", "").endswith(",m0")):
                assert ins[i - 1].op == "s_nop" and ins[i - 1].args.strip() == "2", (name, hex(x.addr), ins[i - 1].op, ins[i - 1].args)
                assert ins[i + 1].op == "s_mov_b32" and ins[i + 1].args.startswith("m0"), (name, hex(x.addr))
                n_dma += 1
            if x.op.startswith("global_store") and {"sc0", "sc1", "nt"} <= set(x.args.split()):  # (printed as "sc0 nt sc1")
                assert ins[i + 1].op == "s_nop" and ins[i + 1].args.strip() == "1", (name, hex(x.addr), ins[i + 1].op, ins[i + 1].args)
                n_st += 1
    assert n_dma >= 20 and n_st >= 8, (n_dma, n_st)


def test_no_kernel_contains_the_packed_shift_the_compiler_misreads(kernels_dis, tuning_dis, tmp_path_factory):
    """v_ashr_pk_u8_i32 / v_ashr_pk_i8_i32 (new on gfx950) write 16 bits and leave the upper half of their destination as it was;
    the compiler (ROCm 7.2) forms them from shift + clamp + pack of two bytes and then treats the result as zero-extended.  BGRA
    bicubic / Lanczos4 came out with channels 2 and 3 OR-ed with a stale weight dword until the clamp got its own statement
    (fixpt_u8, csrc/tile_device.hpp; found by tools/fuzz.py in round 4).  No object of the library may contain the instruction."""
    generic = _disassemble(tmp_path_factory, "kernels.o")
    n = 0
    for name, ins in {**generic, **kernels_dis, **tuning_dis}.items():
        n += len(ins)
        assert not [x for x in ins if x.op.startswith("v_ashr_pk_")], name
    assert n > 100000
