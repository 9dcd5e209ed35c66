"""ctypes mirror of ``include/vr180_remap.h`` (POD types and enum values only)."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

ABI_VERSION = 1
MAX_OPS = 16
MAX_PARAMS = 16

OK, E_INVALID, E_UNSUPPORTED, E_HIP, E_NODEVICE = 0, -1, -2, -3, -4

# cv2 enum values (reference cli.py:57-79 mirrors the same names)
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_AREA, INTER_LANCZOS4 = 0, 1, 2, 3, 4
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101, BORDER_TRANSPARENT = 0, 1, 2, 3, 4, 5
# v1c_plan_last_launch (tests / bench): which kernels served the last launch group; | LAUNCH_FIXUP when a fix-up pass followed
LAUNCH_GENERIC, LAUNCH_TILE, LAUNCH_MIRROR, LAUNCH_CN, LAUNCH_CN_ROT, LAUNCH_BATCH, LAUNCH_ROT_PAIR, LAUNCH_FIXUP = 0, 1, 2, 3, 4, 5, 6, 0x100
LAUNCH_NAMES = {LAUNCH_GENERIC: "generic", LAUNCH_TILE: "tile", LAUNCH_MIRROR: "mirror", LAUNCH_CN: "cn", LAUNCH_CN_ROT: "cn_rot",
                LAUNCH_BATCH: "batch", LAUNCH_ROT_PAIR: "rot_pair"}

OP_NORMALIZE, OP_DENORMALIZE, OP_DENORMALIZE_INV, OP_ZOOM, OP_ZOOM_INV = 1, 2, 3, 4, 5
OP_EQUIRECT_ENC, OP_EQUIRECT_DEC, OP_RADIAL, OP_ROTATE = 6, 7, 8, 9

RAD_ENC_RECTILINEAR, RAD_ENC_STEREOGRAPHIC, RAD_ENC_EQUIDISTANT, RAD_ENC_EQUISOLID, RAD_ENC_ORTHOGRAPHIC = 1, 2, 3, 4, 5
RAD_DEC_RECTILINEAR, RAD_DEC_STEREOGRAPHIC, RAD_DEC_EQUIDISTANT, RAD_DEC_EQUISOLID, RAD_DEC_ORTHOGRAPHIC = 6, 7, 8, 9, 10
RAD_POLYNOMIAL, RAD_RECTDEC_FWD, RAD_RECTDEC_INV = 11, 12, 13


class Op(C.Structure):
    _fields_ = [
        ("opcode", C.c_int32),
        ("iparam", C.c_int32),
        ("nparam", C.c_int32),
        ("reserved", C.c_int32),
        ("p", C.c_double * MAX_PARAMS),
    ]

    def __repr__(self) -> str:  # pragma: no cover - debugging aid
        return f"Op({self.opcode}, {self.iparam}, {list(self.p[: self.nparam])})"


class Chain(C.Structure):
    _fields_ = [("n_ops", C.c_int32), ("reserved", C.c_int32), ("ops", Op * MAX_OPS)]

    def key(self) -> bytes:
        return bytes(self)


class Unit(C.Structure):
    _fields_ = [
        ("src", C.c_void_p),
        ("dst", C.c_void_p),
        ("src_pitch", C.c_int64),
        ("dst_pitch", C.c_int64),
        ("rot", C.c_double * 9),
        ("has_rot", C.c_int32),
        ("reserved", C.c_int32),
    ]


def op(opcode: int, iparam: int = 0, params: Sequence[float] = ()) -> Op:
    params = list(params)
    if len(params) > MAX_PARAMS:
        raise ValueError("too many parameters for one op")
    o = Op()
    o.opcode, o.iparam, o.nparam = int(opcode), int(iparam), len(params)
    for i, v in enumerate(params):
        o.p[i] = float(v)
    return o


def chain(ops: Sequence[Op]) -> Chain:
    if not 1 <= len(ops) <= MAX_OPS:
        raise ValueError("chain length out of range")
    ch = Chain()
    ch.n_ops = len(ops)
    for i, o in enumerate(ops):
        ch.ops[i] = o
    return ch
