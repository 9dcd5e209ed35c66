"""CPU model of the LDS bank conflicts of the bilinear pair tap gather (ds_read2_b64 on 8-byte cells) on the real C2 / C1 maps:
lane -> pixel mapping A (kernel: 16 lanes x 4 rows per wave) vs G (16-lane groups of 4 lane columns x 4 rows) and the box pitch."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import vr180_convert_amd as V  # noqa: E402  (the maps come from the engine's own get_map: run on the GPU box)
from vr180_convert_amd import transformer as T  # noqa: E402

def conflicts(ix, iy, x0, y0, pitch, mapping):
    """ix, iy: (16 rows, 64 px) source integer coords of one tile; returns LDS cycles for all ds_read2_b64 of the tile (ideal: 8 per instr)"""
    total = 0; n = 0
    for wave in range(4):
        for k in range(4):
            lane = np.arange(64)
            if mapping == 'A':
                lx, ly = lane & 15, lane >> 4
            else:
                lx, ly = (lane & 3) + 4 * (lane >> 4), (lane >> 2) & 3
            px = 4 * lx + k; row = 4 * wave + ly
            cell = (iy[row, px] - y0) * pitch + (ix[row, px] - x0)
            for tap_row in (0, 1):
                c0 = cell + tap_row * pitch
                for off in (0, 1):  # the two accesses of ds_read2_b64
                    c = c0 + off
                    for g in range(4):  # 4 groups of 16 lanes
                        cc = np.unique(c[16 * g:16 * g + 16])  # same address: broadcast
                        banks = np.concatenate([(2 * cc) % 32, (2 * cc + 1) % 32])  # 8 bytes = 2 banks; bank = dword mod 32
                        total += np.bincount(banks, minlength=32).max()
                n += 2
    return total, n

def run(name, chain, size, radius):
    xm, ym = V.get_map(chain, radius=radius, size_input=(size, size), size_output=(size, size))
    sx = np.rint(xm.astype(np.float64) * 32).astype(np.int64); sy = np.rint(ym.astype(np.float64) * 32).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    res = {}
    rng = np.random.default_rng(0)
    tiles = [(ty, tx) for ty in range(size // 16) for tx in range(size // 64)]
    sel = rng.choice(len(tiles), 600, replace=False)
    for mapping, podd in (('A', 0), ('A', 1), ('G', 0), ('G', 1)):
        tot = cnt = 0
        for s in sel:
            ty, tx = tiles[s]
            bx, by = ix[16 * ty:16 * ty + 16, 64 * tx:64 * tx + 64], iy[16 * ty:16 * ty + 16, 64 * tx:64 * tx + 64]
            x0 = bx.min() & ~3; y0 = by.min(); cpr = (bx.max() + 2 - x0 + 3) >> 2
            pitch = 4 * cpr + 4 + podd
            t, n = conflicts(bx, by, x0, y0, pitch, mapping)
            tot += t; cnt += n
        res[(mapping, podd)] = tot / cnt
        print(f'{name}: mapping {mapping} pitch 4cpr+{4+podd}: {2 * tot / cnt:.2f} LDS cycles per ds_read2_b64 (conflict-free: 8)')
run('C2', T.EquirectangularEncoder() * T.PolynomialScaler([0, 1, -0.1]) * T.FisheyeDecoder("equidistant"), 4096, 2048.0)
run('C1', T.EquirectangularEncoder() * T.FisheyeDecoder("equidistant"), 2048, 1024.0)
