#!/usr/bin/env python3
"""Can OpenCV's 2-D Lanczos4 weight table (1024 entries x 64 int16 = 128 KB) be folded by symmetry so that it fits LDS next to the
source boxes?  (Round 3's verdict, Next 6: an LDS-resident table for C4.)  No:

* mirror symmetry (fx <-> 32 - fx with the taps reversed, likewise fy) fails for 84 % of the entries by up to 6 units -- the
  1-D coefficients are float32 sums normalised in index order and every product is rounded on its own;
* transposition (fy <-> fx with the 8 x 8 entry transposed) holds for all but 20 entries (fix-up ties), but a transposed entry would
  have to be read column-wise (2-byte strided weight reads) or the pixel window column-wise (8 LDS reads per tap row instead of 4).

So the table stays 128 KB, which leaves 32 KB of the CU's 160 KB for boxes: ONE 64 x 16 tile pair (22 - 24 KB of cells) in flight per CU,
i.e. one wave per SIMD -- the configuration round 1 measured at 16.9 ms against 1.3 ms with the weights through L2 (three workgroups
per CU).  Prints the counts (uses the oracle's restatement of initInterTab2D; the product's v1c_build_itab is tested equal to it).
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import oracle as O  # noqa: E402

O.build()
t = O.build_itab(4).astype(np.int32).reshape(32, 32, 8, 8)  # fy, fx, ky, kx
mx = [np.abs(t[fy, fx] - t[fy, 32 - fx][:, ::-1]).max() for fy in range(32) for fx in range(1, 32)]
my = [np.abs(t[fy, fx] - t[32 - fy, fx][::-1, :]).max() for fy in range(1, 32) for fx in range(32)]
tr = sum(not np.array_equal(t[fy, fx], t[fx, fy].T) for fy in range(32) for fx in range(32))
print(f"x mirror: {sum(v > 0 for v in mx)} of {len(mx)} entries differ, largest difference {max(mx)}")
print(f"y mirror: {sum(v > 0 for v in my)} of {len(my)} entries differ, largest difference {max(my)}")
print(f"transposition: {tr} of 1024 entries differ")
print("table bytes:", t.size * 2, "-> LDS left for boxes:", 160 * 1024 - t.size * 2)
