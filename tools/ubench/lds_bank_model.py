"""Calibrate LDS banking models against tools/ubench/lds_tap_mapping.hip measurements (scale 1.0 / 0.85, angles, pitches)."""
import numpy as np, itertools, math

def lanes_xy(mapping, wave):
    lane = np.arange(64)
    out = []
    for kk in range(4):
        if mapping in 'FG':
            x = 4 * ((lane & 3) + 4 * (lane >> 4)) + kk; y = ((lane >> 2) & 3) + 4 * wave
        elif mapping in 'HI':
            x = 16 * kk + (lane & 15); y = (lane >> 4) + 4 * wave
        else:
            x = 4 * (lane & 15) + kk; y = (lane >> 4) + 4 * wave
        out.append((x, y))
    return out

def cost_access(addr_dw, width_dw, group, nbanks):
    """addr_dw: per-lane dword address; width in dwords; lanes serviced in groups of `group`; returns cycles"""
    tot = 0
    for g in range(0, 64, group):
        a = np.unique(addr_dw[g:g + group])
        banks = np.concatenate([(a + w) % nbanks for w in range(width_dw)])
        tot += np.bincount(banks, minlength=nbanks).max()
    return tot

def model(mapping, scale, deg, pitch, b64, group, nbanks):
    r = math.radians(deg); a, b, c, d = scale * math.cos(r), -scale * math.sin(r), scale * math.sin(r), scale * math.cos(r)
    tot = n = 0
    for wave in range(4):
        for (x, y) in lanes_xy(mapping, wave):
            ix = np.floor(np.float32(40.3) + np.float32(a) * x + np.float32(b) * y).astype(int)
            iy = np.floor(np.float32(30.6) + np.float32(c) * x + np.float32(d) * y).astype(int)
            p = (iy * pitch + ix) & 4095
            for it in range(4):
                pp = p + it
                if b64:
                    pp = pp & 2047
                    for base in (pp, pp + pitch):
                        for off in (0, 1):
                            tot += cost_access(2 * (base + off), 2, group, nbanks)
                        n += 1
                else:
                    for base in (pp, pp + pitch):
                        for off in (0, 1):
                            tot += cost_access(base + off, 1, group, nbanks)
                        n += 1
    return tot / n

meas = {  # (scale, angle, pitch): {mapping: cycles}  from the r02 run (8 waves/SIMD)
 (1.0, 0, 76): dict(A=17.01, D=32.66, F=17.17, G=33.15), (1.0, 0, 77): dict(A=11.01, D=32.99, F=9.81, G=13.40),
 (1.0, 10, 77): dict(A=10.86, D=17.77, F=9.84, G=19.03), (1.0, 25, 76): dict(A=14.09, D=27.70, F=14.65, G=32.57),
 (0.85, 0, 76): dict(A=10.91, D=16.64, F=10.91, G=18.81), (0.85, 10, 77): dict(A=15.50, D=29.42, F=9.89, G=27.74),
 (0.7, 25, 81): dict(A=11.88, D=18.64, F=9.77, G=18.73), (1.0, 45, 77): dict(A=10.93, D=16.80, F=10.44, G=18.95),
}
for (grp32, nb32), (grp64, nb64) in itertools.product([(32, 32), (16, 32), (32, 64)], [(16, 32), (16, 64), (32, 64), (8, 32), (32, 32)]):
    err = 0; rows = []
    for (s, dg, pt), m in meas.items():
        for k, v in m.items():
            b64 = k in 'DG'
            c = model(k, s, dg, pt, b64, grp64 if b64 else grp32, nb64 if b64 else nb32)
            rows.append((k, s, dg, pt, v, c))
            err += (math.log(c / v)) ** 2
    print(f'b32 groups {grp32} banks {nb32} | b64 groups {grp64} banks {nb64}: rms log err {math.sqrt(err / len(rows)):.3f}')
    if err / len(rows) < 0.05:
        for r in rows: print('   ', r)
