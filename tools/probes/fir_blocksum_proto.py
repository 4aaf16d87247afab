#!/usr/bin/env python3
"""Prototype of the block-sum form of the filter bank's FIR filters (fbearmodel.c:399-435):
Hann window = three rectangular windows, running sums over 32-sample blocks, true coefficients on the two
edge blocks.  Checks the algebra against the direct sum in long double."""
import numpy as np

LEN = [1456, 1438, 1406, 1362, 1308, 1244, 1176, 1104, 1030, 956, 884, 814, 748, 686, 626, 570, 520, 472, 430, 390,
       354, 320, 290, 262, 238, 214, 194, 176, 158, 144, 130, 118, 106, 96, 86, 78, 70, 64, 58, 52]
LD = np.longdouble


def fc(b):
    return np.sinh(np.arcsinh(50. / 650.) + b * (np.arcsinh(18000. / 650.) - np.arcsinh(50. / 650.)) / 39.) * 650.


def direct(x, b, T, prec=LD):
    """y_b(t), t = 0..T-1, window coordinates: sample of tap m at u = 32 t + 727 + m"""
    N = LEN[b]
    m = np.arange(-(N // 2 - 1), N // 2)
    w = 2 * LD(np.pi) * LD(fc(b)) / 48000
    win = 4 / LD(N) * np.cos(LD(np.pi) * m / N) ** 2
    h = (win * np.exp(-1j * (w * m).astype(LD))).astype(np.clongdouble)
    out = np.zeros(T, np.clongdouble)
    for t in range(T):
        seg = x[32 * t + 727 + m].astype(prec)
        out[t] = np.sum(h.astype(np.complex128 if prec is float else np.clongdouble) * seg)
    return out


def blocksum(x, b, T):
    N = LEN[b]
    u_lo, u_hi = 728 - N // 2, 726 + N // 2
    cL, cR = u_lo >> 5, u_hi >> 5
    assert cR - 1 >= cL + 1
    w0 = 2 * LD(np.pi) * LD(fc(b)) / 48000
    d = 2 * LD(np.pi) / N
    om = [w0, w0 + d, w0 - d]
    g = [2 / LD(N), 1 / LD(N), 1 / LD(N)]
    q = np.arange(32)

    def rows3(c):          # complex coefficient rows of a block at lag c, the three exponentials
        m = 32 * c + q - 727
        return [(g[i] * np.exp(-1j * (om[i] * m))).astype(np.complex128) for i in range(3)]

    def true_row(c):
        m = 32 * c + q - 727
        win = np.where(np.abs(m) < N // 2, 4 / LD(N) * np.cos(LD(np.pi) * m / N) ** 2, 0)
        return (win * np.exp(-1j * (w0 * m))).astype(np.complex128)
    en, lv, re_, le_ = rows3(cR - 1), rows3(cL), true_row(cR), true_row(cL)
    rot = [np.complex128(np.exp(1j * om[i] * 32)) for i in range(3)]
    blk = lambda k: x[32 * k:32 * k + 32]
    # state at t = -1 is zero when everything before the window is zero: start the walk J steps early instead
    J = cR - 1 - cL
    V = [0j, 0j, 0j]
    out = np.zeros(T, np.complex128)
    for t in range(-J, T):
        for i in range(3):
            e = np.dot(en[i], blk(t + cR - 1)) if t + cR - 1 >= 0 else 0
            l = np.dot(lv[i], blk(t + cL)) if t >= 0 else 0     # nothing leaves during the run-in
            V[i] = rot[i] * V[i] + e - l
        if t >= 0:
            out[t] = V[0] + V[1] + V[2] + np.dot(re_, blk(t + cR)) + np.dot(le_, blk(t + cL))
    return out, J


rng = np.random.default_rng(1)
T = 600
n = 32 * T + 1500
tt = np.arange(n)
cases = {
    "noise": rng.standard_normal(n) * 0.1,
    "saw": 2 * ((tt * 440.0 / 48000) % 1) - 1,
    "tone1k+noise-100dB": np.sin(2 * np.pi * 1000 * tt / 48000) + 1e-5 * rng.standard_normal(n),
    "burst": np.where((tt > 3000) & (tt < 3200), 1e5, 1.0) * rng.standard_normal(n) * 0.01,
}
for name, x in cases.items():
    worst = 0
    print(name)
    for b in [0, 1, 2, 5, 10, 15, 20, 23, 26, 30, 39]:
        N = LEN[b]
        if (726 + N // 2 >> 5) - 1 < (728 - N // 2 >> 5) + 1:
            continue
        ref = direct(x, b, T)
        d64 = direct(x, b, T, float)
        bs, J = blocksum(x, b, T)
        scale = np.max(np.abs(ref))
        e_bs = np.max(np.abs(bs - ref)) / scale
        e_d = np.max(np.abs(d64 - ref)) / scale
        # relative per output (weak outputs)
        r_bs = np.max(np.abs(bs - ref) / np.abs(ref))
        r_d = np.max(np.abs(d64 - ref) / np.abs(ref))
        print(f"  band {b:2d} N {N:4d} J {J:2d}: block-sum err/scale {float(e_bs):.2e} (direct f64 {float(e_d):.2e});"
              f" worst relative {float(r_bs):.2e} ({float(r_d):.2e})")
