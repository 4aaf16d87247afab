#!/usr/bin/env python3
"""How many bits do the operands of the filter bank's FIR filters (fbearmodel.c:399-435) need?  CPU only, numpy.
The 40 complex FIRs are applied to a sawtooth (strong harmonics, weak bands between them) and to a seeded noise
signal with signal and coefficients rounded to B bits below their largest magnitude (what an error-free integer
slicing with B bits would deliver, truncation of cross products aside), against 80-bit accumulation of the
unrounded operands.  Printed: the largest relative error of |A|^2 over bands and time points.
Behind DESIGN.md 5, "Exact integer slicing": 22 bits are the split-FP16 engine's, 40 = five signed 8-bit digits,
48 = six.  profiles/r03_fir_bits.txt holds the output."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import synth_np  # noqa: E402

LD = np.longdouble
LEN = [1456, 1438, 1406, 1362, 1308, 1244, 1176, 1104, 1030, 956, 884, 814, 748, 686, 626, 570, 520, 472, 430, 390, 354,
       320, 290, 262, 238, 214, 194, 176, 158, 144, 130, 118, 106, 96, 86, 78, 70, 64, 58, 52]      # fbearmodel.c:57-61


def ear_weight(f):                                     # earmodel.c:702-709
    fk = f / 1000.
    return 10 ** ((-0.6 * 3.64 * fk ** -0.8 + 6.5 * np.exp(-0.6 * (fk - 3.3) ** 2) - 1e-3 * fk ** 3.6) / 20)


fc = [np.sinh(np.arcsinh(50 / 650) + b * (np.arcsinh(18000 / 650) - np.arcsinh(50 / 650)) / 39) * 650 for b in range(40)]
C = np.zeros((40, 1458), dtype=complex)                # coefficient of band b at delay d (all filters centred at 729)
for b, N in enumerate(LEN):
    D = 1 + (1456 - N) // 2
    for n in range(1, N // 2 + 1):
        s = np.sin(np.pi * n / N)
        c = 4.0 / N * s * s * ear_weight(fc[b]) * np.exp(2j * np.pi * fc[b] * (n - N / 2) / 48000)
        if n == N // 2:
            C[b, D + n] += c.real
        else:
            C[b, D + n] += c
            C[b, 1458 - D - n] += np.conj(c)


def fir(x, cre, cim, times):
    out = np.zeros((40, len(times)), dtype=complex)
    for k, t in enumerate(times):
        seg = x[32 * t - 1457:32 * t][::-1]            # delays 1 .. 1457
        out[:, k] = (cre[:, 1:] @ seg) + 1j * (cim[:, 1:] @ seg)
    return out


def rounded(a, bits):
    m = np.abs(a).max()
    if m == 0:
        return a
    s = 2.0 ** (bits - 1 - int(np.ceil(np.log2(m))))
    return np.round(a * s) / s


n = 48000
t = np.arange(n) / 48000.
signals = {"sawtooth 440 Hz": 0.8 * (2 * ((440 * t) % 1) - 1), "seeded noise": synth_np.pair(5, 1, n)[0][:, 0].astype(np.float64)}
times = np.arange(60, 1400, 7)
for name, x in signals.items():
    exact = np.abs(fir(x.astype(LD), C.real.astype(LD), C.imag.astype(LD), times).astype(complex)) ** 2
    for bits in (22, 35, 40, 48):
        cq = np.stack([rounded(C[b].real, bits) + 1j * rounded(C[b].imag, bits) for b in range(40)])
        got = np.abs(fir(rounded(x, bits).astype(LD), cq.real.astype(LD), cq.imag.astype(LD), times).astype(complex)) ** 2
        rel = np.abs(got - exact) / exact
        print(f"{name:16s} {bits:2d} bits: max relative error of |A|^2 {rel.max():.2e} (band {np.unravel_index(rel.argmax(), rel.shape)[0]}), "
              f"median {np.median(rel):.2e}")
