#!/usr/bin/env python3
"""Design the integer FIR low-pass tables used by include/peaq_synth.h.

Run once (needs scipy); the printed tables are pasted into the header so that
the generator itself is pure 32-bit integer arithmetic and therefore
bit-identical in C, HIP and numpy ("integer PRNG ... same bits
in C++ host, HIP device and the Python golden generator").
"""
import numpy as np
from scipy import signal

NT = 63          # taps (linear phase, delay 31)
FS = 48000.0
cutoffs = [18000.0, 16000.0, 14000.0, 12000.0, 10000.0]   # [0] = reference
for i, fc in enumerate(cutoffs):
    h = signal.firwin(NT, fc, window=("kaiser", 7.0), fs=FS)
    q = np.round(h * 32768).astype(np.int64)
    q[NT // 2] += 32768 - q.sum()      # exact DC gain 2^15
    w, H = signal.freqz(q / 32768.0, worN=4096, fs=FS)
    sb = 20 * np.log10(np.abs(H[w >= fc + 3300]).max())
    print(f"/* fc={fc:.0f} Hz, stop-band (>= fc+3.3 kHz) {sb:.1f} dB */")
    print("{" + ", ".join(str(int(v)) for v in q) + "},")
