#!/usr/bin/env python3
"""Development: raw outputs of the 40 FIR filters of the filter bank (library built with
   make -C gstpeaq_amd/csrc VARIANT=dumpfir EXTRA="-DPEAQ_DEV_PROBES -DPEAQ_DEV_DUMP_FIR")
against the plain sums in numpy (long double), per band.  usage: dbg_fir.py [f64|default] [blocks_per_launch]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
os.environ.setdefault("PEAQ_AMD_LIB", str(ROOT / "gstpeaq_amd" / "libpeaq_amd_dumpfir.so"))
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch, gstpeaq_amd
SUB = int(os.environ.get("DUMP_SUB", "5"))
mode = sys.argv[1] if len(sys.argv) > 1 else "f64"
bpl = int(sys.argv[2]) if len(sys.argv) > 2 else 320
LEN = [1456, 1438, 1406, 1362, 1308, 1244, 1176, 1104, 1030, 956, 884, 814, 748, 686, 626, 570, 520, 472, 430, 390,
       354, 320, 290, 262, 238, 214, 194, 176, 158, 144, 130, 118, 106, 96, 86, 78, 70, 64, 58, 52]
ctx = gstpeaq_amd.Context(0)
if mode == "f64":
    ctx.set_fir_fp64(True)
rng = np.random.default_rng(3)
n = 192 * 45
x = (0.2 * rng.standard_normal(n) + 0.3 * np.sin(2 * np.pi * 1000 * np.arange(n) / 48000)).astype(np.float32)
ref = torch.from_numpy(x[:, None].copy()).cuda()
got = gstpeaq_amd.debug_filterbank(ctx, ref, ref, n // 192, bpl)
# the high-passed, level-scaled signal as the kernel sees it
lev = 10 ** (92 / 20)
hp = np.zeros(n); x1 = x2 = y1a = y2a = y1b = y2b = 0.0
for k in range(n):
    v = float(x[k]) * lev
    ya = v - 2. * x1 + x2 + 1.99517 * y1a - 0.995174 * y2a
    yb = ya - 2. * y1a + y2a + 1.99799 * y1b - 0.997998 * y2b
    x2, x1, y2a, y1a, y2b, y1b = x1, v, y1a, ya, y1b, yb
    hp[k] = yb
sig = np.concatenate([np.zeros(1456), hp]).astype(np.longdouble)      # index 1456 + k = sample k
LD = np.longdouble
def fc(b): return np.sinh(np.arcsinh(50. / 650.) + b * (np.arcsinh(18000. / 650.) - np.arcsinh(50. / 650.)) / 39.) * 650.
def wt(f):
    f = f / 1000.0
    return 10 ** ((-0.6 * 3.64 * f ** -0.8 + 6.5 * np.exp(-0.6 * (f - 3.3) ** 2) - 1e-3 * f ** 3.6) / 20)
nb = n // 192
for b in range(40):
    N = LEN[b]
    m = np.arange(-(N // 2 - 1), N // 2)
    w = 2 * LD(np.pi) * LD(fc(b)) / 48000
    h = 4 / LD(N) * np.cos(LD(np.pi) * m / N) ** 2 * wt(fc(b)) * np.exp(-1j * (w * m))
    exp = np.zeros(nb, np.clongdouble)
    for bl in range(nb):
        t = 192 * bl + 32 * SUB          # newest sample index of sub-sample 5 of block bl: output uses delays 1.. from it
        # window coordinate: centre tap at delay 729 from the newest sample -> sample t - 729
        exp[bl] = np.sum(h * sig[1456 + t - 729 + m])
    g = got[:, 0, b] + 1j * got[:, 0, 80 + b]
    err = np.abs(g - exp.astype(np.complex128))
    sc = np.max(np.abs(exp))
    print(f"band {b:2d} N {N:4d}: max err/scale {float(err.max() / sc):.2e} at block {int(err.argmax())}  nan {int(np.isnan(g).sum())}"
          + ("" if b else "   (band 0: aliased tap not in the numpy sum)"))
