#!/usr/bin/env python3
"""Development: per-block excitation of the filter bank (product library, FP64 engine) against the oracle."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch, gstpeaq_amd
import oracle_lib as orc, cases as case_defs
bpl = int(sys.argv[1]) if len(sys.argv) > 1 else 320
ctx = gstpeaq_amd.Context(0); ctx.set_fir_fp64(True)
ref, test = case_defs.make_inputs(dict(kind="synth", seed=5, channels=1, n=30000))
nb = len(ref) // 192
got = gstpeaq_amd.debug_filterbank(ctx, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(), nb, bpl)
exp = orc.fbear(np.ascontiguousarray(ref[:, 0]), nb)
rel = np.abs(got[:, 0, 0:40] - exp["unsmeared"]) / exp["unsmeared"]
np.set_printoptions(linewidth=250, precision=1)
print("blocks x bands with rel err > 1e-9 (unsmeared):")
for bl in range(min(nb, 40)):
    bad = np.nonzero(rel[bl] > 1e-9)[0]
    print(bl, "max %.1e" % rel[bl].max(), "bands", bad.tolist()[:40])
