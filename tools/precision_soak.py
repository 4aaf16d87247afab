#!/usr/bin/env python3
"""How far does the advanced version's split-FP16 engine (FP32 slopes and spreading; the default until round 4) get from the
all-FP64 one over MANY pairs?  N batches of 4096 seeded 10 s stereo pairs through both; distribution of |dODG|
and |dDI|.  (The ledger, tools/precision_ledger.py, covers the goldens; this is the tail.)
  python tools/precision_soak.py [batches] > profiles/r02_precision_soak.json"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401
import gstpeaq_amd  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = gstpeaq_amd.Context(0)
d_odg, d_di, odg = [], [], []
for b in range(batches):
    seed0 = 1 + 5000 * b
    ref, test = gstpeaq_amd.synth_fill(ctx, seed0, 4096, 2, 480000)
    res = {}
    for mode in ("f16x3", "f64"):
        ctx.set_fir_mode(mode)
        r = torch.empty((4096, 16), dtype=torch.float64, device=ref.device)
        gstpeaq_amd.batch_run(ctx, 1, ref, test, results=r, sync=False)
        torch.cuda.synchronize()
        res[mode] = r.cpu().numpy()
    a, f = res["f16x3"], res["f64"]
    ok = ~(np.isnan(a[:, 12]) | np.isnan(f[:, 12]))
    d_odg.append(np.abs(a[ok, 12] - f[ok, 12]))
    d_di.append(np.abs(a[ok, 11] - f[ok, 11]))
    odg.append(f[ok, 12])
d_odg, d_di, odg = map(np.concatenate, (d_odg, d_di, odg))
q = lambda x, p: float(np.quantile(x, p))
print(json.dumps({"pairs": int(len(d_odg)), "what": "advanced PEAQ, 10 s stereo seeded pairs, split-FP16 engine (PEAQ_FIR_F16X3) vs the all-FP64 engine (the default since round 4: block-sum form)",
                  "odg_range": [float(odg.min()), float(odg.max())],
                  "abs_dODG": {"max": float(d_odg.max()), "p999": q(d_odg, 0.999), "p99": q(d_odg, 0.99), "median": q(d_odg, 0.5)},
                  "abs_dDI": {"max": float(d_di.max()), "p999": q(d_di, 0.999), "p99": q(d_di, 0.99), "median": q(d_di, 0.5)}},
                 indent=1))
