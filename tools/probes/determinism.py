"""two runs of the advanced version on the same pairs: bit-equal?  (development check; the test is in tests/)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import torch
import gstpeaq_amd
ctx = gstpeaq_amd.Context(0)
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ref, test = gstpeaq_amd.synth_fill(ctx, 1, pairs, 2, 480000)
res = []
for i in range(3):
    out = torch.empty((pairs, 16), dtype=torch.float64, device="cuda")
    gstpeaq_amd.batch_run(ctx, 1, ref, test, results=out)
    res.append(out.cpu().numpy().copy())
for i in (1, 2):
    same = np.array_equal(res[0].view(np.uint64), res[i].view(np.uint64))
    d = np.nanmax(np.abs(res[0] - res[i]))
    print("run 0 vs run", i, "bit-equal" if same else "DIFFERENT", "max abs diff", d)
