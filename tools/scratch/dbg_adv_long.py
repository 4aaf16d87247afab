# development: advanced ODG of one synthetic pair at several lengths, FP64 engine vs split-FP16 engine vs the oracle
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import gstpeaq_amd, cases as case_defs, oracle_lib as orc
c64 = gstpeaq_amd.Context(0); c16 = gstpeaq_amd.Context(0); c16.set_fir_mode("f16x3")
for n in (100000, 170000, 330000, 480000):
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=7, channels=1, n=n))
    r = torch.from_numpy(np.ascontiguousarray(ref[None])).cuda(); t = torch.from_numpy(np.ascontiguousarray(test[None])).cuda()
    a = gstpeaq_amd.batch_run(c64, 1, r, t)[0]; b = gstpeaq_amd.batch_run(c16, 1, r, t)[0]
    o = orc.run_pair(1, ref, test) if n <= 170000 else None
    print(n, "blocks", n // 192, "f64", a["odg"], "f16x3", b["odg"], "oracle", None if o is None else o["odg"], flush=True)
