import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import gpu_common as gpu, cases as case_defs, gstpeaq_amd
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
gpu.set_mode(mode)
ref, test = case_defs.make_inputs(dict(kind="synth", seed=22, channels=2, n=31000, test_trim=700))
for piece in (500, 192, 4000, 40000):
    s = gstpeaq_amd.Session(gpu.ctx(), 1, 2)
    for lo in range(0, len(ref), piece):
        s.push_ref(ref[lo:lo + piece]); s.push_test(test[lo:lo + piece])
    s.flush(); r = s.results(); s.close()
    print(mode, "piece", piece, r["odg"], r["movs"])
