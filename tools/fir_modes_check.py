#!/usr/bin/env python3
"""Quick look at the three arithmetics of the advanced version's filter bank (include/peaq_amd.h PEAQ_FIR_*):
per-block excitation of the reduced ones against the FP64 one on three signals, and max |dODG| of each against
the reference's 27 advanced goldens.  Development tool (the tests proper: tests/test_gpu_fir_modes.py; the
ledger: tools/precision_ledger.py).  Needs an MI355X."""
import sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import gstpeaq_amd, cases as case_defs
ctx = gstpeaq_amd.Context(0)
for case in [dict(kind="synth", seed=5, channels=1, n=40000), dict(kind="ats", wave_ref="saw", wave_test="triangle", n=32768, channels=1),
             dict(kind="synth", seed=6, channels=2, n=30000, test_trim=900)]:
    ref, test = case_defs.make_inputs(case)
    nb = min(len(ref), len(test)) // 192
    out = {}
    for mode in ("f64", "f32", "f16x3"):
        ctx.set_fir_mode(mode)
        out[mode] = gstpeaq_amd.debug_filterbank(ctx, torch.from_numpy(ref).cuda(), torch.from_numpy(test).cuda(), nb, 320)
    for mode in ("f32", "f16x3"):
        d = np.abs(out[mode][:, :, :160] - out["f64"][:, :, :160]) / np.maximum(np.abs(out["f64"][:, :, :160]), 1e-300)
        print(case.get("seed", "ats"), mode, "max rel dev of block records vs f64: %.3e" % np.nanmax(d), "nan", np.isnan(out[mode]).sum())
recs = [r for r in json.load(open('/root/repo/tests/golden/ref_e2e.json')) if r["case"]["advanced"]]
for mode in ("f64", "f32", "f16x3"):
    ctx.set_fir_mode(mode)
    worst = 0.0; wm = 0.0
    for rec in recs:
        case = rec["case"]
        ref, test = case_defs.make_inputs(case)
        n = max(len(ref), len(test), 2); n += n & 1
        r = np.zeros((1, n, ref.shape[1]), dtype=np.float32); t = np.zeros_like(r)
        r[0, :len(ref)] = ref; t[0, :len(test)] = test
        got = gstpeaq_amd.batch_run(ctx, 1, torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda(),
                                    np.array([len(ref)], dtype=np.uint32), np.array([len(test)], dtype=np.uint32))[0]
        if not np.isnan(float(rec["odg"])):
            worst = max(worst, abs(got["odg"] - float(rec["odg"])))
            exp = np.array([float(v) for v in rec["movs"]]); ok = ~np.isnan(exp)
            wm = max(wm, np.max(np.abs(got["movs"][:5][ok[:5]] - exp[:5][ok[:5]]) / np.maximum(np.abs(exp[:5][ok[:5]]), 1e-12)))
    print(mode, "max |dODG| vs reference over %d advanced goldens: %.3e, max rel MOV dev %.3e" % (len(recs), worst, wm))
