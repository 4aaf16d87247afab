#!/usr/bin/env python3
"""Mixed-precision ledger: what does FP32 cost in ODG, part by part?

The baseline is the engine with everything in FP64 (PEAQ_AMD_FIR=f64).  Compared with it:
  * the engine's default, in which the filter bank's FIR filters run on the FP16 matrix instruction with both
    operands split into two FP16 parts and three products per term (peaq_fb.hip, fir_mfma_h3) -- this table
    is the evidence behind that default -- and the FP32 matrix instruction (fir_mfma_f32, PEAQ_AMD_FIR=f32);
  * an experimental build with the back end's loudness / detection pow, log, exp in FP32 (report only):
      make -C gstpeaq_amd/csrc VARIANT=fp32be EXTRA=-DPEAQ_LEDGER_FP32_BACKEND
Every end-to-end golden case of tests/golden/ref_e2e*.json (62 cases, basic and advanced) and 8 full-size
seeded pairs (10 s stereo) per version go through each configuration; tabulated are max |dODG|, max |dDI| and
the per-MOV max relative deviation from the all-FP64 baseline (with the case it occurs in).

  python tools/precision_ledger.py [--out profiles/r02_precision_ledger.json] ["name=lib.so[,ENV=VALUE]" ...]
(the worker mode `--dump` is internal: one process per configuration)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def dump():
    import torch
    import cases as case_defs
    import gstpeaq_amd
    ctx = gstpeaq_amd.Context(0)
    out = []
    recs = json.loads((ROOT / "tests" / "golden" / "ref_e2e.json").read_text())
    recs += json.loads((ROOT / "tests" / "golden" / "ref_e2e_level.json").read_text())
    for rec in recs:
        case = rec["case"]
        ref, test = case_defs.make_inputs(case)
        n = max(len(ref), len(test))
        n += n & 1
        ch = ref.shape[1]
        r = np.zeros((1, n, ch), dtype=np.float32)
        t = np.zeros_like(r)
        r[0, : len(ref)] = ref
        t[0, : len(test)] = test
        res = gstpeaq_amd.batch_run(ctx, case["advanced"], torch.from_numpy(r).cuda(), torch.from_numpy(t).cuda(),
                                    np.array([len(ref)], dtype=np.uint32), np.array([len(test)], dtype=np.uint32),
                                    playback_level=case.get("level", 92.0))[0]
        out.append(dict(name=f"{case['name']}{'_adv' if case['advanced'] else ''}", advanced=case["advanced"],
                        movs=[float(v) for v in res["movs"]], di=res["di"], odg=res["odg"]))
    for adv in (0, 1):
        ref, test = gstpeaq_amd.synth_fill(ctx, 1, 8, 2, 480000)
        for i, res in enumerate(gstpeaq_amd.batch_run(ctx, adv, ref, test)):
            out.append(dict(name=f"fullsize_seed{1 + i}{'_adv' if adv else ''}", advanced=adv,
                            movs=[float(v) for v in res["movs"]], di=res["di"], odg=res["odg"]))
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--dump":
        return dump()
    args = sys.argv[1:]
    out_path = ROOT / "profiles" / "r02_precision_ledger.json"
    if args and args[0] == "--out":
        out_path = Path(args[1])
        args = args[2:]
    libs = {"all FP64 (baseline)": ("", {"PEAQ_AMD_FIR": "f64"}),
            "engine default: FIR bank on v_mfma_f32_16x16x32_f16, operands split in two FP16 parts, 3 products per term": ("", {}),
            "FIR bank on v_mfma_f32_16x16x4_f32": ("", {"PEAQ_AMD_FIR": "f32"}),
            "all FP64 again (run-to-run variation)": ("", {"PEAQ_AMD_FIR": "f64"})}
    for a in args:
        name, spec = a.split("=", 1)
        parts = spec.split(",")
        libs[name] = (str(Path(parts[0]).resolve()) if parts[0] else "", dict(p.split("=", 1) for p in parts[1:]))
    results = {}
    for name, (path, extra_env) in libs.items():
        env = dict(os.environ)
        env.pop("PEAQ_AMD_LIB", None)
        env.pop("PEAQ_AMD_FIR_FP64", None)
        env.pop("PEAQ_AMD_FIR", None)
        env.update(extra_env)
        if path:
            env["PEAQ_AMD_LIB"] = path
        o = subprocess.run([sys.executable, __file__, "--dump"], capture_output=True, text=True, env=env, timeout=900)
        if o.returncode != 0:
            raise SystemExit(f"{name}: {o.stderr[-2000:]}")
        results[name] = json.loads(o.stdout.strip().splitlines()[-1])
    from gstpeaq_amd.capi import MOV_NAMES_ADVANCED, MOV_NAMES_BASIC
    base = results["all FP64 (baseline)"]
    ledger = {"_what": __doc__.split("\n\n")[0], "cases": len(base)}
    for name, res in results.items():
        if name == "all FP64 (baseline)":
            continue
        entry = {}
        for adv, label, names in ((0, "basic", MOV_NAMES_BASIC), (1, "advanced", MOV_NAMES_ADVANCED)):
            d_odg = d_di = 0.0
            worst = None
            mov = {n: 0.0 for n in names}
            mov_where = {n: None for n in names}
            nan_mismatch = 0
            for b, v in zip(base, res):
                if b["advanced"] != adv:
                    continue
                if np.isnan(b["odg"]) or np.isnan(v["odg"]):
                    nan_mismatch += int(np.isnan(b["odg"]) != np.isnan(v["odg"]))
                    continue
                if abs(b["odg"] - v["odg"]) > d_odg:
                    d_odg, worst = abs(b["odg"] - v["odg"]), b["name"]
                d_di = max(d_di, abs(b["di"] - v["di"]))
                for n, x, y in zip(names, b["movs"], v["movs"]):
                    if not (np.isnan(x) or np.isnan(y)):
                        rel = abs(x - y) / max(abs(x), abs(y), 1e-12)
                        if rel > mov[n]:
                            mov[n], mov_where[n] = rel, dict(case=b["name"], fp64=x, variant=y)
            entry[label] = dict(max_abs_dODG=d_odg, worst_case=worst, max_abs_dDI=d_di, mov_max_rel=mov,
                                mov_max_rel_where=mov_where, nan_mismatches=nan_mismatch)
        ledger[name] = entry
    out_path.write_text(json.dumps(ledger, indent=1) + "\n")
    print(json.dumps(ledger, indent=1))


if __name__ == "__main__":
    main()
