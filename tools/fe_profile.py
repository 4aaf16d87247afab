#!/usr/bin/env python3
"""Per-phase cycle breakdown of the front-end kernel (development tool).
Needs a library built with -DPEAQ_FE_PROFILE:
  make -C gstpeaq_amd/csrc VARIANT=prof EXTRA=-DPEAQ_FE_PROFILE
  PEAQ_AMD_LIB=gstpeaq_amd/libpeaq_amd_prof.so python tools/fe_profile.py [pairs]
Prints the mean shader cycles a wave spends between consecutive marks (s_memtime), per wave role."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401
import gstpeaq_amd  # noqa: E402

PHASES = ["reductions+flags (after the loads)", "FFT+split", "zero threshold of the bandwidth search", "band grouping", "log/exp per band",
          "upward spreading", "downward+excitation+store", "barrier", "log ratios", "barrier",
          "ref: FFT-512 | test: noise+grouping", "ref: product+inverse", "ref: normalise+FFT-256+peak",
          "work-item decoding (after start-up)", "sample loads + window", "wave start-up + kernel arguments"]
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ADV = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = gstpeaq_amd.Context(0)
ref, test = gstpeaq_amd.synth_fill(ctx, 1, pairs, 2, 480000)
gstpeaq_amd.batch_run(ctx, ADV, ref, test)
buf = (C.c_ulonglong * 64)()
ctx.L.peaq_debug_frontend_profile.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
assert ctx.L.peaq_debug_frontend_profile(ctx.h, buf) == 0
gstpeaq_amd.batch_run(ctx, ADV, ref, test)
assert ctx.L.peaq_debug_frontend_profile(ctx.h, buf) == 0
out = {}
for sig, name in ((0, "ref"), (1, "test")):
    n = buf[32 + sig]
    tot = sum(buf[sig * 16 + i] for i in range(16))
    out[name] = {"waves": n, "total_cycles_per_wave": tot / n,
                 "phases": {f"{i:2d} {PHASES[i]}": round(buf[sig * 16 + i] / n, 1) for i in range(len(PHASES))}}
print(json.dumps(out, indent=1))
