#!/usr/bin/env python3
"""Per-phase cycle breakdown of the filter-bank kernel of the advanced version (development tool).
Needs a library built with -DPEAQ_FB_PROFILE:
  make -C gstpeaq_amd/csrc VARIANT=fbprof EXTRA=-DPEAQ_FB_PROFILE
  PEAQ_AMD_LIB=gstpeaq_amd/libpeaq_amd_fbprof.so python tools/fb_profile.py [pairs]
Prints the mean shader cycles a wave spends between consecutive marks per tile of 10 blocks, per wave."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402,F401
import gstpeaq_amd  # noqa: E402

# (FP64 engine since round 5: 1 = the direct tile, dealt by output; 3 = pick-up only; 4 = slope exchange + barrier; 6 = upward spreading by
# target band + barrier + new A + window shift -- the reduced-precision engines keep the meanings of round 4)
PHASES = ["0 window + zero A + barrier", "1 FIR (own run of K steps / direct tile)", "2 barrier after the FIR", "3 pick up bands (+ barrier)",
          "4 shift window, request next / slope exchange + barrier", "5 log/exp/slope scan (10 bands)", "6 upward spreading", "7 barrier",
          "8 downward spreading + barrier", "9 backward masking + barrier", "10 history + barrier",
          "11 forward masking + barrier", "12 records",
          "13 FP64 engine: block-sum form of bands 0..23 (inside 1)", "14 FP64 engine: barrier, results -> A (inside 1)", "15 records: reads and stores (12 = the rest: next window into LDS)"]
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODE = sys.argv[2] if len(sys.argv) > 2 else "f64"     # "f64" (the engine's default) | "f16x3" | "f32"
ctx = gstpeaq_amd.Context(0)
ctx.set_fir_mode(MODE)
ref, test = gstpeaq_amd.synth_fill(ctx, 1, pairs, 2, 480000)
buf = (C.c_ulonglong * 68)()
ctx.L.peaq_debug_fb_profile.argtypes = [C.POINTER(C.c_ulonglong)]
gstpeaq_amd.batch_run(ctx, 1, ref, test)
assert ctx.L.peaq_debug_fb_profile(buf) == 0
gstpeaq_amd.batch_run(ctx, 1, ref, test)
assert ctx.L.peaq_debug_fb_profile(buf) == 0
out = {"fir_mode": ctx.fir_mode(), "fir_fp64": ctx.fir_fp64()}
for wv in range(4):
    n = buf[64 + wv]
    tot = sum(buf[wv * 16 + i] for i in range(16))
    out[f"wave{wv}"] = {"tiles": n, "cycles_per_tile": round(tot / n, 1),
                        "phases": {PHASES[i]: round(buf[wv * 16 + i] / n, 1) for i in range(len(PHASES))}}
print(json.dumps(out, indent=1))
