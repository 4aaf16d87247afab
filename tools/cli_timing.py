#!/usr/bin/env python3
"""Wall time of the stand-alone CLI (gstpeaq_amd/cli/peaq) on 16-bit stereo WAV files of a few lengths, basic and
advanced: what one user with one file pair sees (process start and HIP initialisation included)."""
import subprocess
import sys
import time
import wave
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import synth_np  # noqa: E402

cli = ROOT / "gstpeaq_amd" / "cli" / "peaq"
out = Path("/tmp/cli_timing")
out.mkdir(exist_ok=True)
for seconds in (10, 60, 300):
    n = seconds * 48000
    r, t = synth_np.pair(7, 2, min(n, 480000))
    reps = -(-n // len(r))
    r, t = np.tile(r, (reps, 1))[:n], np.tile(t, (reps, 1))[:n]
    for name, x in (("ref", r), ("test", t)):
        with wave.open(str(out / f"{name}.wav"), "wb") as w:
            w.setnchannels(2)
            w.setsampwidth(2)
            w.setframerate(48000)
            w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    for mode in ("--basic", "--advanced"):
        t0 = time.perf_counter()
        o = subprocess.run([str(cli), mode, str(out / "ref.wav"), str(out / "test.wav")], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        odg = [ln for ln in o.stdout.splitlines() if ln.startswith("Objective")]
        print(f"{seconds:4d} s stereo {mode:10s} {dt:6.2f} s wall  ({seconds / dt:7.1f} x real time)  {odg[0] if odg else o.stderr[-200:]}")
