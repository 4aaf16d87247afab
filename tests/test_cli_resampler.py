"""The `peaq` CLI on files that are not sampled at 48 kHz, pinned to the REAL reference chain
(`rawaudioparse ! audioconvert ! audioresample ! peaq`, peaq.c:154-209; tests/golden/ref_e2e_resampled.json, made by
tools/make_golden.py resampled).  Runs without a GPU: PEAQ_AMD_CLI_DUMP makes the CLI write what it would hand to
the engine, the oracle (pinned against the reference elsewhere) takes it from there.

What is compared, and how closely two different resamplers CAN agree: the CLI's converter is a Kaiser-windowed
sinc with the parameters measured from GStreamer 1.14's audioresample (cutoff, length, beta, its delay of 1/8 input
sample, its output length); audioresample itself interpolates a tabulated kernel, which the fit follows to 7e-5 of
the peak.  Stated tolerance: |dODG|, |dDI| <= 5e-3 (measured: <= 6e-5 in seven of the eight cases, 2.3e-3 in the eighth; north star 0.02).
A pair whose test signal carries noise right up to ITS OWN Nyquist frequency makes the Bandwidth MOVs read the
converter's transition band: with the converter of round 2 (cutoff 0.96, 134 taps, no delay) the 32 kHz case was
0.035 off -- that sensitivity is the signal's, not the engine's."""
import json
import os
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases as case_defs
import oracle_lib as orc

ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "gstpeaq_amd" / "cli" / "peaq"
GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "ref_e2e_resampled.json").read_text())
TOL = 5e-3


def write_wav_f32(path, x, rate):
    body = x.astype("<f4").tobytes()
    ch = x.shape[1]
    fmt = struct.pack("<HHIIHH", 3, ch, rate, rate * ch * 4, ch * 4, 32)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    Path(path).write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def cli_dump(tmp_path, case):
    ref, test = case_defs.make_inputs(case)
    write_wav_f32(tmp_path / "r.wav", ref, case["rate"])
    write_wav_f32(tmp_path / "t.wav", test, case["rate"])
    env = dict(os.environ, PEAQ_AMD_CLI_DUMP=str(tmp_path / "dump"))
    out = subprocess.run([str(CLI), str(tmp_path / "r.wav"), str(tmp_path / "t.wav")], capture_output=True, text=True,
                         env=env, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    ch = case["channels"]
    return (np.fromfile(str(tmp_path / "dump") + ".ref.f32", dtype="<f4").reshape(-1, ch),
            np.fromfile(str(tmp_path / "dump") + ".test.f32", dtype="<f4").reshape(-1, ch))


@pytest.mark.skipif(not CLI.exists(), reason="gstpeaq_amd/cli/peaq not built (python -c 'import __graft_entry__ as g; g.build()')")
@pytest.mark.parametrize("rec", GOLD["records"], ids=lambda r: f"{r['case']['name']}-{'adv' if r['case']['advanced'] else 'basic'}")
def test_cli_conversion_follows_the_reference_chain(tmp_path, rec):
    case = rec["case"]
    r48, t48 = cli_dump(tmp_path, case)
    assert len(r48) == rec["samples_48k"] == len(t48)            # audioresample's output length
    e = orc.run_pair(case["advanced"], r48, t48)
    assert e["frames"] == rec["frames"]
    assert abs(e["odg"] - float(rec["odg"])) <= TOL and abs(e["di"] - float(rec["di"])) <= TOL, \
        (case["name"], e["odg"], rec["odg"], e["di"], rec["di"])
    print(f"{case['name']} adv={case['advanced']}: dODG {e['odg'] - float(rec['odg']):+.2e} dDI {e['di'] - float(rec['di']):+.2e}")


@pytest.mark.skipif(not CLI.exists(), reason="gstpeaq_amd/cli/peaq not built")
def test_converter_is_a_band_limited_interpolator_with_audioresamples_delay(tmp_path):
    """tones below the cutoff come out at 48 kHz delayed by 1/8 input sample, to the FP32 output's resolution;
    the parameters the CLI runs with are the measured ones of the fixture"""
    p = GOLD["audioresample_prototype"]
    assert abs(p["44100"]["cutoff_of_lower_nyquist"] - 0.94) < 1e-3 and abs(p["44100"]["half_width_input_samples"] - 32.15) < 0.1
    assert abs(p["96000"]["cutoff_of_lower_nyquist"] - 0.921) < 1e-3 and abs(p["96000"]["half_width_input_samples"] - 64.) < 0.1
    for rate in (44100, 32000, 96000, 22050):
        t = np.arange(2 * rate) / rate
        x = (0.3 * np.sin(2 * np.pi * 997 * t) + 0.2 * np.sin(2 * np.pi * 7919 * t))[:, None].astype(np.float32)
        r48, _ = cli_dump(tmp_path, dict(kind="raw", rate=rate, channels=1, _x=x))
        assert len(r48) == int(np.floor((len(x) - 1) * 48000 / rate)) + 1
        t2 = np.arange(len(r48)) / 48000. - 0.125 / rate
        ideal = 0.3 * np.sin(2 * np.pi * 997 * t2) + 0.2 * np.sin(2 * np.pi * 7919 * t2)
        # pass-band ripple of an 85 dB Kaiser design: 6e-5 of the amplitude
        assert np.abs(r48[4000:-4000, 0] - ideal[4000:-4000]).max() < 1e-4, rate
