"""The `peaq` CLI's own RIFF/WAVE reader (gstpeaq_amd/cli/peaq.c; stands in for the reference's
filesrc ! wavparse ! audioconvert ! audioresample, peaq.c:154-209) on every sample format it accepts:
the printed ODG/DI must be those of the oracle on the samples as audioconvert would deliver them
(integer PCM scaled by 1/2^(bits-1)).  Plus files at other sampling rates through the built-in converter.
Needs an MI355X (`-m gpu`)."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases as case_defs
import gst_env
import oracle_lib as orc
from test_conformance_runner import write_wav

pytestmark = pytest.mark.gpu


def run_cli(*args):
    out = subprocess.run([str(gst_env.CLI), *map(str, args)], capture_output=True, text=True, timeout=300)
    return out


def printed(out):
    lines = out.stdout.strip().splitlines()
    assert lines[-2].startswith("Objective Difference Grade: ") and lines[-1].startswith("Distortion Index: "), out.stdout
    return lines[-2].split()[-1], lines[-1].split()[-1]


def quantised(x, bits, flt):
    """what the reader must hand to the engine for a file written by write_wav"""
    if flt:
        return x.astype(np.float32) if bits == 32 else x.astype(np.float64).astype(np.float32)
    scale = float(2 ** (bits - 1))
    q = np.clip(np.round(x.astype(np.float64) * scale), -scale, scale - 1)
    return (q / scale).astype(np.float32)


def write_wav8(path, x):
    body = np.clip(np.round(x * 128.0) + 128, 0, 255).astype(np.uint8).tobytes()
    ch = x.shape[1]
    fmt = struct.pack("<HHIIHH", 1, ch, 48000, 48000 * ch, ch, 8)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    if len(body) & 1:
        chunks += b"\0"
    Path(path).write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
    return ((np.frombuffer(body, dtype=np.uint8).astype(np.float64) - 128) / 128.).astype(np.float32).reshape(x.shape)


@pytest.mark.parametrize("bits,flt,ext", [(8, False, False), (16, False, False), (24, False, False), (24, False, True),
                                          (32, False, False), (32, False, True), (32, True, False), (32, True, True),
                                          (64, True, False)],
                         ids=["pcm8", "pcm16", "pcm24", "pcm24-ext", "pcm32", "pcm32-ext", "f32", "f32-ext", "f64"])
def test_cli_reads_every_sample_format(tmp_path, bits, flt, ext):
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=33, channels=2, n=60000))
    if bits == 8:
        rq, tq = write_wav8(tmp_path / "ref.wav", ref), write_wav8(tmp_path / "test.wav", test)
    else:
        write_wav(tmp_path / "ref.wav", ref, bits=bits, fmt_float=flt, extensible=ext)
        write_wav(tmp_path / "test.wav", test, bits=bits, fmt_float=flt, extensible=ext)
        rq, tq = quantised(ref, bits, flt), quantised(test, bits, flt)
    exp = orc.run_pair(0, rq, tq)
    out = run_cli(tmp_path / "ref.wav", tmp_path / "test.wav")
    assert out.returncode == 0, out.stdout + out.stderr
    assert printed(out) == ("%.3f" % exp["odg"], "%.3f" % exp["di"])


def test_cli_mono_against_stereo_upmixes_the_mono_side(tmp_path):
    ref, test = case_defs.make_inputs(dict(kind="synth", seed=34, channels=2, n=60000))
    mono = ref[:, :1].copy()
    write_wav(tmp_path / "ref.wav", mono, bits=16)
    write_wav(tmp_path / "test.wav", test, bits=16)
    rq = np.repeat(quantised(mono, 16, False), 2, axis=1)
    exp = orc.run_pair(0, rq, quantised(test, 16, False))
    out = run_cli(tmp_path / "ref.wav", tmp_path / "test.wav")
    assert out.returncode == 0, out.stdout + out.stderr
    assert printed(out) == ("%.3f" % exp["odg"], "%.3f" % exp["di"])


def test_cli_converts_other_sampling_rates_like_the_reference_chain(tmp_path):
    """44.1, 32 and 96 kHz pairs through the whole CLI on the GPU: the printed ODG/DI are those of the oracle on
    what the CLI's converter produced (engine parity through this path), and within the stated 5e-3 of the REAL
    reference chain rawaudioparse ! audioconvert ! audioresample ! peaq (tests/golden/ref_e2e_resampled.json;
    tests/test_cli_resampler.py holds the converter itself, on the CPU); --no-resample refuses with status 2"""
    import test_cli_resampler as rs
    for rec in rs.GOLD["records"]:
        case = rec["case"]
        if case["seed"] == 32:
            continue                                    # one 44.1 kHz case is enough here
        r48, t48 = rs.cli_dump(tmp_path, case)          # also writes r.wav / t.wav
        exp = orc.run_pair(case["advanced"], r48, t48)
        out = run_cli("--advanced" if case["advanced"] else "--basic", tmp_path / "r.wav", tmp_path / "t.wav")
        assert out.returncode == 0, out.stdout + out.stderr
        assert printed(out) == ("%.3f" % exp["odg"], "%.3f" % exp["di"]), case["name"]
        assert abs(float(printed(out)[0]) - float(rec["odg"])) <= rs.TOL + 5e-4, (case["name"], printed(out), rec["odg"])
        assert abs(float(printed(out)[1]) - float(rec["di"])) <= rs.TOL + 5e-4, (case["name"], printed(out), rec["di"])
    refuse = run_cli("--no-resample", tmp_path / "r.wav", tmp_path / "t.wav")
    assert refuse.returncode == 2 and "48 kHz" in refuse.stderr
