"""The `peaq` CLI's own RIFF/WAVE reader (gstpeaq_amd/cli/peaq.c; stands in for the reference's
filesrc ! wavparse ! audioconvert ! audioresample, peaq.c:154-209) on every sample format it accepts:
the printed ODG/DI must be those of the oracle on the samples as audioconvert would deliver them
(integer PCM scaled by 1/2^(bits-1)).  Plus the 44.1 kHz path through the built-in resampler.
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


def write_wav_rate(path, x, rate):
    body = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2").tobytes()
    ch = x.shape[1]
    fmt = struct.pack("<HHIIHH", 1, ch, rate, rate * ch * 2, ch * 2, 16)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(body)) + body
    Path(path).write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def resample_np(x, rate):
    """numpy transcription of resample_to_48k() in gstpeaq_amd/cli/peaq.c (Kaiser-windowed sinc, 64 zero crossings)"""
    from scipy.special import i0
    ratio = 48000. / rate
    fc, zc, beta = 0.96 * 0.5 * min(ratio, 1.0), 64, 12.9846
    half = zc / (2 * fc)
    n_out = int(np.floor(len(x) * ratio))
    out = np.zeros((n_out, x.shape[1]))
    for m in range(n_out):
        t = m / ratio
        n = np.arange(max(int(np.ceil(t - half)), 0), min(int(np.floor(t + half)), len(x) - 1) + 1)
        d = t - n
        arg = 2 * np.pi * fc * d
        snc = np.where(np.abs(arg) < 1e-12, 1.0, np.sin(arg) / np.where(arg == 0, 1, arg))
        win = i0(beta * np.sqrt(np.maximum(1 - (d / half) ** 2, 0))) / i0(beta)
        out[m] = (2 * fc * snc * win) @ x[n]
    return out.astype(np.float32)


def test_cli_converts_44100_hz_files(tmp_path):
    """a 44.1 kHz pair (which the reference accepts through audioresample): the CLI converts it to 48 kHz and
    prints what the oracle gives for the same conversion done in numpy; the interpolator itself reproduces
    band-limited tones to the 16-bit quantisation step; --no-resample refuses the files with exit status 2"""
    rate = 44100
    t = np.arange(rate) / rate
    ref = 0.25 * np.sin(2 * np.pi * 997 * t) + 0.125 * np.sin(2 * np.pi * 3301 * t) + 0.06 * np.sin(2 * np.pi * 7919 * t)
    test = ref + 0.004 * np.sin(2 * np.pi * 5003 * t) + 0.002 * np.sin(2 * np.pi * 211 * t)
    ref, test = ref[:, None].astype(np.float32), test[:, None].astype(np.float32)
    write_wav_rate(tmp_path / "r44.wav", ref, rate)
    write_wav_rate(tmp_path / "t44.wav", test, rate)
    r48, t48 = resample_np(quantised(ref, 16, False).astype(np.float64), rate), resample_np(quantised(test, 16, False).astype(np.float64), rate)
    t2 = np.arange(len(r48)) / 48000.
    ideal = 0.25 * np.sin(2 * np.pi * 997 * t2) + 0.125 * np.sin(2 * np.pi * 3301 * t2) + 0.06 * np.sin(2 * np.pi * 7919 * t2)
    assert np.abs(r48[2000:-2000, 0] - ideal[2000:-2000]).max() < 4e-5          # 16-bit step: 3e-5
    exp = orc.run_pair(0, r48, t48)
    out = run_cli(tmp_path / "r44.wav", tmp_path / "t44.wav")
    assert out.returncode == 0, out.stdout + out.stderr
    assert abs(float(printed(out)[0]) - exp["odg"]) <= 2e-3 and abs(float(printed(out)[1]) - exp["di"]) <= 2e-3, (printed(out), exp)
    refuse = run_cli("--no-resample", tmp_path / "r44.wav", tmp_path / "t44.wav")
    assert refuse.returncode == 2 and "48 kHz" in refuse.stderr
