"""numpy transcription of include/peaq_synth.h (bit-identical by construction:
32-bit integer arithmetic only).  Test helper -- not part of the product."""
import re
from pathlib import Path

import numpy as np

_HDR = Path(__file__).resolve().parent.parent / "include" / "peaq_synth.h"
NTAPS = 63
U32 = np.uint32


def _load_taps():
    txt = _HDR.read_text()
    body = txt[txt.index("peaq_synth_taps[5]"):]
    rows = re.findall(r"\{(-?\d+(?:,\s*-?\d+){62})\}", body)
    assert len(rows) == 5
    return np.array([[int(v) for v in r.split(",")] for r in rows], dtype=np.int64)


TAPS = _load_taps()


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    x = (x * 0xC2B2AE35) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _u32(x):
    return int(x) & 0xFFFFFFFF


def params(seed):
    seed = _u32(seed)
    h = int(mix32(_u32(seed * 0x9E3779B9 + 0x7F4A7C15)))
    p = {"seed": seed}
    p["chan_key"] = [int(mix32(_u32(h + 0x632BE5AB * (c + 1)))) for c in range(2)]
    p["tone_inc"], p["tone_amp"], p["tone_ph0"] = [], [], [[0] * 3 for _ in range(2)]
    for i in range(3):
        r = int(mix32(h ^ _u32(0x1000193 * (i + 1))))
        f_hz = 150 + (r % 6144)
        p["tone_inc"].append(_u32(f_hz * 89478 + (r >> 20)))
        p["tone_amp"].append(1024 + ((r >> 8) % 3072))
        for c in range(2):
            p["tone_ph0"][c][i] = int(mix32(_u32(r + 77 * (c + 1))))
    r = int(mix32(h ^ 0xDEADBEEF))
    p["noise_gain"] = 4096 + (r % 12288)
    p["lfo_inc"] = _u32((2 + ((r >> 13) % 14)) * 89478)
    p["lfo_depth"] = (r >> 17) % 16384
    p["test_filter"] = 1 + ((r >> 3) % 4)
    r = int(mix32(h ^ 0x0BADF00D))
    p["test_shift"] = 23 - (8 + (r % 8))
    p["test_gain"] = 248 + ((r >> 4) % 17)
    p["lead_silence"] = 12000 if ((r >> 10) % 4 == 0) else 0
    p["tail_silence"] = 12000 if ((r >> 12) % 4 == 0) else 0
    return p


def _psin(phase):
    x = (phase.astype(np.uint32).view(np.int32) >> 16).astype(np.int64)
    return (x * (32768 - np.abs(x))) >> 13


def pair(seed, channels, n_samples):
    """-> (ref, test) float32 arrays of shape [n_samples, channels]"""
    p = params(seed)
    n = np.arange(n_samples, dtype=np.uint64)
    ref = np.zeros((n_samples, channels), dtype=np.float32)
    test = np.zeros((n_samples, channels), dtype=np.float32)
    live = (n >= p["lead_silence"]) & (n + p["tail_silence"] < n_samples)
    lfo = _psin((n * p["lfo_inc"]) & 0xFFFFFFFF)
    env = 32768 - p["lfo_depth"] + ((lfo * p["lfo_depth"]) >> 15)
    for c in range(channels):
        w = (mix32((p["chan_key"][c] + n * 0x9E3779B1) & 0xFFFFFFFF) >> 17).astype(np.int64) - 16384
        acc_r = np.convolve(w, TAPS[0])[:n_samples]
        acc_t = np.convolve(w, TAPS[p["test_filter"]])[:n_samples]
        acc_r = ((acc_r >> 15) * p["noise_gain"]) >> 15
        acc_t = ((acc_t >> 15) * p["noise_gain"]) >> 15
        tone = np.zeros(n_samples, dtype=np.int64)
        for i in range(3):
            ph = (p["tone_ph0"][c][i] + n * p["tone_inc"][i]) & 0xFFFFFFFF
            tone += (_psin(ph) * p["tone_amp"][i]) >> 15
        vr = ((acc_r + tone) * env) >> 15
        vt = ((acc_t + tone) * env) >> 15
        vt = (vt * p["test_gain"]) >> 8
        vr = vr * 256
        vt = vt * 256
        sh = p["test_shift"]
        vt = ((vt + (1 << (sh - 1))) >> sh) * (1 << sh)
        vr = np.where(live, vr, 0)
        vt = np.where(live, vt, 0)
        ref[:, c] = (vr.astype(np.float32) * np.float32(1.0 / 8388608.0))
        test[:, c] = (vt.astype(np.float32) * np.float32(1.0 / 8388608.0))
    return ref, test


def audiotestsrc(wave, n=131072, freq=440.0, vol=0.8):
    """GStreamer audiotestsrc (sine / saw / triangle), mono F32, 48 kHz --
    verified bit-exact against the real element in the build container
    (tools/make_golden.py).  Inputs of the reference's runtest-1.0.sh."""
    import math
    step = 2 * math.pi * freq / 48000
    acc = 0.0
    y = np.empty(n)
    for i in range(n):
        acc += step
        if acc >= 2 * math.pi:
            acc -= 2 * math.pi
        if wave == "sine":
            y[i] = math.sin(acc) * vol
        elif wave == "saw":
            amp = vol / math.pi
            y[i] = acc * amp if acc < math.pi else (2 * math.pi - acc) * -amp
        elif wave == "triangle":
            amp = vol / (math.pi / 2)
            if acc < math.pi / 2:
                y[i] = acc * amp
            elif acc < 1.5 * math.pi:
                y[i] = (acc - math.pi) * -amp
            else:
                y[i] = (2 * math.pi - acc) * -amp
        else:
            raise ValueError(wave)
    return y.astype(np.float32).reshape(n, 1)
