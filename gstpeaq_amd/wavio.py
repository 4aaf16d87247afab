"""Minimal RIFF/WAVE reader for the tools (host side, numpy): PCM 8/16/24/32 and IEEE float
32/64, plain or WAVE_FORMAT_EXTENSIBLE -> float32 [n, channels] scaled by 1/2^(bits-1), the wire
format of the `peaq` element (audio/x-raw F32LE interleaved, gstpeaq.c:146-152).  Same rules as
the reader of the C CLI (gstpeaq_amd/cli/peaq.c)."""
import struct

import numpy as np


def read_wav(path):
    """-> (float32 [n, channels], sample_rate)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        chunk = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, rate, _, _, bits = struct.unpack_from("<HHIIHH", chunk, 0)
            if tag == 0xFFFE and len(chunk) >= 26:
                tag = struct.unpack_from("<H", chunk, 24)[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            body = chunk
            break
        pos += 8 + size + (size & 1)
    if fmt is None or body is None:
        raise ValueError(f"{path}: fmt or data chunk missing")
    tag, ch, rate, bits = fmt
    if ch < 1:
        raise ValueError(f"{path}: no channels")
    nbytes = bits // 8
    n = len(body) // (nbytes * ch) * ch
    raw = body[:n * nbytes]
    if tag == 3 and bits == 32:
        x = np.frombuffer(raw, "<f4").astype(np.float32)
    elif tag == 3 and bits == 64:
        x = np.frombuffer(raw, "<f8").astype(np.float32)
    elif tag == 1 and bits == 8:
        x = ((np.frombuffer(raw, np.uint8).astype(np.float64) - 128.0) / 128.0).astype(np.float32)
    elif tag == 1 and bits == 16:
        x = (np.frombuffer(raw, "<i2").astype(np.float64) / 32768.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format (tag {tag}, {bits} bit)")
    return np.ascontiguousarray(x.reshape(-1, ch)), rate
