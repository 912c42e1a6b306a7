"""Deterministic test inputs (shared by the parity tests and bench.py)."""
import os
import struct

import numpy as np

SR = 48000.0


def two_tone(n, scale=1.0, f1=220.0, f2=1230.0, a1=0.25, a2=0.10):
    """0.25 sin(2 pi 220 t) + 0.10 sin(2 pi 1230 t) — the input of the reference's
    implementation-vs-implementation parity test (tools/test/test_a2_fast.cpp:118-128)."""
    t = np.arange(n, dtype=np.float64) / SR
    return (a1 * np.sin(2 * np.pi * f1 * scale * t) + a2 * np.sin(2 * np.pi * f2 * scale * t)).astype(np.float32)


def stream_bank(n_streams, n, seed=0):
    """Per-stream two-tone with frequencies scaled by (1 + s/n_streams), plus
    a little seeded noise so no two streams are related by a simple transform."""
    rng = np.random.default_rng(seed)
    x = np.stack([two_tone(n, 1.0 + s / max(n_streams, 1)) for s in range(n_streams)])
    x += rng.uniform(-0.02, 0.02, size=x.shape).astype(np.float32)
    return np.ascontiguousarray(x.astype(np.float32))


def read_wav_mono24(path):
    """Minimal PCM WAV reader (16/24/32-bit int or 32-bit float, mono) -> float32 in [-1, 1)."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    tag, ch, sr, _, _, bits = fmt
    assert ch == 1
    if tag == 3 and bits == 32:
        return np.frombuffer(pcm, dtype="<f4").astype(np.float32), sr
    if bits == 16:
        return (np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0), sr
    if bits == 24:
        b = np.frombuffer(pcm, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        return (v.astype(np.float32) / 8388608.0), sr
    if bits == 32:
        return (np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0), sr
    raise ValueError("unsupported wav")
