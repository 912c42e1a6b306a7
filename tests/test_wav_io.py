"""WAV reader / writer of the render tool (cpp/tools/wav_io.h) against an independent Python decode.
Formats the reference's render accepts through dsp::wav::Load (tools/render.cpp:129-136): mono PCM 16/24/32,
IEEE float 32, WAVE_FORMAT_EXTENSIBLE, extra chunks before `data`; output = SaveWavFloat32's layout
(tools/render.cpp:20-60)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

WAVTOOL = os.path.join(ROOT, "cpp", "tools", "wavtool")


@pytest.fixture(scope="module")
def wavtool():
    subprocess.run(["make", "-C", os.path.join(ROOT, "cpp"), "tools/wavtool"], check=True, capture_output=True)
    return WAVTOOL


def read_f32_wav(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and struct.unpack("<I", b[4:8])[0] == len(b) - 8
    size, fmt, ch, sr, byte_rate, align, bits = struct.unpack("<IHHIIHH", b[16:36])
    assert (size, fmt, ch, bits, align, byte_rate) == (16, 3, 1, 32, 4, sr * 4) and b[36:40] == b"data"
    n = struct.unpack("<I", b[40:44])[0]
    assert 44 + n == len(b)
    return np.frombuffer(b[44:], dtype="<f4"), sr


def write_wav(path, fmt_tag, bits, payload, sr=48000, channels=1, extensible=False, extra_chunks=b""):
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits, 22, bits, 4,
                          fmt_tag, b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, sr, sr * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra_chunks + b"data" + struct.pack("<I", len(payload)) + payload
    if len(payload) & 1:
        body += b"\x00"
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)


def run_copy(wavtool, src, dst):
    return subprocess.run([wavtool, "copy", src, dst], capture_output=True, text=True)


def test_reads_the_reference_example_audio(wavtool, tmp_path):
    src = os.path.join(ROOT, "tests", "golden", "audio", "input.wav")
    dst = str(tmp_path / "copy.wav")
    r = run_copy(wavtool, src, dst)
    assert r.returncode == 0, r.stderr
    y, sr = read_f32_wav(dst)
    assert sr == 48000 and len(y) == 96000  # 2.0 s mono, 24-bit PCM
    # independent 24-bit decode
    b = open(src, "rb").read()
    i = b.index(b"data")
    n = struct.unpack("<I", b[i + 4:i + 8])[0]
    raw = np.frombuffer(b[i + 8:i + 8 + n], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
    v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
    v = np.where(v >= 1 << 23, v - (1 << 24), v)
    np.testing.assert_array_equal(y, (v / 8388608.0).astype(np.float32))
    assert abs(float(np.abs(y).max()) - 0.5) < 1e-3


@pytest.mark.parametrize("kind", ["pcm16", "pcm32", "float32", "float32_extensible", "pcm16_with_list_chunk", "odd_length_pcm24"])
def test_formats(wavtool, tmp_path, kind):
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.9, 0.9, 777).astype(np.float32)
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    if kind.startswith("pcm16"):
        q = np.round(x * 32767).astype("<i2")
        extra = b"LIST" + struct.pack("<I", 5) + b"abcde\x00" if "list" in kind else b""  # odd-sized chunk: padded
        write_wav(src, 1, 16, q.tobytes(), extra_chunks=extra)
        want = q.astype(np.float32) / np.float32(32768.0)
    elif kind == "pcm32":
        q = np.round(x.astype(np.float64) * 2147483647).astype("<i4")
        write_wav(src, 1, 32, q.tobytes())
        want = (q.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif kind == "odd_length_pcm24":
        q = np.round(x * 8388607).astype(np.int32)
        raw = np.stack([q & 255, (q >> 8) & 255, (q >> 16) & 255], axis=1).astype(np.uint8)
        write_wav(src, 1, 24, raw.tobytes())  # 777 * 3 bytes: odd -> pad byte
        want = (q / 8388608.0).astype(np.float32)
    else:
        write_wav(src, 3, 32, x.astype("<f4").tobytes(), extensible="extensible" in kind)
        want = x
    r = run_copy(wavtool, src, dst)
    assert r.returncode == 0, r.stderr
    y, sr = read_f32_wav(dst)
    assert sr == 48000
    np.testing.assert_array_equal(y, want)


def test_rejects_stereo_and_garbage(wavtool, tmp_path):
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    write_wav(src, 1, 16, np.zeros(64, dtype="<i2").tobytes(), channels=2)
    r = run_copy(wavtool, src, dst)
    assert r.returncode != 0 and "mono" in r.stderr
    open(src, "wb").write(b"not a wav file at all")
    r = run_copy(wavtool, src, dst)
    assert r.returncode != 0 and "RIFF" in r.stderr
    write_wav(src, 1, 8, bytes(64))
    r = run_copy(wavtool, src, dst)
    assert r.returncode != 0 and "unsupported" in r.stderr
