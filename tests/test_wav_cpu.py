"""File ingest for BASELINE configs[0]: include/fx_wav.hpp and feature-extractor_amd/wav.py decode the same
files to the same floats (JUCE WavAudioFormat conventions, see the headers)."""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMATS = ["pcm8", "pcm16", "pcm24", "pcm32", "float32", "float64"]


@pytest.fixture(scope="module")
def wav_dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("wavdump") / "wav_dump")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "wav_dump.cpp"), "-o", exe])
    return exe


def _signal(frames, channels, seed=3):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.0, 1.0, (frames, channels))
    x[0, :] = -1.0          # most negative code
    x[1, :] = 1.0           # saturates to the most positive code
    x[2, :] = 0.0
    return x


def _cpp(wav_dump, path, tmp_path):
    out = str(tmp_path / "dump.f32")
    r = subprocess.run([wav_dump, path, out], capture_output=True, text=True)
    if r.returncode != 0:
        return r.returncode, r.stderr.strip(), None
    return 0, [int(v) for v in r.stdout.split()], np.fromfile(out, np.float32)


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("channels", [1, 2])
def test_cpp_and_python_decode_alike(fx, wav_dump, tmp_path, fmt, channels):
    path = str(tmp_path / "t.wav")
    x = _signal(3000 + channels, channels)          # odd payload sizes for pcm8/pcm24 mono
    fx.wav.write_wav(path, 44100, x, fmt)
    sr, y, info = fx.wav.read_wav(path)
    assert sr == 44100 and y.shape == x.shape and y.dtype == np.float32
    assert info["float"] == fmt.startswith("float") and info["bits"] == int(fmt.lstrip("pcmfloat"))
    rc, head, samples = _cpp(wav_dump, path, tmp_path)
    assert rc == 0
    assert head[:5] == [44100, channels, info["bits"], int(info["float"]), x.shape[0]]
    assert head[5] == x.shape[0] // 512 and head[6] == head[5] * 512
    assert np.array_equal(samples.view(np.uint32), y.reshape(-1).view(np.uint32))
    if fmt.startswith("float"):
        assert np.array_equal(y, x.astype(np.float32))
    else:
        bits = info["bits"]
        full = float(1 << (bits - 1))
        q = np.clip(np.rint(x * full), -full, full - 1)
        expect = (q.astype(np.float32) if bits == 32 else q).astype(np.float32) * np.float32(1.0 / full)
        assert np.array_equal(y, expect.astype(np.float32))
        assert y[0, 0] == -1.0 and y[1, 0] == np.float32((full - 1) / full) and y[2, 0] == 0.0


def test_pcm16_matches_stdlib_wave(fx, tmp_path):
    path = str(tmp_path / "std.wav")
    rng = np.random.default_rng(0)
    v = rng.integers(-32768, 32768, (1000, 2), dtype=np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(48000)
        w.writeframes(v.astype("<i2").tobytes())
    sr, y, _ = fx.wav.read_wav(path)
    assert sr == 48000
    assert np.array_equal(y, v.astype(np.float32) / np.float32(32768.0))


def test_extensible_header_and_extra_chunks(fx, wav_dump, tmp_path):
    # WAVE_FORMAT_EXTENSIBLE fmt (40 bytes), an odd-sized LIST chunk before the data, trailing chunk after it
    v = np.arange(-300, 300, dtype=np.int16)
    payload = v.astype("<i2").tobytes()
    guid = struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    fmt = struct.pack("<HHIIHHHHI", 0xFFFE, 1, 48000, 96000, 2, 16, 22, 16, 4) + guid
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 5) + b"abcde\0" \
        + b"data" + struct.pack("<I", len(payload)) + payload + b"cue " + struct.pack("<I", 4) + b"\0\0\0\0"
    path = str(tmp_path / "ext.wav")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    sr, y, info = fx.wav.read_wav(path)
    assert sr == 48000 and info["bits"] == 16 and not info["float"]
    assert np.array_equal(y[:, 0], v.astype(np.float32) / np.float32(32768.0))
    rc, head, samples = _cpp(wav_dump, path, tmp_path)
    assert rc == 0 and head[:5] == [48000, 1, 16, 0, 600]
    assert np.array_equal(samples, y[:, 0])


def test_truncated_data_chunk_reads_what_is_there(fx, wav_dump, tmp_path):
    path = str(tmp_path / "short.wav")
    fx.wav.write_wav(path, 48000, np.linspace(-0.5, 0.5, 1000), "pcm16")
    b = open(path, "rb").read()
    open(path, "wb").write(b[:44 + 1001])           # cut mid-sample
    _, y, _ = fx.wav.read_wav(path)
    assert y.shape == (500, 1)
    rc, head, samples = _cpp(wav_dump, path, tmp_path)
    assert rc == 0 and head[4] == 500 and np.array_equal(samples, y[:, 0])


@pytest.mark.parametrize("blob,reason", [
    (b"RIFX\0\0\0\0WAVE", "not a RIFF/WAVE"),
    (b"RIFF\x04\0\0\0WAVE", "no fmt chunk"),
    (b"RIFF\x1c\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<HHIIHH", 1, 1, 48000, 96000, 2, 16), "no data chunk"),
    (b"RIFF\x24\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<HHIIHH", 2, 1, 48000, 96000, 2, 16) + b"data\0\0\0\0", "unsupported sample format"),
    (b"RIFF\x24\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<HHIIHH", 1, 1, 48000, 96000, 2, 12) + b"data\0\0\0\0", "unsupported sample format"),
    (b"RIFF\x24\0\0\0WAVEfmt \x10\0\0\0" + struct.pack("<HHIIHH", 1, 2, 48000, 96000, 2, 16) + b"data\0\0\0\0", "inconsistent block alignment"),
])
def test_rejected_files(fx, wav_dump, tmp_path, blob, reason):
    path = str(tmp_path / "bad.wav")
    open(path, "wb").write(blob)
    with pytest.raises(fx.wav.WavError, match=reason):
        fx.wav.read_wav(path)
    rc, err, _ = _cpp(wav_dump, path, tmp_path)
    assert rc == 3 and reason in err


def test_hops_of_channel(fx):
    x = np.arange(2 * 1300, dtype=np.float32).reshape(1300, 2)
    h = fx.wav.hops_of_channel(x, 1, 1024)
    assert h.shape == (2, 512) and h[0, 0] == 1.0 and h[1, 511] == 2 * 1023 + 1
