"""RIFF/WAVE reader + writer for the file-input case (BASELINE configs[0]); numpy twin of include/fx_wav.hpp.

In the reference a file reaches the analysers through JUCE's AudioFormatReader / AudioTransportSource
(ref Source/AudioFilePlayer.h:41-61) and AudioDataCollector keeps one channel of it
(ref Source/AudioDataCollector.h:42-64).  Integer PCM of n bits becomes v / 2^(n-1) (8-bit is unsigned,
offset 128) -- JUCE's left-justified int32 x (1.0f / 0x7fffffff) evaluates to exactly that; float files pass
through.  No resampling (the analyser runs at the file's own rate).
"""
import struct

import numpy as np


class WavError(ValueError):
    pass


def read_wav(path):
    """-> (sample_rate, float32 [frames][channels], info dict)."""
    with open(path, "rb") as f:
        b = f.read()
    if len(b) < 12 or b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise WavError("not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(b):
        tag, ln = b[pos:pos + 4], struct.unpack_from("<I", b, pos + 4)[0]
        body = pos + 8
        if tag == b"fmt ":
            if ln < 16 or body + 16 > len(b):
                raise WavError("truncated fmt chunk")
            code, ch, sr, _, align, bits = struct.unpack_from("<HHIIHH", b, body)
            if code == 0xFFFE:
                if ln < 40 or body + 40 > len(b):
                    raise WavError("truncated extensible fmt chunk")
                code = struct.unpack_from("<H", b, body + 24)[0]
            fmt = (code, ch, sr, align, bits)
        elif tag == b"data":
            data = b[body:body + ln]
            break
        pos = body + ln + (ln & 1)
    if fmt is None:
        raise WavError("no fmt chunk")
    if data is None:
        raise WavError("no data chunk")
    code, ch, sr, align, bits = fmt
    if ch < 1:
        raise WavError("no channels")
    if not ((code == 1 and bits in (8, 16, 24, 32)) or (code == 3 and bits in (32, 64))):
        raise WavError("unsupported sample format (format tag %d, %d bits)" % (code, bits))
    if align != bits // 8 * ch:
        raise WavError("inconsistent block alignment")
    n = len(data) // align * ch
    raw = np.frombuffer(data, np.uint8, n * (bits // 8))
    if code == 3:
        x = raw.view("<f4" if bits == 32 else "<f8").astype(np.float32)
    elif bits == 8:
        x = (raw.astype(np.int32) - 128).astype(np.float32) / np.float32(128.0)
    elif bits == 16:
        x = raw.view("<i2").astype(np.float32) / np.float32(32768.0)
    elif bits == 24:
        t = raw.reshape(-1, 3).astype(np.int32)
        v = (t[:, 0] | (t[:, 1] << 8) | (t[:, 2] << 16))
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    else:
        x = raw.view("<i4").astype(np.float32) * np.float32(2.0 ** -31)
    return sr, x.reshape(-1, ch), {"bits": bits, "float": code == 3, "channels": ch}


def write_wav(path, sample_rate, x, fmt="pcm16"):
    """x float [frames] or [frames][channels]; fmt in pcm8 / pcm16 / pcm24 / pcm32 / float32 / float64.
    Integer formats quantise with round-to-nearest and saturate (test and example input only)."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    ch = x.shape[1]
    if fmt in ("float32", "float64"):
        code, bits = 3, int(fmt[5:])
        payload = x.astype("<f4" if bits == 32 else "<f8").tobytes()
    else:
        code, bits = 1, int(fmt[3:])
        full = float(1 << (bits - 1))
        q = np.clip(np.rint(x.astype(np.float64) * full), -full, full - 1).astype(np.int64)
        if bits == 8:
            payload = (q + 128).astype(np.uint8).tobytes()
        elif bits == 16:
            payload = q.astype("<i2").tobytes()
        elif bits == 32:
            payload = q.astype("<i4").tobytes()
        else:
            u = (q & 0xFFFFFF).astype(np.uint32)
            payload = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=-1).astype(np.uint8).tobytes()
    align = bits // 8 * ch
    head = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(payload), b"WAVE", b"fmt ", 16, code, ch, int(sample_rate),
                       int(sample_rate) * align, align, bits, b"data", len(payload))
    with open(path, "wb") as f:
        f.write(head + payload)
        if len(payload) & 1:
            f.write(b"\0")


def hops_of_channel(x, channel, window_size):
    """One channel of a decoded file cut into whole hops [T][window_size/2] (a trailing partial hop is dropped:
    RealTimeAudioDataOverlapper only ever reads whole hops, ref RealTimeAudioAnalysis.h:205-219)."""
    h = window_size // 2
    mono = np.ascontiguousarray(x[:, channel], np.float32)
    t = mono.shape[0] // h
    return mono[:t * h].reshape(t, h)
