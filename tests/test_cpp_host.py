"""The C++ host mirror (include/fx_realtime.hpp) compiled with g++ against libfx_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(fx, tmp_path):
    fx.load_library()
    exe = str(tmp_path / "host_mirror")
    lib_dir = os.path.dirname(fx.library_path())
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp"), "-o", exe,
                           "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir, "-pthread"])
    return exe


def test_cpp_mirror_cpu(fx, tmp_path):
    exe = _build(fx, tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_mirror_gpu(gpu_fx, tmp_path):
    exe = _build(gpu_fx, tmp_path)
    out = subprocess.run([exe, "--gpu"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def _build_example(fx, tmp_path):
    fx.load_library()
    exe = str(tmp_path / "wav_to_osc")
    lib_dir = os.path.dirname(fx.library_path())
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "wav_to_osc.cpp"), "-o", exe,
                           "-L", lib_dir, "-lfx_hip", "-Wl,-rpath," + lib_dir])
    return exe


def _records(path):
    b = open(path, "rb").read()
    out, pos = [], 0
    while pos < len(b):
        n = int.from_bytes(b[pos:pos + 4], "little")
        out.append(b[pos + 4:pos + 4 + n])
        pos += 4 + n
    return out


def test_wav_to_osc_without_gpu_fails_loudly(fx, tmp_path):
    """BASELINE configs[0] plumbing: the example builds, decodes, and -- with no device -- reports the library's error."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = _build_example(fx, tmp_path)
    wav = str(tmp_path / "in.wav")
    fx.wav.write_wav(wav, 48000, fx.synth.samples(1, 4096)[0], "pcm16")
    out = subprocess.run([exe, wav, "--dump", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert out.returncode == 1 and "analysis failed" in out.stderr
    assert subprocess.run([exe, str(tmp_path / "missing.wav")], capture_output=True).returncode == 1
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,window,gain", [("pcm16", 1024, 1.0), ("float32", 2048, 0.5), ("pcm24", 1024, 2.0)])
def test_wav_to_osc_matches_oracle(gpu_fx, oracle, tmp_path, fmt, window, gain):
    """configs[0]: one channel of a 48 kHz WAV -> hops -> analysers -> one OSC feature message per hop; every
    datagram must be byte-identical to the CPU oracle's for the same decoded samples."""
    import numpy as np
    exe = _build_example(gpu_fx, tmp_path)
    wav = str(tmp_path / "in.wav")
    x = gpu_fx.synth.samples(2, 48000 + 333, first_channel=5).T          # stereo, 1 s and a partial hop
    gpu_fx.wav.write_wav(wav, 48000, x, fmt)
    dump = str(tmp_path / "o.bin")
    out = subprocess.run([exe, wav, "--window", str(window), "--channel", "1", "--gain", str(gain), "--address", "/Audio/A1",
                          "--dump", dump, "--batch", "7"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    sr, y, _ = gpu_fx.wav.read_wav(wav)
    hops = gpu_fx.wav.hops_of_channel(y, 1, window)
    _, sm = oracle.push_hops(hops[None], window, float(sr), gain=gain)
    want = [oracle.osc_message("/Audio/A1", sm[0, t]) for t in range(hops.shape[0])]
    got = _records(dump)
    assert len(got) == len(want) == (48000 + 333) // (window // 2)
    assert got == want
    if fmt == "pcm16":
        # the 16-bit samples as the file holds them, widened in the kernels' load stage (FX_SAMPLE_S16): the same datagrams
        dump16 = str(tmp_path / "o16.bin")
        out = subprocess.run([exe, wav, "--window", str(window), "--channel", "1", "--gain", str(gain), "--address", "/Audio/A1",
                              "--dump", dump16, "--batch", "7", "--pcm16-direct"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert _records(dump16) == want
    else:
        out = subprocess.run([exe, wav, "--window", str(window), "--pcm16-direct"], capture_output=True, text=True)
        assert out.returncode == 1 and "16-bit PCM" in out.stderr
    if fmt == "pcm24":
        dump24 = str(tmp_path / "o24.bin")
        out = subprocess.run([exe, wav, "--window", str(window), "--channel", "1", "--gain", str(gain), "--address", "/Audio/A1",
                              "--dump", dump24, "--batch", "7", "--pcm24-direct"], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert _records(dump24) == want

    # the 60 Hz timer view: tick k reads the values after the last hop completed by k/60 s
    dump60 = str(tmp_path / "o60.bin")
    out = subprocess.run([exe, wav, "--window", str(window), "--channel", "1", "--gain", str(gain), "--address", "/Audio/A1",
                          "--dump", dump60, "--rate", "60"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got60 = _records(dump60)
    hop = window // 2
    expect60 = []
    for k in range(1, 10 ** 6):
        done = int(np.floor(k / 60.0 * sr / hop + 1e-9))     # hops complete at time k/60
        if done > hops.shape[0]:
            break
        if done >= 1:
            expect60.append(want[done - 1])
    # ticks are attributed to a hop over [completion, next completion): the last hop serves one more hop period
    assert got60[:len(expect60) - 1] == expect60[:len(expect60) - 1]
    assert abs(len(got60) - len(expect60)) <= 2
