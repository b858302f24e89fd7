"""Differential harness: the reference's own hot-path headers, compiled unmodified against
tools/refdiff/juce_standin.h, single-stepped over hop streams -- against oracle/fx_oracle.c.

BUILD CONTAINER ONLY (needs /root/reference and g++).  Not a build of the reference and not a parity pin:
JUCE's own arithmetic (FFT, getRMSLevel, applyGainRamp, getMagnitude) is a restatement in the stand-in as
it is in the oracle.  What it checks is the feature arithmetic that IS in the reference's headers.

Two builds of the driver:
  cr   : log10(float) correctly rounded -- the oracle's documented convention; the oracle must be BIT-IDENTICAL
  libm : log10(float) = this platform's log10f -- values within 1e-5, and no onset / gate decision may flip
"""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference/Source"
BUILD = os.path.join(HERE, "_build")
HEADERS = ["AudioDataCollector.h", "RealTimeAudioAnalysis.h", "PitchAnalyser.h", "SpectralCharacteristics.h",
           "HarmonicCharacteristics.h", "RealTimeAnalyser.h"]


def available():
    return all(os.path.exists(os.path.join(REFERENCE, h)) for h in HEADERS)


def build(mode):
    """g++ on the driver; the reference headers are included from where they lie (-I /root/reference/Source)."""
    assert mode in ("cr", "libm")
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "refdiff_" + mode)
    srcs = [os.path.join(HERE, "refdiff_driver.cpp"), os.path.join(HERE, "juce_standin.h")]
    if os.path.exists(exe) and all(os.path.getmtime(s) <= os.path.getmtime(exe) for s in srcs):
        return exe
    cmd = ["g++", "-std=c++14", "-O2", "-w", "-ffp-contract=off", "-fno-fast-math", "-I", REFERENCE, "-I", HERE,
           srcs[0], "-o", exe]
    if mode == "cr":
        cmd.insert(1, "-DREFDIFF_LOG10_CR")
    subprocess.check_call(cmd)
    return exe


def run(hops, window_size, order=0, onset_type=1, onset_window=5, onset_sensitivity=0.7, gain=1.0,
        sample_rate=48000.0, mode="cr", workdir=None):
    """hops [C][T][N/2] float32 -> (raw [C][T][12], smoothed [C][T][12]) from the reference's headers."""
    import tempfile
    hops = np.ascontiguousarray(hops, np.float32)
    C, T, half = hops.shape
    assert half * 2 == window_size
    exe = build(mode)
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<6i2fd", window_size, C, T, order, onset_type, onset_window, onset_sensitivity, gain, sample_rate))
            f.write(hops.tobytes())
        subprocess.check_call([exe, fin, fout])
        out = np.fromfile(fout, np.float32)
    n = C * T * 12
    return out[:n].reshape(C, T, 12), out[n:2 * n].reshape(C, T, 12)


# ---- the reference's own collector fed with device blocks of any length (refdiff_blocks.cpp) ----
def build_blocks():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "refdiff_blocks")
    srcs = [os.path.join(HERE, "refdiff_blocks.cpp"), os.path.join(HERE, "juce_standin.h")]
    if os.path.exists(exe) and all(os.path.getmtime(s) <= os.path.getmtime(exe) for s in srcs):
        return exe
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-w", "-ffp-contract=off", "-fno-fast-math", "-DREFDIFF_LOG10_CR", "-I", REFERENCE, "-I", HERE, srcs[0], "-o", exe])
    return exe


def run_blocks(stream, window_size, block, order=0, events=(), sample_rate=48000.0, app_stepping=False):
    """stream [C][total] float32 fed to the reference's AudioDataCollector::audioDeviceIOCallback in blocks of `block` samples (the last one
    shorter); events = [(at_sample, "gain" | "sensitivity" | "onset_window" | "onset_type" | "sample_rate", value) | (at_sample, "clear")], each
    taking effect before the block that starts at or after at_sample (the setters of ref AudioDataCollector.h:122-124 and RealTimeAnalyser.h:111-114,244-258).  -> (raw [C][frames][12], smoothed [C][frames][12]) of the total // (window_size / 2) hops the analysers read.
    app_stepping: the APPLICATION's stepping (--notify-per-block: one pass of the threads' loop at start and one per callback, the reader
    running ahead of the writer as indexesOverlap lets it) instead of the canonical one; the number of frames is then whatever the app analyses."""
    import tempfile
    stream = np.ascontiguousarray(stream, np.float32)
    C, total = stream.shape
    exe = build_blocks()
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<6id", window_size, C, total, block, order, len(events), sample_rate))
            kinds = {"gain": 0, "clear": 1, "sensitivity": 2, "onset_window": 3, "onset_type": 4, "sample_rate": 5}
            for e in events:
                f.write(struct.pack("<iifi", int(e[0]), kinds[e[1]], float(e[2]) if len(e) > 2 else 0.0, 0))
            f.write(stream.tobytes())
        subprocess.run([exe, fin, fout] + (["--notify-per-block"] if app_stepping else []), check=True, timeout=120)
        blob = open(fout, "rb").read()
    frames = struct.unpack("<i", blob[:4])[0]
    out = np.frombuffer(blob[4:], np.float32)
    n = C * frames * 12
    return out[:n].reshape(C, frames, 12).copy(), out[n:2 * n].reshape(C, frames, 12).copy()


# ---- the reference's own OSCFeatureAnalysisOutput against a sender that records what it is handed (refdiff_osc.cpp) ----
def osc_probe(address):
    """-> (host, port, timer Hz, OSC address, the 12 arguments of one sendSpectralFeaturesViaOSC(true) when slot k's getValue() is 100 + k)"""
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "refdiff_osc")
    srcs = [os.path.join(HERE, "refdiff_osc.cpp"), os.path.join(HERE, "juce_standin.h")]
    if not (os.path.exists(exe) and all(os.path.getmtime(s) <= os.path.getmtime(exe) for s in srcs)):
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", "-I", REFERENCE, "-I", HERE, srcs[0], "-o", exe])
    out = subprocess.run([exe, address], capture_output=True, text=True, check=True, timeout=60).stdout
    head, addr, args = [part.strip() for part in out.split("|")]
    host, port, hz = head.split()
    return host, int(port), int(hz), addr, [float(v) for v in args.split()]


# ---- the legacy offline analyser (AudioAnalysis.h / AudioFeatures.h) ----
LEGACY_HEADERS = ["AudioAnalysis.h", "AudioFeatures.h"]


def legacy_available():
    return all(os.path.exists(os.path.join(REFERENCE, h)) for h in LEGACY_HEADERS)


def build_legacy():
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "refdiff_legacy")
    srcs = [os.path.join(HERE, "refdiff_legacy.cpp"), os.path.join(HERE, "juce_standin.h")]
    if os.path.exists(exe) and all(os.path.getmtime(s) <= os.path.getmtime(exe) for s in srcs):
        return exe
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-w", "-ffp-contract=off", "-fno-fast-math", "-DREFDIFF_LOG10_CR",
                           "-I", REFERENCE, "-I", HERE, srcs[0], "-o", exe])
    return exe


def _legacy(op, a, b, c, d, x, payload):
    import tempfile
    exe = build_legacy()
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<5i4xd", op, a, b, c, d, x))          # (struct Header: 5 x int32, padding, float64)
            f.write(np.ascontiguousarray(payload, np.float32).tobytes())
        subprocess.check_call([exe, fin, fout])
        return open(fout, "rb").read()


def legacy_zero_crosses(audio, num_downsamples):
    """AudioAnalyser::analyseNormalisedZeroCrosses of the reference on audio [C][S] -> [C][num_downsamples]."""
    audio = np.ascontiguousarray(audio, np.float32)
    out = _legacy(1, audio.shape[0], audio.shape[1], num_downsamples, 0, 0.0, audio)
    return np.frombuffer(out, np.float32).reshape(audio.shape[0], num_downsamples).copy()


def legacy_log_attack_time(envelope, num_input_samples, num_downsamples, sample_rate):
    envelope = np.ascontiguousarray(envelope, np.float32)
    return np.frombuffer(_legacy(2, envelope.shape[0], num_input_samples, num_downsamples, sample_rate, 0.0, envelope), np.float32)[0]


def legacy_fft_lbp(cur, prev):
    cur, prev = np.ascontiguousarray(cur, np.float32), np.ascontiguousarray(prev, np.float32)
    C, B = cur.shape
    out = np.frombuffer(_legacy(3, C, B, 0, 0, 0.0, np.concatenate([cur, prev])), np.float32).reshape(C, B + 2)
    return out[:, :B].astype(np.uint8), out[:, B].copy(), out[:, B + 1].copy()


def legacy_harmonic_characteristics(mags, nyquist):
    """mags [T][C][B]: T successive frames through one AudioAnalyser per channel -> (out [T][C][3], previousF0 [T][C])."""
    mags = np.ascontiguousarray(mags, np.float32)
    T, C, B = mags.shape
    out = _legacy(4, C, B, T, 0, float(nyquist), mags)
    n = T * C * 3 * 4
    return np.frombuffer(out[:n], np.float32).reshape(T, C, 3).copy(), np.frombuffer(out[n:], np.float64).reshape(T, C).copy()


def legacy_spectral_characteristics(mags, nyquist):
    """mags [T][C][B]: T successive frames through one AudioAnalyser per channel -> (out [T][C][4] = centroid / nyquist, spread, flatness,
    flux; previousBinMagnitudes [C][B] after the last frame)."""
    mags = np.ascontiguousarray(mags, np.float32)
    T, C, B = mags.shape
    out = _legacy(5, C, B, T, 0, float(nyquist), mags)
    n = T * C * 4 * 4
    return np.frombuffer(out[:n], np.float32).reshape(T, C, 4).copy(), np.frombuffer(out[n:], np.float64).reshape(C, B).copy()


def legacy_spectral_slope(mags):
    mags = np.ascontiguousarray(mags, np.float32)
    return np.frombuffer(_legacy(6, mags.shape[0], mags.shape[1], 0, 0, 0.0, mags), np.float32).copy()


def legacy_auto_correlation(data, nyquist):
    """data [C][B][2] -> (products [C][B][2] of getConjugateComplexMultiplicationInPlace, frequency [C] analyseAutoCorrelation prints for them)."""
    data = np.ascontiguousarray(data, np.float32)
    C, B = data.shape[0], data.shape[1]
    out = _legacy(7, C, B, 0, 0, float(nyquist), data)
    n = C * B * 2 * 4
    return np.frombuffer(out[:n], np.float32).reshape(C, B, 2).copy(), np.frombuffer(out[n:], np.float64).copy()
