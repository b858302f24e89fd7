"""ctypes wrapper around oracle/libfx_oracle.so (the CPU oracle).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never by the product package.
PARITY UNPINNED: see oracle/fx_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfx_oracle.so")

NUM_FEATURES = 12
(ONSET, RMS, F0, CENTROID, SPREAD, FLATNESS, LER, FLUX, SLOPE, HER, OER, INHARM) = range(12)
FEATURE_NAMES = ["onset", "rms", "f0", "centroid", "spread", "flatness", "ler", "flux",
                 "slope", "her", "oer", "inharm"]
ONSET_SPECTRAL, ONSET_AMPLITUDE, ONSET_COMBINATION = 0, 1, 2
ORDER_SPECTRAL_THEN_HARMONIC, ORDER_HARMONIC_THEN_SPECTRAL, ORDER_ISOLATED = 0, 1, 2


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "fx_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "fx_offline.c")),
                                                 os.path.getmtime(os.path.join(_HERE, "fx_oracle.h")))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libfx_oracle.so"])
    return _LIB_PATH


_lib = None


class Settings(ctypes.Structure):
    _fields_ = [("gain", ctypes.c_float), ("onset_type", ctypes.c_int), ("onset_sensitivity", ctypes.c_float),
                ("onset_window", ctypes.c_int), ("analysers", ctypes.c_int)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        vp = ctypes.c_void_p
        L.fxo_create.restype = vp
        L.fxo_create.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int]
        L.fxo_destroy.argtypes = [vp]
        L.fxo_reset.argtypes = [vp]
        L.fxo_set_sample_rate.argtypes = [vp, ctypes.c_double]
        L.fxo_set_onset_sensitivity.argtypes = [vp, ctypes.c_float]
        L.fxo_set_onset_window.argtypes = [vp, ctypes.c_int]
        L.fxo_set_onset_type.argtypes = [vp, ctypes.c_int]
        L.fxo_set_gain.argtypes = [vp, ctypes.c_float]
        L.fxo_set_analysers.argtypes = [vp, ctypes.c_int]
        L.fxo_process_frames.argtypes = [vp, fp, ctypes.c_int, fp, fp]
        L.fxo_push_hops.argtypes = [vp, fp, ctypes.c_int, fp, fp]
        L.fxo_batch_frames.restype = ctypes.c_int
        L.fxo_batch_frames.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int]
        L.fxo_batch_hops.restype = ctypes.c_int
        L.fxo_batch_hops.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(Settings), fp, ctypes.c_int, ctypes.c_int, fp, fp, ctypes.c_int]
        L.fxo_fft_complex.argtypes = [ctypes.c_int, ctypes.c_int, fp, fp]
        L.fxo_forward_real.argtypes = [ctypes.c_int, fp, fp]
        L.fxo_bartlett.argtypes = [ctypes.c_int, fp]
        L.fxo_lowpass.argtypes = [ctypes.c_int, fp, fp]
        L.fxo_estimate_pitch.restype = ctypes.c_double
        L.fxo_estimate_pitch.argtypes = [ctypes.c_int, ctypes.c_double, fp, fp, fp]
        L.fxo_lpf_a.restype = ctypes.c_float
        L.fxo_lpf_b.restype = ctypes.c_float
        L.fxo_offline_zero_crosses.argtypes = [fp, ctypes.c_int, ctypes.c_int, fp]
        L.fxo_offline_log_attack_time.restype = ctypes.c_float
        L.fxo_offline_log_attack_time.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.fxo_offline_fft_lbp.argtypes = [fp, fp, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte), fp, fp]
        L.fxo_offline_harmonic_characteristics.argtypes = [fp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double), fp]
        L.fxo_offline_spectral_characteristics.argtypes = [fp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double), fp]
        L.fxo_offline_spectral_slope.restype = ctypes.c_float
        L.fxo_offline_spectral_slope.argtypes = [fp, ctypes.c_int]
        L.fxo_offline_conjugate_multiplication.argtypes = [fp, ctypes.c_int]
        L.fxo_offline_auto_correlation.restype = ctypes.c_int
        L.fxo_offline_auto_correlation.argtypes = [fp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        L.fxo_osc_message.restype = ctypes.c_int
        L.fxo_osc_message.argtypes = [ctypes.c_char_p, fp, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Channel:
    """One analysed input channel (one AnalyserTrackController's analysis state)."""

    def __init__(self, window_size=2048, sample_rate=48000.0, order=ORDER_SPECTRAL_THEN_HARMONIC):
        self.n = int(window_size)
        self._h = lib().fxo_create(self.n, float(sample_rate), int(order))
        if not self._h:
            raise ValueError("window_size must be a power of two >= 4")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fxo_destroy(self._h)
            self._h = None

    def reset(self):
        lib().fxo_reset(self._h)

    def set_sample_rate(self, sr):
        lib().fxo_set_sample_rate(self._h, float(sr))

    def set_onset_sensitivity(self, s):
        lib().fxo_set_onset_sensitivity(self._h, float(s))

    def set_onset_window(self, n):
        lib().fxo_set_onset_window(self._h, int(n))

    def set_onset_type(self, t):
        lib().fxo_set_onset_type(self._h, int(t))

    def set_gain(self, g):
        lib().fxo_set_gain(self._h, float(g))

    def set_analysers(self, mask):
        lib().fxo_set_analysers(self._h, int(mask))

    def process_frames(self, frames):
        """frames [T][N] pre-assembled windows -> (raw [T][12], smoothed [T][12])."""
        frames = _f32(frames).reshape(-1, self.n)
        T = frames.shape[0]
        raw = np.empty((T, NUM_FEATURES), np.float32)
        sm = np.empty((T, NUM_FEATURES), np.float32)
        lib().fxo_process_frames(self._h, _fp(frames), T, _fp(raw), _fp(sm))
        return raw, sm

    def push_hops(self, hops):
        """hops [T][N/2] -> (raw [T][12], smoothed [T][12])."""
        hops = _f32(hops).reshape(-1, self.n // 2)
        T = hops.shape[0]
        raw = np.empty((T, NUM_FEATURES), np.float32)
        sm = np.empty((T, NUM_FEATURES), np.float32)
        lib().fxo_push_hops(self._h, _fp(hops), T, _fp(raw), _fp(sm))
        return raw, sm


def process_frames(frames, window_size, sample_rate=48000.0, order=ORDER_SPECTRAL_THEN_HARMONIC, **settings):
    """frames [C][T][N] -> raw [C][T][12], smoothed [C][T][12] with fresh state per channel."""
    frames = _f32(frames)
    C, T = frames.shape[0], frames.shape[1]
    raw = np.empty((C, T, NUM_FEATURES), np.float32)
    sm = np.empty((C, T, NUM_FEATURES), np.float32)
    for c in range(C):
        ch = Channel(window_size, sample_rate, order)
        _apply(ch, settings)
        raw[c], sm[c] = ch.process_frames(frames[c])
    return raw, sm


def push_hops(hops, window_size, sample_rate=48000.0, order=ORDER_SPECTRAL_THEN_HARMONIC, **settings):
    """hops [C][T][N/2] -> raw [C][T][12], smoothed [C][T][12] with fresh state per channel."""
    hops = _f32(hops)
    C, T = hops.shape[0], hops.shape[1]
    raw = np.empty((C, T, NUM_FEATURES), np.float32)
    sm = np.empty((C, T, NUM_FEATURES), np.float32)
    for c in range(C):
        ch = Channel(window_size, sample_rate, order)
        _apply(ch, settings)
        raw[c], sm[c] = ch.push_hops(hops[c])
    return raw, sm


def batch_frames(frames, window_size, sample_rate=48000.0, order=ORDER_SPECTRAL_THEN_HARMONIC, threads=1):
    """frames [C][T][N] -> (raw, smoothed), analysed by `threads` pthreads inside the C library."""
    frames = _f32(frames)
    C, T = frames.shape[0], frames.shape[1]
    raw = np.empty((C, T, NUM_FEATURES), np.float32)
    sm = np.empty((C, T, NUM_FEATURES), np.float32)
    rc = lib().fxo_batch_frames(int(window_size), float(sample_rate), int(order), _fp(frames), C, T, _fp(raw), _fp(sm), int(threads))
    if rc != 0:
        raise RuntimeError("fxo_batch_frames failed")
    return raw, sm


def batch_hops(hops, window_size, sample_rate=48000.0, order=ORDER_SPECTRAL_THEN_HARMONIC, threads=1, gain=1.0,
               onset_type=ONSET_COMBINATION, onset_sensitivity=None, onset_window=5, analysers=3):
    """hops [C][T][N/2] -> (raw, smoothed) like push_hops(), analysed by `threads` pthreads inside the C library."""
    hops = _f32(hops)
    C, T = hops.shape[0], hops.shape[1]
    raw = np.empty((C, T, NUM_FEATURES), np.float32)
    sm = np.empty((C, T, NUM_FEATURES), np.float32)
    if onset_sensitivity is None:
        raise ValueError("onset_sensitivity must be given (the channel default is set through the same setter)")
    st = Settings(float(gain), int(onset_type), float(onset_sensitivity), int(onset_window), int(analysers))
    rc = lib().fxo_batch_hops(int(window_size), float(sample_rate), int(order), ctypes.byref(st), _fp(hops), C, T, _fp(raw), _fp(sm), int(threads))
    if rc != 0:
        raise RuntimeError("fxo_batch_hops failed")
    return raw, sm


def _apply(ch, settings):
    if "gain" in settings:
        ch.set_gain(settings["gain"])
    if "onset_type" in settings:
        ch.set_onset_type(settings["onset_type"])
    if "onset_sensitivity" in settings:
        ch.set_onset_sensitivity(settings["onset_sensitivity"])
    if "onset_window" in settings:
        ch.set_onset_window(settings["onset_window"])
    if "analysers" in settings:
        ch.set_analysers(settings["analysers"])


# ---- taps ----
def fft_complex(x, inverse=False):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n = x.shape[0]
    out = np.empty(n, np.complex64)
    lib().fxo_fft_complex(n, int(inverse), _fp(x.view(np.float32)), _fp(out.view(np.float32)))
    return out


def forward_real(x):
    x = _f32(x)
    out = np.empty(2 * x.shape[0], np.float32)
    lib().fxo_forward_real(x.shape[0], _fp(x), _fp(out))
    return out


def bartlett(x):
    x = _f32(x).copy()
    lib().fxo_bartlett(x.shape[0], _fp(x))
    return x


def lowpass(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().fxo_lowpass(x.shape[0], _fp(x), _fp(out))
    return out


def estimate_pitch(spec2n, nyquist=24000.0):
    spec2n = _f32(spec2n)
    n = spec2n.shape[0] // 2
    cnd = np.empty(2 * n, np.float32)
    lag = ctypes.c_float(0)
    f0 = lib().fxo_estimate_pitch(n, float(nyquist), _fp(spec2n), _fp(cnd), ctypes.byref(lag))
    return f0, lag.value, cnd


def lpf_constants():
    return float(lib().fxo_lpf_a()), float(lib().fxo_lpf_b())


def osc_message(address, smoothed12):
    v = _f32(smoothed12)
    buf = (ctypes.c_ubyte * 256)()
    n = lib().fxo_osc_message(address.encode(), _fp(v), buf, 256)
    if n < 0:
        raise ValueError("address too long")
    return bytes(buf[:n])


# ---- the legacy offline analyser (ref AudioAnalysis.h; oracle/fx_offline.c) ----
def offline_zero_crosses(audio, num_downsamples):
    """audio [C][num_samples] -> [C][num_downsamples] (ref AudioAnalysis.h:517-541)."""
    audio = _f32(audio)
    out = np.empty((audio.shape[0], num_downsamples), np.float32)
    for c in range(audio.shape[0]):
        lib().fxo_offline_zero_crosses(_fp(audio[c]), audio.shape[1], int(num_downsamples), _fp(out[c]))
    return out


def offline_log_attack_time(envelope, num_input_samples, num_downsamples, sample_rate):
    """ref AudioAnalysis.h:611-622; envelope = channel 0 of the energy envelope."""
    envelope = _f32(envelope)
    return np.float32(lib().fxo_offline_log_attack_time(_fp(envelope), envelope.shape[0], int(num_input_samples), int(num_downsamples), int(sample_rate)))


def offline_fft_lbp(cur, prev):
    """cur, prev [C][num_bins] -> (bits [C][num_bins] uint8, highest_ratio [C], activity_ratio [C]) (ref AudioAnalysis.h:543-564)."""
    cur, prev = _f32(cur), _f32(prev)
    C, B = cur.shape
    bits = np.empty((C, B), np.uint8)
    hi, act = np.empty(C, np.float32), np.empty(C, np.float32)
    for c in range(C):
        lib().fxo_offline_fft_lbp(_fp(cur[c]), _fp(prev[c]), B, bits[c].ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), _fp(hi[c:c + 1]), _fp(act[c:c + 1]))
    return bits, hi, act


def offline_harmonic_characteristics(mags, nyquist, previous_f0):
    """mags [C][num_bins], previous_f0 [C] float64 (updated in place) -> [C][3] = f0, harmonic energy ratio, inharmonicity
    (ref AudioAnalysis.h:253-303)."""
    mags = _f32(mags)
    C, B = mags.shape
    out = np.empty((C, 3), np.float32)
    for c in range(C):
        pf = ctypes.c_double(float(previous_f0[c]))
        lib().fxo_offline_harmonic_characteristics(_fp(mags[c]), B, float(nyquist), ctypes.byref(pf), _fp(out[c]))
        previous_f0[c] = pf.value
    return out


def offline_spectral_characteristics(mags, nyquist, previous_bins):
    """mags [C][num_bins], previous_bins [C][num_bins] float64 (previousBinMagnitudes, updated in place) -> [C][4] = centroid / nyquist,
    spread, flatness, flux (ref AudioAnalysis.h:463-515)."""
    mags = _f32(mags)
    C, B = mags.shape
    out = np.empty((C, 4), np.float32)
    for c in range(C):
        lib().fxo_offline_spectral_characteristics(_fp(mags[c]), B, float(nyquist), previous_bins[c].ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _fp(out[c]))
    return out


def offline_spectral_slope(mags):
    """mags [C][num_bins] -> [C] (ref AudioAnalysis.h:566-609)."""
    mags = _f32(mags)
    return np.array([lib().fxo_offline_spectral_slope(_fp(mags[c]), mags.shape[1]) for c in range(mags.shape[0])], np.float32)


def offline_conjugate_multiplication(data):
    """data [C][num_items][2] (r, i) -> each item times its conjugate, as the reference's float arithmetic writes it (ref AudioAnalysis.h:623-633)."""
    out = _f32(data).copy()
    for c in range(out.shape[0]):
        lib().fxo_offline_conjugate_multiplication(_fp(out[c]), out.shape[1])
    return out


def offline_auto_correlation(data, nyquist):
    """data [C][num_items][2] -> (peak bin [C] int32, frequency [C] float64) (ref AudioAnalysis.h:636-665)."""
    data = _f32(data)
    peaks, freqs = np.empty(data.shape[0], np.int32), np.empty(data.shape[0], np.float64)
    for c in range(data.shape[0]):
        f = ctypes.c_double()
        peaks[c] = lib().fxo_offline_auto_correlation(_fp(data[c]), data.shape[1], float(nyquist), ctypes.byref(f))
        freqs[c] = f.value
    return peaks, freqs
