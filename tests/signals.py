"""Test signals: the input classes SURVEY.md 8(c) lists, as hop streams [C][T][N/2]."""
import numpy as np


def tone_vibrato_noise(C, T, N, seed=1, sr=48000.0):
    rng = np.random.default_rng(seed)
    n = np.arange(T * N // 2)
    out = np.empty((C, n.size), np.float32)
    for c in range(C):
        f = 110.0 * 2 ** (c / 5.0)
        vib = 1.0 + 0.01 * np.sin(2 * np.pi * 5.0 * n / sr)
        ph = 2 * np.pi * np.cumsum(f * vib) / sr
        x = 0.5 * np.sin(ph) + 0.25 * np.sin(2 * ph) + 0.12 * np.sin(3 * ph) + rng.normal(0, 0.05, n.size)
        out[c] = x
    return out.reshape(C, T, N // 2)


def silence(C, T, N):
    return np.zeros((C, T, N // 2), np.float32)


def loud_noise(C, T, N, seed=2):
    return np.random.default_rng(seed).normal(0, 1.0, (C, T, N // 2)).astype(np.float32)


def quiet_noise(C, T, N, seed=3, sigma=1e-3):
    return np.random.default_rng(seed).normal(0, sigma, (C, T, N // 2)).astype(np.float32)


def impulse(C, T, N, at_hop=6):
    x = np.zeros((C, T, N // 2), np.float32)
    for c in range(C):
        x[c, at_hop, (17 * c + 5) % (N // 2)] = 0.9
    return x


def impulse_on_boundary(C, T, N):
    """impulses exactly at the first sample of a hop: in one window the impulse is at sample N/2, in the
    next at sample 0, and re^2 is perfectly flat -- every `mag > mean` of the peak picker is an exact tie
    decided by the rounding of the reference's serial sum"""
    x = np.zeros((C, T, N // 2), np.float32)
    for c in range(C):
        x[c, 2 + c % 3, 0] = 0.7 / (c + 1)
        if T > 8:
            x[c, 8, 0] = -0.31 * (c + 1)
    return x


def sine(C, T, N, amp=0.9, sr=48000.0):
    n = np.arange(T * N // 2)
    out = np.stack([amp * np.sin(2 * np.pi * (220.0 * (c + 1)) * n / sr) for c in range(C)])
    return out.astype(np.float32).reshape(C, T, N // 2)


def dc(C, T, N, level=0.5):
    return np.full((C, T, N // 2), level, np.float32)


def bursts(C, T, N, seed=4):
    """tone bursts separated by exact digital silence: exercises onset, the magSum<=0.05 skip of
    the flux state and the low-pass filter's decay into denormals"""
    rng = np.random.default_rng(seed)
    x = tone_vibrato_noise(C, T, N, seed=seed)
    gate = (rng.random((C, T, 1)) > 0.5).astype(np.float32)
    return (x * gate).astype(np.float32)


def levels(C, T, N, seed=5):
    """noise at per-channel levels 0.003 .. 1: walks the serial flatness product through underflow
    to 0, the normal range and overflow to inf"""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1.0, (C, T, N // 2))
    g = np.logspace(-2.5, 0, C)[:, None, None]
    return (x * g).astype(np.float32)


def flat_edge(C, T, N, seed=6):
    """Noise whose level puts the END of the serial flatness product inside the fp64 subnormal band
    (2^-1074 .. 2^-1022) for some channels: there `magnitudeProduct *= binMagnitude` loses mantissa
    bits before it reaches 0 and flatness depends on IEEE gradual underflow (up to ~0.1 %).
    Gains were calibrated offline so that frame 5 of the stream straddles the band."""
    edge = {512: 0.036494, 1024: 0.056501, 2048: 0.062250, 4096: 0.054533}.get(N, 0.05)
    band = {512: 0.08, 1024: 0.037, 2048: 0.022, 4096: 0.009}.get(N, 0.04)
    base = np.random.default_rng(seed).normal(0, 1.0, (8, N // 2))
    x = base[np.arange(T) % 8][None, :, :]
    g = edge * np.linspace(0.995, 1.0 + 1.3 * band, C)[:, None, None]
    return (x * g).astype(np.float32)


def low_tones(C, T, N, seed=7, sr=48000.0):
    """Fundamentals of 9 .. 70 Hz (a pedal note, mains hum, a rumble), a slow drift and a DC offset under a tone: the
    lag search (ref PitchAnalyser.h:161-190) is not decided within the first few hundred lags, or never dips below its
    threshold and falls back to the global minimum -- the frame kernel's second, whole-lag-array pitch pass"""
    rng = np.random.default_rng(seed)
    n = np.arange(T * N // 2)
    out = np.empty((C, n.size), np.float32)
    for c in range(C):
        f = 9.0 * (70.0 / 9.0) ** (c / max(1, C - 1))
        x = 0.6 * np.sin(2 * np.pi * f * n / sr + 0.3 * c)
        if c % 3 == 1: x = x + 0.4                                   # DC under the tone
        if c % 3 == 2: x = x + 0.2 * n / n.size                      # slow drift
        out[c] = x + rng.normal(0, 1e-3, n.size)
    return out.reshape(C, T, N // 2)


ALL = {
    "tone": tone_vibrato_noise, "silence": silence, "loud_noise": loud_noise, "quiet_noise": quiet_noise,
    "impulse": impulse, "sine": sine, "dc": dc, "bursts": bursts, "levels": levels,
    "flat_edge": flat_edge, "impulse_on_boundary": impulse_on_boundary, "low_tones": low_tones,
}


def assert_features_close(got, want, rtol=1e-5, names=None, what=""):
    """<= rtol relative on every float slot; onset (slot 0) bit-exact; NaN == NaN, inf == inf."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    g = got.reshape(-1, 12).astype(np.float64)
    w = want.reshape(-1, 12).astype(np.float64)
    assert np.array_equal(g[:, 0], w[:, 0], equal_nan=True), "%s onset column differs at rows %s" % (what, np.nonzero(g[:, 0] != w[:, 0])[0][:8])
    same = (g == w) | (np.isnan(g) & np.isnan(w))
    with np.errstate(invalid="ignore", divide="ignore"):
        err = np.abs(g - w) / np.abs(w)
    err = np.where(same, 0.0, err)
    err = np.where(np.isnan(err), np.inf, err)
    bad = np.argwhere(err > rtol)
    if bad.size:
        r, f = bad[0]
        raise AssertionError("%s: %d values beyond rtol=%g; first: row %d slot %d (%s): got %r want %r"
                             % (what, len(bad), rtol, r, f, names[f] if names else f, got.reshape(-1, 12)[r, f], want.reshape(-1, 12)[r, f]))
    return float(err.max()) if err.size else 0.0
