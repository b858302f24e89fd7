"""Second, independent restatement of the reference path in numpy (small cases only).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see fx_oracle.h): like fx_oracle.c this is written from
the reference sources (/root/reference/Source, cited as "ref:") and the published JUCE 4.2 algorithms,
not checked against a run of the reference.  Its only job is to catch transcription slips in
fx_oracle.c: the two restatements were written separately (this one array-at-a-time, that one
loop-at-a-time) and tests/test_oracle.py requires them to agree.
"""
import math

import numpy as np

f32 = np.float32
f64 = np.float64


# --------------------------------------------------------------------------------------------
# JUCE 4.2 juce::FFT (kiss-style radix-4 / radix-2 decimation in time), SURVEY.md App. A.1
# --------------------------------------------------------------------------------------------
def twiddles(n, inverse):
    i = np.arange(n, dtype=f64)
    phase = (2.0 if inverse else -2.0) * math.pi * i / n
    return np.cos(phase).astype(f32), np.sin(phase).astype(f32)


def factors(n):
    out = []
    root = int(math.sqrt(n))
    div, m = 4, n
    while m > 1:
        while m % div:
            div = 3 if div == 2 else (2 if div == 4 else div + 2)
            if div > root:
                div = m
        m //= div
        out.append((div, m))
    return out


def _cmul(ar, ai, br, bi):
    return (ar * br - ai * bi).astype(f32), (ar * bi + ai * br).astype(f32)


def fft(re, im, inverse=False):
    """complex fp32 FFT with the reference's exact rounding sequence; returns (re, im)."""
    n = re.shape[0]
    twr, twi = twiddles(n, inverse)
    facs = factors(n)

    def perform(idx, stride, level):
        # idx: input indices of this sub-transform, in order
        radix, length = facs[level]
        if length == 1:
            outr = re[idx].astype(f32).copy()
            outi = im[idx].astype(f32).copy()
        else:
            parts = [perform(idx[j::radix], stride * radix, level + 1) for j in range(radix)]
            outr = np.concatenate([p[0] for p in parts])
            outi = np.concatenate([p[1] for p in parts])
        k = np.arange(length)
        if radix == 2:
            sr, si = _cmul(outr[length:], outi[length:], twr[k * stride], twi[k * stride])
            r0, i0 = outr[:length].copy(), outi[:length].copy()
            outr[length:], outi[length:] = r0 - sr, i0 - si
            outr[:length], outi[:length] = r0 + sr, i0 + si
        elif radix == 4:
            L = length
            d0r, d0i = outr[:L].copy(), outi[:L].copy()
            s0r, s0i = _cmul(outr[L:2 * L], outi[L:2 * L], twr[k * stride], twi[k * stride])
            s1r, s1i = _cmul(outr[2 * L:3 * L], outi[2 * L:3 * L], twr[k * stride * 2], twi[k * stride * 2])
            s2r, s2i = _cmul(outr[3 * L:], outi[3 * L:], twr[k * stride * 3], twi[k * stride * 3])
            s3r, s3i = s0r + s2r, s0i + s2i
            s4r, s4i = s0r - s2r, s0i - s2i
            s5r, s5i = d0r - s1r, d0i - s1i
            ar, ai = d0r + s1r, d0i + s1i
            outr[2 * L:3 * L], outi[2 * L:3 * L] = ar - s3r, ai - s3i
            outr[:L], outi[:L] = ar + s3r, ai + s3i
            if inverse:
                outr[L:2 * L], outi[L:2 * L] = s5r - s4i, s5i + s4r
                outr[3 * L:], outi[3 * L:] = s5r + s4i, s5i - s4r
            else:
                outr[L:2 * L], outi[L:2 * L] = s5r + s4i, s5i - s4r
                outr[3 * L:], outi[3 * L:] = s5r - s4i, s5i + s4r
        else:
            raise ValueError("unexpected radix %d" % radix)
        return outr, outi

    return perform(np.arange(n), 1, 0)


def forward_real(x):
    """ref RealTimeAudioAnalysis.h:255-278: n reals -> n (re, im) pairs (returned as two arrays)."""
    x = np.asarray(x, f32)
    return fft(x, np.zeros_like(x), False)


# --------------------------------------------------------------------------------------------
# frame-level pieces
# --------------------------------------------------------------------------------------------
def bartlett(x):
    """ref RealTimeAudioAnalysis.h:141-151 (two float gain ramps; exact for power-of-two n)."""
    n = x.shape[0]
    h = n // 2
    inc = f32(1.0) / f32(h)
    up = (np.arange(h, dtype=f32) * inc).astype(f32)
    down = (f32(1.0) - np.arange(h, dtype=f32) * inc).astype(f32)
    return (np.asarray(x, f32) * np.concatenate([up, down])).astype(f32)


def lowpass(x):
    """ref RealTimeAudioAnalysis.h:106-125."""
    a = f32(np.pi) / f32(2.0)
    # expf(-1.5707964f): the correctly rounded value (numpy's own float32 exp is 1 ulp low here)
    b = f32(math.exp(-float(f32(np.pi) / f32(2.0))))
    y = np.empty_like(x, dtype=f32)
    y[0] = x[0]
    for n in range(1, x.shape[0]):
        y[n] = f32(a * x[n]) + f32(b * y[n - 1])
    return y


def log_rms(x):
    """ref RealTimeAnalyser.h:207-208 + JUCE getRMSLevel (float squares, double sum)."""
    x = np.asarray(x, f32)
    s = np.sum((x * x).astype(f32).astype(f64))  # pairwise in numpy: ~1e-16 from the serial sum
    rms = f32(math.sqrt(s / x.shape[0]))
    return f32(math.log10(float(f32(rms * f32(9.0) + f32(1.0)))))      # correctly rounded float log10 (see fx_oracle.c)


def spectral(re, im, prev_mag, lrms, nyquist):
    """ref SpectralCharacteristics.h:62-143 and :145-200.  Returns (5 features, slope, new prev)."""
    n = re.shape[0]
    M = n // 2
    v = re[:M].astype(f64)
    mag = v * v
    eps = 0.01 * f64(lrms)
    rpb = nyquist / M
    fc = np.arange(M, dtype=f64) * rpb + rpb / 2.0
    diff = mag - prev_mag
    flux = float(np.sum(np.where(diff > 0, (diff + np.abs(diff)) / 2.0, 0.0)))
    mag_sum = float(np.sum(mag))
    lhr = float(np.sum(mag[: M // 5 + 1]))
    incl = mag > eps
    flat_sum = float(np.sum(mag[incl]))
    prod = 1.0
    with np.errstate(over="ignore", under="ignore"):
        for m_ in mag[incl]:           # serial on purpose: overflow / underflow are order-dependent
            prod = prod * m_
    cnt = float(np.count_nonzero(incl))
    weighted = float(np.sum(fc * mag))
    flux /= float(f32(M * (M + 1)) / f32(2.0))

    # slope :145-200
    inter = np.empty(2 * n, f32)
    inter[0::2], inter[1::2] = re, im
    max_e = float(np.max(np.abs(inter[:M])))
    max_e = max(max_e, float(np.max(mag)))
    if not (max_e > 0.0001):
        slope = f32(0.0)
    else:
        ne = mag / max_e
        mean_e = float(np.sum(ne)) / M
        prod_sum = float(np.sum(np.arange(M, dtype=f64) * ne))
        bin_var = 0.0
        for i in range(M):
            bin_var += (i / M - 0.5) * (i / M - 0.5)
        bin_var /= M
        e_var = float(np.sum((ne - mean_e) ** 2)) / M
        with np.errstate(divide="ignore", invalid="ignore"):
            bs, es = math.sqrt(bin_var), math.sqrt(e_var)
            r = (prod_sum - (M * mean_e * 0.5)) / float(f32(M) - f32(1.0)) * es * bs
            slope = f32(r * (bs / es)) if es != 0 else f32(np.float64(r) * np.float64(bs) / np.float64(0.0))

    if not (mag_sum > 0.05):
        return (f32(0), f32(0), f32(0), f32(0), f32(0)), slope, prev_mag
    centroid = f32(weighted / mag_sum)
    inv_n = 1.0 / (cnt if cnt > 0 else 1.0)
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        flatness = f32(np.float64(prod) ** inv_n / (inv_n * flat_sum)) if flat_sum > eps else f32(0)
        log_flat = f32(np.log10(f64(flatness) * 9.0 + 1.0))
    c = f32(centroid / f32(nyquist / 2.0))
    log_centroid = f32(math.log10(float(f32(c * f32(9.0) + f32(1.0)))))
    var = float(np.sum(((fc / nyquist) - (f64(centroid) / nyquist)) ** 2 * mag))
    max_spread = f32((f64(centroid) / nyquist) * (1.0 - f64(centroid) / nyquist))
    with np.errstate(divide="ignore", invalid="ignore"):
        spread = f32(np.float64(var / mag_sum) / f64(max_spread))
    return (log_centroid, spread, log_flat, f32(lhr / mag_sum), f32(flux)), slope, mag


def pitch(fre, nyquist):
    """ref PitchAnalyser.h:24-59, :83-217.  fre = real parts of the LPF+windowed spectrum."""
    n = fre.shape[0]
    p = (fre * fre).astype(f32)
    ar, ai = fft(p, np.zeros(n, f32), True)
    scale = f32(1.0) / f32(n)
    d = np.concatenate([(ar * scale).astype(f32), (ai * scale).astype(f32)])
    v = ((d * d).astype(f32) * np.arange(2 * n, dtype=f32)).astype(f32)
    cnd = np.empty(2 * n, f32)
    cnd[0] = 1.0
    s = f32(0.0)
    for i in range(1, 2 * n):
        s = f32(s + v[i])
        cnd[i] = f32(v[i] / s) if s != 0 else f32(0)
    gmin, gidx, lag = f32(100.0), -1.0, -1.0
    i = 2
    while i < n:
        if cnd[i] < gmin:
            gmin, gidx = cnd[i], float(i)
        if cnd[i] < f32(0.01):
            while i + 1 < n and cnd[i + 1] < cnd[i]:
                i += 1
            right = i + 1
            lag = float(i) if cnd[i] <= cnd[right] else float(right)
            break
        i += 1
    lag = gidx if lag == -1.0 else lag
    return (nyquist * 2.0) / lag, lag, cnd


def harmonic(re, f0, nyquist):
    """ref HarmonicCharacteristics.h:46-244.  re = real parts of the RAW frame's spectrum."""
    n = re.shape[0]
    M = n // 2
    v = re[:M].astype(f64)
    mag = v * v
    mag_sum, max_mag = float(np.sum(mag)), float(np.max(mag))
    with np.errstate(divide="ignore", invalid="ignore"):
        normed_d = mag / max_mag
    normed = normed_d.astype(f32)
    sum_normed = float(np.sum(normed_d))
    mean = mag_sum / M
    if mag_sum < 0.005:
        return f32(0), f32(0)
    peaks = []
    for b in range(M):
        if not mag[b] > mean:
            continue
        lo = 0 if b < 2 else b - 2
        hi = b + 2 - (2 - ((M - 1) - b) if b >= M - 2 else 0)
        if all(not (mag[q] > mag[b]) for q in range(lo, hi) if q != b):
            peaks.append(b)
    fr = nyquist / M

    def nb_max(c):
        lo, hi = max(c - 2, 0), min(c + 2, M)
        m_ = normed[c]
        for q in range(lo, hi):
            if normed[q] > m_:
                m_ = normed[q]
        return float(m_)

    f0_bin = int(math.floor(f0 / fr))
    score = 0.0
    for k in range(1, 16):
        b = int(math.floor((f0 / 2.0 ** k) / fr))
        if b == f0_bin:
            continue
        score += nb_max(b)
    for h in range(1, 4):
        b = int(math.floor(f0 * h / fr))
        if b >= M:
            break
        score += nb_max(b)
    her = min(max(score / sum_normed, 0.0), 1.0)
    her = float(f32(her))
    inh = 0.0
    if f0 > 0:
        for b in peaks:
            if b == f0_bin:
                continue
            fs = b * fr if b * fr != 0.0 else fr * 0.5
            fe = (b + 1) * fr
            rs = 1.0 if fs == f0 else max(fs, f0) / min(fs, f0)
            re_ = 1.0 if fe == f0 else max(fe, f0) / min(fe, f0)
            if math.floor(rs) != math.floor(re_):
                continue
            r = min(rs, re_)
            inh += (r - math.floor(r)) * (mag[b] / mag_sum)
    return f32(math.log10(her * 9.0 + 1.0)), f32(math.log10(inh * 9.0 + 1.0))


def raw_features(frames, sample_rate=48000.0):
    """frames [T][N] pre-assembled windows of ONE channel -> raw [T][12] (onset slot left 0).
    ref RealTimeAnalyser.h:141-177, :201-234."""
    frames = np.asarray(frames, f32)
    T, n = frames.shape
    nyq = sample_rate / 2.0
    prev = np.zeros(n // 2, f64)
    out = np.zeros((T, 12), f32)
    for t in range(T):
        x = frames[t]
        lr = log_rms(x)
        sre, sim = forward_real(bartlett(x))
        (cen, spr, flat, ler, flux), slope, prev = spectral(sre, sim, prev, lr, nyq)
        fre, _ = forward_real(bartlett(lowpass(x)))
        rre, _ = forward_real(x)
        f0, _, _ = pitch(fre, nyq)
        her, inh = harmonic(rre, f0, nyq)
        out[t] = [0, lr, f32(f0 / 5000.0), cen, spr, flat, ler, flux, slope, her, her, inh]
    return out
