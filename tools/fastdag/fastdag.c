/*
 * fastdag.c -- CPU study of a TOLERANCE-MODE transform DAG for the frame kernels (round 5, review item 1, stage A).
 *
 * TEST INFRASTRUCTURE, build container only; nothing here is part of the product.  It answers one question with data: if the four
 * transforms of a frame stop following the reference FFT's rounding DAG (which forfeits FMA and the real-input symmetries), how far
 * do the twelve values move from the oracle's, how often does a discrete decision flip, and what would a per-frame guard + exact
 * replay cost?
 *
 * The exact side is oracle/fx_oracle.c itself (included below, so that its static functions are reachable): the reference's
 * arithmetic line by line.  The fast side computes the same features with the same feature code on spectra from a variant DAG:
 *   - twiddle products contracted to fused multiply-adds (a.r*b.r - a.i*b.i -> fma(a.r, b.r, -(a.i*b.i)));
 *   - TWO real frames per complex transform: the Bartlett-windowed frame (spectral analyser, ref RealTimeAnalyser.h:212-215) and the
 *     raw frame (harmonic analyser, ref :161) are the real and imaginary part of ONE N-point complex transform, split by the Hermitian
 *     symmetries;
 *   - the low-passed, windowed frame (pitch, ref :152-160) as an N/2-point complex transform of its even / odd samples;
 *   - the inverse transform of re^2 (ref PitchAnalyser.h:110-127) as a HALF-LENGTH real-even transform: re^2 is real and even, so its
 *     inverse is, and only lags below N are ever read (ref :161-190).
 * Signals: the mix of tools/stress_signals.py restated in C (eight kinds, levels 1e-5 .. 31.6), random order mode / onset settings /
 * gain per case, as tools/stress_parity.py draws them.
 *
 * Output: per window size and signal kind, per slot -- frames, values beyond 1e-5 relative, worst finite error, NaN / inf mismatches;
 * flips of the lag, the peak list, the flatness gate, the three level gates and the onset; the share of frames with ANY violation (the
 * replay rate of a PERFECT guard: a lower bound for any real one); and, for the guard described in guard_frame(), the frames it
 * taints and the violations it misses.
 *
 *   gcc -O2 -std=gnu11 -mfma -ffp-contract=off -o fastdag fastdag.c -lm -lpthread
 *   ./fastdag <frames per window size, e.g. 3e7> [threads] [seed] [windows, e.g. 1024,2048,4096] [guard K] [onset type, -1 = random] [variant 0 | 1 | 2 = null test] [kind, -1 = the mix, 8 = bench.py's signal]
 */
#define _GNU_SOURCE
#include "../../oracle/fx_oracle.c"

#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>

static int VARIANT = 0;      /* 0: FMA + two frames per transform + half-length transforms; 1: the reference's own DAG with fused twiddle products only */

/* ------------------------------------------------------------------------- */
/* the variant DAG                                                            */
/* ------------------------------------------------------------------------- */
static inline cpx c_mul_f(cpx a, cpx b)
{
    if (VARIANT == 2) return c_mul(a, b);          /* the harness's own null test: the reference's products, so the "fast" side IS the exact one */
    cpx c = { fmaf(a.r, b.r, -(a.i * b.i)), fmaf(a.r, b.i, a.i * b.r) };
    return c;
}

static void fast_butterfly2(const fft_cfg* c, cpx* data, int stride, int length)
{
    cpx* end = data + length;
    const cpx* tw = c->tw;
    for (int i = length; --i >= 0;) {
        const cpx s = c_mul_f(*end, *tw);
        tw += stride;
        *end = c_sub(*data, s);
        end++;
        *data = c_add(*data, s);
        data++;
    }
}

static void fast_butterfly4(const fft_cfg* c, cpx* data, int stride, int length)
{
    const int l2 = length * 2, l3 = length * 3;
    const cpx *t1 = c->tw, *t2 = c->tw, *t3 = c->tw;
    for (int i = length; --i >= 0;) {
        const cpx s0 = c_mul_f(data[length], *t1);
        const cpx s1 = c_mul_f(data[l2], *t2);
        const cpx s2 = c_mul_f(data[l3], *t3);
        const cpx s3 = c_add(s0, s2);
        const cpx s4 = c_sub(s0, s2);
        const cpx s5 = c_sub(*data, s1);
        *data = c_add(*data, s1);
        data[l2] = c_sub(*data, s3);
        t1 += stride; t2 += stride * 2; t3 += stride * 3;
        *data = c_add(*data, s3);
        data[length].r = s5.r + s4.i; data[length].i = s5.i - s4.r;
        data[l3].r     = s5.r - s4.i; data[l3].i     = s5.i + s4.r;
        ++data;
    }
}

/* forward complex transform, the same factorisation as the reference's, fused twiddle products */
static void fast_perform(const fft_cfg* c, const cpx* in, cpx* out, int stride, const fft_factor* facs)
{
    const fft_factor f = *facs++;
    if (f.radix == 1) { *out = *in; return; }
    if (f.length == 1) {
        for (int j = 0; j < f.radix; j++) out[j] = in[j * stride];
    } else {
        for (int j = 0; j < f.radix; j++)
            fast_perform(c, in + j * stride, out + j * f.length, stride * f.radix, facs);
    }
    if (f.radix == 2)      fast_butterfly2(c, out, stride, f.length);
    else if (f.radix == 4) fast_butterfly4(c, out, stride, f.length);
}

typedef struct {
    int n;
    fft_cfg full, half;          /* forward tables of N and N/2 points */
    cpx *z, *Z, *g, *G;
} fast_cfg;

static void fast_init(fast_cfg* f, int n)
{
    f->n = n;
    fft_cfg_init(&f->full, n, 0);
    fft_cfg_init(&f->half, n / 2, 0);
    f->z = (cpx*) malloc(sizeof(cpx) * (size_t) n); f->Z = (cpx*) malloc(sizeof(cpx) * (size_t) n);
    f->g = (cpx*) malloc(sizeof(cpx) * (size_t) n); f->G = (cpx*) malloc(sizeof(cpx) * (size_t) n);
}
static void fast_free(fast_cfg* f) { fft_cfg_free(&f->full); fft_cfg_free(&f->half); free(f->z); free(f->Z); free(f->g); free(f->G); }

/* two real frames a, b -> their spectra as 2N interleaved floats each (every bin, as performRealOnlyForwardTransform leaves them) */
static void fast_two_for_one(fast_cfg* f, const float* a, const float* b, float* spec_a, float* spec_b)
{
    const int n = f->n;
    for (int i = 0; i < n; i++) { f->z[i].r = a[i]; f->z[i].i = b[i]; }
    fast_perform(&f->full, f->z, f->Z, 1, f->full.factors);
    for (int k = 0; k < n; k++) {
        const cpx p = f->Z[k], q = f->Z[(n - k) & (n - 1)];
        spec_a[2 * k]     = 0.5f * (p.r + q.r);
        spec_a[2 * k + 1] = 0.5f * (p.i - q.i);
        spec_b[2 * k]     = 0.5f * (p.i + q.i);
        spec_b[2 * k + 1] = -0.5f * (p.r - q.r);
    }
}

/* one real frame through an N/2-point complex transform of its even / odd samples: re, im of bins 0 .. N/2 */
static void fast_real_half(fast_cfg* f, const float* x, float* re, float* im)
{
    const int n = f->n, h = n / 2;
    for (int i = 0; i < h; i++) { f->g[i].r = x[2 * i]; f->g[i].i = x[2 * i + 1]; }
    fast_perform(&f->half, f->g, f->G, 1, f->half.factors);
    for (int k = 0; k <= h; k++) {
        const cpx p = f->G[k & (h - 1)], q = f->G[(h - k) & (h - 1)];
        const cpx e = { 0.5f * (p.r + q.r), 0.5f * (p.i - q.i) };             /* spectrum of the even samples */
        const cpx o = { 0.5f * (p.i + q.i), -0.5f * (p.r - q.r) };            /* spectrum of the odd samples */
        const cpx w = (k < h) ? f->full.tw[k] : (cpx) { -1.0f, 0.0f };        /* e^{-2 pi i k / N} */
        const cpx t = c_mul_f(o, w);
        re[k] = e.r + t.r;
        if (im) im[k] = e.i + t.i;
    }
}

/* ------------------------------------------------------------------------- */
/* one frame on either side: the pieces of run_frame() (fx_oracle.c), with the values the study compares kept                        */
/* ------------------------------------------------------------------------- */
typedef struct {
    float  raw[FXO_NUM_FEATURES];
    float  lag;
    float  log_rms;
    int    num_peaks;
    int    peaks[2048];
    double mag_sum_w, max_e_w, mag_sum_r;     /* the three level gates' operands (:121, :165, HarmonicCharacteristics.h:88) */
    float  *wspec;                            /* windowed spectrum, 2N floats (kept by the caller's buffers) */
    /* guard inputs of the fast side */
    float  *cnd;                              /* cumulative normalised difference, N floats */
    float  *acf;                              /* d[s], N floats (before the squaring) */
    float  top_pair, top_filt;                /* largest |re|, |im| of the windowed / raw spectra, of the filtered frame's */
} frame_view;

/* ref PitchAnalyser.h:129-190 on an auto-correlation d[0..N) already scaled by 1/N (the lag scan reads nothing beyond N) */
static float lag_from_acf(int N, const float* d, float* cnd)
{
    float sum = 0.0f;
    cnd[0] = 1.0f;
    for (int s = 1; s < N; s++) {
        const float v = d[s] * d[s] * s;
        sum += v;
        cnd[s] = (sum != 0.0f) ? v / sum : 0.0f;
    }
    const float threshold = 0.01f;
    float gmin_idx = -1.0f, gmin = 100.0f, lag = -1.0f;
    for (int s = 2; s < N; s++) {
        if (cnd[s] < gmin) { gmin_idx = (float) s; gmin = cnd[s]; }
        if (cnd[s] < threshold) {
            while (s + 1 < N && cnd[s + 1] < cnd[s]) s++;
            const int right = s + 1;
            /* (cnd[N] is read by the reference when the walk ends at N - 1: it is the first lag of the imaginary half, ~0 -- only reachable
             * for a cnd that falls all the way to the end; both sides treat it as the reference's value would be compared: as larger) */
            lag = (right >= N || cnd[s] <= cnd[right]) ? (float) s : (float) right;
            break;
        }
    }
    return (lag == -1.0f) ? gmin_idx : lag;
}

static void finish_frame(fxo_channel* c, const spec_frame* s, const harm_frame* h, float* raw12)
{
    afeatures* fs = &c->feat;
    afeatures* fh = (c->order_mode == FXO_ORDER_ISOLATED) ? &c->feat_harm : &c->feat;
    float onset = 0.0f;
    if (c->order_mode == FXO_ORDER_HARMONIC_THEN_SPECTRAL) {
        af_update(fh, FXO_RMS, h->log_rms);  harmonic_writes(fh, h);
        af_update(fs, FXO_RMS, s->log_rms);  onset = spectral_writes(c, fs, s);
    } else {
        af_update(fs, FXO_RMS, s->log_rms);  onset = spectral_writes(c, fs, s);
        af_update(fh, FXO_RMS, h->log_rms);  harmonic_writes(fh, h);
    }
    raw12[FXO_ONSET] = onset;         raw12[FXO_RMS] = s->log_rms;   raw12[FXO_F0] = h->f0_feature;
    raw12[FXO_CENTROID] = s->centroid; raw12[FXO_SPREAD] = s->spread; raw12[FXO_FLATNESS] = s->flatness;
    raw12[FXO_LER] = s->ler;          raw12[FXO_FLUX] = s->flux;     raw12[FXO_SLOPE] = s->slope;
    raw12[FXO_HER] = h->her;          raw12[FXO_OER] = h->oer;       raw12[FXO_INHARM] = h->inharm;
}

static void level_operands(const fxo_channel* c, const float* wspec, const float* rspec, frame_view* v)
{
    const int M = c->m;
    double ms = 0.0, mx = buf_magnitude(wspec, M), mr = 0.0;
    for (int i = 0; i < M; i++) {
        const double a = wspec[2 * i], b = rspec[2 * i];
        ms += a * a; if (a * a > mx) mx = a * a;
        mr += b * b;
    }
    v->mag_sum_w = ms; v->max_e_w = mx; v->mag_sum_r = mr;
}

static void list_peaks(fxo_channel* c, frame_view* v)
{
    /* c->mags holds the raw-frame magnitudes after harmonic_characteristics */
    double sum = 0.0;
    for (int i = 0; i < c->m; i++) sum += c->mags[i];
    const double mean = sum / (double) c->m;
    v->num_peaks = 0;
    if (sum < 0.005) return;
    for (int b = 0; b < c->m; b++) if (bin_is_peak(b, c->mags, c->m, mean)) v->peaks[v->num_peaks++] = b;
}

/* the oracle's frame: spectral_compute + harmonic_compute of fx_oracle.c, with the lag and the windowed spectrum kept */
static void exact_frame(fxo_channel* c, frame_view* v, float* wspec_keep)
{
    const spec_frame s = spectral_compute(c);
    memcpy(wspec_keep, c->spec, sizeof(float) * 2 * (size_t) c->n);
    harm_frame h;
    {
        /* harmonic_compute, with the lag */
        const float rms = buf_rms(c->overlap, c->n);
        h.log_rms = log10_float(rms * 9.0f + 1.0f);
        fxo_lowpass(c->n, c->overlap, c->filt);
        fxo_bartlett(c->n, c->filt);
        forward_real(&c->fwd, c->filt, c->fspec, c->scratch);
        forward_real(&c->fwd, c->overlap, c->spec, c->scratch);
        const double f0 = estimate_pitch(&c->inv, c->nyquist, c->fspec, c->work, c->scratch, &v->lag);
        h.f0_feature = (float) (f0 / 5000.0);
        const harmonic_out ho = harmonic_characteristics(c, c->spec, f0);
        h.her = ho.her; h.oer = ho.her; h.inharm = ho.inharm;
    }
    v->log_rms = s.log_rms;
    v->wspec = wspec_keep;
    level_operands(c, wspec_keep, c->spec, v);
    list_peaks(c, v);
    finish_frame(c, &s, &h, v->raw);
}

typedef struct { float *win, *wspec, *rspec, *fre, *pw, *acf, *cnd; } fast_bufs;

static void fast_frame(fxo_channel* c, fast_cfg* f, fast_bufs* b, frame_view* v)
{
    const int N = c->n, H = N / 2;
    spec_frame s; harm_frame h;
    const float rms = buf_rms(c->overlap, N);
    s.log_rms = h.log_rms = log10_float(rms * 9.0f + 1.0f);
    memcpy(b->win, c->overlap, sizeof(float) * (size_t) N);
    fxo_bartlett(N, b->win);
    if (VARIANT >= 1) {
        for (int i = 0; i < N; i++) { f->z[i].r = b->win[i]; f->z[i].i = 0.0f; }
        fast_perform(&f->full, f->z, (cpx*) b->wspec, 1, f->full.factors);
        for (int i = 0; i < N; i++) { f->z[i].r = c->overlap[i]; f->z[i].i = 0.0f; }
        fast_perform(&f->full, f->z, (cpx*) b->rspec, 1, f->full.factors);
    } else
    fast_two_for_one(f, b->win, c->overlap, b->wspec, b->rspec);
    {
        const spectral_out so = spectral_characteristics(c, b->wspec, s.log_rms);
        s.centroid = so.centroid; s.spread = so.spread; s.flatness = so.flatness; s.ler = so.ler; s.flux = so.flux;
        s.slope = spectral_slope(c, b->wspec);
    }
    fxo_lowpass(N, c->overlap, c->filt);
    fxo_bartlett(N, c->filt);
    const float scale = 1.0f / N;
    if (VARIANT >= 1) {
        /* the reference's DAG, fused: forward of the filtered frame, re^2 with imag := 0, the inverse as the forward transform of the
         * conjugate (= of the same real data), planar real part scaled by 1/N */
        for (int i = 0; i < N; i++) { f->z[i].r = c->filt[i]; f->z[i].i = 0.0f; }
        fast_perform(&f->full, f->z, f->Z, 1, f->full.factors);
        for (int k = 0; k < N; k++) { if (k <= H) b->fre[k] = f->Z[k].r; b->pw[k] = f->Z[k].r * f->Z[k].r; f->z[k].r = b->pw[k]; f->z[k].i = 0.0f; }
        fast_perform(&f->full, f->z, f->Z, 1, f->full.factors);      /* real input: the inverse is the conjugate of this, same real part */
        for (int sIdx = 0; sIdx < N; sIdx++) b->acf[sIdx] = f->Z[sIdx].r * scale;
    } else {
    fast_real_half(f, c->filt, b->fre, NULL);                              /* re of bins 0 .. N/2 */
    /* re^2 of every bin (ref PitchAnalyser.h:83-108), real and even: bins above N/2 mirror those below */
    for (int k = 0; k <= H; k++) b->pw[k] = b->fre[k] * b->fre[k];
    for (int k = H + 1; k < N; k++) b->pw[k] = b->pw[N - k];
    /* its inverse, real and even: the forward real transform's real part, lags 0 .. N/2, mirrored */
    fast_real_half(f, b->pw, b->acf, NULL);
    for (int sIdx = 0; sIdx <= H; sIdx++) b->acf[sIdx] *= scale;
    for (int sIdx = H + 1; sIdx < N; sIdx++) b->acf[sIdx] = b->acf[N - sIdx];
    }
    v->lag = lag_from_acf(N, b->acf, b->cnd);
    const double f0 = (c->nyquist * 2.0f) / v->lag;
    h.f0_feature = (float) (f0 / 5000.0);
    {
        const harmonic_out ho = harmonic_characteristics(c, b->rspec, f0);
        h.her = ho.her; h.oer = ho.her; h.inharm = ho.inharm;
    }
    v->log_rms = s.log_rms;
    v->wspec = b->wspec; v->cnd = b->cnd; v->acf = b->acf;
    v->top_pair = v->top_filt = 0.0f;
    for (int k = 0; k < 2 * N; k++) { const float a = fabsf(b->wspec[k]), q = fabsf(b->rspec[k]); if (a > v->top_pair) v->top_pair = a; if (q > v->top_pair) v->top_pair = q; }
    for (int k = 0; k <= H; k++) if (fabsf(b->fre[k]) > v->top_filt) v->top_filt = fabsf(b->fre[k]);
    level_operands(c, b->wspec, b->rspec, v);
    list_peaks(c, v);
    finish_frame(c, &s, &h, v->raw);
}

/* ------------------------------------------------------------------------- */
/* A guard the fast kernel could evaluate from its OWN values: a frame is tainted (= replayed on the exact DAG) when a discrete        */
/* decision was taken inside the error band of its operands, or when a slot's conditioning says its error may exceed the bar.          */
/* e_abs: the transform error scale of the frame, eps32 * sqrt(log2 N) * (L2 norm of the transform's input) * GUARD_K.               */
/* ------------------------------------------------------------------------- */
static double GUARD_K = 4.0;

enum { T_LAG = 1, T_GATE = 2, T_PEAKS = 4, T_LEVEL = 8, T_FLUX = 16, T_FLAT = 32, T_SLOPE = 64, T_PROD = 128 };

static double l2(const float* x, int n) { double s = 0.0; for (int i = 0; i < n; i++) s += (double) x[i] * x[i]; return sqrt(s); }

static int guard_frame(const fxo_channel* c, const fast_bufs* b, const frame_view* v, const double* prev_mag_before)
{
    const int N = c->n, M = c->m;
    int taint = 0;
    const double eps32 = 5.96e-8, lg = sqrt(log2((double) N));
    /* the two-for-one transform carries both frames: its error scale is that of the pair */
    (void) lg;
    /* a transform's error follows its LARGEST bins (a tone's energy sits in a few bins whose magnitude every late butterfly near them
     * carries): eps32 x the largest magnitude of the pair's spectra, of the filtered frame's */
    const double e_pair = GUARD_K * eps32 * (double) v->top_pair;
    const double e_filt = GUARD_K * eps32 * (double) v->top_filt;
    /* (1) flatness gate |re| ~ sqrt(0.01 logRMS), ref SpectralCharacteristics.h:89-94 */
    {
        const double thr = sqrt(0.01 * (double) v->log_rms);
        for (int m = 0; m < M; m++) if (fabs(fabs((double) b->wspec[2 * m]) - thr) <= e_pair) { taint |= T_GATE; break; }
    }
    /* (2) the level gates :121, :165, HarmonicCharacteristics.h:88: sums of squares move by ~2 e sqrt(sum) */
    {
        const double dw = 2.0 * e_pair * sqrt(v->mag_sum_w) + 1e-12, dr = 2.0 * e_pair * sqrt(v->mag_sum_r) + 1e-12;
        if (fabs(v->mag_sum_w - 0.05) <= dw || fabs(v->max_e_w - 0.0001) <= 2.0 * e_pair * sqrt(v->max_e_w) + 1e-12 || fabs(v->mag_sum_r - 0.005) <= dr) taint |= T_LEVEL;
    }
    /* (3) peaks, ref HarmonicCharacteristics.h:127-145: mag > mean and the three neighbour compares, on bins above the mean */
    {
        double sum = 0.0;
        for (int i = 0; i < M; i++) sum += (double) b->rspec[2 * i] * b->rspec[2 * i];
        const double mean = sum / M;
        if (sum >= 0.005) {
            for (int i = 0; i < M && !(taint & T_PEAKS); i++) {
                const double a = fabs((double) b->rspec[2 * i]), mag = a * a, band = 2.0 * e_pair * a + e_pair * e_pair;
                if (fabs(mag - mean) <= band + 2.0 * e_pair * sqrt(sum) / M) { taint |= T_PEAKS; break; }
                if (mag <= mean) continue;
                const int left = i < 2 ? 2 - i : 0, right = i >= M - 2 ? 2 - ((M - 1) - i) : 0;
                for (int nb = i - (2 - left); nb < i + (2 - right); nb++) {
                    if (nb == i) continue;
                    const double o = fabs((double) b->rspec[2 * nb]);
                    if (fabs(o * o - mag) <= band + 2.0 * e_pair * o + e_pair * e_pair) { taint |= T_PEAKS; break; }
                }
            }
        }
    }
    /* (4) the lag, ref PitchAnalyser.h:161-190: every comparison the search made, against the band of cnd = d^2 s / sum */
    {
        /* error of d[s]: the forward transform's error goes through the squaring (2 |re| e) and the second transform */
        double pw_l2 = 0.0, re_max = 0.0;
        for (int k = 0; k < N; k++) { pw_l2 += (double) b->pw[k] * b->pw[k]; }
        for (int k = 0; k <= N / 2; k++) if (fabs((double) b->fre[k]) > re_max) re_max = fabs((double) b->fre[k]);
        double pw_top = 0.0;
        for (int k = 0; k < N; k++) if (b->pw[k] > pw_top) pw_top = b->pw[k];
        (void) pw_l2;
        /* d = (1/N) sum pw cos: the second transform's own error follows d[0] = sum pw / N; the first transform's goes through the squaring */
        const double e_d = GUARD_K * eps32 * fabs((double) b->acf[0]) + 2.0 * re_max * e_filt * sqrt((double) N) / N;
        (void) pw_top;
        const float* cnd = b->cnd; const float* d = b->acf;
        double run = 0.0;
        int decided = 0;
        float gmin = 100.0f;
        for (int s = 1; s < N && !decided; s++) {
            run += (double) d[s] * d[s] * s;
            if (s < 2) continue;
            const double band = run > 0.0 ? (2.0 * fabs((double) d[s]) * e_d + e_d * e_d) * s / run + 4e-7 * cnd[s] : 0.0;
            if (fabs((double) cnd[s] - 0.01) <= band) { taint |= T_LAG; break; }
            if (cnd[s] < gmin) gmin = cnd[s];
            if (cnd[s] < 0.01f) {
                int t = s;
                double r2 = run;
                for (;;) {
                    if (t + 1 >= N) break;
                    r2 += (double) d[t + 1] * d[t + 1] * (t + 1);
                    const double b2 = r2 > 0.0 ? ((2.0 * fabs((double) d[t + 1]) * e_d + e_d * e_d) * (t + 1) + (2.0 * fabs((double) d[t]) * e_d + e_d * e_d) * t) / r2 + 4e-7 * (cnd[t] + cnd[t + 1]) : 0.0;
                    if (fabs((double) cnd[t + 1] - (double) cnd[t]) <= b2) { taint |= T_LAG; break; }
                    if (!(cnd[t + 1] < cnd[t])) break;
                    t++;
                }
                decided = 1;
            }
        }
        if (!decided && !(taint & T_LAG)) {
            /* never below the threshold: the global minimum must be unique within the band */
            int at = -1; float best = 100.0f;
            for (int s = 2; s < N; s++) if (cnd[s] < best) { best = cnd[s]; at = s; }
            double run2 = 0.0;
            for (int s = 1; s < N; s++) {
                run2 += (double) d[s] * d[s] * s;
                if (s < 2 || s == at) continue;
                const double band = run2 > 0.0 ? (2.0 * fabs((double) d[s]) * e_d + e_d * e_d) * s / run2 * 2.0 + 8e-7 * cnd[s] : 0.0;
                if ((double) cnd[s] - (double) best <= band) { taint |= T_LAG; break; }
            }
        }
    }
    /* (5) conditioning of the continuous slots */
    {
        /* flux = sum of rectified differences of re^2 (ref :76-79): error ~ 2 e sqrt(sum over counted bins of re_t^2 + re_prev^2) */
        double flux = 0.0, q = 0.0;
        for (int m = 0; m < M; m++) {
            const double a = (double) b->wspec[2 * m] * b->wspec[2 * m], p = prev_mag_before[m];
            if (a - p > 0.0) { flux += a - p; q += a + p; }
        }
        if (v->mag_sum_w > 0.05 || 1) {
            if (flux > 0.0 && 2.0 * e_pair * sqrt(q) > 0.25e-5 * flux) taint |= T_FLUX;
        }
        /* flatness: geometric mean of the gated magnitudes: relative error ~ (2 e / cnt) sqrt(sum 1 / re^2) */
        const double eps_gate = 0.01 * (double) v->log_rms;
        double inv2 = 0.0, cnt = 0.0, lsum = 0.0, lmin = 0.0, lmax = 0.0;
        for (int m = 0; m < M; m++) {
            const double a = (double) b->wspec[2 * m] * b->wspec[2 * m];
            if (a > eps_gate) { inv2 += 1.0 / a; cnt += 1.0; lsum += log2(a); if (lsum < lmin) lmin = lsum; if (lsum > lmax) lmax = lsum; }
        }
        if (cnt > 0.0 && 2.0 * e_pair * sqrt(inv2) / cnt > 0.25e-5) taint |= T_FLAT;
        /* the serial product's excursions (ref :92): inside the normal range, or decided far outside it */
        if ((lmax > 1020.0 && lmax < 1028.0) || (lmin < -1018.0 && lmin > -1080.0) || (lsum < -1018.0 && lsum > -1080.0)) taint |= T_PROD;
        /* slope: r ~ sum (i - M/2) e_i (ref :175-198): cancellation when the spectrum is balanced about M/2 */
        double num = 0.0, den = 0.0;
        for (int m = 0; m < M; m++) {
            const double a = (double) b->wspec[2 * m] * b->wspec[2 * m];
            num += ((double) m - 0.5 * M) * a;
            den += ((double) m - 0.5 * M) * ((double) m - 0.5 * M) * a;
        }
        if (v->max_e_w > 0.0001 && 2.0 * e_pair * sqrt(den) > 0.25e-5 * fabs(num)) taint |= T_SLOPE;
        /* low energy ratio (ref :86-87,125): the partial sum up to bin M/5 against its own error */
        double lhr = 0.0;
        for (int m = 0; m <= M / 5; m++) lhr += (double) b->wspec[2 * m] * b->wspec[2 * m];
        if (v->mag_sum_w > 0.05 && 2.0 * e_pair * sqrt(lhr) > 0.25e-5 * lhr) taint |= T_SLOPE;
        /* harmonic energy ratio / inharmonicity (ref HarmonicCharacteristics.h:147-244): sums of a few raw-frame magnitudes over the total */
        if (v->mag_sum_r >= 0.005) {
            double top = 0.0;
            for (int m = 0; m < M; m++) { const double a = (double) b->rspec[2 * m] * b->rspec[2 * m]; if (a > top) top = a; }
            /* every probe is a magnitude within two bins of a multiple / sub-octave of f0: bound its error by the neighbourhood's largest */
            const double rpb = c->nyquist / (double) M, f0 = (c->nyquist * 2.0f) / v->lag;
            double score = 0.0, err = 0.0;
            for (int k = 1; k <= 18; k++) {
                const double fq = k <= 15 ? f0 / pow(2.0, (double) k) : f0 * (double) (k - 15);
                const int bin = (int) floor(fq / rpb);
                if (bin < 0 || bin >= M) continue;
                double mx = 0.0;
                for (int q = bin - 2 < 0 ? 0 : bin - 2; q < (bin + 2 < M ? bin + 2 : M); q++) { const double a = (double) b->rspec[2 * q] * b->rspec[2 * q]; if (a > mx) mx = a; }
                { const double a = (double) b->rspec[2 * bin] * b->rspec[2 * bin]; if (a > mx) mx = a; }
                score += mx; err += 2.0 * e_pair * sqrt(mx);
            }
            if (score > 0.0 && err > 0.25e-5 * score) taint |= T_PEAKS;
            double inh = 0.0, ierr = 0.0;
            for (int q = 0; q < v->num_peaks; q++) { const double a = (double) b->rspec[2 * v->peaks[q]] * b->rspec[2 * v->peaks[q]]; inh += a; ierr += 2.0 * e_pair * sqrt(a); }
            if (inh > 0.0 && ierr > 0.25e-5 * inh) taint |= T_PEAKS;
        }
    }
    return taint;
}

/* ------------------------------------------------------------------------- */
/* signals: tools/stress_signals.py restated                                                                                         */
/* ------------------------------------------------------------------------- */
typedef struct { uint64_t s; int have; double spare; } rng_t;
static uint64_t rng_u64(rng_t* r) { uint64_t z = (r->s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double rng_uniform(rng_t* r) { return (double) (rng_u64(r) >> 11) * (1.0 / 9007199254740992.0); }
static double rng_range(rng_t* r, double a, double b) { return a + (b - a) * rng_uniform(r); }
static int rng_int(rng_t* r, int lo, int hi) { return lo + (int) (rng_u64(r) % (uint64_t) (hi - lo)); }
static double rng_normal(rng_t* r)
{
    if (r->have) { r->have = 0; return r->spare; }
    double u, v, q;
    do { u = 2.0 * rng_uniform(r) - 1.0; v = 2.0 * rng_uniform(r) - 1.0; q = u * u + v * v; } while (q >= 1.0 || q == 0.0);
    const double m = sqrt(-2.0 * log(q) / q);
    r->spare = v * m; r->have = 1;
    return u * m;
}

enum { NUM_KINDS = 9, MIX_KINDS = 8 };
static const char* KIND_NAMES[NUM_KINDS] = { "harmonic tone", "noise", "tone + noise", "sparse impulses", "gated bursts", "chirp", "DC + tiny noise", "silence + one loud hop",
                                             "bench.py's synthetic mix (not part of the stress mix)" };
static int ONLY_KIND = -1;       /* -1: the stress mix (kinds 0 .. 7); 8: the signal bench.py analyses, reported by itself */

static int make_signal(rng_t* r, int T, int N, float* out)
{
    const int n = T * N / 2, H = N / 2;
    const int kind = ONLY_KIND >= 0 ? ONLY_KIND : rng_int(r, 0, MIX_KINDS);
    const double level = kind == 8 ? 1.0 : pow(10.0, rng_range(r, -5.0, 1.5));
    const double two_pi = 6.283185307179586;
    double* x = (double*) calloc((size_t) n, sizeof(double));
    if (kind == 0) {
        const double f = rng_range(r, 40, 6000);
        const int harm = rng_int(r, 1, 8);
        for (int h = 0; h < harm; h++) {
            const double a = rng_uniform(r) / (h + 1), ph0 = rng_range(r, 0, 6);
            for (int t = 0; t < n; t++) x[t] += a * sin((h + 1) * two_pi * f * t / 48000.0 + ph0);
        }
    } else if (kind == 1) {
        for (int t = 0; t < n; t++) x[t] = rng_normal(r);
    } else if (kind == 2) {
        const double f = rng_range(r, 40, 6000), g = rng_range(r, 0, 0.3);
        for (int t = 0; t < n; t++) x[t] = sin(two_pi * f * t / 48000.0) + g * rng_normal(r);
    } else if (kind == 3) {
        const int cnt = n / 3000 > 1 ? n / 3000 : 1;
        for (int i = 0; i < cnt; i++) x[rng_int(r, 0, n)] = rng_normal(r);
    } else if (kind == 4) {
        const double f = rng_range(r, 80, 2000);
        for (int h = 0; h < T; h++) {
            if (rng_uniform(r) > 0.5) for (int i = 0; i < H; i++) { const int t = h * H + i; x[t] = sin(two_pi * f * t / 48000.0); }
        }
    } else if (kind == 5) {
        const double f0 = rng_range(r, 50, 500), f1 = rng_range(r, 500, 12000);
        double acc = 0.0;
        for (int t = 0; t < n; t++) { acc += f0 + (f1 - f0) * t / (n > 1 ? n - 1 : 1); x[t] = sin(two_pi * acc / 48000.0); }
    } else if (kind == 6) {
        const double dc = rng_range(r, -1, 1);
        for (int t = 0; t < n; t++) x[t] = dc + 1e-3 * rng_normal(r);
    } else if (kind == 8) {
        /* SURVEY.md 8(d) / feature-extractor_amd/synth.py: three harmonics of a pitch from the 72-step scale above 55 Hz + uniform noise of +-0.05 */
        const double f = 55.0 * pow(2.0, (double) rng_int(r, 0, 72) / 12.0), ph0 = rng_range(r, 0, two_pi);
        for (int t = 0; t < n; t++) {
            const double ph = ph0 + two_pi * f * t / 48000.0;
            x[t] = 0.4 * sin(ph) + 0.2 * sin(2 * ph) + 0.1 * sin(3 * ph) + rng_range(r, -0.05, 0.05);
        }
    } else {
        const int h = rng_int(r, 0, T);
        for (int i = 0; i < H; i++) x[h * H + i] = rng_normal(r);
    }
    for (int t = 0; t < n; t++) out[t] = (float) (x[t] * level);
    free(x);
    return kind;
}

/* ------------------------------------------------------------------------- */
/* bookkeeping                                                                                                                        */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t frames;
    uint64_t beyond[FXO_NUM_FEATURES];      /* values beyond 1e-5 relative (slot 0: onset flips) */
    uint64_t special[FXO_NUM_FEATURES];     /* NaN / inf / zero on one side only */
    double   worst[FXO_NUM_FEATURES];       /* worst finite relative error */
    uint64_t lag_flips, peak_flips, gate_flip_frames, level_flips;
    uint64_t violating;                     /* frames with any of the above that shows in a value */
    uint64_t violating_harm;                /* ... in a slot of the harmonic analyser (f0, HER, OER, inharmonicity): what a HYBRID -- spectral analyser on the exact DAG,
                                               only the pitch pair and the raw-frame transform on the variant -- would still have to replay */
    uint64_t tainted, tainted_by[8], missed; /* the guard: frames it taints, violating frames it lets through */
    uint64_t missed_slot[FXO_NUM_FEATURES];
    double   worst_ratio_pair, worst_ratio_filt;     /* observed transform error / the guard's scale (GUARD_K = 1) */
} stats_t;

static void stats_merge(stats_t* a, const stats_t* b)
{
    a->frames += b->frames; a->lag_flips += b->lag_flips; a->peak_flips += b->peak_flips; a->gate_flip_frames += b->gate_flip_frames;
    a->level_flips += b->level_flips; a->violating += b->violating; a->violating_harm += b->violating_harm; a->tainted += b->tainted; a->missed += b->missed;
    for (int i = 0; i < 8; i++) a->tainted_by[i] += b->tainted_by[i];
    for (int i = 0; i < FXO_NUM_FEATURES; i++) {
        a->beyond[i] += b->beyond[i]; a->special[i] += b->special[i]; a->missed_slot[i] += b->missed_slot[i];
        if (b->worst[i] > a->worst[i]) a->worst[i] = b->worst[i];
    }
    if (b->worst_ratio_pair > a->worst_ratio_pair) a->worst_ratio_pair = b->worst_ratio_pair;
    if (b->worst_ratio_filt > a->worst_ratio_filt) a->worst_ratio_filt = b->worst_ratio_filt;
}

typedef struct { int N; uint64_t target; uint64_t seed; stats_t by_kind[NUM_KINDS]; int onset_type_fixed; } job_t;

static void* worker(void* arg)
{
    job_t* j = (job_t*) arg;
    const int N = j->N, H = N / 2, T = 40;
    rng_t r = { j->seed, 0, 0.0 };
    fast_cfg fc; fast_init(&fc, N);
    fast_bufs b;
    b.win = (float*) malloc(sizeof(float) * (size_t) N); b.wspec = (float*) malloc(sizeof(float) * 2 * (size_t) N);
    b.rspec = (float*) malloc(sizeof(float) * 2 * (size_t) N); b.fre = (float*) malloc(sizeof(float) * (size_t) N);
    b.pw = (float*) malloc(sizeof(float) * (size_t) N); b.acf = (float*) malloc(sizeof(float) * (size_t) N); b.cnd = (float*) malloc(sizeof(float) * (size_t) N);
    float* hops = (float*) malloc(sizeof(float) * (size_t) T * H);
    float* wkeep = (float*) malloc(sizeof(float) * 2 * (size_t) N);
    double* prev_before = (double*) malloc(sizeof(double) * (size_t) H);
    uint64_t done = 0;
    while (done < j->target) {
        const int kind = make_signal(&r, T, N, hops);
        const int order = rng_int(&r, 0, 3), otype = j->onset_type_fixed >= 0 ? j->onset_type_fixed : rng_int(&r, 0, 3), owin = rng_int(&r, 1, 22);
        const float sens = (float) rng_range(&r, 0, 2);
        const float gains[4] = { 1.0f, 1.0f, 0.5f, 3.0f };
        const float gain = gains[rng_int(&r, 0, 4)];
        fxo_channel* ce = fxo_create(N, 48000.0, order);
        fxo_channel* cf = fxo_create(N, 48000.0, order);
        fxo_channel* both[2] = { ce, cf };
        for (int k = 0; k < 2; k++) {
            fxo_set_gain(both[k], gain); fxo_set_onset_type(both[k], otype); fxo_set_onset_sensitivity(both[k], sens); fxo_set_onset_window(both[k], owin);
        }
        stats_t* st = &j->by_kind[kind];
        for (int t = 0; t < T; t++) {
            for (int k = 0; k < 2; k++) {
                fxo_channel* c = both[k];
                for (int s = N - 1; s >= H; s--) c->overlap[s - H] = c->overlap[s];
                for (int i = 0; i < H; i++) c->overlap[H + i] = hops[t * H + i] * c->gain;
            }
            frame_view ve, vf;
            memcpy(prev_before, cf->prev_mag, sizeof(double) * (size_t) H);
            exact_frame(ce, &ve, wkeep);
            fast_frame(cf, &fc, &b, &vf);
            st->frames++;
            int violated = 0, slot_bad[FXO_NUM_FEATURES] = { 0 };
            for (int i = 0; i < FXO_NUM_FEATURES; i++) {
                const float g = vf.raw[i], w = ve.raw[i];
                if (g == w || (isnan(g) && isnan(w))) continue;
                if (i == FXO_ONSET) { st->beyond[i]++; violated = 1; slot_bad[i] = 1; continue; }
                if (isnan(g) || isnan(w) || isinf(g) || isinf(w) || w == 0.0f) { st->special[i]++; violated = 1; slot_bad[i] = 1; continue; }
                const double e = fabs((double) g - (double) w) / fabs((double) w);
                if (e > st->worst[i]) st->worst[i] = e;
                if (e > 1e-5) { st->beyond[i]++; violated = 1; slot_bad[i] = 1; }
            }
            if (ve.lag != vf.lag) st->lag_flips++;
            if (ve.num_peaks != vf.num_peaks || memcmp(ve.peaks, vf.peaks, sizeof(int) * (size_t) ve.num_peaks)) st->peak_flips++;
            {
                const double eg = 0.01 * (double) ve.log_rms;
                int flips = 0;
                for (int m = 0; m < H; m++) {
                    const double a = (double) ve.wspec[2 * m] * ve.wspec[2 * m], c2 = (double) vf.wspec[2 * m] * vf.wspec[2 * m];
                    flips += (a > eg) != (c2 > eg);
                }
                if (flips) st->gate_flip_frames++;
            }
            if ((ve.mag_sum_w > 0.05) != (vf.mag_sum_w > 0.05) || (ve.max_e_w > 0.0001) != (vf.max_e_w > 0.0001) || (ve.mag_sum_r < 0.005) != (vf.mag_sum_r < 0.005)) st->level_flips++;
            if (violated) st->violating++;
            if (slot_bad[FXO_F0] || slot_bad[FXO_HER] || slot_bad[FXO_OER] || slot_bad[FXO_INHARM]) st->violating_harm++;
            /* transform error against the guard's scale (for the choice of GUARD_K) */
            {
                const double eps32 = 5.96e-8;
                const double sp = eps32 * (double) vf.top_pair;
                double worst = 0.0;
                for (int m = 0; m < 2 * N; m += 2) { const double d = fabs((double) vf.wspec[m] - (double) ve.wspec[m]); if (d > worst) worst = d; }
                for (int m = 0; m < H; m++) { const double d = fabs((double) b.rspec[2 * m] - (double) ce->spec[2 * m]); if (d > worst) worst = d; }
                if (sp > 0.0 && worst / sp > st->worst_ratio_pair) st->worst_ratio_pair = worst / sp;
                const double sf = eps32 * (double) vf.top_filt;
                double wf = 0.0;
                for (int k = 0; k <= H; k++) { const double d = fabs((double) b.fre[k] - (double) ce->fspec[2 * k]); if (d > wf) wf = d; }
                if (sf > 0.0 && wf / sf > st->worst_ratio_filt) st->worst_ratio_filt = wf / sf;
            }
            const int taint = guard_frame(cf, &b, &vf, prev_before);
            if (taint) { st->tainted++; for (int q = 0; q < 8; q++) if (taint & (1 << q)) st->tainted_by[q]++; }
            else if (violated) { st->missed++; for (int i = 0; i < FXO_NUM_FEATURES; i++) st->missed_slot[i] += (uint64_t) slot_bad[i]; }
            /* a tainted frame is replayed: the fast channel continues from the exact side's state (flux state, histories) */
            if (taint) {
                memcpy(cf->prev_mag, ce->prev_mag, sizeof(double) * (size_t) H);
                cf->feat = ce->feat; cf->feat_harm = ce->feat_harm; cf->onset = ce->onset;
            }
        }
        fxo_destroy(ce); fxo_destroy(cf);
        done += (uint64_t) T;
    }
    free(b.win); free(b.wspec); free(b.rspec); free(b.fre); free(b.pw); free(b.acf); free(b.cnd); free(hops); free(wkeep); free(prev_before);
    fast_free(&fc);
    return NULL;
}

static const char* SLOT_NAMES[FXO_NUM_FEATURES] = { "onset", "rms", "f0", "centroid", "spread", "flatness", "ler", "flux", "slope", "her", "oer", "inharm" };
static const char* TAINT_NAMES[8] = { "lag", "flatness gate", "peaks", "level gates", "flux conditioning", "flatness conditioning", "slope conditioning", "product range" };

static void print_stats(const char* title, const stats_t* s)
{
    if (!s->frames) return;
    printf("  %-24s %12" PRIu64 " frames | violating %10" PRIu64 " (%.4f %%) | guard taints %10" PRIu64 " (%.4f %%), misses %" PRIu64 "\n", title, s->frames,
           s->violating, 100.0 * (double) s->violating / (double) s->frames, s->tainted, 100.0 * (double) s->tainted / (double) s->frames, s->missed);
    printf("      frames with a harmonic-analyser slot (f0, HER, OER, inharmonicity) outside the bar: %" PRIu64 " (%.4f %%)\n", s->violating_harm, 100.0 * (double) s->violating_harm / (double) s->frames);
    printf("      flips: lag %" PRIu64 ", peak list %" PRIu64 ", flatness-gate frames %" PRIu64 ", level gates %" PRIu64 ", onset %" PRIu64 "\n",
           s->lag_flips, s->peak_flips, s->gate_flip_frames, s->level_flips, s->beyond[FXO_ONSET]);
    printf("      slot       beyond 1e-5   NaN/inf/0 mismatch   worst finite rel err   missed by the guard\n");
    for (int i = 1; i < FXO_NUM_FEATURES; i++)
        if (s->beyond[i] || s->special[i] || s->worst[i] > 0.0)
            printf("      %-9s %12" PRIu64 " %12" PRIu64 "          %10.3e          %10" PRIu64 "\n", SLOT_NAMES[i], s->beyond[i], s->special[i], s->worst[i], s->missed_slot[i]);
    printf("      guard taints by cause:");
    for (int q = 0; q < 8; q++) if (s->tainted_by[q]) printf(" %s %.4f %%;", TAINT_NAMES[q], 100.0 * (double) s->tainted_by[q] / (double) s->frames);
    printf("\n      largest transform error / (eps32 x largest bin): pair %.2f, low-passed frame %.2f\n", s->worst_ratio_pair, s->worst_ratio_filt);
}

int main(int argc, char** argv)
{
    const double per_size = argc > 1 ? atof(argv[1]) : 1e5;
    const int threads = argc > 2 ? atoi(argv[2]) : 4;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 0) : 1;
    int sizes[8], ns = 0;
    if (argc > 4) { char* p = argv[4]; while (*p && ns < 8) { sizes[ns++] = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; } }
    else { sizes[0] = 1024; sizes[1] = 2048; sizes[2] = 4096; ns = 3; }
    if (argc > 5) GUARD_K = atof(argv[5]);
    const int otype_fixed = argc > 6 ? atoi(argv[6]) : -1;
    if (argc > 7) VARIANT = atoi(argv[7]);
    if (argc > 8) ONLY_KIND = atoi(argv[8]);
    printf("fastdag: %.3g frames per window size, %d threads, seed %" PRIu64 ", guard K = %.2f, onset type %s\nvariant %d: %s\n", per_size, threads, seed, GUARD_K,
           otype_fixed < 0 ? "random" : (otype_fixed == 1 ? "amplitude (the reference's default)" : "fixed"), VARIANT,
           VARIANT == 2 ? "NULL TEST: the reference's own transform DAG and products through this tool's own driver (must give 0 violating frames)"
           : VARIANT == 1 ? "the reference's own transform DAG with fused (FMA) twiddle products, nothing else changed"
                        : "FMA twiddle products + windowed || raw frame in one complex transform + half-length real transforms for the pitch pair");
    for (int si = 0; si < ns; si++) {
        const int N = sizes[si];
        const time_t t0 = time(NULL);
        pthread_t th[64]; job_t* jobs = (job_t*) calloc((size_t) threads, sizeof(job_t));
        for (int i = 0; i < threads; i++) {
            jobs[i].N = N; jobs[i].target = (uint64_t) (per_size / threads) + 1; jobs[i].seed = seed * 1000003ull + (uint64_t) N * 7919ull + (uint64_t) i; jobs[i].onset_type_fixed = otype_fixed;
            pthread_create(&th[i], NULL, worker, &jobs[i]);
        }
        stats_t by_kind[NUM_KINDS], total;
        memset(by_kind, 0, sizeof by_kind); memset(&total, 0, sizeof total);
        for (int i = 0; i < threads; i++) { pthread_join(th[i], NULL); for (int k = 0; k < NUM_KINDS; k++) stats_merge(&by_kind[k], &jobs[i].by_kind[k]); }
        for (int k = 0; k < NUM_KINDS; k++) stats_merge(&total, &by_kind[k]);
        printf("\n== window %d points (%ld s) ==\n", N, (long) (time(NULL) - t0));
        print_stats("ALL KINDS", &total);
        for (int k = 0; k < NUM_KINDS; k++) print_stats(KIND_NAMES[k], &by_kind[k]);
        fflush(stdout);
        free(jobs);
    }
    return 0;
}
