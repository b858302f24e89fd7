/*
 * fx_oracle.c -- CPU oracle: a plain-C restatement of the reference's per-frame
 * feature path  RealTimeAnalyser -> SpectralCharacteristics /
 * HarmonicCharacteristics / PitchAnalyser.
 *
 * TEST INFRASTRUCTURE ONLY (see fx_oracle.h).  PARITY UNPINNED (see fx_oracle.h):
 * the reference holds no golden vectors and cannot be built here (JUCE 4.2.3 is
 * absent), so this file is checked only against mathematics (numpy fp64 DFT,
 * closed forms) and an independent numpy restatement (oracle/fx_numpy.py).
 *
 * Every function cites the reference lines it follows ("ref:", relative to
 * /root/reference/Source/).  Third-party arithmetic that is NOT in
 * /root/reference -- JUCE 4.2.3 juce_audio_basics (AudioSampleBuffer, FFT),
 * pinned by Feature-Extractor.jucer:5 jucerVersion="4.2.3" -- is restated from
 * its published algorithm and marked "JUCE:".
 *
 * log10 of a float argument (ref RealTimeAnalyser.h:149,208; SpectralCharacteristics.h:134): the
 * original toolchains pick a float overload whose last bit differs between libms (and from
 * glibc's log10f).  logRMS feeds the discrete test `binMagnitude > 0.01*logRMS`, so a 1-ulp
 * difference can flip a bin.  The oracle therefore evaluates these as (float) log10((double) x) --
 * the correctly rounded float, which every good log10f approximates -- and the GPU does the same.
 *
 * Floating-point discipline: the original toolchains (VS2015 / Xcode, x86-64
 * SSE2) evaluate float expressions in float and double expressions in double
 * with no FMA contraction; unqualified log10/exp on a float argument resolve to
 * the float overloads.  Build this file with
 * -ffp-contract=off and without -ffast-math.
 */
#include "fx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define FXO_MAX_HIST 64

/* ------------------------------------------------------------------------- */
/* ValueHistory, ref: RealTimeAudioAnalysis.h:40-96                           */
/* ------------------------------------------------------------------------- */
typedef struct {
    float h[FXO_MAX_HIST];
    int   len;
    int   recorded;
} vhist;

/* ref: RealTimeAudioAnalysis.h:73-81 (setHistoryLength resets recordedHistory) */
static void vh_set_length(vhist* v, int len)
{
    if (len < 0) len = 0;
    if (len > FXO_MAX_HIST) len = FXO_MAX_HIST;
    v->recorded = 0;
    v->len = len;
    for (int i = 0; i < FXO_MAX_HIST; i++) v->h[i] = 0.0f;
}

/* ref: RealTimeAudioAnalysis.h:49-57 (left-to-right float sum) */
static float vh_total(const vhist* v)
{
    float total = 0.0f;
    for (int i = 0; i < v->len; i++) total += v->h[i];
    return total;
}

/* ref: RealTimeAudioAnalysis.h:59-71 */
static void vh_insert(vhist* v, float x)
{
    if (v->len <= 0) return;
    for (int i = 0; i < v->len - 1; i++) v->h[i] = v->h[i + 1];
    v->h[v->len - 1] = x;
    if (v->recorded < v->len) v->recorded++;
}

/* ------------------------------------------------------------------------- */
/* AudioFeatures, ref: RealTimeAnalyser.h:14-92                               */
/* ------------------------------------------------------------------------- */
typedef struct { vhist f[FXO_NUM_FEATURES]; } afeatures;

/* ref: RealTimeAnalyser.h:70-74 : history 1 for Onset and Flux, else 10 */
static void af_init(afeatures* a)
{
    for (int i = 0; i < FXO_NUM_FEATURES; i++)
        vh_set_length(&a->f[i], (i == FXO_ONSET || i == FXO_FLUX) ? 1 : 10);
}
/* ref: RealTimeAnalyser.h:76-82 */
static void af_update(afeatures* a, int slot, float v) { vh_insert(&a->f[slot], v); }
/* ref: RealTimeAnalyser.h:84-88 : total / recordedHistory (float / int -> float;
 * 0/0 = NaN before the first insert, as in the reference) */
static float af_value(const afeatures* a, int slot)
{
    return vh_total(&a->f[slot]) / (float) a->f[slot].recorded;
}

/* ------------------------------------------------------------------------- */
/* JUCE: juce::FFT (juce_audio_basics/effects/juce_FFT.cpp, JUCE 4.2.x)       */
/* kiss-style mixed-radix decimation in time, radix 4 then 2, fp32.           */
/* Call sites: ref RealTimeAudioAnalysis.h:162-177.  Spec: SURVEY.md App. A.1 */
/* ------------------------------------------------------------------------- */
typedef struct { float r, i; } cpx;
typedef struct { int radix, length; } fft_factor;
typedef struct {
    int        size;
    int        inverse;
    cpx*       tw;
    fft_factor factors[32];
} fft_cfg;

static cpx c_mul(cpx a, cpx b) { cpx c = { a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r }; return c; }
static cpx c_add(cpx a, cpx b) { cpx c = { a.r + b.r, a.i + b.i }; return c; }
static cpx c_sub(cpx a, cpx b) { cpx c = { a.r - b.r, a.i - b.i }; return c; }

static void fft_cfg_init(fft_cfg* c, int size, int inverse)
{
    c->size = size;
    c->inverse = inverse;
    c->tw = (cpx*) malloc(sizeof(cpx) * (size_t) size);
    for (int i = 0; i < size; i++) {
        /* JUCE: phase in double, table entries rounded to float */
        const double phase = (inverse ? 2.0 : -2.0) * 3.14159265358979323846 * i / size;
        c->tw[i].r = (float) cos(phase);
        c->tw[i].i = (float) sin(phase);
    }
    /* JUCE: factor list, 4s first then 2 (N is a power of two here) */
    const int root = (int) sqrt((double) size);
    int divisor = 4, n = size;
    for (int i = 0; i < 32; i++) {
        while ((n % divisor) != 0) {
            if (divisor == 2)      divisor = 3;
            else if (divisor == 4) divisor = 2;
            else                   divisor += 2;
            if (divisor > root)    divisor = n;
        }
        n /= divisor;
        c->factors[i].radix = divisor;
        c->factors[i].length = n;
    }
}
static void fft_cfg_free(fft_cfg* c) { free(c->tw); c->tw = NULL; }

static void fft_butterfly2(const fft_cfg* c, cpx* data, int stride, int length)
{
    cpx* end = data + length;
    const cpx* tw = c->tw;
    for (int i = length; --i >= 0;) {
        const cpx s = c_mul(*end, *tw);
        tw += stride;
        *end = c_sub(*data, s);
        end++;
        *data = c_add(*data, s);
        data++;
    }
}

static void fft_butterfly4(const fft_cfg* c, cpx* data, int stride, int length)
{
    const int l2 = length * 2, l3 = length * 3;
    const cpx *t1 = c->tw, *t2 = c->tw, *t3 = c->tw;
    for (int i = length; --i >= 0;) {
        const cpx s0 = c_mul(data[length], *t1);
        const cpx s1 = c_mul(data[l2], *t2);
        const cpx s2 = c_mul(data[l3], *t3);
        const cpx s3 = c_add(s0, s2);
        const cpx s4 = c_sub(s0, s2);
        const cpx s5 = c_sub(*data, s1);
        *data = c_add(*data, s1);
        data[l2] = c_sub(*data, s3);
        t1 += stride; t2 += stride * 2; t3 += stride * 3;
        *data = c_add(*data, s3);
        if (c->inverse) {
            data[length].r = s5.r - s4.i; data[length].i = s5.i + s4.r;
            data[l3].r     = s5.r + s4.i; data[l3].i     = s5.i - s4.r;
        } else {
            data[length].r = s5.r + s4.i; data[length].i = s5.i - s4.r;
            data[l3].r     = s5.r - s4.i; data[l3].i     = s5.i + s4.r;
        }
        ++data;
    }
}

static void fft_perform(const fft_cfg* c, const cpx* in, cpx* out, int stride, const fft_factor* facs)
{
    const fft_factor f = *facs++;
    if (f.radix == 1) { *out = *in; return; }          /* only for size 1 */
    if (f.length == 1) {
        for (int j = 0; j < f.radix; j++) out[j] = in[j * stride];
    } else {
        for (int j = 0; j < f.radix; j++)
            fft_perform(c, in + j * stride, out + j * f.length, stride * f.radix, facs);
    }
    if (f.radix == 2)      fft_butterfly2(c, out, stride, f.length);
    else if (f.radix == 4) fft_butterfly4(c, out, stride, f.length);
}

void fxo_fft_complex(int n, int inverse, const float* in, float* out)
{
    fft_cfg c;
    fft_cfg_init(&c, n, inverse);
    fft_perform(&c, (const cpx*) in, (cpx*) out, 1, c.factors);
    fft_cfg_free(&c);
}

/* JUCE: FFT::performRealOnlyForwardTransform -- copy n reals into a complex
 * scratch (imag 0), transform, n interleaved (re,im) pairs out, no scaling.
 * ref: RealTimeAudioAnalysis.h:255-278 (getFrequencyData), :167-171 */
static void forward_real(const fft_cfg* fwd, const float* x, float* spec2n, cpx* scratch)
{
    for (int i = 0; i < fwd->size; i++) { scratch[i].r = x[i]; scratch[i].i = 0.0f; }
    fft_perform(fwd, scratch, (cpx*) spec2n, 1, fwd->factors);
}

/* JUCE: FFT::performRealOnlyInverseTransform -- d is n interleaved complex;
 * result planar: d[i] = re_i/n, d[i+n] = im_i/n.  ref: RealTimeAudioAnalysis.h:173-177 */
static void inverse_real(const fft_cfg* inv, float* d2n, cpx* scratch)
{
    const int n = inv->size;
    fft_perform(inv, (const cpx*) d2n, scratch, 1, inv->factors);
    const float scale = 1.0f / n;
    for (int i = 0; i < n; i++) {
        d2n[i]     = scratch[i].r * scale;
        d2n[i + n] = scratch[i].i * scale;
    }
}

void fxo_forward_real(int n, const float* x, float* spec2n)
{
    fft_cfg c; fft_cfg_init(&c, n, 0);
    cpx* scratch = (cpx*) malloc(sizeof(cpx) * (size_t) n);
    forward_real(&c, x, spec2n, scratch);
    free(scratch); fft_cfg_free(&c);
}

/* ------------------------------------------------------------------------- */
/* JUCE: AudioSampleBuffer methods on the path           */
/* ------------------------------------------------------------------------- */
/* getRMSLevel: float square, double accumulate.  ref: RealTimeAnalyser.h:148,207 */
static float buf_rms(const float* x, int n)
{
    double sum = 0.0;
    for (int i = 0; i < n; i++) { const float s = x[i]; sum += s * s; }
    return (float) sqrt(sum / n);
}
/* applyGainRamp: float gain accumulated by a float increment */
static void buf_gain_ramp(float* x, int n, float g0, float g1)
{
    if (g0 == g1) { for (int i = 0; i < n; i++) x[i] *= g0; return; }
    const float inc = (g1 - g0) / n;
    for (int i = 0; i < n; i++) { x[i] *= g0; g0 += inc; }
}
/* getMagnitude: max |x|.  ref: SpectralCharacteristics.h:153 */
static float buf_magnitude(const float* x, int n)
{
    float mn = x[0], mx = x[0];
    for (int i = 1; i < n; i++) { if (x[i] < mn) mn = x[i]; if (x[i] > mx) mx = x[i]; }
    float r = mn;
    if (-mn > r) r = -mn;
    if (mx > r)  r = mx;
    if (-mx > r) r = -mx;
    return r;
}

/* ref: RealTimeAudioAnalysis.h:141-151 (two gain ramps 0->1, 1->0) */
void fxo_bartlett(int n, float* x)
{
    buf_gain_ramp(x, n / 2, 0.0f, 1.0f);
    buf_gain_ramp(x + n / 2, n / 2, 1.0f, 0.0f);
}

/* ref: RealTimeAudioAnalysis.h:106-127.  float_Pi / m and exp(-float_Pi / m)
 * are float expressions (exp(float) -> expf). */
static const float k_float_pi = 3.14159265358979323846f;

/* log10 (float) -> float, correctly rounded (see the header comment) */
static float log10_float(float x) { return (float) log10((double) x); }
float fxo_lpf_a(void) { return k_float_pi / 2.0f; }
float fxo_lpf_b(void) { return expf(-k_float_pi / 2.0f); }

void fxo_lowpass(int n, const float* in, float* out)
{
    const float a = fxo_lpf_a();
    const float b = fxo_lpf_b();
    if (n > 0) out[0] = in[0];
    for (int s = 1; s < n; s++)
        out[s] = (a * in[s]) + (b * out[s - 1]);
}

/* ------------------------------------------------------------------------- */
/* OnsetDetector, ref: SpectralCharacteristics.h:210-312                      */
/* ------------------------------------------------------------------------- */
typedef struct {
    vhist flux_hist, amp_hist;
    int   type;
    float multiplier;
} onset_det;

static void onset_init(onset_det* o)
{
    vh_set_length(&o->flux_hist, 5);       /* :238-239 */
    vh_set_length(&o->amp_hist, 5);
    o->type = FXO_ONSET_AMPLITUDE;         /* :240 */
    o->multiplier = 1.7f;                  /* :311 */
}

/* ref: SpectralCharacteristics.h:249-306 */
static int onset_detect(const onset_det* o)
{
    const vhist* sf = &o->flux_hist;
    const vhist* am = &o->amp_hist;
    if (am->recorded == 0 || sf->recorded == 0) return 0;
    if (sf->recorded < sf->len || am->recorded < am->len) return 0;

    const float mean_sf  = vh_total(sf) / sf->recorded;
    const float mean_amp = vh_total(am) / am->recorded;

    int cand = sf->len - 1;
    if (o->type == FXO_ONSET_SPECTRAL || o->type == FXO_ONSET_COMBINATION) cand = sf->len / 2;

    const float cand_sf  = sf->h[cand];
    const float cand_amp = am->h[cand];
    if (cand_amp < 0.01f) return 0;

    for (int i = 0; i < sf->len; i++) {
        if (i == cand) continue;
        if (am->h[i] >= cand_amp && (o->type == FXO_ONSET_AMPLITUDE || o->type == FXO_ONSET_COMBINATION)) return 0;
        if (sf->h[i] >= cand_sf  && (o->type == FXO_ONSET_SPECTRAL  || o->type == FXO_ONSET_COMBINATION)) return 0;
    }
    const int on_sf  = cand_sf  > mean_sf  * o->multiplier;
    const int on_amp = cand_amp > mean_amp * o->multiplier;
    switch (o->type) {
        case FXO_ONSET_AMPLITUDE:   return on_amp;
        case FXO_ONSET_SPECTRAL:    return on_sf;
        case FXO_ONSET_COMBINATION: return on_amp && on_sf;
        default:                    return 0;
    }
}

/* ------------------------------------------------------------------------- */
/* channel state                                                              */
/* ------------------------------------------------------------------------- */
struct fxo_channel {
    int     n, m;              /* window size, numMagnitudes = n/2 */
    double  nyquist;
    float   gain;
    int     order_mode;
    int     analysers;         /* bit 0 spectral, bit 1 harmonic */
    fft_cfg fwd, inv;
    float*  overlap;           /* ref: RealTimeAudioAnalysis.h:236 overlappedAudio */
    double* prev_mag;          /* ref: SpectralCharacteristics.h:203 previousBinMagnitudes */
    afeatures feat;            /* shared AudioFeatures (or the spectral one when isolated) */
    afeatures feat_harm;       /* harmonic analyser's own AudioFeatures when isolated */
    onset_det onset;
    /* scratch */
    float*  win;  float* filt; float* spec; float* fspec; float* work; cpx* scratch;
    double* mags; double* fcs; float* normed; int* peaks;
};

fxo_channel* fxo_create(int window_size, double sample_rate, int order_mode)
{
    if (window_size < 4 || (window_size & (window_size - 1))) return NULL;
    fxo_channel* c = (fxo_channel*) calloc(1, sizeof(*c));
    c->n = window_size; c->m = window_size / 2;
    c->nyquist = sample_rate / 2.0;                 /* ref: RealTimeAudioAnalysis.h:251 */
    c->gain = 1.0f;                                 /* ref: AudioDataCollector.h:129 */
    c->order_mode = order_mode;
    c->analysers = 3;
    fft_cfg_init(&c->fwd, window_size, 0);
    fft_cfg_init(&c->inv, window_size, 1);
    const size_t n = (size_t) window_size;
    c->overlap  = (float*)  calloc(n, sizeof(float));
    c->prev_mag = (double*) calloc(n / 2, sizeof(double));
    c->win   = (float*) calloc(n, sizeof(float));
    c->filt  = (float*) calloc(n, sizeof(float));
    c->spec  = (float*) calloc(2 * n, sizeof(float));
    c->fspec = (float*) calloc(2 * n, sizeof(float));
    c->work  = (float*) calloc(2 * n, sizeof(float));
    c->scratch = (cpx*) calloc(n, sizeof(cpx));
    c->mags  = (double*) calloc(n / 2, sizeof(double));
    c->fcs   = (double*) calloc(n / 2, sizeof(double));
    c->normed = (float*) calloc(n / 2, sizeof(float));
    c->peaks = (int*) calloc(n / 2, sizeof(int));
    onset_init(&c->onset);
    fxo_reset(c);
    return c;
}

void fxo_destroy(fxo_channel* c)
{
    if (!c) return;
    fft_cfg_free(&c->fwd); fft_cfg_free(&c->inv);
    free(c->overlap); free(c->prev_mag); free(c->win); free(c->filt); free(c->spec);
    free(c->fspec); free(c->work); free(c->scratch); free(c->mags); free(c->fcs);
    free(c->normed); free(c->peaks);
    free(c);
}

void fxo_reset(fxo_channel* c)
{
    memset(c->overlap, 0, sizeof(float) * (size_t) c->n);        /* RealTimeAudioAnalysis.h:202 */
    for (int i = 0; i < c->m; i++) c->prev_mag[i] = 0.0;         /* SpectralCharacteristics.h:36-37 */
    af_init(&c->feat);
    af_init(&c->feat_harm);
    /* histories emptied; type / multiplier / window length are settings and survive */
    const int len = c->onset.flux_hist.len;
    vh_set_length(&c->onset.flux_hist, len);
    vh_set_length(&c->onset.amp_hist, len);
}

/* ref: RealTimeAnalyser.h:111-114 */
void fxo_set_sample_rate(fxo_channel* c, double sr) { c->nyquist = sr / 2.0; }
/* ref: RealTimeAnalyser.h:244-248 */
void fxo_set_onset_sensitivity(fxo_channel* c, float s) { c->onset.multiplier = 1.0f + s; }
/* ref: RealTimeAnalyser.h:250-254 (both histories reset) */
void fxo_set_onset_window(fxo_channel* c, int length)
{
    vh_set_length(&c->onset.amp_hist, length);
    vh_set_length(&c->onset.flux_hist, length);
}
/* ref: RealTimeAnalyser.h:258 */
void fxo_set_onset_type(fxo_channel* c, int t) { c->onset.type = t; }
/* ref: AudioDataCollector.h:124 */
void fxo_set_gain(fxo_channel* c, float g) { c->gain = g; }
void fxo_set_analysers(fxo_channel* c, int mask) { c->analysers = mask & 3; }

/* ------------------------------------------------------------------------- */
/* SpectralCharacteristicsAnalyser, ref: SpectralCharacteristics.h:31-206     */
/* ------------------------------------------------------------------------- */
typedef struct { float centroid, spread, flatness, ler, flux; } spectral_out;

/* ref: SpectralCharacteristics.h:100-143 (+ fillIntermediateValues :62-97) */
static spectral_out spectral_characteristics(fxo_channel* c, const float* spec2n, double rms)
{
    const int M = c->m;
    const double nyquist = c->nyquist;
    const double eps = 0.01 * rms;                                  /* :108 */
    double weighted = 0.0, var = 0.0, mag_sum = 0.0, prod = 1.0;
    double flat_sum = 0.0, flux = 0.0, lhr = 0.0, cnt = 0.0;

    /* fillIntermediateValues :62-97 */
    const double range_per_bin = nyquist / M;
    const int lower_portion = M / 5;
    for (int m = 0; m < M; m++) {
        const double fc = (double) m * range_per_bin + (range_per_bin / 2.0);
        c->fcs[m] = fc;
        const double v = (double) spec2n[2 * m];                    /* real part only :68-72 */
        const double mag = v * v;
        const double diff = fabs(mag) - fabs(c->prev_mag[m]);       /* :76 */
        const double rect = (diff + fabs(diff)) / 2.0;
        if (diff > 0.0) flux += rect;
        c->mags[m] = mag;
        mag_sum += mag;
        if (m == lower_portion) lhr = mag_sum;                      /* :86-87 */
        if (mag > eps) { flat_sum += mag; prod *= mag; cnt++; }     /* :89-94 */
        weighted += fc * mag;
    }

    const float max_flux = (M * (M + 1)) / 2.0f;                    /* :111 */
    flux /= max_flux;

    spectral_out z = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    if (!(mag_sum > 0.05)) return z;                                /* :121-123, prev NOT updated */

    lhr /= mag_sum;
    const float centroid = (float) (weighted / mag_sum);            /* :127 */
    const double inv_n = 1.0 / (cnt > 0.0 ? cnt : 1.0);             /* :129-130 */
    const float flatness = flat_sum > eps                           /* :57-60 */
        ? (float) (pow(prod, inv_n) / (inv_n * flat_sum)) : 0.0f;
    const float log_flat = (float) log10(flatness * 9.0 + 1.0);     /* :132 (double log10) */
    const float cc = centroid / (float) (nyquist / 2.0);            /* :133 */
    const float log_centroid = log10_float(cc * 9.0f + 1.0f);       /* :134 (float log10) */
    for (int i = 0; i < M; i++) {                                   /* :135-139 */
        var += pow((c->fcs[i] / nyquist) - (centroid / nyquist), 2.0) * c->mags[i];
        c->prev_mag[i] = c->mags[i];
    }
    const float max_spread = (float) ((centroid / nyquist) * (1.0 - (centroid / nyquist)));
    const float spread = (float) ((var / mag_sum) / max_spread);    /* :140-141 */
    spectral_out o = { log_centroid, spread, log_flat, (float) lhr, (float) flux };
    return o;
}

/* ref: SpectralCharacteristics.h:145-200 */
static float spectral_slope(fxo_channel* c, const float* spec2n)
{
    const int M = c->m;
    const double mean_bin = 0.5;
    double mean_e = 0.0, prod_sum = 0.0;
    double max_mag = buf_magnitude(spec2n, M);      /* :153 raw interleaved floats [0,M) */
    for (int i = 0; i < M; i++) {
        const double v = spec2n[i * 2];
        const double mag = v * v;
        c->mags[i] = mag;
        if (mag > max_mag) max_mag = mag;
    }
    if (!(max_mag > 0.0001)) return 0.0f;           /* :165-167 */
    for (int i = 0; i < M; i++) {
        const double ne = c->mags[i] / max_mag;
        mean_e += ne;
        prod_sum += (double) i * ne;
    }
    mean_e /= (double) M;
    double bin_var = 0.0, e_var = 0.0;
    for (double i = 0.0; i < M; i++) {              /* :182-188 */
        const double ni = i / (double) M;
        bin_var += (ni - mean_bin) * (ni - mean_bin);
        const double ne = c->mags[(int) i] / max_mag;
        e_var += (ne - mean_e) * (ne - mean_e);
    }
    bin_var /= (double) M;
    e_var /= (double) M;
    const double bin_std = sqrt(bin_var), e_std = sqrt(e_var);
    /* :195  (numMagnitudes - 1.0f) is a float */
    const double r = (prod_sum - (M * mean_e * mean_bin)) / (M - 1.0f) * e_std * bin_std;
    const double grad = r * (bin_std / e_std);      /* :198 */
    return (float) grad;
}

/* ------------------------------------------------------------------------- */
/* PitchAnalyser, ref: PitchAnalyser.h:17-219                                 */
/* ------------------------------------------------------------------------- */
static double estimate_pitch(const fft_cfg* inv, double nyquist, const float* spec2n,
                             float* work2n, cpx* scratch, float* lag_out)
{
    const int N = inv->size, two_n = 2 * N;
    /* getComplexConjugateMultiplication :83-108 : re*re, imag := 0 */
    for (int k = 0; k < two_n; k += 2) {
        const float re = spec2n[k];
        work2n[k] = re * re;
        work2n[k + 1] = 0.0f;
    }
    /* getAutoCorrelationFromConjugateMultiplication :110-127 */
    inverse_real(inv, work2n, scratch);
    for (int s = 0; s < two_n; s++) work2n[s] = work2n[s] * work2n[s] * s;
    /* getCumulativeNormalisedDifference... :129-159 (in place: cnd[s] only needs ac[s]) */
    {
        float sum = 0.0f;
        work2n[0] = 1.0f;
        for (int s = 1; s < two_n; s++) {
            const float v = work2n[s];
            sum += v;
            work2n[s] = (sum != 0.0f) ? v / sum : 0.0f;
        }
    }
    /* getLagEstimateFromCumulativeDifference :161-190 */
    const float* cnd = work2n;
    const float threshold = 0.01f;
    float gmin_idx = -1.0f, gmin = 100.0f, lag = -1.0f;
    for (int s = 2; s < N; s++) {
        if (cnd[s] < gmin) { gmin_idx = (float) s; gmin = cnd[s]; }
        if (cnd[s] < threshold) {
            while (s + 1 < N && cnd[s + 1] < cnd[s]) s++;
            /* getInterpolatedValley... :192-217 : leftNeighbour == lagEstimate for
             * every lag >= 1, so only the first branch (:201-203) is reachable */
            const int right = s + ((s < N + 1) ? 1 : 0);
            lag = (cnd[s] <= cnd[right]) ? (float) s : (float) right;
            break;
        }
    }
    const float lag_est = (lag == -1.0f) ? gmin_idx : lag;          /* :188-189 */
    if (lag_out) *lag_out = lag_est;
    return (nyquist * 2.0f) / lag_est;                              /* :57 */
}

double fxo_estimate_pitch(int n, double nyquist, const float* spec2n, float* cnd2n, float* lag_out)
{
    fft_cfg inv; fft_cfg_init(&inv, n, 1);
    float* work = (float*) malloc(sizeof(float) * 2 * (size_t) n);
    cpx* scratch = (cpx*) malloc(sizeof(cpx) * (size_t) n);
    const double f0 = estimate_pitch(&inv, nyquist, spec2n, work, scratch, lag_out);
    if (cnd2n) memcpy(cnd2n, work, sizeof(float) * 2 * (size_t) n);
    free(work); free(scratch); fft_cfg_free(&inv);
    return f0;
}

/* ------------------------------------------------------------------------- */
/* HarmonicCharacteristicsAnalyser, ref: HarmonicCharacteristics.h:40-261     */
/* ------------------------------------------------------------------------- */
static int bin_for_frequency(double f, double range_per_bin) { return (int) floor(f / range_per_bin); } /* :246-249 */

/* :251-259 */
static double frequency_ratio(double f1, double f2)
{
    if (f1 == f2) return 1.0;
    const double higher = f1 > f2 ? f1 : f2;
    const double lower  = higher == f1 ? f2 : f1;
    return higher / lower;
}

/* :200-210 */
static double max_bin_in_neighbourhood(int centre, int range, const float* normed, int num_bins)
{
    const int start = centre - range >= 0 ? centre - range : 0;
    const int end   = centre + range < num_bins ? centre + range : num_bins;
    double mx = normed[centre];
    for (int b = start; b < end; b++)
        if (normed[b] > mx) mx = normed[b];
    return mx;
}

/* :127-145 */
static int bin_is_peak(int bin, const double* mags, int M, double mean)
{
    const double mag = mags[bin];
    if (mag <= mean) return 0;
    const int left  = bin < 2 ? 2 - bin : 0;
    const int right = bin >= M - 2 ? 2 - ((M - 1) - bin) : 0;
    for (int nb = bin - (2 - left); nb < bin + (2 - right); nb++)
        if (nb != bin && mags[nb] > mag) return 0;
    return 1;
}

typedef struct { float her, oer, inharm; } harmonic_out;

/* ref: HarmonicCharacteristics.h:46-106 */
static harmonic_out harmonic_characteristics(fxo_channel* c, const float* spec2n, double f0)
{
    const int M = c->m;
    double mag_sum = 0.0, max_mag = 0.0;
    for (int i = 0; i < M; i++) {                                   /* :61-69 */
        const double v = (double) spec2n[i * 2];
        const double mag = v * v;
        c->mags[i] = mag;
        mag_sum += mag;
        if (mag > max_mag) max_mag = mag;
    }
    double sum_normed = 0.0;
    for (int b = 0; b < M; b++) {                                   /* :73-78 */
        const double nm = c->mags[b] / max_mag;
        c->normed[b] = (float) nm;
        sum_normed += nm;
    }
    const double mean_mag = mag_sum / (double) M;                   /* :86 */
    harmonic_out z = { 0.0f, 0.0f, 0.0f };
    if (mag_sum < 0.005) return z;                                  /* :88-89 */

    int num_peaks = 0;                                              /* fillPeakBins :115-125 */
    for (int b = 0; b < M; b++)
        if (bin_is_peak(b, c->mags, M, mean_mag)) c->peaks[num_peaks++] = b;

    const double range_per_bin = c->nyquist / (double) M;           /* :93 */

    /* calculateHarmonicEnergyCharacteristics(normed, f0, rpb, sumNormed, 15.0, 3.0) :147-198 */
    double score = 0.0, even_e = 0.0, odd_e = 0.0;
    for (double lower = 1.0; lower < 15.0 + 1.0; ++lower) {
        const double lf = f0 / pow(2.0, lower);
        const int lb = bin_for_frequency(lf, range_per_bin);
        if (lb == bin_for_frequency(f0, range_per_bin)) continue;
        score += max_bin_in_neighbourhood(lb, 2, c->normed, M);
    }
    for (double h = 1.0; h < 3.0 + 1.0; h++) {
        const double hf = f0 * h;
        const int hb = bin_for_frequency(hf, range_per_bin);
        if (hb >= M) break;
        const double bm = max_bin_in_neighbourhood(hb, 2, c->normed, M);
        if ((int) h % 2 == 0) even_e += bm; else odd_e += bm;
        score += bm;
    }
    double her = score / sum_normed;
    if (her > 1.0) her = 1.0;
    if (her < 0.0) her = 0.0;
    double oer = 1.0;
    if (odd_e > 0.0) oer = even_e / odd_e;
    if (oer > 1.0) oer = 1.0;
    if (oer < 0.0) oer = 0.0;
    const double her_d = (double) (float) her;                      /* struct holds floats :197, :95-96 */
    const double oer_d = (double) (float) oer;

    /* calculateInharmonicity :212-244 */
    double inharm = 0.0;
    if (f0 > 0.0) {                                                 /* :98 */
        const int f0_bin = bin_for_frequency(f0, range_per_bin);
        for (int p = 0; p < num_peaks; p++) {
            const int bin = c->peaks[p];
            if (f0_bin == bin) continue;
            double start_f = bin * range_per_bin;
            if (start_f == 0.0) start_f = range_per_bin * 0.5;
            const double end_f = (double) (bin + 1) * range_per_bin;
            const double rs = frequency_ratio(start_f, f0);
            const double re = frequency_ratio(end_f, f0);
            if (floor(rs) != floor(re)) continue;
            const double r = rs < re ? rs : re;
            const double prop = r - floor(r);
            inharm += prop * (c->mags[bin] / mag_sum);
        }
    }
    harmonic_out o;
    o.her    = (float) log10(her_d * 9.0 + 1.0);                    /* :101-105 */
    o.inharm = (float) log10(inharm * 9.0 + 1.0);
    o.oer    = (float) log10(oer_d * 9.0 + 1.0);
    return o;
}

/* ------------------------------------------------------------------------- */
/* per-frame drivers, ref: RealTimeAnalyser.h:141-177 and :201-242            */
/* ------------------------------------------------------------------------- */
typedef struct { float log_rms, centroid, spread, flatness, ler, flux, slope; } spec_frame;
typedef struct { float log_rms, f0_feature, her, oer, inharm; } harm_frame;

/* ref: RealTimeAnalyser.h:206-224 (everything except the feature writes) */
static spec_frame spectral_compute(fxo_channel* c)
{
    spec_frame r;
    const float rms = buf_rms(c->overlap, c->n);                    /* :207 un-windowed */
    r.log_rms = log10_float(rms * 9.0f + 1.0f);                     /* :208 */
    memcpy(c->win, c->overlap, sizeof(float) * (size_t) c->n);      /* :206 copy */
    fxo_bartlett(c->n, c->win);                                     /* :212 */
    forward_real(&c->fwd, c->win, c->spec, c->scratch);             /* :215 */
    const spectral_out s = spectral_characteristics(c, c->spec, r.log_rms);  /* :218 */
    r.centroid = s.centroid; r.spread = s.spread; r.flatness = s.flatness;
    r.ler = s.ler; r.flux = s.flux;
    r.slope = spectral_slope(c, c->spec);                           /* :224 */
    return r;
}

/* ref: RealTimeAnalyser.h:147-172 (everything except the feature writes) */
static harm_frame harmonic_compute(fxo_channel* c)
{
    harm_frame r;
    const float rms = buf_rms(c->overlap, c->n);                    /* :148 */
    r.log_rms = log10_float(rms * 9.0f + 1.0f);                     /* :149 */
    fxo_lowpass(c->n, c->overlap, c->filt);                         /* :152-154 */
    fxo_bartlett(c->n, c->filt);                                    /* :157 */
    forward_real(&c->fwd, c->filt, c->fspec, c->scratch);           /* :160 */
    forward_real(&c->fwd, c->overlap, c->spec, c->scratch);         /* :161 raw, un-windowed */
    const double f0 = estimate_pitch(&c->inv, c->nyquist, c->fspec, c->work, c->scratch, NULL); /* :164 */
    r.f0_feature = (float) (f0 / 5000.0);                           /* :165-166 */
    const harmonic_out h = harmonic_characteristics(c, c->spec, f0);/* :169 */
    r.her = h.her;
    r.oer = h.her;                                                  /* :171 OER slot receives HER */
    r.inharm = h.inharm;
    return r;
}

/* feature writes of the spectral thread, ref: RealTimeAnalyser.h:209,219-226,236-242 */
static float spectral_writes(fxo_channel* c, afeatures* f, const spec_frame* s)
{
    af_update(f, FXO_CENTROID, s->centroid);
    af_update(f, FXO_FLATNESS, s->flatness);
    af_update(f, FXO_LER,      s->ler);
    af_update(f, FXO_SPREAD,   s->spread);
    af_update(f, FXO_FLUX,     s->flux);
    af_update(f, FXO_SLOPE,    s->slope);
    /* detectOnset :236-242 */
    vh_insert(&c->onset.flux_hist, af_value(f, FXO_FLUX));
    vh_insert(&c->onset.amp_hist,  af_value(f, FXO_RMS));
    const float onset = onset_detect(&c->onset) ? 1.0f : 0.0f;
    af_update(f, FXO_ONSET, onset);
    return onset;
}

/* feature writes of the harmonic thread, ref: RealTimeAnalyser.h:150,166,170-172 */
static void harmonic_writes(afeatures* f, const harm_frame* h)
{
    af_update(f, FXO_F0,     h->f0_feature);
    af_update(f, FXO_HER,    h->her);
    af_update(f, FXO_OER,    h->oer);
    af_update(f, FXO_INHARM, h->inharm);
}

static void run_frame(fxo_channel* c, float* raw12, float* smoothed12)
{
    const int do_spec = c->analysers & 1, do_harm = c->analysers & 2;
    spec_frame s; harm_frame h;
    memset(&s, 0, sizeof s); memset(&h, 0, sizeof h);
    if (do_spec) s = spectral_compute(c);
    if (do_harm) h = harmonic_compute(c);
    afeatures* fs = &c->feat;
    afeatures* fh = (c->order_mode == FXO_ORDER_ISOLATED) ? &c->feat_harm : &c->feat;
    float onset = 0.0f;
    if (c->order_mode == FXO_ORDER_HARMONIC_THEN_SPECTRAL) {
        if (do_harm) { af_update(fh, FXO_RMS, h.log_rms);  harmonic_writes(fh, &h); }
        if (do_spec) { af_update(fs, FXO_RMS, s.log_rms);  onset = spectral_writes(c, fs, &s); }
    } else {
        if (do_spec) { af_update(fs, FXO_RMS, s.log_rms);  onset = spectral_writes(c, fs, &s); }
        if (do_harm) { af_update(fh, FXO_RMS, h.log_rms);  harmonic_writes(fh, &h); }
    }
    if (raw12) {
        raw12[FXO_ONSET] = onset;       raw12[FXO_RMS] = do_spec ? s.log_rms : h.log_rms;   raw12[FXO_F0] = h.f0_feature;
        raw12[FXO_CENTROID] = s.centroid; raw12[FXO_SPREAD] = s.spread; raw12[FXO_FLATNESS] = s.flatness;
        raw12[FXO_LER] = s.ler;         raw12[FXO_FLUX] = s.flux;     raw12[FXO_SLOPE] = s.slope;
        raw12[FXO_HER] = h.her;         raw12[FXO_OER] = h.oer;       raw12[FXO_INHARM] = h.inharm;
    }
    if (smoothed12) {
        for (int i = 0; i < FXO_NUM_FEATURES; i++) {
            const int harm_slot = (i == FXO_F0 || i == FXO_HER || i == FXO_OER || i == FXO_INHARM);
            /* with one analyser only, the RMS slot lives in that analyser's AudioFeatures */
            const afeatures* src = harm_slot ? fh : fs;
            if (i == FXO_RMS && !do_spec) src = fh;
            smoothed12[i] = af_value(src, i);
        }
    }
}

/* ref: RealTimeAudioAnalysis.h:205-219 + AudioDataCollector.h:88 (gain) */
void fxo_push_hop(fxo_channel* c, const float* hop, float* raw12, float* smoothed12)
{
    const int half = c->n / 2;
    for (int s = c->n - 1; s >= half; s--) c->overlap[s - half] = c->overlap[s];
    for (int i = 0; i < half; i++) c->overlap[half + i] = hop[i] * c->gain;
    run_frame(c, raw12, smoothed12);
}

void fxo_process_frame(fxo_channel* c, const float* frame, float* raw12, float* smoothed12)
{
    memcpy(c->overlap, frame, sizeof(float) * (size_t) c->n);
    run_frame(c, raw12, smoothed12);
}

void fxo_process_frames(fxo_channel* c, const float* frames, int T, float* raw, float* smoothed)
{
    for (int t = 0; t < T; t++)
        fxo_process_frame(c, frames + (size_t) t * c->n,
                          raw ? raw + (size_t) t * FXO_NUM_FEATURES : NULL,
                          smoothed ? smoothed + (size_t) t * FXO_NUM_FEATURES : NULL);
}

void fxo_push_hops(fxo_channel* c, const float* hops, int T, float* raw, float* smoothed)
{
    for (int t = 0; t < T; t++)
        fxo_push_hop(c, hops + (size_t) t * (c->n / 2),
                     raw ? raw + (size_t) t * FXO_NUM_FEATURES : NULL,
                     smoothed ? smoothed + (size_t) t * FXO_NUM_FEATURES : NULL);
}

/* ------------------------------------------------------------------------- */
/* batch driver for the CPU baseline                                          */
/* ------------------------------------------------------------------------- */
typedef struct {
    int window, order, c0, c1, T;
    double sr;
    const float* frames;
    float *raw, *smoothed;
    const fxo_settings* settings;   /* NULL: defaults, `frames` holds pre-assembled windows; else `frames` holds hops */
} batch_job;

static void* batch_worker(void* arg)
{
    batch_job* j = (batch_job*) arg;
    for (int c = j->c0; c < j->c1; c++) {
        fxo_channel* ch = fxo_create(j->window, j->sr, j->order);
        if (!ch) return NULL;
        const size_t o = (size_t) c * j->T;
        float* raw = j->raw ? j->raw + o * FXO_NUM_FEATURES : NULL;
        float* sm = j->smoothed ? j->smoothed + o * FXO_NUM_FEATURES : NULL;
        if (j->settings) {
            fxo_set_gain(ch, j->settings->gain);
            fxo_set_onset_type(ch, j->settings->onset_type);
            fxo_set_onset_sensitivity(ch, j->settings->onset_sensitivity);
            fxo_set_onset_window(ch, j->settings->onset_window);
            fxo_set_analysers(ch, j->settings->analysers);
            fxo_push_hops(ch, j->frames + o * (size_t) (j->window / 2), j->T, raw, sm);
        } else {
            fxo_process_frames(ch, j->frames + o * j->window, j->T, raw, sm);
        }
        fxo_destroy(ch);
    }
    return NULL;
}

static int run_batch(int window_size, double sample_rate, int order_mode, const float* data, const fxo_settings* settings,
                     int C, int T, float* raw, float* smoothed, int threads);

int fxo_batch_frames(int window_size, double sample_rate, int order_mode, const float* frames,
                     int C, int T, float* raw, float* smoothed, int threads)
{
    return run_batch(window_size, sample_rate, order_mode, frames, NULL, C, T, raw, smoothed, threads);
}

int fxo_batch_hops(int window_size, double sample_rate, int order_mode, const fxo_settings* settings, const float* hops,
                   int C, int T, float* raw, float* smoothed, int threads)
{
    if (!settings) return -1;
    return run_batch(window_size, sample_rate, order_mode, hops, settings, C, T, raw, smoothed, threads);
}

static int run_batch(int window_size, double sample_rate, int order_mode, const float* frames, const fxo_settings* settings,
                     int C, int T, float* raw, float* smoothed, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > C) threads = C;
    if (C <= 0 || T <= 0) return 0;
    pthread_t* th = (pthread_t*) malloc(sizeof(pthread_t) * (size_t) threads);
    batch_job* jobs = (batch_job*) malloc(sizeof(batch_job) * (size_t) threads);
    int started = 0, rc = 0;
    for (int i = 0; i < threads; i++) {
        batch_job j = { window_size, order_mode, (int) ((long long) C * i / threads),
                        (int) ((long long) C * (i + 1) / threads), T, sample_rate, frames, raw, smoothed, settings };
        jobs[i] = j;
        if (pthread_create(&th[i], NULL, batch_worker, &jobs[i]) != 0) { rc = -1; break; }
        started++;
    }
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* OSC sink wire format, ref: OSCFeatureAnalysisOutput.h:107                  */
/* OSCSender::send(address, 12 floats): OSC 1.0 message = address string      */
/* (NUL-terminated, padded to 4), type tags ",ffffffffffff" (padded to 4),    */
/* 12 big-endian float32 in the order onset,rms,f0,centroid,slope,spread,     */
/* flatness,ler,flux,her,oer,inharm.                                          */
/* ------------------------------------------------------------------------- */
int fxo_osc_message(const char* address, const float* v, unsigned char* out, int cap)
{
    static const int order[12] = { FXO_ONSET, FXO_RMS, FXO_F0, FXO_CENTROID, FXO_SLOPE, FXO_SPREAD,
                                   FXO_FLATNESS, FXO_LER, FXO_FLUX, FXO_HER, FXO_OER, FXO_INHARM };
    const int alen = (int) strlen(address);
    const int apad = (alen + 4) & ~3;
    const int tpad = 16;                      /* ",ffffffffffff" = 13 chars + NUL -> 16 */
    const int total = apad + tpad + 48;
    if (total > cap) return -1;
    memset(out, 0, (size_t) total);
    memcpy(out, address, (size_t) alen);
    memcpy(out + apad, ",ffffffffffff", 13);
    unsigned char* p = out + apad + tpad;
    for (int i = 0; i < 12; i++) {
        unsigned int bits;
        memcpy(&bits, &v[order[i]], 4);
        p[0] = (unsigned char) (bits >> 24); p[1] = (unsigned char) (bits >> 16);
        p[2] = (unsigned char) (bits >> 8);  p[3] = (unsigned char) bits;
        p += 4;
    }
    return total;
}
