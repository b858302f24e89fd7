/*
 * fx_oracle.h -- CPU oracle for the RealTimeAnalyser hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path is the HIP library declared in
 * include/fx.h and it never calls into this file.
 *
 * PARITY UNPINNED: the reference (SeanSoraghan/Feature-Extractor) has no
 * tests, golden vectors or fixtures, and its hot path cannot be compiled in
 * this image because it depends on JUCE 4.2.3 (Feature-Extractor.jucer:5),
 * which is not vendored under /root/reference and is not installed.  The
 * JUCE arithmetic (FFT, getRMSLevel, applyGainRamp, getMagnitude) is
 * restated here from its published algorithm (see fx_oracle.c headers).
 *
 * All "ref:" citations are relative to /root/reference/Source/.
 */
#ifndef FX_ORACLE_H
#define FX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Feature slots, ref: RealTimeAnalyser.h:17-32 (AudioFeatures::eAudioFeature). */
enum {
    FXO_ONSET = 0, FXO_RMS, FXO_F0, FXO_CENTROID, FXO_SPREAD, FXO_FLATNESS,
    FXO_LER, FXO_FLUX, FXO_SLOPE, FXO_HER, FXO_OER, FXO_INHARM, FXO_NUM_FEATURES
};

/* Onset types, ref: SpectralCharacteristics.h:213-219. */
enum { FXO_ONSET_SPECTRAL = 0, FXO_ONSET_AMPLITUDE = 1, FXO_ONSET_COMBINATION = 2 };

/* How the two analyser threads' writes to the one shared AudioFeatures
 * object (ref: AnalyserTrackController.h:20-21,201) are serialised per hop.
 * The reference leaves this to a data race; the oracle fixes an order. */
enum {
    FXO_ORDER_SPECTRAL_THEN_HARMONIC = 0, /* SURVEY.md 8(a) canonical order */
    FXO_ORDER_HARMONIC_THEN_SPECTRAL = 1,
    FXO_ORDER_ISOLATED = 2               /* one AudioFeatures per analyser */
};

typedef struct fxo_channel fxo_channel;

/* One analysed input channel = one AnalyserTrackController's analysis state
 * (ref: AnalyserTrackController.h:199-206): overlapper, both analysers, the
 * shared AudioFeatures, flux history, onset detector. */
fxo_channel* fxo_create(int window_size, double sample_rate, int order_mode);
void fxo_destroy(fxo_channel*);
void fxo_reset(fxo_channel*);

/* ref: RealTimeAnalyser.h:111-114, :244-258; AudioDataCollector.h:124 */
void fxo_set_sample_rate(fxo_channel*, double sample_rate);
void fxo_set_onset_sensitivity(fxo_channel*, float s);
void fxo_set_onset_window(fxo_channel*, int length);
void fxo_set_onset_type(fxo_channel*, int type);
void fxo_set_gain(fxo_channel*, float gain);
/* Which of the two analyser threads exist (ref AnalyserTrackController.h:20-21 constructs both): bit 0 =
 * RealTimeSpectralAnalyser, bit 1 = RealTimeHarmonicAnalyser.  Slots an absent analyser would write stay
 * at their initial state (raw 0; getValue = 0/0 = NaN, as AudioFeatures::getValue returns before any insert). */
void fxo_set_analysers(fxo_channel*, int mask);

/* Push one hop of window_size/2 samples (ref: RealTimeAudioAnalysis.h:205-219)
 * and run one spectral + one harmonic frame.  raw12 = the 12 values passed to
 * updateFeature this hop; smoothed12 = getValue() of every slot afterwards.
 * Either output may be NULL. */
void fxo_push_hop(fxo_channel*, const float* hop, float* raw12, float* smoothed12);

/* Same, but the caller supplies the already assembled window (N samples,
 * gain NOT applied again).  The overlap buffer is replaced by the frame. */
void fxo_process_frame(fxo_channel*, const float* frame, float* raw12, float* smoothed12);

/* Batch helpers used by tests and by bench.py's cpu_baseline leg:
 * frames[T][N] (pre-assembled) or hops[T][N/2]; raw[T][12]; smoothed[T][12]. */
void fxo_process_frames(fxo_channel*, const float* frames, int T, float* raw, float* smoothed);
void fxo_push_hops(fxo_channel*, const float* hops, int T, float* raw, float* smoothed);

/* Many channels at once, one fresh fxo_channel per channel, `threads` worker threads over disjoint
 * channel blocks (the reference runs one thread pair per channel, AnalyserTrackController.h:184-185).
 * frames [C][T][N]; raw/smoothed [C][T][12] (either may be NULL).  Returns 0 on success. */
int fxo_batch_frames(int window_size, double sample_rate, int order_mode, const float* frames,
                     int C, int T, float* raw, float* smoothed, int threads);
/* The same over hops [C][T][N/2] with the runtime settings applied to every channel before its first hop. */
typedef struct { float gain; int onset_type; float onset_sensitivity; int onset_window; int analysers; } fxo_settings;
int fxo_batch_hops(int window_size, double sample_rate, int order_mode, const fxo_settings* settings, const float* hops,
                   int C, int T, float* raw, float* smoothed, int threads);

/* ---- taps (stateless building blocks, exposed for unit tests) ---- */
/* JUCE 4.2 FFT restatement; in/out are interleaved complex, size n. */
void fxo_fft_complex(int n, int inverse, const float* in, float* out);
/* ref: RealTimeAudioAnalysis.h:255-278 : n reals -> 2n floats (re,im pairs). */
void fxo_forward_real(int n, const float* x, float* spec2n);
/* ref: RealTimeAudioAnalysis.h:141-151 */
void fxo_bartlett(int n, float* x);
/* ref: RealTimeAudioAnalysis.h:106-125 */
void fxo_lowpass(int n, const float* in, float* out);
/* ref: PitchAnalyser.h:24-59 ; spec2n = forward_real of LPF+window frame.
 * cnd2n (may be NULL) receives the cumulative normalised difference buffer. */
double fxo_estimate_pitch(int n, double nyquist, const float* spec2n, float* cnd2n, float* lag_out);
/* constants used by the low-pass (so the GPU side can be checked against them) */
float fxo_lpf_a(void);
float fxo_lpf_b(void);

/* OSC 1.0 message as OSCSender::send(address, 12 floats) would emit it
 * (ref: OSCFeatureAnalysisOutput.h:107).  Returns bytes written (<= cap) or -1. */
int fxo_osc_message(const char* address, const float* smoothed12, unsigned char* out, int cap);

#ifdef __cplusplus
}
#endif
#endif
