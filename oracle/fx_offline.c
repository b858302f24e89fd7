/*
 * fx_offline.c -- CPU oracle for the reference's LEGACY offline analyser (struct AudioAnalyser, ref AudioAnalysis.h),
 * zero crossings, log attack time, FFT-LBP, histogram F0 / harmonic energy ratio / inharmonicity.
 *
 * TEST INFRASTRUCTURE ONLY (see fx_oracle.h): only tests/ may load this.  PARITY UNPINNED in the sense of fx_oracle.h: the
 * reference holds no vectors for these functions either; tools/refdiff/refdiff_legacy.cpp compiles AudioAnalysis.h /
 * AudioFeatures.h unmodified against a JUCE stand-in and tests/test_refdiff_cpu.py requires this file to agree with it bit
 * for bit (the four functions below touch no JUCE arithmetic beyond AudioSampleBuffer::getSample / findMinMax).
 * "ref:" citations are relative to /root/reference/Source/.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ref: AudioAnalysis.h:517-541 analyseNormalisedZeroCrosses.  audio [num_samples] of one channel; out [num_downsamples].
 * stepSize = numInputSamples / numDownsamples (int division, :521); per step the pairs (s, s+1) for s in [0, stepSize - 1)
 * (:529); a pair counts if first > 0 and (first - second) > first, or first < 0 and (first - second) < first (float
 * arithmetic, :534-535); value = numZeroCrosses * 2.0f / (float) stepSize (:538). */
void fxo_offline_zero_crosses(const float* audio, int num_samples, int num_downsamples, float* out)
{
    const int step = num_samples / num_downsamples;
    for (int i = 0; i < num_downsamples; i++) {
        float n = 0;
        const int sample = i * step;
        for (int s = 0; s < step - 1; ++s) {
            const float first = audio[sample + s];
            const float second = audio[sample + s + 1];
            const float d = first - second;
            if ((first > 0.0f && d > first) || (first < 0.0f && d < first)) n++;
        }
        out[i] = n * 2.0f / (float) step;
    }
}

/* ref: AudioAnalysis.h:611-622 setLogAttackTime.  envelope [n] = channel 0 of features.energyEnvelope; maxEnergy = the range's
 * end (:615); i = first index holding it (:616-618); msPerSample = 1.0 / (double) (sampleRate / 1000) with sampleRate an
 * int (AudioFeatures.h:69,303: integer division, :619); samplesPerStep = numInputSamples / numDownsamples (:620);
 * log10 ((double) (float) (i * samplesPerStep) * (float) msPerSample), as a float (:621). */
float fxo_offline_log_attack_time(const float* envelope, int n, int num_input_samples, int num_downsamples, int sample_rate)
{
    float max_energy = envelope[0];
    for (int k = 1; k < n; k++) if (envelope[k] > max_energy) max_energy = envelope[k];
    int i = 0;
    while (i < n && envelope[i] != max_energy) i++;
    const double ms_per_sample = 1.0 / (double) (sample_rate / 1000);
    const int samples_per_step = num_input_samples / num_downsamples;
    return (float) log10((double) (float) (i * samples_per_step) * (float) ms_per_sample);
}

/* ref: AudioAnalysis.h:543-564 calculateFFTLBP.  The reference only PRINTS: the bit pattern lbp[i] = |cur - prev| > 0.1f (:554-555),
 * then highestBinActivity / numBins (the last bin over the threshold, as a float, :559-560) and totalDiffs / totalSum (:563).
 * Returned here: bits [num_bins], *highest_ratio, *activity_ratio. */
void fxo_offline_fft_lbp(const float* cur, const float* prev, int num_bins, unsigned char* bits, float* highest_ratio, float* activity_ratio)
{
    const float threshold = 0.1f;
    float total_diffs = 0.0f, highest = 0.0f, total_sum = 0.0f;
    for (int i = 0; i < num_bins; i++) {
        total_sum++;
        const float diff = fabsf(cur[i] - prev[i]);
        const int b = diff > threshold;
        bits[i] = (unsigned char) b;
        total_diffs += (float) b;
        if (diff > threshold) highest = (float) i;
    }
    *highest_ratio = highest / (float) num_bins;
    *activity_ratio = total_diffs / total_sum;
}

/* ---- histogram F0 (ref: AudioAnalysis.h:253-303 calculateHarmonicCharacteristics and what it calls) ---- */
typedef struct { int interval, count; double her, freq; } cand_t;

/* ref :375-393 binIsPeak: above the mean, and no bin of [bin - (2 - leftOffset), bin + (2 - rightOffset)) larger */
static int bin_is_peak(int bin, const float* mag, int num_bins, double mean)
{
    const double m = (double) mag[bin];
    if (m <= mean) return 0;
    const int left = bin < 2 ? 2 - bin : 0;
    const int right = bin >= num_bins - 2 ? 2 - ((num_bins - 1) - bin) : 0;
    for (int nb = bin - (2 - left); nb < bin + (2 - right); nb++)
        if (nb != bin && (double) mag[nb] > m) return 0;
    return 1;
}

/* ref :80-98 F0Candidate::updateHarmonicEnergyRatio */
static double harmonic_energy_ratio(const float* mag, int num_bins, double frequency, double frpb, double total, double num_harmonics)
{
    double score = 0.0;
    for (double h = 1.0; h < num_harmonics + 1.0; h++) {
        const double hf = frequency * h;
        const int bin = (int) ceil(hf / frpb);
        if (bin >= num_bins) break;
        score += (double) mag[bin];
    }
    return score / total;
}

/* magnitudes [num_bins] of one channel's frame; *previous_f0 is the analyser's state (ref :273,293; untouched on the early
 * return :269-270); out3 = {f0, harmonicEnergyRatio, inharmonicity} as floats (:302). */
void fxo_offline_harmonic_characteristics(const float* mag, int num_bins, double nyquist, double* previous_f0, float* out3)
{
    double sum = 0.0;
    for (int i = 0; i < num_bins; i++) sum += (double) mag[i];                       /* :264-268 */
    const double mean = sum / (double) num_bins;
    if (sum < 0.001) { out3[0] = out3[1] = out3[2] = 0.0f; return; }                 /* :270-271 */
    int* peaks = (int*) malloc(sizeof(int) * (size_t) num_bins);
    cand_t* hist = (cand_t*) malloc(sizeof(cand_t) * (size_t) num_bins);
    int np = 0, nh = 0;
    for (int bin = 0; bin < num_bins; bin++) {                                       /* :350-364 */
        if (!bin_is_peak(bin, mag, num_bins, mean)) continue;
        peaks[np++] = bin;                                                           /* :397 */
        for (int p = 0; p < np - 1; p++) {                                           /* :398-407 */
            const int interval = bin - peaks[p];
            int pos = -1;
            for (int q = 0; q < nh; q++) if (hist[q].interval == interval) { pos = q; break; }   /* :410-417 */
            if (pos != -1) hist[pos].count++;
            else { hist[nh].interval = interval; hist[nh].count = 1; hist[nh].her = 0.0; hist[nh].freq = 0.0; nh++; }
        }
    }
    const double frpb = nyquist / (double) num_bins;                                 /* :428 */
    double max_weighted = 0.0, f0 = 0.0, her = 0.0;                                  /* :424-426 */
    for (int q = 0; q < nh; q++) {                                                   /* :429-439 */
        const double freq = (double) hist[q].interval * frpb;                        /* :59-62 */
        const double h = harmonic_energy_ratio(mag, num_bins, freq, frpb, sum, 15.0);
        const double weighted = (double) hist[q].count * h;                          /* :64-67 */
        if (weighted > max_weighted) { max_weighted = weighted; f0 = freq; her = h; }
    }
    if (*previous_f0 != f0 && *previous_f0 > 10.0) {                                 /* :273-291 */
        const double top = *previous_f0 > f0 ? *previous_f0 : f0;
        const double bottom = top == *previous_f0 ? f0 : *previous_f0;
        const double ratio = top / bottom;
        if (ratio > 2.0) {
            const double eps = 0.1;
            if (ratio - floor(ratio) < eps) {
                f0 = *previous_f0;
                her = harmonic_energy_ratio(mag, num_bins, f0, nyquist / (double) num_bins, sum, 15.0);
            }
        }
    }
    *previous_f0 = f0;                                                               /* :293 */
    double inh = 0.0;
    if (f0 > 0.0) {                                                                  /* :296-297, :305-336 */
        const int f0_bin = (int) ceil(f0 / (nyquist / (double) num_bins));           /* :443-448 getBinForFrequency */
        for (int k = 0; k < np; k++) {
            const int bin = peaks[k];
            if (f0_bin == bin) continue;
            double fs = bin * frpb;
            if (fs == 0.0) fs = frpb * 0.5;
            const double fe = (double) (bin + 1) * frpb;
            const double rs = fs == f0 ? 1.0 : (fs > f0 ? fs : f0) / (fs > f0 ? f0 : fs);       /* :338-346 */
            const double re = fe == f0 ? 1.0 : (fe > f0 ? fe : f0) / (fe > f0 ? f0 : fe);
            if (floor(rs) != floor(re)) continue;
            const double r = rs < re ? rs : re;
            inh += (r - floor(r)) * ((double) mag[bin] / sum);
        }
    }
    out3[0] = (float) f0; out3[1] = (float) her; out3[2] = (float) inh;
    free(peaks); free(hist);
}

/* ref: AudioAnalysis.h:463-515 calculateSpectralCharacteristics (the legacy, full-spectrum form: magnitudes as they are, no gate on
 * single bins).  mags [num_bins] of one channel; prev [num_bins] = previousBinMagnitudes (:700; zeros in a new analyser, :120-121),
 * replaced by this frame's magnitudes only when the frame passes the 0.001 gate (:498-500, :510).  out4 = centroid / nyquist,
 * spread, flatness, flux -- the struct's four floats in constructor order (:19-29, :514).
 * Every sum runs in double in bin order (:479-496); flux adds the rectified difference of |magnitude| (:486-489: (diff + |diff|) / 2
 * where diff > 0); the product is the plain serial IEEE product (:494: inf and 0 are sticky); frequencyRangePerBin = nyquist /
 * numBins with numBins a size_t (:465-466); invNumBins = 1.0 / numBins (:504); pow (x, 2.0) as the reference writes it (:509). */
void fxo_offline_spectral_characteristics(const float* mags, int num_bins, double nyquist, double* prev, float* out4)
{
    const size_t n = (size_t) num_bins;
    const double frpb = nyquist / n;
    double weighted = 0.0, var = 0.0, sum = 0.0, product = 1.0, flux = 0.0;
    double* centre = (double*) malloc(sizeof(double) * n);
    for (size_t i = 0; i < n; ++i) {
        const double fc = (double) i * frpb + (frpb / 2.0);
        centre[i] = fc;
        const double m = (double) mags[i];
        const double diff = fabs(m) - fabs(prev[i]);
        const double rectified = (diff + fabs(diff)) / 2.0;
        if (diff > 0.0) flux += rectified;
        sum += m;
        product *= m;
        weighted += fc * m;
    }
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0f;
    const double eps = 0.001;
    if (sum > eps) {
        const float centroid = (float) (weighted / sum);
        const double inv = 1.0 / n;
        const float flatness = (float) (pow(product, inv) / (inv * sum));
        for (size_t i = 0; i < n; ++i) {
            var += pow((centre[i] / nyquist) - (centroid / nyquist), 2.0) * (double) mags[i];
            prev[i] = (double) mags[i];
        }
        const float max_spread = (float) ((centroid / nyquist) * (1.0 - (centroid / nyquist)));
        const float spread = (float) ((var / sum) / max_spread);
        out4[0] = centroid / (float) nyquist; out4[1] = spread; out4[2] = flatness; out4[3] = (float) flux;
    }
    free(centre);
}

/* ref: AudioAnalysis.h:566-609 calculateNormalisedSpectralSlope (legacy form: the magnitudes themselves, normalised by
 * AudioSampleBuffer::getMagnitude = max |x| of the frame, :573).  The loops and the final expression are those of
 * SpectralCharacteristics.h:145-200, which descends from this function: `(numBins - 1.0f)` with numBins a double (:602). */
float fxo_offline_spectral_slope(const float* mags, int num_bins)
{
    const double bins = (double) num_bins, mean_bin = 0.5;
    float peak = 0.0f;
    for (int i = 0; i < num_bins; i++) { const float a = fabsf(mags[i]); if (a > peak) peak = a; }
    const double magnitude = (double) peak;
    if (!(magnitude > 0.0001)) return 0.0f;
    double mean_energy = 0.0, prod_sum = 0.0;
    for (int i = 0; i < (int) bins; i++) {
        const double e = mags[i] / magnitude;
        mean_energy += e;
        prod_sum += (double) i * e;
    }
    mean_energy /= bins;
    double bin_var = 0.0, energy_var = 0.0;
    for (double i = 0.0; i < bins; i++) {
        const double ni = i / bins;
        bin_var += (ni - mean_bin) * (ni - mean_bin);
        const double e = mags[(int) i] / magnitude;
        energy_var += (e - mean_energy) * (e - mean_energy);
    }
    bin_var /= bins;
    energy_var /= bins;
    const double bin_std = sqrt(bin_var), energy_std = sqrt(energy_var);
    const double r = (prod_sum - (bins * mean_energy * mean_bin)) / (bins - 1.0f) * energy_std * bin_std;
    return (float) (r * (bin_std / energy_std));
}

/* ref: AudioAnalysis.h:623-633 getConjugateComplexMultiplicationInPlace: data [num_items] interleaved (r, i), each item times its own
 * conjugate in float arithmetic exactly as written: ((r * r) - (i * (-i)), (r * (-i)) + (r * i)). */
void fxo_offline_conjugate_multiplication(float* data, int num_items)
{
    for (int k = 0; k < num_items; k++) {
        const float r = data[2 * k], i = data[2 * k + 1];
        const float cr = r, ci = -i;
        data[2 * k] = (r * cr) - (i * ci);
        data[2 * k + 1] = (r * ci) + (cr * i);
    }
}

/* ref: AudioAnalysis.h:636-648 analyseAutoCorrelation (+ getMaxIndex, :650-665): the bin of the largest real part -- the first one on
 * ties, `data[i] > currentMax` -- and the frequency the reference prints for it: peakBin * (nyquist / numItems) + half a bin. */
int fxo_offline_auto_correlation(const float* data, int num_items, double nyquist, double* frequency)
{
    int peak = 0;
    float current = data[0];
    for (int i = 0; i < num_items; i++)
        if (data[2 * i] > current) { peak = i; current = data[2 * i]; }
    const double frpb = nyquist / (double) num_items;
    *frequency = (peak * frpb) + (frpb / 2.0);
    return peak;
}
