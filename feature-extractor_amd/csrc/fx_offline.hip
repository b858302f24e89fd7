// fx_offline.hip -- the reference's LEGACY offline analyser (struct AudioAnalyser, ref AudioAnalysis.h) on
// gfx950: zero crossings, log attack time, FFT-LBP, the histogram F0 / harmonic energy ratio / inharmonicity and (round 4) the
// full-spectrum characteristics, the legacy slope and the auto-correlation peak, behind the fx_offline_* entries of include/fx.h.  "ref:" citations are relative to the reference's Source/.
//
// None of this is on the real-time path (the reference never instantiates AudioAnalyser); the kernels are written for exact
// agreement with the reference's arithmetic -- integer counts, fp32 comparisons as written, fp64 sums in the reference's own
// order where the order decides a comparison -- one workgroup per channel (per downsample step for the zero crossings).
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <vector>

#include "fx_context.h"

#pragma clang fp contract(off)

struct fx_offline {
    int device = 0;
    int C = 0;
    double nyquist = 24000.0;
    hipStream_t stream = nullptr;
    double* d_prev_f0 = nullptr;        // [C] previousF0 of each channel's analyser (ref AudioAnalysis.h:109,293,697)
    double* d_prev_bins = nullptr;      // [C][prev_bins] previousBinMagnitudes of each channel's analyser (ref :120-121,510,700); null until first used
    int     prev_bins = 0;
    void*   d_in = nullptr;  size_t in_cap = 0;      // staging for host buffers
    void*   d_out = nullptr; size_t out_cap = 0;
};

namespace {

constexpr int NT = 256;
constexpr int MAX_BINS = 4097;          // windowSize / 2 + 1 for windows up to 8192 (ref AudioAnalysis.h:113)

// The reference's serial sums, as ONE LANE runs them: t[0] + t[1] + ... in index order, from 0.0, nothing re-associated.  What can be done for
// speed without touching the order: the loads of a batch of eight terms are issued before the batch ahead of it is added, and independent chains
// of one kernel run on different LANES of a wave (a lane per chain costs what one chain costs) or on different waves.  What is left is the
// latency of a dependent v_add_f64, ~10 ns: measured (tools/offline_timing.py, 1024 analysers x 1025 bins, us per call, as shipped / in an
// experiment build with the chains cut out): spectral characteristics 40.9 / 13.2 (round 5: 77), slope 40.2 / 5.1 (49), auto-correlation 10.4 (40: its 256-entry serial
// maximum became a butterfly); at 4097 bins 424 / 70 and 357 / 26 (the slope 205 since its workgroup fits a CU twice: 16 bytes of LDS per bin).  Batches of 16 / 32: 42.2 / 51.9 and 36.5 / 44.2 -- not kept.  The harmonic
// characteristics (85 -> 53 us): peak positions by a shuffle prefix sum, the histogram's pairs a wave per new peak, the best candidate by butterflies.
template <typename T, bool PRODUCT>
__device__ __forceinline__ double serial_chain(const T* t, int n)
{
    constexpr int B = 8;
    double acc = PRODUCT ? 1.0 : 0.0;
    int i = 0;
    if (n >= B) {
        T cur[B], nxt[B];
#pragma unroll
        for (int u = 0; u < B; u++) cur[u] = t[u];
        for (; i + 2 * B <= n; i += B) {                                   // the next batch's loads are in flight while this batch is added
#pragma unroll
            for (int u = 0; u < B; u++) nxt[u] = t[i + B + u];
#pragma unroll
            for (int u = 0; u < B; u++) { if (PRODUCT) acc *= (double) cur[u]; else acc += (double) cur[u]; }
#pragma unroll
            for (int u = 0; u < B; u++) cur[u] = nxt[u];
        }
#pragma unroll
        for (int u = 0; u < B; u++) { if (PRODUCT) acc *= (double) cur[u]; else acc += (double) cur[u]; }
        i += B;
    }
    for (; i < n; i++) { if (PRODUCT) acc *= (double) t[i]; else acc += (double) t[i]; }
    return acc;
}
template <typename T>
__device__ __forceinline__ double serial_sum(const T* t, int n) { return serial_chain<T, false>(t, n); }
__device__ __forceinline__ double serial_product(const double* t, int n) { return serial_chain<double, true>(t, n); }      // IEEE product in index order, from 1.0 (inf / 0 sticky as they fall)
// order-free reductions (a maximum, an integer sum) over the block: butterflies inside a wave, the four wave results through LDS
__device__ __forceinline__ float block_max(float v, float* s_wave /* [NT / 64] */)
{
    for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o, 64); if (w > v) v = w; }
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = s_wave[0];
    for (int w = 1; w < NT / 64; w++) if (s_wave[w] > m) m = s_wave[w];
    __syncthreads();                                                      // (s_wave may be used again)
    return m;
}

// ---- ref AudioAnalysis.h:517-541 analyseNormalisedZeroCrosses: block = (channel, downsample step) ----
__global__ void __launch_bounds__(NT) zero_crosses_kernel(const float* audio, int num_samples, int num_downsamples, float* out)
{
    __shared__ int s_cnt[NT / 64];
    const int c = blockIdx.x / num_downsamples, i = blockIdx.x % num_downsamples;
    const int step = num_samples / num_downsamples;                        // :521 (int division)
    const float* a = audio + (size_t) c * num_samples + (size_t) i * step; // :528
    int n = 0;
    for (int s = threadIdx.x; s < step - 1; s += NT) {                     // :529
        const float first = a[s], second = a[s + 1];
        const float d = first - second;
        if ((first > 0.0f && d > first) || (first < 0.0f && d < first)) n++;        // :534-535
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
        for (int w = 0; w < NT / 64; w++) total += s_cnt[w];
        out[(size_t) c * num_downsamples + i] = (float) total * 2.0f / (float) step;       // :538 (a float count is exact below 2^24)
    }
}

// ---- ref AudioAnalysis.h:611-622 setLogAttackTime: one block ----
__global__ void __launch_bounds__(NT) log_attack_time_kernel(const float* env, int n, int num_input_samples, int num_downsamples, int sample_rate, float* out)
{
    __shared__ float s_max[NT];
    __shared__ int s_idx[NT];
    float m = -__builtin_huge_valf();
    bool any = false;
    for (int k = threadIdx.x; k < n; k += NT) { const float v = env[k]; if (!any || v > m) { m = v; any = true; } }   // findMinMax(...).getEnd(), :615
    s_max[threadIdx.x] = any ? m : -__builtin_huge_valf();
    __syncthreads();
    if (threadIdx.x == 0) { float mm = s_max[0]; for (int k = 1; k < NT; k++) if (s_max[k] > mm) mm = s_max[k]; s_max[0] = mm; }
    __syncthreads();
    const float max_energy = s_max[0];
    int first = n;                                                         // first index holding the maximum, :616-618
    for (int k = threadIdx.x; k < n; k += NT) if (env[k] == max_energy) { first = k; break; }
    s_idx[threadIdx.x] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
        int i = n;
        for (int k = 0; k < NT; k++) if (s_idx[k] < i) i = s_idx[k];
        const double ms_per_sample = 1.0 / (double) (sample_rate / 1000);  // :619 (sampleRate is an int: AudioFeatures.h:303)
        const int samples_per_step = num_input_samples / num_downsamples;  // :620
        out[0] = (float) log10((double) (float) (i * samples_per_step) * (float) ms_per_sample);   // :621
    }
}

// ---- ref AudioAnalysis.h:543-564 calculateFFTLBP: block = channel ----
__global__ void __launch_bounds__(NT) fft_lbp_kernel(const float* cur, const float* prev, int num_bins, unsigned char* bits, float* highest_ratio, float* activity_ratio)
{
    __shared__ int s_cnt[NT / 64], s_hi[NT / 64];
    const int c = blockIdx.x;
    const float threshold = 0.1f;                                          // :546
    int cnt = 0, hi = 0;
    for (int i = threadIdx.x; i < num_bins; i += NT) {
        const float diff = fabsf(cur[(size_t) c * num_bins + i] - prev[(size_t) c * num_bins + i]);    // :554
        const int b = diff > threshold;
        bits[(size_t) c * num_bins + i] = (unsigned char) b;               // :555
        cnt += b;
        if (b && i > hi) hi = i;                                           // :558-559: the last bin over the threshold
    }
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); const int h = __shfl_xor(hi, o, 64); if (h > hi) hi = h; }      // (integers: any order)
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = cnt; s_hi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0, highest = 0;
        for (int k = 0; k < NT / 64; k++) { total += s_cnt[k]; if (s_hi[k] > highest) highest = s_hi[k]; }
        highest_ratio[c] = (float) highest / (float) num_bins;             // :563 (float counts are exact here)
        activity_ratio[c] = (float) total / (float) num_bins;
    }
}

// ---- ref AudioAnalysis.h:253-303 calculateHarmonicCharacteristics: block = channel ----
__device__ __forceinline__ bool bin_is_peak(int bin, const float* mag, int num_bins, double mean)        // :375-393
{
    const double m = (double) mag[bin];
    if (m <= mean) return false;
    const int left = bin < 2 ? 2 - bin : 0;
    const int right = bin >= num_bins - 2 ? 2 - ((num_bins - 1) - bin) : 0;
    for (int nb = bin - (2 - left); nb < bin + (2 - right); nb++)
        if (nb != bin && (double) mag[nb] > m) return false;
    return true;
}
__device__ __forceinline__ double harmonic_energy_ratio(const float* mag, int num_bins, double frequency, double frpb, double total)   // :80-98, 15 harmonics
{
    double score = 0.0;
    for (double h = 1.0; h < 15.0 + 1.0; h++) {
        const double hf = frequency * h;
        const int bin = (int) ceil(hf / frpb);
        if (bin >= num_bins) break;
        score += (double) mag[bin];
    }
    return score / total;
}

__global__ void __launch_bounds__(NT) harmonic_characteristics_kernel(const float* mags, int num_bins, double nyquist, double* prev_f0, float* out3)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* mag = reinterpret_cast<float*>(smem);                           // [num_bins]
    int* peaks = reinterpret_cast<int*>(mag + num_bins + (num_bins & 1));  // [num_bins] ascending peak bins
    int* cnt = peaks + num_bins + (num_bins & 1);                          // [num_bins] histogram: pairs of peaks that many bins apart
    int* key = cnt + num_bins + (num_bins & 1);                            // [num_bins] where the interval first entered the histogram
    double* term = reinterpret_cast<double*>(cnt);                         // [<= num_bins] inharmonicity terms (cnt / key are free by then)
    __shared__ double s_sum, s_f0, s_her;
    __shared__ int s_wave_peaks[NT / 64];
    __shared__ double s_w[NT / 64], s_fr[NT / 64], s_hr[NT / 64];
    __shared__ int s_k[NT / 64];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < num_bins; i += NT) { mag[i] = mags[(size_t) c * num_bins + i]; cnt[i] = 0; key[i] = 0x7fffffff; }
    __syncthreads();
    if (tid == 0) {                                                        // :264-268: the reference's serial sum (its last bit decides `mag > mean`)
        s_sum = serial_sum(mag, num_bins);
    }
    __syncthreads();
    const double sum = s_sum;
    const double mean = sum / (double) num_bins;
    if (sum < 0.001) {                                                     // :270-271 (previousF0 untouched)
        if (tid < 3) out3[(size_t) c * 3 + tid] = 0.0f;
        return;
    }
    // peaks, in ascending order (:350-364): every thread a contiguous run of bins
    const int run = (num_bins + NT - 1) / NT, b0 = tid * run, b1 = min(num_bins, b0 + run);
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += bin_is_peak(b, mag, num_bins, mean) ? 1 : 0;
    int before = mine;                                                     // where this thread's peaks go: a prefix sum over the block (integers: any order)
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(before, o, 64); if (lane >= o) before += v; }
    if (lane == 63) s_wave_peaks[wave] = before;
    __syncthreads();
    int np = 0;
    {
        int at = before - mine;
        for (int w = 0; w < NT / 64; w++) { if (w < wave) at += s_wave_peaks[w]; np += s_wave_peaks[w]; }
        for (int b = b0; b < b1; b++) if (bin_is_peak(b, mag, num_bins, mean)) peaks[at++] = b;
    }
    __syncthreads();
    // the frequency histogram (:395-408): one entry per distinct distance between two peaks, counting the pairs; the reference
    // appends an interval when it first meets it -- scanning new peaks j upwards and, for each, earlier peaks p upwards -- and
    // that order breaks ties below
    // (a wave per new peak j, its lanes across the earlier peaks p: the pairs of the longest rows are not one thread's)
    for (int j = wave; j < np; j += NT / 64) {
        const int pj = peaks[j];
        for (int p = lane; p < j; p += 64) {
            const int d = pj - peaks[p];
            atomicAdd(&cnt[d], 1);
            atomicMin(&key[d], j * 8192 + p);
        }
    }
    __syncthreads();
    // estimateF0AndHERFromFrequencyHistogram (:419-441): the candidate with the largest count x harmonic energy ratio, the
    // first such in histogram order (`>` against a running maximum that starts at 0)
    const double frpb = nyquist / (double) num_bins;                       // :428
    double best_w = 0.0, best_f = 0.0, best_h = 0.0;
    int best_k = 0x7fffffff;
    for (int d = 1 + tid; d < num_bins; d += NT) {
        if (cnt[d] == 0) continue;
        const double freq = (double) d * frpb;                             // :59-62
        const double her = harmonic_energy_ratio(mag, num_bins, freq, frpb, sum);
        const double w = (double) cnt[d] * her;                            // :64-67
        if (w > best_w || (w == best_w && w > 0.0 && key[d] < best_k)) { best_w = w; best_f = freq; best_h = her; best_k = key[d]; }
    }
    // (largest weight, then first in histogram order) is a total order on the candidates with a weight above 0 -- the only ones ever taken; the
    // others all read (0, 0, 0, max): butterflies inside a wave, then the four wave results
    for (int o = 32; o > 0; o >>= 1) {
        const double w = __shfl_xor(best_w, o, 64), f = __shfl_xor(best_f, o, 64), h = __shfl_xor(best_h, o, 64);
        const int k = __shfl_xor(best_k, o, 64);
        if (w > best_w || (w == best_w && w > 0.0 && k < best_k)) { best_w = w; best_f = f; best_h = h; best_k = k; }
    }
    if (lane == 0) { s_w[wave] = best_w; s_fr[wave] = best_f; s_hr[wave] = best_h; s_k[wave] = best_k; }
    __syncthreads();
    if (tid == 0) {
        double w = 0.0, f0 = 0.0, her = 0.0;
        int k = 0x7fffffff;
        for (int q = 0; q < NT / 64; q++)
            if (s_w[q] > w || (s_w[q] == w && w > 0.0 && s_k[q] < k)) { w = s_w[q]; f0 = s_fr[q]; her = s_hr[q]; k = s_k[q]; }
        const double previous = prev_f0[c];
        if (previous != f0 && previous > 10.0) {                           // :273-291
            const double top = previous > f0 ? previous : f0;
            const double bottom = top == previous ? f0 : previous;
            const double ratio = top / bottom;
            if (ratio > 2.0) {
                const double eps = 0.1;
                if (ratio - floor(ratio) < eps) {
                    f0 = previous;
                    her = harmonic_energy_ratio(mag, num_bins, f0, nyquist / (double) num_bins, sum);
                }
            }
        }
        prev_f0[c] = f0;                                                   // :293
        s_f0 = f0; s_her = her;
    }
    __syncthreads();                                                       // (cnt / key are no longer read: `term` may overwrite them)
    const double f0 = s_f0;
    // calculateInharmonicity (:305-336): one term per peak, added in peak order
    if (f0 > 0.0) {
        const int f0_bin = (int) ceil(f0 / (nyquist / (double) num_bins)); // :443-448
        for (int k = tid; k < np; k += NT) {
            const int bin = peaks[k];
            double t = 0.0;                                                // (a skipped peak adds nothing: x + 0.0 == x)
            if (f0_bin != bin) {
                double fs = bin * frpb;
                if (fs == 0.0) fs = frpb * 0.5;
                const double fe = (double) (bin + 1) * frpb;
                const double rs = fs == f0 ? 1.0 : (fs > f0 ? fs : f0) / (fs > f0 ? f0 : fs);       // :338-346
                const double re = fe == f0 ? 1.0 : (fe > f0 ? fe : f0) / (fe > f0 ? f0 : fe);
                if (floor(rs) == floor(re)) {
                    const double r = rs < re ? rs : re;
                    t = (r - floor(r)) * ((double) mag[bin] / sum);
                }
            }
            term[k] = t;
        }
    }
    __syncthreads();
    if (tid == 0) {
        double inh = 0.0;
        if (f0 > 0.0) inh = serial_sum(term, np);
        out3[(size_t) c * 3 + 0] = (float) f0; out3[(size_t) c * 3 + 1] = (float) s_her; out3[(size_t) c * 3 + 2] = (float) inh;   // :302
    }
}

size_t harmonic_lds_bytes(int num_bins) { return 4 * (size_t) (num_bins + (num_bins & 1)) * sizeof(float); }

fx_status grow_bytes(void** ptr, size_t* cap, size_t need)
{
    if (need <= *cap) return FX_OK;
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    HIP_TRY(hipMalloc(ptr, need));
    *cap = need;
    return FX_OK;
}

// input(s) on the device: the caller's pointers, or a staged copy of host buffers (two inputs back to back)
fx_status stage_in(fx_offline* o, int mem_kind, const void* a, size_t a_bytes, const void* b, size_t b_bytes, const void** da, const void** db)
{
    if (mem_kind == FX_MEM_DEVICE) { *da = a; *db = b; return FX_OK; }
    fx_status st = grow_bytes(&o->d_in, &o->in_cap, a_bytes + b_bytes);
    if (st != FX_OK) return st;
    HIP_TRY(hipMemcpyAsync(o->d_in, a, a_bytes, hipMemcpyHostToDevice, o->stream));
    *da = o->d_in;
    *db = nullptr;
    if (b) {
        HIP_TRY(hipMemcpyAsync(static_cast<char*>(o->d_in) + a_bytes, b, b_bytes, hipMemcpyHostToDevice, o->stream));
        *db = static_cast<char*>(o->d_in) + a_bytes;
    }
    return FX_OK;
}

fx_status check_args(fx_offline* o, const void* in, const void* out, int mem_kind)
{
    if (!o || !in || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (mem_kind != FX_MEM_HOST && mem_kind != FX_MEM_DEVICE) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown memory kind %d", mem_kind);
    HIP_TRY(hipSetDevice(o->device));
    return FX_OK;
}

// ---- ref AudioAnalysis.h:463-515 calculateSpectralCharacteristics: block = channel.  Every sum of the reference is a serial double
// sum in bin order, and the product a serial IEEE product (inf / 0 sticky).  What is serial is only the ADDITIONS: each bin's terms --
// the rectified difference, fc * m, the pow() of the spread -- are a function of that bin alone, so the whole block forms them side by side
// into LDS (the same expressions, so the same roundings: no contraction in this file), and single lanes then add them up in bin order.
// (Round 4 had one thread evaluate everything, 1025 pow() calls in a row: 1.98 ms per call of 1024 analysers; round 5, one thread adding four
// chains with every LDS read waited for: 77 us; bench.py `offline`.)
// Round 6: the three sums of the first pass run on three lanes of wave 0 and the product on a lane of wave 1, at the same time, each with its
// loads ahead of its additions (serial_sum); "not added" became "+ 0.0" (the flux starts at +0.0 and only grows: x + 0.0 == x bit for bit).
// LDS: a[bins] | b[bins] | m[bins], doubles = 24 bytes per bin (98 KB at 4097 bins).
constexpr size_t SPECTRAL_LDS_PER_BIN = 3 * sizeof(double);
__global__ void __launch_bounds__(NT) spectral_characteristics_kernel(const float* mags, int num_bins, double nyquist, double* prev_bins, float* out4)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ta = reinterpret_cast<double*>(smem);                           // per-bin terms, first pass: rectified difference (or 0); second: spread term
    double* tb = ta + num_bins;                                             // first pass: fc * m
    double* tm = tb + num_bins;                                             // (double) mag
    __shared__ double s_chain[4];                                           // flux, sum, weighted, product
    const int c = blockIdx.x;
    double* gprev = prev_bins + (size_t) c * num_bins;
    const size_t n = (size_t) num_bins;
    const double frpb = nyquist / n;                                        // :466
    for (int i = threadIdx.x; i < num_bins; i += NT) {
        const float mf = mags[(size_t) c * num_bins + i];
        const double pv = gprev[i];
        const double fc = (double) i * frpb + (frpb / 2.0);                 // :479-496, the per-bin part
        const double m = (double) mf;
        const double diff = fabs(m) - fabs(pv);
        const double rectified = (diff + fabs(diff)) / 2.0;
        ta[i] = diff > 0.0 ? rectified : 0.0;                               // (the reference's `if (diff > 0.0) flux += rectified`)
        tb[i] = fc * m;
        tm[i] = m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {                                                  // the additions, in bin order: a lane per sum
        const double* src = threadIdx.x == 0 ? ta : (threadIdx.x == 1 ? tm : tb);
        s_chain[threadIdx.x] = serial_sum(src, num_bins);
    } else if (threadIdx.x == 64) {
        s_chain[3] = serial_product(tm, num_bins);
    }
    __syncthreads();
    const double flux = s_chain[0], sum = s_chain[1], weighted = s_chain[2], product = s_chain[3];
    if (!(sum > 0.001)) {                                                   // :498-500
        if (threadIdx.x == 0) { out4[4 * c] = 0.0f; out4[4 * c + 1] = 0.0f; out4[4 * c + 2] = 0.0f; out4[4 * c + 3] = 0.0f; }
        return;
    }
    const float centroid = (float) (weighted / sum);
    for (int i = threadIdx.x; i < num_bins; i += NT) {
        const double fc = (double) i * frpb + (frpb / 2.0);                 // binCentreFrequencies[i], the same expression
        ta[i] = pow((fc / nyquist) - (centroid / nyquist), 2.0) * tm[i];    // :509, the per-bin part
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double var = serial_sum(ta, num_bins);
        const double inv = 1.0 / n;
        const float flatness = (float) (pow(product, inv) / (inv * sum));                 // :505
        const float max_spread = (float) ((centroid / nyquist) * (1.0 - (centroid / nyquist)));
        out4[4 * c] = centroid / (float) nyquist; out4[4 * c + 1] = (float) ((var / sum) / max_spread); out4[4 * c + 2] = flatness; out4[4 * c + 3] = (float) flux;
    }
    for (int i = threadIdx.x; i < num_bins; i += NT) gprev[i] = tm[i];                    // :510
}

// ---- ref AudioAnalysis.h:566-609 calculateNormalisedSpectralSlope: block = channel; the maximum in parallel (order does not matter),
// each bin's quotient and squares by the whole block, the double sums by one thread in the reference's order ----
// LDS: e[bins] doubles | t[bins] doubles = 16 bytes per bin (66 KB at 4097 bins: two workgroups per CU; the magnitudes are read twice from
// global memory -- the second time from the L2 -- instead of being kept)
constexpr size_t SLOPE_LDS_PER_BIN = 2 * sizeof(double);
__global__ void __launch_bounds__(NT) spectral_slope_kernel(const float* mags, int num_bins, float* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* te = reinterpret_cast<double*>(smem);                           // e_i = mag[i] / magnitude
    double* tt = te + num_bins;                                             // first pass: i * e_i; second: (e_i - mean)^2
    __shared__ float s_wave[NT / 64];
    __shared__ double s_chain[2];
    const int c = blockIdx.x;
    const float* mag = mags + (size_t) c * num_bins;
    float peak = 0.0f;
    for (int i = threadIdx.x; i < num_bins; i += NT) { const float a = fabsf(mag[i]); if (a > peak) peak = a; }
    peak = block_max(peak, s_wave);
    const double bins = (double) num_bins, mean_bin = 0.5, magnitude = (double) peak;        // getMagnitude, :573
    if (!(magnitude > 0.0001)) { if (threadIdx.x == 0) out[c] = 0.0f; return; }
    for (int i = threadIdx.x; i < num_bins; i += NT) {
        const double e = mag[i] / magnitude;
        te[i] = e;
        tt[i] = (double) i * e;
    }
    __syncthreads();
    if (threadIdx.x < 2) s_chain[threadIdx.x] = serial_sum(threadIdx.x == 0 ? te : tt, num_bins);     // the two sums of :577-583, a lane each
    __syncthreads();
    const double mean_energy = s_chain[0] / bins, prod_sum = s_chain[1];
    __syncthreads();                                                        // (s_chain is written again below)
    for (int i = threadIdx.x; i < num_bins; i += NT) {
        const double e = te[i], ni = (double) i / bins;
        tt[i] = (e - mean_energy) * (e - mean_energy);
        te[i] = (ni - mean_bin) * (ni - mean_bin);                             // (e is not needed again)
    }
    __syncthreads();
    if (threadIdx.x < 2) s_chain[threadIdx.x] = serial_sum(threadIdx.x == 0 ? te : tt, num_bins);
    __syncthreads();
    if (threadIdx.x == 0) {
        double bin_var = s_chain[0], energy_var = s_chain[1];
        bin_var /= bins;
        energy_var /= bins;
        const double bin_std = sqrt(bin_var), energy_std = sqrt(energy_var);
        const double r = (prod_sum - (bins * mean_energy * mean_bin)) / (bins - 1.0f) * energy_std * bin_std;      // :602
        out[c] = (float) (r * (bin_std / energy_std));
    }
}

// ---- ref AudioAnalysis.h:623-633 getConjugateComplexMultiplicationInPlace: thread = item ----
__global__ void __launch_bounds__(NT) conjugate_multiplication_kernel(float* data, long long items)
{
    const long long k = (long long) blockIdx.x * NT + threadIdx.x;
    if (k >= items) return;
    const float r = data[2 * k], i = data[2 * k + 1];
    const float cr = r, ci = -i;
    data[2 * k] = (r * cr) - (i * ci);
    data[2 * k + 1] = (r * ci) + (cr * i);
}

// ---- ref AudioAnalysis.h:636-665 analyseAutoCorrelation + getMaxIndex: block = channel; the FIRST bin holding the largest real part ----
__global__ void __launch_bounds__(NT) auto_correlation_kernel(const float* data, int num_items, double nyquist, int* peak_bin, double* frequency)
{
    __shared__ float s_val[NT / 64];
    __shared__ int s_idx[NT / 64];
    const int c = blockIdx.x;
    const float* d = data + (size_t) c * num_items * 2;
    float best = 0.0f; int at = -1;
    // ascending i: the first on ties.  NaNs are skipped as getMaxIndex skips them (`data[i] > currentMax` is false for one), so a NaN early in a
    // thread's stride cannot hide the maxima behind it; a NaN at data[0] is thread 0's business below
    for (int i = threadIdx.x; i < num_items; i += NT) { const float v = d[2 * i]; if (v == v && (at < 0 || v > best)) { best = v; at = i; } }
    // (larger value, then smaller bin) is a total order on the candidates that exist (at >= 0; none of them a NaN): butterflies inside a wave
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(best, o, 64); const int k = __shfl_xor(at, o, 64);
        if (k >= 0 && (at < 0 || v > best || (v == best && k < at))) { best = v; at = k; }
    }
    if ((threadIdx.x & 63) == 0) { s_val[threadIdx.x >> 6] = best; s_idx[threadIdx.x >> 6] = at; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // getMaxIndex starts from data[0] and moves on `data[i] > currentMax` only: a NaN at 0 keeps bin 0, NaNs elsewhere never win
        int peak = 0; float current = d[0];
        for (int k = 0; k < NT / 64; k++)
            if (s_idx[k] >= 0 && (s_val[k] > current || (s_val[k] == current && s_idx[k] < peak))) { current = s_val[k]; peak = s_idx[k]; }
        const double frpb = nyquist / (double) num_items;
        peak_bin[c] = peak;
        frequency[c] = (peak * frpb) + (frpb / 2.0);
    }
}

} // namespace

extern "C" {

fx_status fx_offline_create(fx_offline** out, int device_id, int num_channels, double nyquist)
{
    if (!out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null output pointer");
    *out = nullptr;
    if (num_channels <= 0 || !(nyquist > 0.0)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "num_channels and nyquist must be positive");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void) hipGetLastError();
        return fx_fail(FX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    if (device_id < 0 || device_id >= count) return fx_fail(FX_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, count);
    HIP_TRY(hipSetDevice(device_id));
    fx_offline* o = new (std::nothrow) fx_offline();
    if (!o) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    o->device = device_id; o->C = num_channels; o->nyquist = nyquist;
    hipError_t e = hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**) &o->d_prev_f0, sizeof(double) * (size_t) num_channels);
    if (e == hipSuccess) e = hipMemsetAsync(o->d_prev_f0, 0, sizeof(double) * (size_t) num_channels, o->stream);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&harmonic_characteristics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int) harmonic_lds_bytes(MAX_BINS));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectral_characteristics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int) (SPECTRAL_LDS_PER_BIN * MAX_BINS));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spectral_slope_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int) (SLOPE_LDS_PER_BIN * MAX_BINS));
    if (e == hipSuccess) e = hipStreamSynchronize(o->stream);
    if (e != hipSuccess) { fx_offline_destroy(o); return fx_fail(FX_ERR_HIP, "setting up the offline analyser failed: %s", hipGetErrorString(e)); }
    *out = o;
    return FX_OK;
}

fx_status fx_offline_destroy(fx_offline* o)
{
    if (!o) return FX_OK;
    (void) hipSetDevice(o->device);
    if (o->stream) (void) hipStreamSynchronize(o->stream);
    if (o->d_prev_f0) (void) hipFree(o->d_prev_f0);
    if (o->d_prev_bins) (void) hipFree(o->d_prev_bins);
    if (o->d_in) (void) hipFree(o->d_in);
    if (o->d_out) (void) hipFree(o->d_out);
    if (o->stream) (void) hipStreamDestroy(o->stream);
    delete o;
    return FX_OK;
}

fx_status fx_offline_reset(fx_offline* o)
{
    if (!o) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipMemsetAsync(o->d_prev_f0, 0, sizeof(double) * (size_t) o->C, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    if (o->d_prev_bins) { HIP_TRY(hipFree(o->d_prev_bins)); o->d_prev_bins = nullptr; o->prev_bins = 0; }      // a new analyser: any frame size again
    return FX_OK;
}

fx_status fx_offline_get_previous_f0(fx_offline* o, double* out)
{
    if (!o || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipMemcpyAsync(out, o->d_prev_f0, sizeof(double) * (size_t) o->C, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return FX_OK;
}

fx_status fx_offline_zero_crosses(fx_offline* o, const float* audio, int num_samples, int num_downsamples, float* out, int mem_kind)
{
    fx_status st = check_args(o, audio, out, mem_kind);
    if (st != FX_OK) return st;
    if (num_downsamples < 1 || num_samples < num_downsamples) return fx_fail(FX_ERR_INVALID_ARGUMENT, "need 1 <= num_downsamples <= num_samples (stepSize > 0, ref AudioAnalysis.h:521)");
    const size_t in_bytes = sizeof(float) * (size_t) o->C * num_samples, out_bytes = sizeof(float) * (size_t) o->C * num_downsamples;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, audio, in_bytes, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_out = out;
    if (mem_kind == FX_MEM_HOST) { if ((st = grow_bytes(&o->d_out, &o->out_cap, out_bytes)) != FX_OK) return st; d_out = static_cast<float*>(o->d_out); }
    hipLaunchKernelGGL(zero_crosses_kernel, dim3((unsigned) (o->C * num_downsamples)), dim3(NT), 0, o->stream, static_cast<const float*>(da), num_samples, num_downsamples, d_out);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) { HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, o->stream)); HIP_TRY(hipStreamSynchronize(o->stream)); }
    return FX_OK;
}

fx_status fx_offline_log_attack_time(fx_offline* o, const float* envelope, int n, int num_input_samples, int num_downsamples, int sample_rate,
                                     float* out, int mem_kind)
{
    fx_status st = check_args(o, envelope, out, mem_kind);
    if (st != FX_OK) return st;
    if (n < 1 || num_downsamples < 1 || sample_rate < 1000) return fx_fail(FX_ERR_INVALID_ARGUMENT, "need n >= 1, num_downsamples >= 1, sample_rate >= 1000 (sampleRate / 1000 is an int division, ref AudioAnalysis.h:619)");
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, envelope, sizeof(float) * (size_t) n, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_out = out;
    if (mem_kind == FX_MEM_HOST) { if ((st = grow_bytes(&o->d_out, &o->out_cap, sizeof(float))) != FX_OK) return st; d_out = static_cast<float*>(o->d_out); }
    hipLaunchKernelGGL(log_attack_time_kernel, dim3(1), dim3(NT), 0, o->stream, static_cast<const float*>(da), n, num_input_samples, num_downsamples, sample_rate, d_out);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) { HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(float), hipMemcpyDeviceToHost, o->stream)); HIP_TRY(hipStreamSynchronize(o->stream)); }
    return FX_OK;
}

fx_status fx_offline_fft_lbp(fx_offline* o, const float* cur, const float* prev, int num_bins, unsigned char* bits, float* highest_ratio,
                             float* activity_ratio, int mem_kind)
{
    fx_status st = check_args(o, cur, bits, mem_kind);
    if (st != FX_OK) return st;
    if (!prev || !highest_ratio || !activity_ratio || num_bins < 1) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument or num_bins < 1");
    const size_t in_bytes = sizeof(float) * (size_t) o->C * num_bins;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, cur, in_bytes, prev, in_bytes, &da, &db)) != FX_OK) return st;
    unsigned char* d_bits = bits; float* d_hi = highest_ratio; float* d_act = activity_ratio;
    const size_t bits_bytes = (size_t) o->C * num_bins, bits_pad = (bits_bytes + 15) & ~(size_t) 15;
    if (mem_kind == FX_MEM_HOST) {
        if ((st = grow_bytes(&o->d_out, &o->out_cap, bits_pad + 2 * sizeof(float) * (size_t) o->C)) != FX_OK) return st;
        d_bits = static_cast<unsigned char*>(o->d_out);
        d_hi = reinterpret_cast<float*>(d_bits + bits_pad);
        d_act = d_hi + o->C;
    }
    hipLaunchKernelGGL(fft_lbp_kernel, dim3((unsigned) o->C), dim3(NT), 0, o->stream, static_cast<const float*>(da), static_cast<const float*>(db), num_bins, d_bits, d_hi, d_act);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(bits, d_bits, bits_bytes, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipMemcpyAsync(highest_ratio, d_hi, sizeof(float) * (size_t) o->C, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipMemcpyAsync(activity_ratio, d_act, sizeof(float) * (size_t) o->C, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipStreamSynchronize(o->stream));
    }
    return FX_OK;
}

fx_status fx_offline_harmonic_characteristics(fx_offline* o, const float* magnitudes, int num_bins, float* out3, int mem_kind)
{
    fx_status st = check_args(o, magnitudes, out3, mem_kind);
    if (st != FX_OK) return st;
    if (num_bins < 4 || num_bins > MAX_BINS) return fx_fail(FX_ERR_INVALID_ARGUMENT, "num_bins must be in [4, %d]", MAX_BINS);
    const size_t in_bytes = sizeof(float) * (size_t) o->C * num_bins, out_bytes = sizeof(float) * 3 * (size_t) o->C;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, magnitudes, in_bytes, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_out = out3;
    if (mem_kind == FX_MEM_HOST) { if ((st = grow_bytes(&o->d_out, &o->out_cap, out_bytes)) != FX_OK) return st; d_out = static_cast<float*>(o->d_out); }
    hipLaunchKernelGGL(harmonic_characteristics_kernel, dim3((unsigned) o->C), dim3(NT), harmonic_lds_bytes(num_bins), o->stream,
                       static_cast<const float*>(da), num_bins, o->nyquist, o->d_prev_f0, d_out);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) { HIP_TRY(hipMemcpyAsync(out3, d_out, out_bytes, hipMemcpyDeviceToHost, o->stream)); HIP_TRY(hipStreamSynchronize(o->stream)); }
    return FX_OK;
}

fx_status fx_offline_spectral_characteristics(fx_offline* o, const float* magnitudes, int num_bins, float* out4, int mem_kind)
{
    fx_status st = check_args(o, magnitudes, out4, mem_kind);
    if (st != FX_OK) return st;
    if (num_bins < 1 || num_bins > MAX_BINS) return fx_fail(FX_ERR_INVALID_ARGUMENT, "num_bins must be in [1, %d]", MAX_BINS);
    if (o->d_prev_bins && o->prev_bins != num_bins)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "this analyser's previousBinMagnitudes holds %d bins (fixed by its window size, ref AudioAnalysis.h:120-121); "
                                                "fx_offline_reset before frames of %d", o->prev_bins, num_bins);
    if (!o->d_prev_bins) {
        HIP_TRY(hipMalloc((void**) &o->d_prev_bins, sizeof(double) * (size_t) o->C * num_bins));
        HIP_TRY(hipMemsetAsync(o->d_prev_bins, 0, sizeof(double) * (size_t) o->C * num_bins, o->stream));
        o->prev_bins = num_bins;
    }
    const size_t in_bytes = sizeof(float) * (size_t) o->C * num_bins, out_bytes = sizeof(float) * 4 * (size_t) o->C;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, magnitudes, in_bytes, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_out = out4;
    if (mem_kind == FX_MEM_HOST) { if ((st = grow_bytes(&o->d_out, &o->out_cap, out_bytes)) != FX_OK) return st; d_out = static_cast<float*>(o->d_out); }
    hipLaunchKernelGGL(spectral_characteristics_kernel, dim3((unsigned) o->C), dim3(NT), SPECTRAL_LDS_PER_BIN * (size_t) num_bins, o->stream,
                       static_cast<const float*>(da), num_bins, o->nyquist, o->d_prev_bins, d_out);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) { HIP_TRY(hipMemcpyAsync(out4, d_out, out_bytes, hipMemcpyDeviceToHost, o->stream)); HIP_TRY(hipStreamSynchronize(o->stream)); }
    return FX_OK;
}

fx_status fx_offline_get_previous_bins(fx_offline* o, double* out, int num_bins)
{
    if (!o || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (!o->d_prev_bins || o->prev_bins != num_bins) return fx_fail(FX_ERR_INVALID_ARGUMENT, "previousBinMagnitudes holds %d bins", o->prev_bins);
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipMemcpyAsync(out, o->d_prev_bins, sizeof(double) * (size_t) o->C * num_bins, hipMemcpyDeviceToHost, o->stream));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return FX_OK;
}

fx_status fx_offline_spectral_slope(fx_offline* o, const float* magnitudes, int num_bins, float* out, int mem_kind)
{
    fx_status st = check_args(o, magnitudes, out, mem_kind);
    if (st != FX_OK) return st;
    if (num_bins < 1 || num_bins > MAX_BINS) return fx_fail(FX_ERR_INVALID_ARGUMENT, "num_bins must be in [1, %d]", MAX_BINS);
    const size_t in_bytes = sizeof(float) * (size_t) o->C * num_bins, out_bytes = sizeof(float) * (size_t) o->C;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, magnitudes, in_bytes, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_out = out;
    if (mem_kind == FX_MEM_HOST) { if ((st = grow_bytes(&o->d_out, &o->out_cap, out_bytes)) != FX_OK) return st; d_out = static_cast<float*>(o->d_out); }
    hipLaunchKernelGGL(spectral_slope_kernel, dim3((unsigned) o->C), dim3(NT), SLOPE_LDS_PER_BIN * (size_t) num_bins, o->stream, static_cast<const float*>(da), num_bins, d_out);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) { HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, o->stream)); HIP_TRY(hipStreamSynchronize(o->stream)); }
    return FX_OK;
}

fx_status fx_offline_auto_correlation(fx_offline* o, float* data, int num_items, int* peak_bin, double* frequency, int mem_kind)
{
    fx_status st = check_args(o, data, peak_bin, mem_kind);
    if (st != FX_OK) return st;
    if (!frequency || num_items < 1) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument or num_items < 1");
    const size_t in_bytes = sizeof(float) * 2 * (size_t) o->C * num_items;
    const void *da, *db;
    if ((st = stage_in(o, mem_kind, data, in_bytes, nullptr, 0, &da, &db)) != FX_OK) return st;
    float* d_data = const_cast<float*>(static_cast<const float*>(da));
    int* d_peak = peak_bin; double* d_freq = frequency;
    if (mem_kind == FX_MEM_HOST) {
        if ((st = grow_bytes(&o->d_out, &o->out_cap, (sizeof(double) + sizeof(double)) * (size_t) o->C)) != FX_OK) return st;
        d_freq = static_cast<double*>(o->d_out);
        d_peak = reinterpret_cast<int*>(d_freq + o->C);
    }
    const long long items = (long long) o->C * num_items;
    hipLaunchKernelGGL(conjugate_multiplication_kernel, dim3((unsigned) ((items + NT - 1) / NT)), dim3(NT), 0, o->stream, d_data, items);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(auto_correlation_kernel, dim3((unsigned) o->C), dim3(NT), 0, o->stream, d_data, num_items, o->nyquist, d_peak, d_freq);
    HIP_TRY(hipGetLastError());
    if (mem_kind == FX_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(data, d_data, in_bytes, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipMemcpyAsync(peak_bin, d_peak, sizeof(int) * (size_t) o->C, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipMemcpyAsync(frequency, d_freq, sizeof(double) * (size_t) o->C, hipMemcpyDeviceToHost, o->stream));
        HIP_TRY(hipStreamSynchronize(o->stream));
    }
    return FX_OK;
}

fx_status fx_offline_sync(fx_offline* o)
{
    if (!o) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(o->device));
    HIP_TRY(hipStreamSynchronize(o->stream));
    return FX_OK;
}

} // extern "C"
