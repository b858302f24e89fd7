// fx_tail_kernels.hip.h -- fx_finalise_kernel, fx_epilogue_kernel, fx_history_kernel: scalar tails, smoothing, onset
// Included by fx_kernels.hip inside namespace fxk (one translation unit: every kernel sees the same
// inlined helpers); not a stand-alone header.  The __global__ kernels are compiled only where FX_WITH_TAIL_KERNELS is
// defined (one object of the library); the device functions are also what fx_hop_kernel finishes a hop with.

// a captured step (hipGraph) reads what changes from call to call from device memory
__device__ __forceinline__ EpilogueParams with_dyn(const EpilogueParams& in)
{
    EpilogueParams p = in;
    if (p.dyn) {
        p.nyquist = p.dyn->nyquist;
        p.frames_before = p.dyn->frames_before;
        p.hist_base = p.dyn->hist_base;
        p.onset_reset_frame = p.dyn->onset_reset_frame;
        p.onset_window = p.dyn->onset_window;
        p.onset_type = p.dyn->onset_type;
        p.onset_multiplier = p.dyn->onset_multiplier;
        epilogue_constants(p);                        // (the host's are for the nyquist it knew)
    }
    return p;
}

// ---------------------------------------------------------------------------------------------
// fx_finalise_kernel: thread = frame.  The scalar tail of calculateSpectralCharacteristicsFrom-
// Intermediates (ref SpectralCharacteristics.h:116-143), calculateNormalisedSpectralSlope (:189-199),
// the harmonic logs (ref HarmonicCharacteristics.h:101-105) and the slot mapping of
// RealTimeAnalyser.h:165-172,219-224.  Output: raw[C][T][12] with the onset slot still 0.
// ---------------------------------------------------------------------------------------------
// Five of the twelve values end in (float) log10(x) -- logRMS, flatness, centroid, HER, inharmonicity -- and those logarithms are
// half of this function's instructions.  `logs[5]` receives their arguments (1.0, whose logarithm is +0, where the reference
// leaves the slot at 0); the caller takes the logarithms -- one after the other in a thread that owns a frame, or side by
// side in five lanes of a wavefront that finishes one hop -- and stores them with finalise_logs().
enum { LOG_RMS = 0, LOG_FLATNESS = 1, LOG_CENTROID = 2, LOG_HER = 3, LOG_INHARM = 4, NUM_LOGS = 5 };
__device__ __forceinline__ void finalise_values(const EpilogueParams& p, const FramePart& f, float (&out)[FX_NUM_FEATURES], double (&logs)[NUM_LOGS])
{
    const int M = p.window / 2;
    const double nyquist = p.nyquist;
#pragma unroll
    for (int i = 0; i < FX_NUM_FEATURES; i++) out[i] = 0.0f;
    // a2, ref RealTimeAnalyser.h:207-208: log10 of a float, correctly rounded -- it also gates bins through eps, so it
    // must equal the CPU oracle's to the last bit (see oracle/fx_oracle.c)
    const float rms = (float) sqrt(f.sum_sq * (0.5 * p.inv_bins));               // sum / window: a power of two, the product is the quotient
    logs[LOG_RMS] = (double) (rms * 9.0f + 1.0f);
    logs[LOG_FLATNESS] = 1.0; logs[LOG_CENTROID] = 1.0; logs[LOG_HER] = 1.0; logs[LOG_INHARM] = 1.0;

    const bool spec = p.analysers & 1, harm = p.analysers & 2;
    // weightedMagnitudeSum (:95) from the moments: fc[m] = (m + 1/2) * frpb (:70), so sum fc * mag = frpb * (b1 + mag_sum / 2)
    const double wsum = p.frpb * (f.b1 + 0.5 * f.mag_sum);
    if (spec && f.mag_sum > 0.05) {                                            // :121-123
        const float centroid = (float) (wsum / f.mag_sum);                     // :127
        const double dcnt = (double) f.cnt;
        const double inv_n = 1.0 / (dcnt > 0.0 ? dcnt : 1.0);                  // :129-130
        // :57-60 `flatnessMagnitudeSum > epsilon`: every gated bin alone exceeds epsilon (>= 0) and the terms are positive,
        // so the test is "at least one bin passed the gate"
        const float flatness = f.cnt > 0.0f ? (float) (pow(f.prod, inv_n) / (inv_n * f.flat_sum)) : 0.0f;
        logs[LOG_FLATNESS] = (double) flatness * 9.0 + 1.0;                    // :132
        const float cc = centroid / (float) (nyquist / 2.0);                   // :133
        logs[LOG_CENTROID] = (double) (cc * 9.0f + 1.0f);                      // :134
        const double cn = (double) centroid / nyquist;
        const float max_spread = (float) (cn * (1.0 - cn));                    // :140
        // the spread's sum ((fc - centroid) / nyq)^2 * mag (:135-139) = (b2 + b1 + mag_sum/4) / M^2 - 2 cn (b1 + mag_sum/2) / M + cn^2 mag_sum;
        // where that form loses too much (the frame kernel's test) the kernel has taken the sum as the reference writes it
        double var = f.var;
        if (!f.refined) {
            const double cm = (double) centroid * p.inv_nyquist, rm = p.inv_bins;
            var = ((f.b2 + f.b1 + 0.25 * f.mag_sum) * rm - (cm + cm) * (f.b1 + 0.5 * f.mag_sum)) * rm + (cm * cm) * f.mag_sum;
        }
        out[FX_SPREAD] = (float) ((var / f.mag_sum) / (double) max_spread);    // :141
        out[FX_LER] = (float) (f.lhr / f.mag_sum);                             // :125
        const float max_flux = (float) (M * (M + 1)) / 2.0f;                   // :111
        out[FX_FLUX] = (float) (f.flux / (double) max_flux);
    }
    if (spec && f.max_e > 0.0001) {                                            // :165-167
        // normedEnergy = mag / max (ref :172): the sums over bins were taken before the division
        const double rmax = 1.0 / f.max_e;
        const double se = f.mag_sum * rmax;
        const double frpb = p.frpb;
        const double s1 = (wsum - (frpb / 2.0) * f.mag_sum) / frpb;            // sum m * mag
        const double ps = s1 * rmax;                                           // :175
        const double mean_e = se * p.inv_bins;                                 // :177  (/ M, a power of two)
        const double ev = f.vsum * rmax * rmax * p.inv_bins;                   // :187,190
        const double bin_std = p.bin_std, e_std = sqrt(ev);                    // :191-192
        const double r = (ps - ((double) M * mean_e * 0.5)) / (double) ((float) M - 1.0f) * e_std * bin_std;   // :195
        out[FX_SLOPE] = (float) (r * (bin_std / e_std));                       // :198
    }
    if (harm) {
        const double f0 = (nyquist * 2.0) / (double) f.lag;                    // ref PitchAnalyser.h:57
        out[FX_F0] = (float) (f0 / 5000.0);                                    // ref RealTimeAnalyser.h:165-166
    }
    if (harm && (f.flags & 1)) {
        double her = f.her_score / f.sum_normed;                               // ref HarmonicCharacteristics.h:186-188
        if (her > 1.0) her = 1.0;
        if (her < 0.0) her = 0.0;
        her = (double) (float) her;                                            // struct of floats, :197
        logs[LOG_HER] = her * 9.0 + 1.0;                                       // :101
        logs[LOG_INHARM] = f.inh * 9.0 + 1.0;                                  // :102
    }
}
__device__ __forceinline__ void finalise_logs(const float (&y)[NUM_LOGS], float (&out)[FX_NUM_FEATURES])
{
    out[FX_RMS] = y[LOG_RMS];
    out[FX_FLATNESS] = y[LOG_FLATNESS];
    out[FX_CENTROID] = y[LOG_CENTROID];
    out[FX_HER] = y[LOG_HER];
    out[FX_OER] = y[LOG_HER];                                                  // ref RealTimeAnalyser.h:171 writes HER into the OER slot
    out[FX_INHARM] = y[LOG_INHARM];
}

// one thread, one frame: the twelve raw values (onset slot still 0)
__device__ __forceinline__ void finalise_thread(const EpilogueParams& p, const FramePart& f, float (&out)[FX_NUM_FEATURES])
{
    double logs[NUM_LOGS];
    float y[NUM_LOGS];
    finalise_values(p, f, out, logs);
#pragma unroll
    for (int i = 0; i < NUM_LOGS; i++) y[i] = (float) log10(logs[i]);
    finalise_logs(y, out);
}

__device__ __forceinline__ void finalise_frame(const EpilogueParams& p, long long idx)
{
    const FramePart f = p.part[idx];
    float out[FX_NUM_FEATURES];
    finalise_thread(p, f, out);
    f4* dst = reinterpret_cast<f4*>(p.raw + idx * FX_NUM_FEATURES);
    dst[0] = f4{out[0], out[1], out[2], out[3]};
    dst[1] = f4{out[4], out[5], out[6], out[7]};
    dst[2] = f4{out[8], out[9], out[10], out[11]};
}

// A GROUP of lanes that works on one channel's hop together: the whole wavefront (the one-hop kernels: a workgroup is a channel)
// or a quarter of one (fx_tail_fused_kernel: four channels per wavefront).  `lane` is the lane in the wavefront throughout;
// group_lane() its index in the group, group_get() the value lane i of the caller's group holds.
template <int GROUP> __device__ __forceinline__ int group_lane(int lane) { return lane & (GROUP - 1); }
template <int GROUP> __device__ __forceinline__ float group_get(float v, int lane, int i)
{
    if (GROUP == 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
    return __shfl(v, (lane & ~(GROUP - 1)) + i, 64);
}

// The same for the one frame of a one-hop call, by a group of lanes: every lane forms the values (they are uniform in the group),
// lanes 0..4 take one logarithm each; `out` is the full vector in every lane.  Bit for bit what finalise_frame writes.
template <int GROUP = 64>
__device__ __forceinline__ void finalise_wave(const EpilogueParams& p, const FramePart& f, int lane, float (&out)[FX_NUM_FEATURES])
{
    double logs[NUM_LOGS];
    float y[NUM_LOGS];
    finalise_values(p, f, out, logs);
    double mine = logs[0];
#pragma unroll
    for (int i = 1; i < NUM_LOGS; i++) mine = group_lane<GROUP>(lane) == i ? logs[i] : mine;
    const float ym = (float) log10(mine);
#pragma unroll
    for (int i = 0; i < NUM_LOGS; i++) y[i] = group_get<GROUP>(ym, lane, i);
    finalise_logs(y, out);
}
#ifdef FX_WITH_TAIL_KERNELS
__global__ void __launch_bounds__(256)
fx_finalise_kernel(const EpilogueParams p_arg)
{
    const EpilogueParams p = with_dyn(p_arg);
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long) p.C * p.T) return;
    finalise_frame(p, idx);
}
#endif

// ---------------------------------------------------------------------------------------------
// smoothing (ValueHistory, ref RealTimeAudioAnalysis.h:40-96; AudioFeatures, ref
// RealTimeAnalyser.h:70-88) and onset detection (ref SpectralCharacteristics.h:249-306,
// RealTimeAnalyser.h:236-242).
//
// A ValueHistory of length L after its k-th insert holds the last min(k, L) inserted values,
// oldest first, padded on the left with the zeros it was created with; getTotal() adds them left to
// right in fp32.  Every smoothed value and every onset decision of frame t is therefore a pure
// function of the raw values of frames t-HLEN+1 .. t, and all (channel, frame) pairs are evaluated
// in parallel: thread = (channel, frame).  Frames before this call come from `hist`, a ring of HLEN rows per channel
// (row of the frame with global index g = g mod HLEN; EpilogueParams::hist_base = the row of this call's first frame).
// ---------------------------------------------------------------------------------------------
// Rows of raw values around a frame: this call's rows in `raw`, the rows before the call in the channel's ring `hist`
// (global memory or an LDS copy of the ring, same row positions), or an LDS tile holding rows tile_first .. of one
// channel with TILE_STRIDE floats per row (13: odd, so that threads one row apart fall on different banks).
constexpr int TILE_STRIDE = 13;
// ring row of the frame tau frames from this call's first one, -HLEN <= tau < 0 (hist_base in [0, HLEN))
__device__ __forceinline__ int hist_row_before(int hist_base, int tau) { const int r = hist_base + tau; return r < 0 ? r + HLEN : r; }
// rows before a frame that its smoothed values and its onset decision read: nine for the 10-deep slots (RealTimeAnalyser.h:73); the
// detector's L candidates each want an RMS mean that reaches nine frames back with one insert per hop, five with two
__device__ __forceinline__ int hist_rows_read(int onset_window) { const int n = onset_window + 8; return n < 9 ? 9 : (n > HLEN ? HLEN : n); }
struct RawView {
    const float* raw; const float* hist; int T; long long frames_before; int hist_base;
    const float* tile = nullptr; int tile_first = 0; bool tiled = false;     // (a flag, not `tile != nullptr`: a dynamic-LDS tile may sit at LDS address 0)
    // raw value of slot s at frame index tau relative to this call (tau may be negative);
    // frames before the stream began read as "not recorded"
    __device__ __forceinline__ bool valid(int tau) const { return frames_before + (long long) tau >= 0 && tau > -HLEN - 1; }
    __device__ __forceinline__ float get(int tau, int s) const
    {
        if (tiled) return tile[(tau - tile_first) * TILE_STRIDE + s];
        return tau >= 0 ? raw[(size_t) tau * FX_NUM_FEATURES + s] : hist[hist_row_before(hist_base, tau) * FX_NUM_FEATURES + s];
    }
};

// smoothed RMS as AudioFeatures::getValue(enRMS) returns it when `pushes_after` of frame tau's
// two RMS inserts have happened (shared AudioFeatures: two inserts per hop; isolated: one)
__device__ __forceinline__ float rms_value(const RawView& v, int tau, int order_mode, int pushes_of_tau)
{
    float total = 0.0f;
    long long recorded;
    if (order_mode == FX_ORDER_ISOLATED) {
        // spectral analyser's own AudioFeatures: one insert per frame, window = frames tau-9 .. tau
#pragma unroll
        for (int i = 0; i < 10; i++) { const int f = tau - 9 + i; total += v.valid(f) ? v.get(f, FX_RMS) : 0.0f; }
        recorded = v.frames_before + tau + 1;
    } else {
        // inserts are numbered 2*g (first analyser of frame g) and 2*g+1; the newest insert present
        // is 2*tau + pushes_of_tau - 1 and the history holds the 10 newest
        const long long newest = 2 * (v.frames_before + tau) + pushes_of_tau - 1;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const long long q = newest - 9 + i;                     // global insert index
            const long long g = q >> 1;                             // its frame (floor for q >= 0)
            const int f = (int) (g - v.frames_before);
            total += (q >= 0 && v.valid(f)) ? v.get(f, FX_RMS) : 0.0f;
        }
        recorded = newest + 1;
    }
    if (recorded > 10) recorded = 10;
    return total / (float) recorded;
}

// OnsetDetector::detectOnset, ref SpectralCharacteristics.h:249-306, for frame t.  The detector's histories
// hold (getValue(enFlux), getValue(enRMS)) as seen by detectOnset() of each frame
// (ref RealTimeAnalyser.h:236-242): flux of that frame, and the RMS mean at that moment.
// `amp` (optional): amp[f - amp_first] = rms_value(v, f, order_mode, pushes at detection) for the frames around t,
// evaluated once per frame by the caller's workgroup instead of once per (frame, candidate) here.
__device__ __forceinline__ bool detect_onset(const EpilogueParams& p, const RawView& v, int t, int order_mode, bool spec,
                                             const float* amp = nullptr, int amp_first = 0)
{
    const int L = p.onset_window;
    const int rms_pushes_at_detect = (order_mode == FX_ORDER_HARMONIC_THEN_SPECTRAL) ? 2 : 1;
    const long long g = p.frames_before + t;
    long long recorded = g - p.onset_reset_frame + 1;
    if (recorded > L) recorded = L;
    bool onset = false;
    if (spec && recorded >= L && L > 0) {                           // :253-258 both histories full
        int cand = L - 1;                                           // :263-266
        const bool use_amp = p.onset_type == FX_ONSET_AMPLITUDE || p.onset_type == FX_ONSET_COMBINATION;
        const bool use_flux = p.onset_type == FX_ONSET_SPECTRAL || p.onset_type == FX_ONSET_COMBINATION;
        if (use_flux) cand = L / 2;
        const float cand_amp = amp ? amp[t - L + 1 + cand - amp_first] : rms_value(v, t - L + 1 + cand, order_mode, rms_pushes_at_detect);
        const float cand_sf = (0.0f + v.get(t - L + 1 + cand, FX_FLUX)) / 1.0f;
        bool ok = !(cand_amp < 0.01f);                              // :271-274
        float tot_amp = 0.0f, tot_flux = 0.0f;
#pragma unroll 1
        for (int i = 0; i < L; i++) {                               // :260-261 totals, :276-289 neighbours
            const int f = t - L + 1 + i;
            const float amp_i = amp ? amp[f - amp_first] : rms_value(v, f, order_mode, rms_pushes_at_detect);
            const float flx_i = (0.0f + v.get(f, FX_FLUX)) / 1.0f;
            tot_amp += amp_i;
            tot_flux += flx_i;
            if (i != cand) {
                if (amp_i >= cand_amp && use_amp) ok = false;
                if (flx_i >= cand_sf && use_flux) ok = false;
            }
        }
        const float mean_flux = tot_flux / (float) recorded;
        const float mean_amp = tot_amp / (float) recorded;
        const bool on_sf = cand_sf > mean_flux * p.onset_multiplier;    // :291-292
        const bool on_amp = cand_amp > mean_amp * p.onset_multiplier;
        bool res = false;
        if (p.onset_type == FX_ONSET_AMPLITUDE) res = on_amp;
        else if (p.onset_type == FX_ONSET_SPECTRAL) res = on_sf;
        else if (p.onset_type == FX_ONSET_COMBINATION) res = on_amp && on_sf;
        onset = ok && res;
    }
    return onset;
}

template <bool TILED = false>
__device__ __forceinline__ void epilogue_frame(const EpilogueParams& p, int c, int t, const float* tile = nullptr, int tile_first = 0,
                                               const float* amp = nullptr, int amp_first = 0)
{
    RawView v;
    v.raw = p.raw + (size_t) c * p.T * FX_NUM_FEATURES;
    v.hist = p.hist + (size_t) c * HLEN * FX_NUM_FEATURES;
    v.T = p.T;
    v.frames_before = p.frames_before;
    v.hist_base = p.hist_base;
    v.tile = tile;
    v.tile_first = tile_first;
    v.tiled = TILED;

    float sm[FX_NUM_FEATURES];
    float rw[FX_NUM_FEATURES];
#pragma unroll
    for (int s = 0; s < FX_NUM_FEATURES; s++) rw[s] = v.get(t, s);

    const bool spec = p.analysers & 1, harm = p.analysers & 2;
    // with a single analyser the RMS slot gets one insert per hop, like an isolated AudioFeatures
    const int order_mode = (spec && harm) ? p.order_mode : FX_ORDER_ISOLATED;
    const float never = __int_as_float(0x7fc00000);                  // getValue() of a slot nobody wrote: 0.0f / 0
    // 10-deep slots (ref RealTimeAnalyser.h:73)
    long long rec10 = p.frames_before + t + 1; if (rec10 > 10) rec10 = 10;
#pragma unroll
    for (int s = 0; s < FX_NUM_FEATURES; s++) {
        if (s == FX_ONSET || s == FX_FLUX || s == FX_RMS) continue;
        const bool harm_slot = s == FX_F0 || s == FX_HER || s == FX_OER || s == FX_INHARM;
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; i++) { const int f = t - 9 + i; total += v.valid(f) ? v.get(f, s) : 0.0f; }
        sm[s] = (harm_slot ? harm : spec) ? total / (float) rec10 : never;
    }
    sm[FX_FLUX] = spec ? (0.0f + rw[FX_FLUX]) / 1.0f : never;        // history length 1
    // RMS after every analyser of this hop has inserted (what the OSC timer samples)
    sm[FX_RMS] = rms_value(v, t, order_mode, 2);

    const bool onset = detect_onset(p, v, t, order_mode, spec, amp, amp_first);
    rw[FX_ONSET] = onset ? 1.0f : 0.0f;
    sm[FX_ONSET] = spec ? (0.0f + rw[FX_ONSET]) / 1.0f : never;      // history length 1

    const size_t o = ((size_t) c * p.T + t) * FX_NUM_FEATURES;
    if (p.out_raw) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.out_raw[o + s] = rw[s];
    }
    if (p.out_smoothed) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.out_smoothed[o + s] = sm[s];
    }
    if (t == p.T - 1) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.latest[(size_t) c * FX_NUM_FEATURES + s] = sm[s];
    }
}

// detect_onset for a group of lanes: the L <= 32 candidates' (RMS mean, flux) pairs -- each RMS mean is ten history
// reads and the index arithmetic of the double insert -- are evaluated by the group's lanes side by side into `scratch` (the
// group's own [64] floats of LDS); the detector's serial loop then only reads them back.  Same values, same order of
// additions; the result is uniform in the group.
template <int GROUP = 64>
__device__ __forceinline__ bool detect_onset_wave(const EpilogueParams& p, const RawView& v, int t, int order_mode, bool spec, int lane, float* scratch)
{
    const int L = p.onset_window;
    const int rms_pushes_at_detect = (order_mode == FX_ORDER_HARMONIC_THEN_SPECTRAL) ? 2 : 1;
    const long long g = p.frames_before + t;
    long long recorded = g - p.onset_reset_frame + 1;
    if (recorded > L) recorded = L;
    if (!(spec && recorded >= L && L > 0)) return false;            // :253-258 both histories full
    for (int k = group_lane<GROUP>(lane); k < L; k += GROUP) {
        const int f = t - L + 1 + k;
        scratch[k] = rms_value(v, f, order_mode, rms_pushes_at_detect);
        scratch[32 + k] = (0.0f + v.get(f, FX_FLUX)) / 1.0f;
    }
    wave_fence();
    int cand = L - 1;                                               // :263-266
    const bool use_amp = p.onset_type == FX_ONSET_AMPLITUDE || p.onset_type == FX_ONSET_COMBINATION;
    const bool use_flux = p.onset_type == FX_ONSET_SPECTRAL || p.onset_type == FX_ONSET_COMBINATION;
    if (use_flux) cand = L / 2;
    const float cand_amp = scratch[cand];
    const float cand_sf = scratch[32 + cand];
    bool ok = !(cand_amp < 0.01f);                                  // :271-274
    float tot_amp = 0.0f, tot_flux = 0.0f;
#pragma unroll 1
    for (int i = 0; i < L; i++) {                                   // :260-261 totals, :276-289 neighbours
        const float amp_i = scratch[i];
        const float flx_i = scratch[32 + i];
        tot_amp += amp_i;
        tot_flux += flx_i;
        if (i != cand) {
            if (amp_i >= cand_amp && use_amp) ok = false;
            if (flx_i >= cand_sf && use_flux) ok = false;
        }
    }
    const float mean_flux = tot_flux / (float) recorded;
    const float mean_amp = tot_amp / (float) recorded;
    const bool on_sf = cand_sf > mean_flux * p.onset_multiplier;    // :291-292
    const bool on_amp = cand_amp > mean_amp * p.onset_multiplier;
    bool res = false;
    if (p.onset_type == FX_ONSET_AMPLITUDE) res = on_amp;
    else if (p.onset_type == FX_ONSET_SPECTRAL) res = on_sf;
    else if (p.onset_type == FX_ONSET_COMBINATION) res = on_amp && on_sf;
    return ok && res;
}

// The same for the single frame of a one-hop call (T == 1) with the twelve slots spread over twelve lanes: lane s
// evaluates slot s (the onset lane the detector, the RMS lane the double-insert mean), so the hop's tail takes the time
// of its longest slot instead of the sum of all.  Same expressions, same order of the fp32 additions as epilogue_frame.
// All lanes of the calling wavefront must take part (the onset detector's candidates are spread over the group's lanes);
// `live` = false: a group beyond the last channel, which keeps in step and stores nothing.
template <int GROUP = 64>
__device__ __forceinline__ void epilogue_hop(const EpilogueParams& p, int c, int lane, float* scratch, bool live = true)
{
    RawView v;
    v.raw = p.raw + (size_t) c * p.T * FX_NUM_FEATURES;
    v.hist = p.hist + (size_t) c * HLEN * FX_NUM_FEATURES;
    v.T = p.T;
    v.frames_before = p.frames_before;
    v.hist_base = p.hist_base;
    const int t = 0, s = group_lane<GROUP>(lane);
    const bool spec = p.analysers & 1, harm = p.analysers & 2;
    const int order_mode = (spec && harm) ? p.order_mode : FX_ORDER_ISOLATED;
    const float never = __int_as_float(0x7fc00000);
    const bool onset = detect_onset_wave<GROUP>(p, v, t, order_mode, spec, lane, scratch);       // all lanes
    const bool mine = s < FX_NUM_FEATURES;
    float rw = 0.0f, sm = 0.0f;
    if (mine) {
        rw = v.get(t, s);
        if (s == FX_ONSET) {
            rw = onset ? 1.0f : 0.0f;
            sm = spec ? (0.0f + rw) / 1.0f : never;
        } else if (s == FX_FLUX) {
            sm = spec ? (0.0f + rw) / 1.0f : never;
        } else if (s == FX_RMS) {
            sm = rms_value(v, t, order_mode, 2);
        } else {
            long long rec10 = p.frames_before + t + 1; if (rec10 > 10) rec10 = 10;
            const bool harm_slot = s == FX_F0 || s == FX_HER || s == FX_OER || s == FX_INHARM;
            float total = 0.0f;
#pragma unroll
            for (int i = 0; i < 10; i++) { const int f = t - 9 + i; total += v.valid(f) ? v.get(f, s) : 0.0f; }
            sm = (harm_slot ? harm : spec) ? total / (float) rec10 : never;
        }
    }
    const size_t o = p.out_stride ? ((size_t) c * (size_t) p.out_stride + (size_t) p.out_t0) * FX_NUM_FEATURES : ((size_t) c * p.T + t) * FX_NUM_FEATURES;
    // (Round 3 also tried the twelve slots as three 16-byte stores per vector -- a one-hop call's results go to a pinned host slot,
    // every store a transaction across PCIe: no measurable difference in the hop's round trip, not kept.)
    if (!mine || !live) return;
    if (p.out_raw) p.out_raw[o + s] = rw;
    if (p.out_smoothed) p.out_smoothed[o + s] = sm;
    p.latest[(size_t) c * FX_NUM_FEATURES + s] = sm;
}

// thread = (channel, frame); a block is EPI_TILE consecutive frames of ONE channel and first stages the rows it needs
// (its own and the HLEN - 1 before them) in LDS: each thread's ~70 reads of neighbouring rows then hit LDS instead of L2
constexpr int EPI_TILE = 256;
#ifdef FX_WITH_TAIL_KERNELS
__global__ void __launch_bounds__(EPI_TILE)
fx_epilogue_kernel(const EpilogueParams p_arg)
{
    __shared__ float tile[(EPI_TILE + HLEN) * TILE_STRIDE];
    __shared__ float amp[EPI_TILE + MAX_ONSET_WINDOW];
    const EpilogueParams p = with_dyn(p_arg);
    const int tiles = (p.T + EPI_TILE - 1) / EPI_TILE;
    const int c = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * EPI_TILE;
    const int first = t0 - HLEN;                                     // first row of the tile (may be before the call)
    const int rows = (p.T - t0 < EPI_TILE ? p.T - t0 : EPI_TILE) + HLEN;
    const float* raw = p.raw + (size_t) c * p.T * FX_NUM_FEATURES;
    const float* hist = p.hist + (size_t) c * HLEN * FX_NUM_FEATURES;
    for (int i = threadIdx.x; i < rows * FX_NUM_FEATURES; i += EPI_TILE) {
        const int r = i / FX_NUM_FEATURES, s = i % FX_NUM_FEATURES, tau = first + r;
        // rows more than HLEN before the call do not exist (and are never read: RawView::valid)
        tile[r * TILE_STRIDE + s] = tau >= 0 ? raw[(size_t) tau * FX_NUM_FEATURES + s]
                                             : (tau >= -HLEN ? hist[hist_row_before(p.hist_base, tau) * FX_NUM_FEATURES + s] : 0.0f);
    }
    __syncthreads();
    // The detector of frame t compares the RMS means of frames t - L + 1 .. t as they stood at each frame's own
    // detection (ref RealTimeAnalyser.h:236-242): one value per FRAME, wanted by up to L frames -- evaluated once here
    // (ten history reads and the double-insert index arithmetic each) for this tile's frames and the MAX_ONSET_WINDOW
    // before them.
    {
        RawView v;
        v.raw = raw; v.hist = hist; v.T = p.T; v.frames_before = p.frames_before; v.hist_base = p.hist_base; v.tile = tile; v.tile_first = first; v.tiled = true;
        const bool both = (p.analysers & 1) && (p.analysers & 2);
        const int order_mode = both ? p.order_mode : FX_ORDER_ISOLATED;
        const int pushes = (order_mode == FX_ORDER_HARMONIC_THEN_SPECTRAL) ? 2 : 1;
        for (int j = threadIdx.x; j < EPI_TILE + MAX_ONSET_WINDOW; j += EPI_TILE) {
            const int f = t0 - MAX_ONSET_WINDOW + j;
            amp[j] = (f < p.T) ? rms_value(v, f, order_mode, pushes) : 0.0f;
        }
    }
    __syncthreads();
    const int t = t0 + (int) threadIdx.x;
    if (t < p.T) epilogue_frame<true>(p, c, t, tile, first, amp, t0 - MAX_ONSET_WINDOW);
}
#endif

// The call's newest min(T, HLEN) frames go to their rows of the channel's ring: row (hist_base + tau) mod HLEN for the frame
// tau of this call.  Nothing else moves -- a one-hop call writes ONE row per channel (until round 4 the whole table was
// copied from a `hist_in` to a `hist_out` to shift it by the call's frames: 33 of a live 8192-channel call's 107 us).
__device__ __forceinline__ int hist_rows_written(int T) { return T < HLEN ? T : HLEN; }
__device__ __forceinline__ void history_value(const EpilogueParams& p, int c, int h, int s)
{
    const int tau = p.T - hist_rows_written(p.T) + h;                // frame relative to this call, 0 <= h < hist_rows_written
    p.hist[((size_t) c * HLEN + (size_t) ((p.hist_base + tau) % HLEN)) * FX_NUM_FEATURES + s] = p.raw[((size_t) c * p.T + tau) * FX_NUM_FEATURES + s];
}

#ifdef FX_WITH_TAIL_KERNELS
__global__ void __launch_bounds__(256)
fx_history_kernel(const EpilogueParams p_arg)
{
    const EpilogueParams p = with_dyn(p_arg);
    const int rows = hist_rows_written(p.T);
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (p.clear_queue && idx < p.clear_count) p.clear_queue[idx] = 0u;      // (the frame kernel that used them finished before this step's tail began)
    if (idx >= (long long) p.C * rows * FX_NUM_FEATURES) return;
    const int s = (int) (idx % FX_NUM_FEATURES);
    const int h = (int) ((idx / FX_NUM_FEATURES) % rows);
    const int c = (int) (idx / ((long long) FX_NUM_FEATURES * rows));
    history_value(p, c, h, s);
}
#endif

// The three kernels above in one launch for calls of a few frames per channel (one hop per call: the reference's own cadence,
// AudioDataCollector.h:66-94), where two kernel boundaries cost more than the work.  A QUARTER of a wavefront per channel: the
// tail of a hop has 12 slots, 5 logarithms and (by default) 5 onset candidates to spread over lanes, which 16 lanes hold as well
// as 64 -- and a call over thousands of channels then issues a quarter of the instructions.  (Measured at 8192 channels x one
// 1024-pt hop: a wavefront per channel 24.6 us; a THREAD per channel -- the other extreme, 64 channels per instruction --
// 33 us, all of it the latency of one lane's chain of ~5 000 dependent fp64 instructions; profiles/r04_live_cadence.txt.)
// Lanes 0..T-1 of a group finalise the call's frames; the raw values of the frames before the call that the smoothing reads
// (hist_rows_read: 13 of the ring's 48 rows with the default onset window) and of the call's own frames are staged in LDS (one
// round trip to memory instead of the ~70 dependent ones of the smoothing / onset loops), then a lane per slot (one hop) or a
// lane per frame smooths from there, and the call's own rows go to the ring.
constexpr int TAIL_GROUP = 16;                          // lanes per channel
constexpr int TAIL_CHANNELS = 64 / TAIL_GROUP;          // channels per wavefront = per workgroup
static_assert(FUSED_TAIL_MAX_FRAMES <= TAIL_GROUP && FX_NUM_FEATURES <= TAIL_GROUP && NUM_LOGS <= TAIL_GROUP, "a lane per frame / slot / logarithm");

// ONE hop per channel, the four channels of a wavefront (first channel c_first): ring rows to LDS, the scalar tail by the whole group
// straight into the LDS row, a lane per smoothed slot, this hop's row to the ring.  The wavefront's LDS: s_hist[4][HLEN * 12],
// s_raw[4][16], s_scratch[4][64] floats (ONE_HOP_TAIL_BYTES).  Called by every lane of the wavefront; the two barriers are the
// workgroup's (fx_tail_fused_kernel: a workgroup IS one wavefront; fx_frame_tail_kernel: its tail wavefronts all pass here).
constexpr int ONE_HOP_TAIL_FLOATS = TAIL_CHANNELS * (HLEN * FX_NUM_FEATURES + 16 + 64);
constexpr int ONE_HOP_TAIL_BYTES = 4 * ONE_HOP_TAIL_FLOATS;
// `c_end`: one past the last channel this wavefront may finish -- the context's channel count, or the end of the calling workgroup's own
// block of channels (fx_frame_tail_kernel with a channel count per workgroup that is no multiple of TAIL_CHANNELS: the group beyond
// it belongs to the NEXT workgroup, which may still be analysing that channel's frame).
__device__ __forceinline__ void tail_one_hop(const EpilogueParams& p, int c_first, int c_end, int lane, float* lds)
{
    const int g = lane / TAIL_GROUP, gl = lane % TAIL_GROUP;
    float* s_hist = lds + g * (HLEN * FX_NUM_FEATURES);
    float* s_raw = lds + TAIL_CHANNELS * (HLEN * FX_NUM_FEATURES) + g * 16;
    float* s_scratch = lds + TAIL_CHANNELS * (HLEN * FX_NUM_FEATURES + 16) + g * 64;
    const int c_mine = c_first + g;
    const bool live = c_mine < c_end;
    const int c = live ? c_mine : c_end - 1;              // a group beyond the last channel keeps in step on the last one and stores nothing
    const float* ring = p.hist + (size_t) c * HLEN * FX_NUM_FEATURES;
    {
        // the LDS copy keeps the ring's row positions; rows nobody reads are not fetched (a row is 48 bytes: three 16-byte pieces)
        const int need = hist_rows_read(p.onset_window);
        for (int i = gl; i < need * 3; i += TAIL_GROUP) {
            const int r = hist_row_before(p.hist_base, -need + i / 3), q = i % 3;
            reinterpret_cast<uint4*>(s_hist)[r * 3 + q] = reinterpret_cast<const uint4*>(ring)[r * 3 + q];
        }
    }
    {
        // the scalar tail by the whole group (its logarithms side by side), straight into the LDS row -- the raw vector never makes
        // the trip through global memory
        const FramePart f = p.part[c];
        float out[FX_NUM_FEATURES];
        finalise_wave<TAIL_GROUP>(p, f, lane, out);
        if (gl < FX_NUM_FEATURES) {
            float mine = out[0];
#pragma unroll
            for (int k = 1; k < FX_NUM_FEATURES; k++) mine = gl == k ? out[k] : mine;
            s_raw[gl] = mine;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_barrier();
    EpilogueParams q = p;                                 // the smoothing reads channel c's rows through (raw + c*12, hist + c*HLEN*12)
    q.raw = s_raw - (size_t) c * FX_NUM_FEATURES;
    q.hist = s_hist - (size_t) c * HLEN * FX_NUM_FEATURES;
    epilogue_hop<TAIL_GROUP>(q, c, lane, s_scratch, live);   // a lane per slot, the onset detector's candidates side by side
    q.hist = p.hist;
    if (live && gl < FX_NUM_FEATURES) history_value(q, c, 0, gl);
}

// fx_frame_kernel's one-frame form (DIRECT: one wavefront per channel, flux state in global memory) and the hop's tail in ONE launch:
// when the frames are done, the first ceil(channels / 4) wavefronts of the workgroup finish its channels' hops as tail_one_hop does,
// in the transform buffers nobody needs any more; the others leave.  What it buys a live call over thousands of channels is the
// second launch and its ramp: the tails of the workgroups that finish first run beside the frames of those still at work.
#if FX_PART != 3
template <int N, bool BLOCKS = false>
__global__ void __launch_bounds__(Occ<N>::MAX_THREADS, Occ<N>::WAVES_PER_SIMD)
fx_frame_tail_kernel(const FrameParams p, const EpilogueParams ep_arg)
{
    frame_kernel_body<N, true, true, true, BLOCKS>(p);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // the frames' records (global memory) ...
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");            // ... are read below by other wavefronts of this workgroup
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int tail_waves = (p.ch_per_wg + TAIL_CHANNELS - 1) / TAIL_CHANNELS;
    if (wave >= tail_waves) return;
    const EpilogueParams ep = with_dyn(ep_arg);
    float* lds = reinterpret_cast<float*>(smem + sizeof(f2) * FrameLds<N>::TW_ENTRIES) + (size_t) wave * ONE_HOP_TAIL_FLOATS;
    const int c_begin = (int) blockIdx.x * p.ch_per_wg;
    tail_one_hop(ep, c_begin + TAIL_CHANNELS * wave, min(c_begin + p.ch_per_wg, p.C), (int) (threadIdx.x & 63), lds);
}
#endif
#ifdef FX_WITH_TAIL_KERNELS
__global__ void __launch_bounds__(64)
fx_tail_fused_kernel(const EpilogueParams p_arg)
{
    // [channels of the wavefront] ring copy | raw rows of the call's frames | onset scratch; one hop per call: tail_one_hop's layout
    __shared__ __attribute__((aligned(16))) float s_all[TAIL_CHANNELS * (HLEN * FX_NUM_FEATURES + FUSED_TAIL_MAX_FRAMES * FX_NUM_FEATURES + 64)];
    static_assert(FUSED_TAIL_MAX_FRAMES * FX_NUM_FEATURES >= 16, "tail_one_hop fits the same memory");
    const EpilogueParams p = with_dyn(p_arg);
    const int lane = threadIdx.x, g = lane / TAIL_GROUP, gl = lane % TAIL_GROUP;
    // a call cut into work units whose tail is this kernel (<= FUSED_TAIL_MAX_FRAMES frames, cut by an explicit plan or small units):
    // its ticket counter and hand-over counts go back to zero here, as fx_history_kernel does for longer calls (16 threads per channel)
    {
        const long long idx = (long long) blockIdx.x * 64 + lane;
        if (p.clear_queue && idx < p.clear_count) p.clear_queue[idx] = 0u;
    }
    if (p.T == 1) { tail_one_hop(p, blockIdx.x * TAIL_CHANNELS, p.C, lane, s_all); return; }
    float* s_hist = s_all + g * (HLEN * FX_NUM_FEATURES);
    float* s_raw = s_all + TAIL_CHANNELS * (HLEN * FX_NUM_FEATURES) + g * (FUSED_TAIL_MAX_FRAMES * FX_NUM_FEATURES);
    const int c_mine = blockIdx.x * TAIL_CHANNELS + g;
    const bool live = c_mine < p.C;
    const int c = live ? c_mine : p.C - 1;                // a group beyond the last channel keeps in step on the last one and stores nothing
    const float* ring = p.hist + (size_t) c * HLEN * FX_NUM_FEATURES;
    {
        const int need = hist_rows_read(p.onset_window);            // what the call's FIRST frame reaches back; later frames less
        for (int i = gl; i < need * 3; i += TAIL_GROUP) {
            const int r = hist_row_before(p.hist_base, -need + i / 3), q = i % 3;
            reinterpret_cast<uint4*>(s_hist)[r * 3 + q] = reinterpret_cast<const uint4*>(ring)[r * 3 + q];
        }
    }
    if (gl < p.T && live) finalise_frame(p, (long long) c * p.T + gl);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");           // the wave's raw values are written before any lane reads them back
    __builtin_amdgcn_s_barrier();
    for (int i = gl; i < p.T * FX_NUM_FEATURES; i += TAIL_GROUP) s_raw[i] = p.raw[(size_t) c * p.T * FX_NUM_FEATURES + i];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_barrier();
    // the smoothing reads channel c's rows through (raw + c*T*12, hist + c*HLEN*12): point those at the LDS copies
    EpilogueParams q = p;
    q.raw = s_raw - (size_t) c * p.T * FX_NUM_FEATURES;
    q.hist = s_hist - (size_t) c * HLEN * FX_NUM_FEATURES;
    if (gl < p.T && live) epilogue_frame(q, c, gl);                  // a lane per frame
    // this call's rows of the ring: T <= FUSED_TAIL_MAX_FRAMES different rows, which held frames HLEN before these -- further back
    // than anything reads (hist_rows_read <= 40), and what was read went through the LDS copy above in any case
    q.hist = p.hist;
    if (live)
        for (int i = gl; i < p.T * FX_NUM_FEATURES; i += TAIL_GROUP) history_value(q, c, i / FX_NUM_FEATURES, i % FX_NUM_FEATURES);
}
#endif
