// fx_capi.cpp -- extern "C" shim declared in include/fx.h.  Host-side plumbing only: device
// buffers, per-channel state residency in HBM, stream ordering, launch of the gfx950 kernels.
// There is no CPU path: without a usable gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "fx_kernels.h"

#include "fx_context.h"

thread_local std::string g_fx_err;

fx_status fx_fail(fx_status code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_fx_err = buf;
    return code;
}

namespace {

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// Which kernel family a context runs (include/fx.h, FX_LOW_LATENCY): frames across a PAIR of wavefronts -- windows of 2048 / 4096
// points with both analysers -- when the create flag asks for it, or when the tuning knob forces either (experiments, tests).
bool uses_pairs(const fx_context* c, int waves_per_frame)
{
    if (!fxk::pair_kernel_available(c->N) || (c->flags & (FX_SPECTRAL_ONLY | FX_HARMONIC_ONLY))) return false;
    return waves_per_frame == 2 || (waves_per_frame == 0 && (c->flags & FX_LOW_LATENCY));
}
bool known_format(int f) { return f == FX_SAMPLE_F32 || f == FX_SAMPLE_F16 || f == FX_SAMPLE_S16 || f == FX_SAMPLE_S24; }
size_t sample_size(int f) { return f == FX_SAMPLE_F32 ? 4 : (f == FX_SAMPLE_S24 ? 3 : 2); }

} // namespace


namespace {

fx_status zero_state(fx_context* c)
{
    const size_t half = (size_t) c->C * (c->N / 2);
    HIP_TRY(hipMemsetAsync(c->d_prev, 0, half * sizeof(float), c->stream));
    for (int i = 0; i < 2; i++) HIP_TRY(hipMemsetAsync(c->d_tail[i], 0, half * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_hist, 0, (size_t) c->C * fxk::HLEN * FX_NUM_FEATURES * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_latest, 0, (size_t) c->C * FX_NUM_FEATURES * sizeof(float), c->stream));
    // the work units' ticket counter and hand-over counts: zero between calls (a cut call's last kernel leaves them so; a call that
    // failed half way is followed by this reset)
    HIP_TRY(hipMemsetAsync(c->d_queue, 0, sizeof(unsigned) * (1 + (size_t) c->C), c->stream));
    c->frames_seen = 0;
    c->onset_reset_frame = 0;
    c->carry_count = 0;                     // (a freshly constructed AudioDataCollector: nothing written, nothing pending)
    return FX_OK;
}

// Scratch that follows the largest call seen.  Growing frees and reallocates (hipFree waits for the device), so a
// buffer that has grown once grows by at least half again: a caller ramping its batch size up does not pay per call.
template <typename T> fx_status grow(T** ptr, size_t* cap, size_t need)
{
    if (need <= *cap) return FX_OK;
    size_t want = need;
    if (*ptr) {
        if (want < *cap + *cap / 2) want = *cap + *cap / 2;
        // the pointer is forgotten BEFORE the free is attempted: a free that reports a failure must not be repeated by the next call
        // (found by tests/cpp/host_sanitize.cpp: the old order freed the block twice)
        T* old = *ptr;
        *ptr = nullptr;
        *cap = 0;
        HIP_TRY(hipFree(old));
    }
    *ptr = nullptr;
    *cap = 0;
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && want != need) { (void) hipGetLastError(); want = need; e = hipMalloc(&p, want); }
    if (e != hipSuccess) return fx_fail(e == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e));
    *ptr = static_cast<T*>(p);
    *cap = want;
    return FX_OK;
}

} // namespace

// Long calls are cut in time as well (FrameParams::num_chunks): work units of ~200 us -- long enough to carry a
// workgroup's prologue and the hand-over, short enough for many rounds of them.  Measured (1024 ch x 512 frames, best
// of three interleaved runs): 8 frames per wavefront for the full bundle at 1024 points (2.80 against 3.01 ms uncut),
// twice that with the harmonic analyser alone (1.93 / 2.04), more for the small windows (512 points +3 %, 256 points
// +2 %); nothing for the spectral analyser alone, and a LOSS of 3-8 % at 2048 and 4096 points, whose workgroups carry
// a 16-32 KB twiddle table each and already run in 4-8 rounds at their usual shapes: those are never cut.  Calls of up
// to 8 units are cut into equal units; longer ones into units of decreasing length -- a third of what is left each
// time (at most four units' worth), down to a quarter unit -- long units first (little overhead), short ones last (the
// launch's tail is one short unit deep): 512 frames = 168, 112, 80, 48, 32, 24, 16, 16, 16 (2.74 against 2.80 ms for
// eight units of 64 at the bench shape).  fx_tuning::frames_per_unit overrides the unit (0 = never cut), fx_tuning::unit_plan
// gives the lengths outright (experiments).  Host-only arithmetic, pure (no environment, no device): declared in
// include/fx.h so that the CPU tests can hold it to its invariants.
// the reference's table for a window size: phase in double, entries rounded to float (JUCE 4.2 FFT::FFTConfig)
static std::vector<float> reference_twiddles(int window_size)
{
    std::vector<float> tw(2 * (size_t) window_size);
    for (int i = 0; i < window_size; i++) {
        const double phase = -2.0 * 3.14159265358979323846 * i / window_size;
        tw[2 * i] = (float) std::cos(phase);
        tw[2 * i + 1] = (float) std::sin(phase);
    }
    return tw;
}

extern "C" int fx_twiddle_symmetry(int window_size)
{
    if (!is_pow2(window_size) || window_size < 256 || window_size > 4096) return 0;
    const std::vector<float> tw = reference_twiddles(window_size);
    std::vector<float> ordered(tw.size());
    float first[18];
    fxk::build_pass_twiddles(window_size, tw.data(), ordered.data());
    fxk::fill_first_pass_twiddles(window_size, ordered.data(), first);
    return (fxk::first_pass_twiddles_hermitian(window_size, first) ? 1 : 0) | (fxk::twiddles_have_quarter_turn(window_size, tw.data()) ? 2 : 0);
}

extern "C" int fx_plan_units(int window_size, unsigned flags, int waves_per_channel, int num_frames, const fx_tuning* tuning, int* sizes, int cap)
{
    const int k = waves_per_channel, T = num_frames;
    if (!sizes || cap < 1 || k < 1 || T < 1) return 0;
    if (T < 2 * k) { sizes[0] = T; return 1; }          // (a few hops: nothing to cut, and the one-hop path is latency-critical)
    int per_wave = window_size <= 256 ? 32 : (window_size == 512 ? 16 : 8);
    if (flags & FX_HARMONIC_ONLY) per_wave *= 2;
    int unit = (window_size > 1024 || (flags & FX_SPECTRAL_ONLY)) ? 0 : k * per_wave;
    if (tuning && tuning->frames_per_unit >= 0) unit = tuning->frames_per_unit;
    int n = 0;
    if (tuning && tuning->unit_plan_len > 0 && tuning->unit_plan_len <= cap && tuning->unit_plan_len <= FX_MAX_UNITS) {
        int sum = 0;
        for (int i = 0; i < tuning->unit_plan_len; i++) { const int v = tuning->unit_plan[i]; if (v <= 0) { sum = -1; break; } sizes[n++] = v; sum += v; }
        if (sum != T) n = 0;
    }
    if (n == 0 && unit >= k && unit > 0) {
        if (T >= 8 * unit) {
            const int least = unit / 4 > k ? unit / 4 / k * k : k;
            for (int rem = T; rem > 0 && n < cap; ) {
                int sz = (rem / 3 + k / 2) / k * k;
                if (sz < least) sz = least;
                if (sz > 4 * unit && T <= 4 * unit * (cap - 10)) sz = 4 * unit;       // (no unit longer than ~1 ms)
                if (rem - sz < least || n == cap - 1) sz = rem;
                sizes[n++] = sz;
                rem -= sz;
            }
        } else {
            const int cnt = (2 * T + unit) / (2 * unit);                  // T / unit, rounded
            if (cnt >= 2 && cnt <= cap) {
                int per = (T + cnt - 1) / cnt;
                per = (per + k - 1) / k * k;                               // whole rounds of the k wavefronts
                for (int at = 0; at < T; at += per) sizes[n++] = at + per < T ? per : T - at;
            }
        }
    }
    if (n < 2) { sizes[0] = T; n = 1; }
    return n;
}

extern "C" void fx_tuning_defaults(fx_tuning* t)
{
    if (!t) return;
    memset(t, 0, sizeof *t);
    t->frames_per_unit = -1;
    t->stream_graph = t->stream_hop_kernel = t->stream_zero_copy = t->one_hop_kernel = t->call_timing = t->stream_fill_streaming = -1;
}

// The ONLY place the library reads the environment: called once per context, by fx_create.
extern "C" void fx_tuning_from_env(fx_tuning* t)
{
    if (!t) return;
    fx_tuning_defaults(t);
    auto geti = [](const char* name, int* out, int lo) { if (const char* e = getenv(name)) { const int v = atoi(e); if (v >= lo) *out = v; } };
    geti("FX_WAVES", &t->waves_per_channel, 1);
    geti("FX_CHANNELS_PER_WG", &t->channels_per_workgroup, 1);
    geti("FX_WAVES_PER_FRAME", &t->waves_per_frame, 1);
    geti("FX_FRAMES_PER_CHUNK", &t->frames_per_unit, 0);
    if (const char* plan = getenv("FX_CHUNK_PLAN")) {
        int n = 0;
        for (const char* q = plan; *q && n < FX_MAX_UNITS; ) { const int v = atoi(q); if (v > 0) t->unit_plan[n++] = v; while (*q && *q != ',') q++; if (*q == ',') q++; }
        t->unit_plan_len = n;
    }
    geti("FX_STREAM_GRAPH", &t->stream_graph, 0);
    geti("FX_STREAM_HOP_KERNEL", &t->stream_hop_kernel, 0);
    geti("FX_STREAM_ZEROCOPY", &t->stream_zero_copy, 0);
    geti("FX_ONE_HOP_KERNEL", &t->one_hop_kernel, 0);
    geti("FX_CALL_TIMING", &t->call_timing, 0);
    geti("FX_HANDOVER_SPINS", &t->handover_spin_limit, 1);
    geti("FX_STREAM_FILL_STREAMING", &t->stream_fill_streaming, 0);
}

extern "C" fx_status fx_get_tuning(fx_context* c, fx_tuning* out)
{
    if (!c || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *out = c->tuning;
    return FX_OK;
}

extern "C" fx_status fx_set_tuning(fx_context* c, const fx_tuning* t)
{
    if (!c || !t) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (t->waves_per_channel < 0 || t->channels_per_workgroup < 0 || t->waves_per_frame < 0 || t->waves_per_frame > 2 ||
        t->unit_plan_len < 0 || t->unit_plan_len > FX_MAX_UNITS || t->handover_spin_limit < 0)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "tuning value out of range");
    // the kernel family is fixed while a history exists: the two families' continuous slots may differ in the last bit, and a
    // channel's smoothing window must not hold values of both
    if (c->frames_seen > 0 && uses_pairs(c, t->waves_per_frame) != uses_pairs(c, c->tuning.waves_per_frame))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "waves_per_frame selects the kernel family and %lld frames have been analysed with the other one; "
                                                "fx_reset_state first", c->frames_seen);
    c->tuning = *t;              // (nothing below can fail: a refused call leaves the old knobs in place)
    return FX_OK;
}

// tests only (csrc/fx_kernels.h; not in include/fx.h)
extern "C" fx_status fx_set_tuning_internal(fx_context* c, unsigned test_hooks)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    c->test_hooks = test_hooks;
    return FX_OK;
}

// A frame-kernel work unit that gave up waiting for its predecessor's flux state stores 1 to c->h_err (pinned host
// memory).  Sticky: every synchronising entry point reports it until fx_reset_state.
fx_status fx_check_device_error(fx_context* c)
{
    if (c && c->h_err && *(volatile unsigned*) c->h_err != 0)
        return fx_fail(FX_ERR_HIP, "a frame-kernel work unit timed out waiting for the flux state of the unit before it; "
                                   "the results of that call and of every call since are not valid (fx_reset_state clears this)");
    return FX_OK;
}

namespace {

// Everything one analysis step launches: kernel arguments and the wavefront count of the frame kernel.
struct Step {
    fxk::FrameParams    fp;
    fxk::EpilogueParams ep;
    int analysers = 3;
    int waves = 1;
    bool pair = false;      // fx_pair_kernel: one frame across two wavefronts (fp.waves_per_ch counts pairs)
    bool hop_pairs = false; // a one-frame call through fx_hop_pair_kernel (six wavefronts per channel) instead of fx_hop_kernel (three)
};

hipError_t launch_frames(const fx_context* c, const Step& st)
{
    return st.pair ? fxk::launch_pair_kernel(c->N, st.fp, c->stream) : fxk::launch_frame_kernel(c->N, st.fp, st.analysers, c->stream);
}

// Fill the kernel arguments of a step over T frames per channel from the context's current state.  `part` / `raw`
// default to the context's own (growable) scratch; a captured step passes buffers it owns, because a graph keeps the
// addresses it was captured with.
fx_status prepare_step(fx_context* c, const void* d_in, int T, int sample_format, int hop_mode, float* d_or, float* d_os,
                       fxk::FramePart* part, float* raw, const fxk::DynParams* dyn, Step* st)
{
    fxk::FrameParams& fp = st->fp;
    fp.in = d_in;
    fp.sample_format = sample_format;
    fp.hop_mode = hop_mode;
    fp.T = T;
    fp.C = c->C;
    fp.gain = c->gain;
    fp.tail_in = c->d_tail[c->cur];
    fp.tail_out = c->d_tail[c->cur ^ 1];
    fp.prev_re = c->d_prev;
    fp.tw = c->d_tw;
    fp.tw_image = c->d_tw + 2 * (size_t) c->N;
    fp.part = part;
    fp.nyquist = c->sample_rate / 2.0;          // ref RealTimeAudioAnalysis.h:251, RealTimeAnalyser.h:113
    fp.bin_var = c->bin_var;
    fp.lpf_a = c->lpf_a;
    fp.lpf_b = c->lpf_b;
    fp.dyn = dyn;
    for (int i = 0; i < 18; i++) fp.first_tw[i] = c->first_tw[i];
    fp.tw_quarter_turn = (c->tw_quarter_turn && !(c->test_hooks & FX_HOOK_NO_QUARTER_TURN)) ? 1 : 0;
    fp.tw_at_quarter[0] = c->tw_at_quarter[0]; fp.tw_at_quarter[1] = c->tw_at_quarter[1];
    fp.block_mode = 0; fp.blk_carry_bytes = fp.blk_carry_row_bytes = 0; fp.blk_in_row_bytes = 0; fp.blk_carry_in = nullptr; fp.blk_carry_out = nullptr;
    fp.blk_hop0 = 0; fp.blk_keep_rest = 0; fp.in_hop_stride = 0; fp.in_hop0 = 0;

    // Workgroup shape: channels per workgroup x wavefronts per channel (= frames of one channel in flight): the
    // measured-best shape for this window size, fewer waves when the call has fewer frames, fewer channels when the
    // context has fewer or the LDS holds fewer (one twiddle table per workgroup, one flux state per channel, one
    // transform buffer per wave).  fx_tuning overrides for experiments.
    const size_t lds_cu = 160 * 1024;
    st->analysers = (c->flags & FX_SPECTRAL_ONLY) ? 1 : ((c->flags & FX_HARMONIC_ONLY) ? 2 : 3);
    // The low-latency family (opt-in: FX_LOW_LATENCY, or fx_tuning::waves_per_frame = 2): windows of 2048 / 4096 points with both
    // analysers run one frame across a PAIR of wavefronts -- fx_pair_kernel for calls of several frames, fx_hop_pair_kernel for
    // one frame per call.  The default family keeps a frame in one wavefront at every size (profiles/NOTEBOOK_design_r1-r5.md: pairs are the faster
    // path for one hop, not for throughput).
    st->pair = st->hop_pairs = uses_pairs(c, c->tuning.waves_per_frame);
    {
        const int kcap = st->pair ? fxk::pair_kernel_max_pairs(c->N) : fxk::frame_kernel_max_waves(c->N);
        // one frame per call through the batch kernels (both analysers): the flux state stays in global memory (FrameParams::direct_state)
        const bool direct = T == 1 && !st->pair && st->analysers == 3;
        fp.direct_state = direct ? 1 : 0;
        auto lds_bytes = [&](int ch_, int k_) { return st->pair ? fxk::pair_kernel_lds_bytes(c->N, ch_, k_) : fxk::frame_kernel_lds_bytes(c->N, ch_, k_, direct); };
        int ch = 1, k = 1;
        if (st->pair) { ch = 1; k = kcap; }
        else fxk::frame_kernel_preferred_shape(c->N, &ch, &k);
        if (c->tuning.waves_per_channel >= 1) k = c->tuning.waves_per_channel;
        if (c->tuning.channels_per_workgroup >= 1) ch = c->tuning.channels_per_workgroup;
        if (k > T) k = T;
        if (k > kcap) k = kcap;
        // one frame per call through the batch kernels: one wavefront per channel, so channels share a workgroup's twiddle table.
        // With the flux state in global memory (direct), up to 1024 points as many as a workgroup may hold (1024 points: 8 channels =
        // 76 KB, two workgroups and 16 wavefronts per CU -- what the LDS holds of the batch shape too); at the split sizes the registers
        // allow 8 wavefronts per CU whatever the shape, and a CU does better with two workgroups of four (staggered) than with one of
        // eight in lockstep (2048 points), or with one workgroup of eight from the channel count at which every CU has one (4096 points,
        // whose eight wavefronts are all a CU holds).  Measured, us per call of one hop per channel, channels per workgroup 4 / 8 --
        // profiles/r04_live_cadence.txt:
        //   1024 points  4096 ch 43.4 / 39.1   8192 ch 66.6 / 63.3   16384 ch 116.8 / 111.1
        //   2048 points  2048 ch 44.8 / 43.2   4096 ch 72.0 / 73.0    8192 ch 125.4 / 133.6
        //   4096 points  1024 ch 61.7 / 71.0   2048 ch 108.7 / 75.7   4096 ch 205.4 / 139.6
        // Without the direct form (one analyser only): four (2048 points, 4096 channels x 1 hop 152 us against 193 us with one).
        if (T == 1 && !st->pair && c->tuning.channels_per_workgroup < 1)
            ch = !direct ? 4 : (c->N <= 1024 ? kcap : (c->N == 2048 ? 4 : (c->C >= 2048 ? kcap : 4)));
        // Two frames per call (a 1024-sample device block against a 1024-point window: the live cadence of hosts with larger buffers): two channels
        // per workgroup share the twiddle table.  Measured (tools/device_blocks.py, us per call of two hops, channels per workgroup 1 / 2 / 4):
        // 8192 channels x 1024 points 164.8 / 144.4 / 183.3; 4096 channels x 2048 points 169.8 / 163.5 / 172.9.  Four frames per call: one.
        if (T == 2 && !st->pair && c->N <= 2048 && c->tuning.channels_per_workgroup < 1) ch = 2;
        if (ch > c->C) ch = c->C;
        while (ch > 1 && (ch * k > kcap || lds_bytes(ch, k) > lds_cu)) ch--;
        while (k > 1 && lds_bytes(ch, k) > lds_cu) k--;
        if (lds_bytes(ch, k) > lds_cu)
            return fx_fail(FX_ERR_UNSUPPORTED, "window size %d does not fit the LDS", c->N);
        fp.ch_per_wg = ch;
        fp.waves_per_ch = k;
        st->waves = ch * k * (st->pair ? 2 : 1);
        fp.num_chunks = 1;
        fp.queue = nullptr;
        fp.err = c->d_err;
        fp.spin_limit = c->tuning.handover_spin_limit > 0 ? (unsigned) c->tuning.handover_spin_limit : (1u << 22);
        fp.debug_flags = c->test_hooks;
        for (int i = 0; i <= fxk::FX_MAX_CHUNKS; i++) fp.chunk_begin[i] = 0;
        if (!dyn && c->d_queue) {
            int sizes[fxk::FX_MAX_CHUNKS];
            const int n = fx_plan_units(c->N, c->flags, k, T, &c->tuning, sizes, fxk::FX_MAX_CHUNKS);
            if (n >= 2) {
                fp.num_chunks = n;
                fp.queue = c->d_queue;
                for (int i = 0; i < n; i++) fp.chunk_begin[i + 1] = fp.chunk_begin[i] + sizes[i];
            }
        }
    }

    fxk::EpilogueParams& ep = st->ep;
    ep.part = part;
    ep.raw = raw;
    ep.nyquist = c->sample_rate / 2.0;
    ep.bin_var = c->bin_var;
    ep.window = c->N;
    fxk::epilogue_constants(ep);
    ep.hist = c->d_hist;
    ep.hist_base = (int) (c->frames_seen % fxk::HLEN);
    ep.out_raw = d_or;
    ep.out_smoothed = d_os;
    ep.out_stride = 0; ep.out_t0 = 0;
    ep.latest = c->d_latest;
    ep.C = c->C;
    ep.T = T;
    ep.frames_before = c->frames_seen;
    ep.onset_reset_frame = c->onset_reset_frame;
    ep.onset_window = c->onset_window;
    ep.onset_type = c->onset_type;
    ep.onset_multiplier = c->onset_multiplier;
    ep.order_mode = (int) (c->flags & FX_ORDER_MASK);
    ep.analysers = st->analysers;
    ep.dyn = dyn;
    ep.clear_queue = fp.num_chunks > 1 ? c->d_queue : nullptr;
    ep.clear_count = 1 + c->C;
    return FX_OK;
}

void fill_dyn(const fx_context* c, fxk::DynParams* d)
{
    d->nyquist = c->sample_rate / 2.0;
    d->frames_before = c->frames_seen;
    d->hist_base = (int) (c->frames_seen % fxk::HLEN);
    d->onset_reset_frame = c->onset_reset_frame;
    d->gain = c->gain;
    d->onset_multiplier = c->onset_multiplier;
    d->onset_window = c->onset_window;
    d->onset_type = c->onset_type;
}

void advance(fx_context* c, int T)
{
    c->cur ^= 1;
    c->frames_seen += T;
}

// fx_push_samples' call that completes exactly one hop, without the re-blocking pass: `in` of run() is then the device BLOCK of every
// channel (rows of in_row_bytes) and the one-frame kernels read the hop from [pending samples | block] themselves and write the new
// pending samples (FrameParams::block_mode, csrc/fx_blocks.hip.h)
struct BlockFeed {
    const unsigned char* carry_in;
    unsigned char*       carry_out;
    int                  carry_bytes, carry_row_bytes;
    long long            in_row_bytes;
};

// Whether this context's one-hop calls can take blocks directly: windows from 1024 points, both analysers, the default kernel family
// (the pair family and the single-analyser forms read hops: those calls go through fx_reblock_kernel).
bool blocks_feed_kernels(const fx_context* c)
{
    return c->N >= 1024 && !(c->flags & (FX_SPECTRAL_ONLY | FX_HARMONIC_ONLY)) && !uses_pairs(c, c->tuning.waves_per_frame) &&
           !(c->test_hooks & FX_HOOK_NO_BLOCK_FEED);
}

// in_kind / out_kind: where the caller's samples and result buffers live (fx_push_samples hands over hops it has assembled in device
// memory with results that may go to the host)
fx_status run(fx_context* c, const void* in, int T, int sample_format, int in_kind, int out_kind, int hop_mode,
              float* out_raw, float* out_smoothed, const BlockFeed* blocks = nullptr)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (T < 0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "negative frame count");
    if (T == 0) return FX_OK;
    if (!in) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null input buffer");
    if (!known_format(sample_format))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown sample format %d", sample_format);
    if ((in_kind != FX_MEM_HOST && in_kind != FX_MEM_DEVICE) || (out_kind != FX_MEM_HOST && out_kind != FX_MEM_DEVICE))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown memory kind %d", in_kind != FX_MEM_HOST && in_kind != FX_MEM_DEVICE ? in_kind : out_kind);
    HIP_TRY(hipSetDevice(c->device));
    { const fx_status es = fx_check_device_error(c); if (es != FX_OK) return es; }     // sticky: an earlier call's hand-over failed

    const size_t esz = sample_size(sample_format);
    const size_t per_frame = hop_mode ? (size_t) c->N / 2 : (size_t) c->N;
    const size_t in_bytes = (size_t) c->C * T * per_frame * esz;
    const size_t out_elems = (size_t) c->C * T * FX_NUM_FEATURES;

    fx_status st;
    const size_t raw_bytes = out_elems * sizeof(float);
    if ((st = grow(&c->d_raw, &c->raw_cap, raw_bytes)) != FX_OK) return st;
    if ((st = grow(&c->d_part, &c->part_cap, (size_t) c->C * T * sizeof(fxk::FramePart))) != FX_OK) return st;

    const void* d_in = in;
    float* d_or = out_raw;
    float* d_os = out_smoothed;
    if (out_kind == FX_MEM_HOST) {
        if (out_raw || out_smoothed) {
            // one allocation, two halves
            if (2 * raw_bytes > c->out_cap) {
                float* old = c->d_out_raw;
                c->d_out_raw = nullptr; c->out_cap = 0;          // (forgotten first: see grow())
                if (old) HIP_TRY(hipFree(old));
                void* p = nullptr;
                HIP_TRY(hipMalloc(&p, 2 * raw_bytes));
                c->d_out_raw = static_cast<float*>(p);
                c->out_cap = 2 * raw_bytes;
            }
            c->d_out_sm = c->d_out_raw + out_elems;
        }
        d_or = out_raw ? c->d_out_raw : nullptr;
        d_os = out_smoothed ? c->d_out_sm : nullptr;
    }
    if (in_kind == FX_MEM_HOST) {
        if ((st = grow(reinterpret_cast<unsigned char**>(&c->d_in), &c->in_cap, in_bytes)) != FX_OK) return st;
        HIP_TRY(hipMemcpyAsync(c->d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->d_in;
    } else {
        if (reinterpret_cast<uintptr_t>(in) % (blocks ? 4 : 16) != 0)
            return fx_fail(FX_ERR_INVALID_ARGUMENT, "device input must be %d-byte aligned", blocks ? 4 : 16);
    }
    if (blocks && (T < 1 || T > 4096 || !hop_mode || in_kind != FX_MEM_DEVICE)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "a block feed is 1 .. 4096 hops per channel from device memory");

    // A call of TWO hops per channel (a 1024-sample device buffer against a 1024-point window, 960-sample blocks every other call ...) runs
    // as two one-frame launches over the same buffers -- the second reads hop 1 and writes frame 1 (FrameParams::in_hop_stride / in_hop0,
    // EpilogueParams::out_stride / out_t0) -- and, like a one-frame call, records no timing events unless asked to.  Measured
    // (tools/device_blocks.py, us per call of two hops: batch form with its events / batch form without / two one-frame launches):
    // 8192 channels x 1024 points 144 / 136.6 / 133.8; 1024 x 1024 52 / 42.6 / 39.7; 4096 x 2048 165 / 154.9 / 157.1; 512 x 2048 - / 50.4 / 46.3;
    // 1024 x 4096 137 / 127.1 / 123.4; 256 x 4096 - / 66.7 / 57.1.  Most of what a two-hop call cost over two one-hop calls was the three event
    // records (barrier packets); the launches themselves are worth 0 - 14 %.  What this form really buys is the block feed: a block that
    // completes two hops is read by the kernels directly (1000-sample blocks at 8192 channels: 177 -> 132 us per call).
    const bool both = !(c->flags & (FX_SPECTRAL_ONLY | FX_HARMONIC_ONLY));
    const bool in_two = hop_mode && T == 2 && both && c->N >= 1024 && !uses_pairs(c, c->tuning.waves_per_frame) && !(c->test_hooks & FX_HOOK_NO_TWO_LAUNCHES);
    const bool one_frame_launches = in_two || (blocks && T <= 2);        // (a block feed of more hops is ONE launch of the batch kernel's block-fed form)
    const int parts = one_frame_launches ? T : 1, part_T = one_frame_launches ? 1 : T;

    // The three events fx_last_kernel_ms() reads.  Each is a barrier packet between launches, which a call of milliseconds does not
    // notice and a one-frame call does (back to back 27 us per call with them, 14.6 without): those record none unless asked to.
    // (a call made of one-frame launches is a live call: no events by default, like a one-frame call)
    const bool timed = c->profiling || c->tuning.call_timing == 1 || (c->tuning.call_timing < 0 && part_T > 1);
#define FX_EV(e) do { if (timed) HIP_TRY(hipEventRecord(e, c->stream)); } while (0)
    hipEvent_t e0 = c->ev[0], e1 = c->ev[1], e2 = c->ev[2];
    bool last_valid = timed;
    if (c->profiling && c->prof_used + 3 <= 3 * 4096) {
        while (c->prof_events.size() < c->prof_used + 3) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            c->prof_events.push_back(e);
        }
        e0 = c->prof_events[c->prof_used]; e1 = c->prof_events[c->prof_used + 1]; e2 = c->prof_events[c->prof_used + 2];
        c->prof_used += 3;
        last_valid = false;
    }
    for (int part = 0; part < parts; part++) {
        const bool first = part == 0, last = part == parts - 1;       // (the events bracket the whole call: frame-kernel time is only split out of one-part calls)
        Step step;
        if ((st = prepare_step(c, d_in, part_T, sample_format, hop_mode, d_or, d_os, c->d_part, c->d_raw, nullptr, &step)) != FX_OK) return st;
        if (parts > 1) {
            step.fp.in_hop_stride = T; step.fp.in_hop0 = part;
            step.ep.out_stride = T;    step.ep.out_t0 = part;
        }
        if (blocks) {
            step.fp.block_mode = 1;
            step.fp.blk_hop0 = one_frame_launches ? part : 0;
            step.fp.blk_keep_rest = last ? 1 : 0;
            step.fp.blk_carry_in = blocks->carry_in;
            step.fp.blk_carry_out = blocks->carry_out;
            step.fp.blk_carry_bytes = blocks->carry_bytes;
            step.fp.blk_carry_row_bytes = blocks->carry_row_bytes;
            step.fp.blk_in_row_bytes = blocks->in_row_bytes;
            if (step.pair || step.analysers != 3) return fx_fail(FX_ERR_INVALID_ARGUMENT, "this context's kernels do not read blocks");
        }
        // ONE frame per channel -- the reference's own cadence, an analysis per hop as it arrives (AudioDataCollector.h:66-94,
        // RealTimeAnalyser.h:201-234) -- is one launch of fx_hop_kernel: three wavefronts per channel (pitch / spectral /
        // harmonic) and the hop's tail, instead of one wavefront per channel and a second launch.
        // (measured, tools/live_cadence.py, profiles/r04_live_cadence.txt: once the call holds more than the chip takes in one round of
        // workgroups -- 1024 channels of 1024 points, 512 of 2048, and 1024 of 4096 since a 4096-point workgroup is 80 KB and a CU holds two --
        // the batch kernels take over: one wavefront per channel with the flux state left in global memory (prepare_step), then the fused
        // tail on a quarter wavefront per channel: 63 against 120 us at 8192 channels x 1024-pt, 76 against 186 us at 2048 channels x 4096-pt;
        // below it the hop kernel wins, 19.9 against 22.9 us at 1024 x 1024-pt, 58.7 against 61.8 us at 1024 x 4096-pt.  The pair family's
        // hop kernel -- six wavefronts and 100 KB per channel -- keeps 2^20 at every size)
        const bool one_hop = part_T == 1 && step.analysers == 3 && fxk::hop_kernel_available(c->N) &&
                             (c->tuning.one_hop_kernel == 1 || (c->tuning.one_hop_kernel < 0 && (long long) c->C * c->N <= ((c->N == 4096 && !step.hop_pairs) ? (1ll << 22) : (1ll << 20))));
        if (one_hop) {
            const fxk::HopSignal none = {nullptr, nullptr, 0u, 0u, nullptr};
            if (first) FX_EV(e0);
            HIP_TRY(fxk::launch_hop_kernel(c->N, step.fp, step.ep, none, c->stream, step.hop_pairs));
            if (last) { FX_EV(e1); FX_EV(e2); }
        } else {
            if (first) FX_EV(e0);
            // One frame per channel through the batch kernels: frames and tails in ONE launch (fx_frame_tail_kernel) while the chip holds all
            // of the call's workgroups at once -- two per CU at these sizes, one of eight channels at 4096 points.  Beyond that a workgroup whose
            // first wavefronts are finishing its hops keeps the LDS the next workgroup is waiting for, and the tail is better off as a launch
            // of its own.  Measured (us per call, one launch / two; profiles/r04_live_cadence.txt): 1024 points 2048 channels 26.8 / 28.6, 4096
            // channels 41.0 / 41.7, 8192 channels 70.2 / 68.2; 2048 points 2048 channels 44.9 / 45.7, 4096 channels 77.5 / 73.8; windows of 512
            // points and fewer lose either way (4096 channels 34.7 / 33.2): their frames are no longer than the tail.
            const long long groups = ((long long) c->C + step.fp.ch_per_wg - 1) / step.fp.ch_per_wg;
            const long long one_round = (long long) c->compute_units * ((c->N == 4096 && step.fp.ch_per_wg > 4) ? 1 : 2);
            const bool one_launch = step.fp.direct_state && fxk::frame_tail_kernel_available(c->N) &&
                                    ((c->test_hooks & FX_HOOK_TAIL_ALWAYS_FUSED) || (!(c->test_hooks & FX_HOOK_TAIL_NEVER_FUSED) && groups <= one_round));
            if (one_launch) {
                HIP_TRY(fxk::launch_frame_tail_kernel(c->N, step.fp, step.ep, c->stream));
                if (last) FX_EV(e1);
            } else {
                HIP_TRY(launch_frames(c, step));
                if (last) FX_EV(e1);
                HIP_TRY(fxk::launch_epilogue_kernels(step.ep, c->stream));
            }
            if (last) FX_EV(e2);
        }
        advance(c, part_T);
    }
#undef FX_EV
    c->ev_valid = last_valid;

    if (out_kind == FX_MEM_HOST) {
        if (out_raw) HIP_TRY(hipMemcpyAsync(out_raw, c->d_out_raw, raw_bytes, hipMemcpyDeviceToHost, c->stream));
        if (out_smoothed) HIP_TRY(hipMemcpyAsync(out_smoothed, c->d_out_sm, raw_bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return fx_check_device_error(c);
    }
    return FX_OK;
}

} // namespace

extern "C" {

int fx_abi_version(void) { return FX_ABI_VERSION; }
const char* fx_last_error(void) { return g_fx_err.c_str(); }

fx_status fx_create(fx_context** out, int device_id, int num_channels, int window_size, double sample_rate, unsigned flags)
{
    if (!out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null output pointer");
    *out = nullptr;
    if (num_channels <= 0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "num_channels must be positive");
    if (!is_pow2(window_size) || window_size < 256 || window_size > 4096)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "window_size must be a power of two in [256, 4096], got %d", window_size);
    if (!(sample_rate > 0.0)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "sample_rate must be positive");
    if ((flags & FX_ORDER_MASK) == 3u || (flags & ~(FX_ORDER_MASK | FX_SPECTRAL_ONLY | FX_HARMONIC_ONLY | FX_LOW_LATENCY)) ||
        ((flags & FX_SPECTRAL_ONLY) && (flags & FX_HARMONIC_ONLY)))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown or contradictory flags 0x%x", flags);

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void) hipGetLastError();
        return fx_fail(FX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    }
    if (device_id < 0 || device_id >= count) return fx_fail(FX_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fx_fail(FX_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));

    fx_context* c = new (std::nothrow) fx_context();
    if (!c) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    if (prop.multiProcessorCount > 0) c->compute_units = prop.multiProcessorCount;
    c->device = device_id;
    c->C = num_channels;
    c->N = window_size;
    c->sample_rate = sample_rate;
    c->flags = flags;
    fx_tuning_from_env(&c->tuning);          // once; nothing on the analysis path reads the environment

    fx_status st = FX_OK;
    auto cleanup = [&](fx_status s) { fx_destroy(c); return s; };
    {
        hipError_t e = fxk::prepare_kernels(window_size);
        if (e == hipSuccess) e = fxk::prepare_hop_kernel(window_size);
        if (e == hipSuccess) e = fxk::prepare_pair_kernel(window_size);
        if (e != hipSuccess) return cleanup(fx_fail(FX_ERR_HIP, "kernel preparation failed: %s", hipGetErrorString(e)));
    }
#define TRY_OR_CLEAN(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
    TRY_OR_CLEAN(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) TRY_OR_CLEAN(hipEventCreate(&c->ev[i]));
    const size_t half = (size_t) num_channels * (window_size / 2);
    TRY_OR_CLEAN(hipMalloc((void**) &c->d_tw, sizeof(float) * 4 * window_size));      // the pass-ordered table, then the frame kernel's LDS image of it
    TRY_OR_CLEAN(hipMalloc((void**) &c->d_prev, sizeof(float) * half));
    for (int i = 0; i < 2; i++) TRY_OR_CLEAN(hipMalloc((void**) &c->d_tail[i], sizeof(float) * half));
    TRY_OR_CLEAN(hipMalloc((void**) &c->d_hist, sizeof(float) * (size_t) num_channels * fxk::HLEN * FX_NUM_FEATURES));
    TRY_OR_CLEAN(hipMalloc((void**) &c->d_latest, sizeof(float) * (size_t) num_channels * FX_NUM_FEATURES));
    TRY_OR_CLEAN(hipMalloc((void**) &c->d_queue, sizeof(unsigned) * (1 + (size_t) num_channels)));
    for (int i = 0; i < 2; i++) TRY_OR_CLEAN(hipMalloc((void**) &c->d_carry[i], half * 4));
    TRY_OR_CLEAN(hipHostMalloc((void**) &c->h_err, 64, hipHostMallocCoherent));
    *c->h_err = 0;
    { void* q = nullptr; TRY_OR_CLEAN(hipHostGetDevicePointer(&q, c->h_err, 0)); c->d_err = static_cast<unsigned*>(q); }

    // Twiddle table exactly as the reference's FFT builds it (JUCE 4.2 FFT::FFTConfig, SURVEY.md
    // App. A.1): phase in double, entries rounded to float.  The inverse table is its conjugate.
    {
        const std::vector<float> tw = reference_twiddles(window_size);
        std::vector<float> ordered(tw.size());
        fxk::build_pass_twiddles(window_size, tw.data(), ordered.data());    // same values, pass access order
        fxk::fill_first_pass_twiddles(window_size, ordered.data(), c->first_tw);
        if (!fxk::first_pass_twiddles_hermitian(window_size, c->first_tw))
            return cleanup(fx_fail(FX_ERR_UNSUPPORTED, "this host's cos/sin produce a twiddle table without the mirror symmetry the kernels rely on"));
        c->tw_quarter_turn = fxk::twiddles_have_quarter_turn(window_size, tw.data());
        c->tw_at_quarter[0] = tw[2 * (size_t) (window_size / 4)]; c->tw_at_quarter[1] = tw[2 * (size_t) (window_size / 4) + 1];     // (false only costs the 4096-point kernel two global reads per item)
        TRY_OR_CLEAN(hipMemcpy(c->d_tw, ordered.data(), ordered.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<float> image(ordered.size(), 0.0f);
        fxk::build_twiddle_image(window_size, ordered.data(), image.data());
        TRY_OR_CLEAN(hipMemcpy(c->d_tw + ordered.size(), image.data(), image.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // ref SpectralCharacteristics.h:180-189: binVar does not depend on the signal
    {
        const int M = window_size / 2;
        double bv = 0.0;
        for (double i = 0.0; i < M; i++) {
            const double ni = i / (double) M;
            bv += (ni - 0.5) * (ni - 0.5);
        }
        c->bin_var = bv / (double) M;
    }
    // ref RealTimeAudioAnalysis.h:122,127: float_Pi / m and exp(-float_Pi / m), m = 2.0f, in fp32
    {
        const float float_pi = 3.14159265358979323846f;
        c->lpf_a = float_pi / 2.0f;
        c->lpf_b = std::exp(-float_pi / 2.0f);
    }
    if ((st = zero_state(c)) != FX_OK) return cleanup(st);
    TRY_OR_CLEAN(hipStreamSynchronize(c->stream));
#undef TRY_OR_CLEAN
    *out = c;
    return FX_OK;
}

fx_status fx_destroy(fx_context* c)
{
    if (!c) return FX_OK;
    (void) hipSetDevice(c->device);
    fx_comm_release(c);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    void* bufs[] = {c->d_tw, c->d_prev, c->d_tail[0], c->d_tail[1], c->d_hist, c->d_latest,
                    c->d_raw, c->d_part, c->d_in, c->d_out_raw, c->d_queue, c->d_carry[0], c->d_carry[1], c->d_hops, c->d_osc};
    for (void* b : bufs) if (b) (void) hipFree(b);
    if (c->h_err) (void) hipHostFree(c->h_err);
    for (int i = 0; i < 3; i++) if (c->ev[i]) (void) hipEventDestroy(c->ev[i]);
    for (hipEvent_t e : c->prof_events) (void) hipEventDestroy(e);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
    return FX_OK;
}

fx_status fx_reset_state(fx_context* c)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_err) *c->h_err = 0;
    return zero_state(c);
}

fx_status fx_set_sample_rate(fx_context* c, double sr)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (!(sr > 0.0)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "sample_rate must be positive");
    c->sample_rate = sr;
    return FX_OK;
}

fx_status fx_set_onset_sensitivity(fx_context* c, float s)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (!(s >= 0.0f)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "sensitivity must be >= 0");   // jassert, RealTimeAnalyser.h:246
    c->onset_multiplier = 1.0f + s;
    return FX_OK;
}

fx_status fx_set_onset_window(fx_context* c, int length)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (length < 1 || length > fxk::MAX_ONSET_WINDOW)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "onset window must be in [1,%d]", fxk::MAX_ONSET_WINDOW);
    c->onset_window = length;
    c->onset_reset_frame = c->frames_seen;      // both histories emptied, RealTimeAudioAnalysis.h:73-81
    return FX_OK;
}

fx_status fx_set_onset_type(fx_context* c, int type)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (type < FX_ONSET_SPECTRAL || type > FX_ONSET_COMBINATION) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown onset type %d", type);
    c->onset_type = type;
    return FX_OK;
}

fx_status fx_set_gain(fx_context* c, float gain)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    c->gain = gain;
    return FX_OK;
}

fx_status fx_push_hops(fx_context* c, const void* hops, int num_hops, int sample_format, int mem_kind,
                       float* out_raw, float* out_smoothed)
{
    if (c && c->carry_count > 0)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "%d samples per channel are pending from fx_push_samples; whole hops would overtake them "
                                                "(finish the stream with fx_push_samples, or fx_reset_state)", c->carry_count);
    return run(c, hops, num_hops, sample_format, mem_kind, mem_kind, 1, out_raw, out_smoothed);
}

// ---- the collector's real interface: device blocks of any length (ref AudioDataCollector.h:36-94) ----
int fx_pending_samples(fx_context* c) { return c ? c->carry_count : 0; }

fx_status fx_clear_pending(fx_context* c)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(fxk::clear_carry(c->d_carry[c->carry_cur], (size_t) c->C * (c->N / 2) * 4, c->stream));
    return FX_OK;
}

fx_status fx_push_samples(fx_context* c, const void* samples, int num_samples, int sample_format, int mem_kind,
                          float* out_raw, float* out_smoothed, int* frames_out)
{
    if (frames_out) *frames_out = 0;
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (num_samples < 0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "negative sample count");
    if (!known_format(sample_format)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown sample format %d", sample_format);
    if (mem_kind != FX_MEM_HOST && mem_kind != FX_MEM_DEVICE) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown memory kind %d", mem_kind);
    if (num_samples == 0) return FX_OK;
    if (!samples) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null input buffer");
    if (c->carry_count > 0 && sample_format != c->carry_format)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "%d pending samples per channel are in sample format %d, this block in %d: a stream keeps one format "
                                                "between hop boundaries", c->carry_count, c->carry_format, sample_format);
    const int H = c->N / 2;
    const size_t esz = sample_size(sample_format);
    const long long have = (long long) c->carry_count + num_samples;
    const long long hops64 = have / H;
    if (hops64 > (1ll << 24)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "block of %d samples per channel is too long for one call", num_samples);
    const int hops = (int) hops64, rest = (int) (have - hops64 * H);
    HIP_TRY(hipSetDevice(c->device));
    { const fx_status es = fx_check_device_error(c); if (es != FX_OK) return es; }

    // whole hops from an aligned device buffer and nothing pending: the block IS the hop buffer
    if (mem_kind == FX_MEM_DEVICE && c->carry_count == 0 && rest == 0 && reinterpret_cast<uintptr_t>(samples) % 16 == 0) {
        const fx_status st = run(c, samples, hops, sample_format, FX_MEM_DEVICE, FX_MEM_DEVICE, 1, out_raw, out_smoothed);
        if (st == FX_OK && frames_out) *frames_out = hops;
        return st;
    }
    // ... and so is a host block of whole hops (512-sample callbacks against a 1024-point window): one copy in, no re-blocking
    if (mem_kind == FX_MEM_HOST && c->carry_count == 0 && rest == 0) {
        const fx_status st = run(c, samples, hops, sample_format, FX_MEM_HOST, FX_MEM_HOST, 1, out_raw, out_smoothed);
        if (st == FX_OK && frames_out) *frames_out = hops;
        return st;
    }
    fx_status st;
    const unsigned char* d_block = static_cast<const unsigned char*>(samples);
    const size_t block_bytes = (size_t) c->C * (size_t) num_samples * esz;
    if (mem_kind == FX_MEM_HOST) {
        if ((st = grow(reinterpret_cast<unsigned char**>(&c->d_in), &c->in_cap, block_bytes)) != FX_OK) return st;
        HIP_TRY(hipMemcpyAsync(c->d_in, samples, block_bytes, hipMemcpyHostToDevice, c->stream));
        d_block = static_cast<const unsigned char*>(c->d_in);
    } else if (reinterpret_cast<uintptr_t>(samples) % 4 != 0) {
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "device input must be 4-byte aligned (16-byte aligned to be analysed in place)");
    }
    if (hops >= 1 && hops <= (c->N == 1024 ? 4096 : 2) && blocks_feed_kernels(c)) {
        // A block that completes hops is read by the analysis kernels directly from [pending | block]: the live case -- 441 / 480 / 512 ...
        // samples, one hop; 960 / 1000 / 1024, two -- as one-frame launches, longer blocks (1024-point windows) as one launch of the batch
        // kernel's block-fed form; the call's last frame leaves the rest in the other carry buffer.  No pass over the samples beside the
        // analysis.  (Windows of 2048 / 4096 points re-block calls of more than two hops: measured faster there, fx_kernels.hip launch_t.)
        const BlockFeed feed = {c->d_carry[c->carry_cur], c->d_carry[c->carry_cur ^ 1], (int) ((size_t) c->carry_count * esz), H * 4,
                                (long long) num_samples * (long long) esz};
        st = run(c, d_block, hops, sample_format, FX_MEM_DEVICE, mem_kind, 1, out_raw, out_smoothed, &feed);
        if (st != FX_OK) return st;             // (the stream is no longer the caller's: fx_reset_state, as the contract says)
        c->carry_cur ^= 1;
        c->carry_count = rest;
        c->carry_format = sample_format;
        if (frames_out) *frames_out = hops;
        return FX_OK;
    }
    const size_t hop_bytes = (size_t) c->C * (size_t) hops * H * esz;
    if ((st = grow(&c->d_hops, &c->hops_cap, hop_bytes)) != FX_OK) return st;
    fxk::ReblockParams rp;
    rp.in = d_block;
    rp.carry_in = c->d_carry[c->carry_cur];
    rp.hops_out = c->d_hops;
    rp.carry_out = c->d_carry[c->carry_cur ^ 1];
    rp.in_row_bytes = (long long) num_samples * (long long) esz;
    rp.out_row_bytes = (long long) hops * H * (long long) esz;
    rp.carry_bytes = (int) ((size_t) c->carry_count * esz);
    rp.carry_row_bytes = H * 4;
    rp.C = c->C;
    HIP_TRY(fxk::launch_reblock_kernel(rp, c->stream));
    // the stream holds the new carry whatever happens to the analysis below
    c->carry_cur ^= 1;
    c->carry_count = rest;
    c->carry_format = sample_format;
    if (hops > 0) {
        st = run(c, c->d_hops, hops, sample_format, FX_MEM_DEVICE, mem_kind, 1, out_raw, out_smoothed);
        if (st != FX_OK) return st;
    } else if (mem_kind == FX_MEM_HOST) {
        HIP_TRY(hipStreamSynchronize(c->stream));            // the caller's block may be reused on return
    }
    if (frames_out) *frames_out = hops;
    return FX_OK;
}

fx_status fx_process_frames(fx_context* c, const void* frames, int num_frames, int sample_format, int mem_kind,
                            float* out_raw, float* out_smoothed)
{
    if (c && c->carry_count > 0)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "%d samples per channel are pending from fx_push_samples (finish the stream with fx_push_samples, or fx_reset_state)", c->carry_count);
    return run(c, frames, num_frames, sample_format, mem_kind, mem_kind, 0, out_raw, out_smoothed);
}

fx_status fx_get_smoothed(fx_context* c, float* out, int mem_kind)
{
    if (!c || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t) c->C * FX_NUM_FEATURES * sizeof(float);
    HIP_TRY(hipMemcpyAsync(out, c->d_latest, bytes,
                           mem_kind == FX_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    if (mem_kind != FX_MEM_DEVICE) { HIP_TRY(hipStreamSynchronize(c->stream)); return fx_check_device_error(c); }
    return FX_OK;
}


// ref OSCFeatureAnalysisOutput.h:89-113 for every track at once: the messages are formed by fx_osc_kernel from `latest`
fx_status fx_get_osc_datagrams(fx_context* c, const char* prefix, int first_channel, unsigned char* out, int stride, int* lengths, int mem_kind)
{
    if (!c || !prefix || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (mem_kind != FX_MEM_HOST && mem_kind != FX_MEM_DEVICE) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown memory kind %d", mem_kind);
    const int longest = first_channel < 0 || first_channel > 0x7fffffff - c->C ? -1 : fx_osc_message_bytes(prefix, first_channel + c->C - 1);
    if (longest < 0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "OSC prefix longer than %d bytes, or a channel number out of range", fxk::FX_OSC_PREFIX_MAX);
    if (stride < longest || (stride & 3)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "stride %d: must be a multiple of 4 and hold the longest message (%d bytes)", stride, longest);
    if (mem_kind == FX_MEM_DEVICE && (reinterpret_cast<uintptr_t>(out) & 3)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "a device buffer of messages must start on a 4-byte boundary");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t) c->C * (size_t) stride;
    if (mem_kind == FX_MEM_HOST && c->osc_cap < bytes) {
        if (c->d_osc) { unsigned char* old = c->d_osc; c->d_osc = nullptr; c->osc_cap = 0; HIP_TRY(hipFree(old)); }
        HIP_TRY(hipMalloc((void**) &c->d_osc, bytes));
        c->osc_cap = bytes;
    }
    fxk::OscParams p = {};
    p.latest = c->d_latest;
    p.out = mem_kind == FX_MEM_HOST ? c->d_osc : out;
    p.C = c->C;
    p.stride = stride;
    p.first_channel = first_channel;
    p.prefix_len = (int) strlen(prefix);
    memcpy(p.prefix, prefix, (size_t) p.prefix_len);
    HIP_TRY(fxk::launch_osc_kernel(p, c->stream));
    if (lengths) for (int i = 0; i < c->C; i++) lengths[i] = fx_osc_message_bytes(prefix, first_channel + i);
    if (mem_kind == FX_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, c->d_osc, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return fx_check_device_error(c);
    }
    return FX_OK;
}

fx_status fx_host_alloc(void** out, size_t bytes)
{
    if (!out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void) hipGetLastError(); return fx_fail(FX_ERR_NO_DEVICE, "no HIP device available (page-locked memory is the runtime's)"); }
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    *out = p;
    return FX_OK;
}

fx_status fx_host_free(void* p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return FX_OK;
}

fx_status fx_sync(fx_context* c)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return fx_check_device_error(c);
}

fx_status fx_get_stream(fx_context* c, void** stream)
{
    if (!c || !stream) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *stream = (void*) c->stream;
    return FX_OK;
}

fx_status fx_last_kernel_ms(fx_context* c, float* frame_ms, float* epi_ms)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    if (!c->ev_valid) return fx_fail(FX_ERR_INVALID_ARGUMENT, "the last analysis call recorded no timing (none made yet, a profiled one, or a one-frame call without fx_tuning::call_timing = 1)");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[2]));
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, c->ev[1], c->ev[2]));
    if (frame_ms) *frame_ms = a;
    if (epi_ms) *epi_ms = b;
    return fx_check_device_error(c);
}

fx_status fx_profile_begin(fx_context* c)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    c->profiling = true;
    c->prof_used = 0;
    return FX_OK;
}

fx_status fx_profile_end(fx_context* c, double* frame_ms, double* epi_ms, int* calls)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double a = 0.0, b = 0.0;
    for (size_t i = 0; i + 3 <= c->prof_used; i += 3) {
        float x = 0.f, y = 0.f;
        HIP_TRY(hipEventElapsedTime(&x, c->prof_events[i], c->prof_events[i + 1]));
        HIP_TRY(hipEventElapsedTime(&y, c->prof_events[i + 1], c->prof_events[i + 2]));
        a += x; b += y;
    }
    if (frame_ms) *frame_ms = a;
    if (epi_ms) *epi_ms = b;
    if (calls) *calls = (int) (c->prof_used / 3);
    c->profiling = false;
    c->prof_used = 0;
    return fx_check_device_error(c);
}

// ---------------------------------------------------------------------------------------------
// streaming ingest (fx_stream_*): pinned host ring, H2D on a side stream, analysis behind an event
// ---------------------------------------------------------------------------------------------
} // extern "C"

// The producer's copy into a pinned slot, by several host threads (fx_stream_push): a caller whose audio sits in ordinary memory
// has to move every sample once more before PCIe sees it, and one memcpy thread moves ~12 GB/s where the link takes 55.  A small
// persistent pool: workers sleep on a generation counter, each copies its share of the bytes, the last one wakes the caller.
// A slot is written once by the producer and read next by the DMA engine, never again by the core that wrote it: non-temporal stores
// skip the read-for-ownership of every destination line (a third of the copy's memory traffic; glibc's memcpy only switches to them
// far above the ~8 MB a fill thread copies).  x86-64 only; elsewhere, and for the head / tail of a piece, plain memcpy.
#if defined(__x86_64__)
#include <emmintrin.h>
static void copy_streaming(unsigned char* d, const unsigned char* s, size_t n)
{
    size_t head = (64 - (reinterpret_cast<uintptr_t>(d) & 63)) & 63;
    if (head > n) head = n;
    memcpy(d, s, head); d += head; s += head; n -= head;
    for (size_t blocks = n / 64; blocks > 0; blocks--) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 16)),
                      c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 32)), e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(d), a); _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c); _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
        s += 64; d += 64;
    }
    _mm_sfence();
    memcpy(d, s, n & 63);
}
#else
static void copy_streaming(unsigned char* d, const unsigned char* s, size_t n) { memcpy(d, s, n); }
#endif

class FillPool {
public:
    ~FillPool() { resize(0); }
    bool streaming = true;          // fx_tuning::stream_fill_streaming (taken by fx_stream_create)
    void copy(void* dst, const void* src, size_t bytes, int threads)
    {
        if (threads <= 1 || bytes < (1u << 20)) { memcpy(dst, src, bytes); return; }
        if ((int) workers_.size() != threads - 1) resize(threads - 1);
        const size_t piece = ((bytes + (size_t) threads - 1) / (size_t) threads + 4095) & ~(size_t) 4095;
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = static_cast<unsigned char*>(dst); src_ = static_cast<const unsigned char*>(src); bytes_ = bytes; piece_ = piece;
            pending_ = (int) workers_.size();
            generation_++;
        }
        wake_.notify_all();
        slice(threads - 1);                                   // the caller copies the last piece itself
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
    }
private:
    void slice(int k)
    {
        const size_t at = piece_ * (size_t) k;
        if (at >= bytes_) return;
        const size_t n = bytes_ - at < piece_ ? bytes_ - at : piece_;
        if (streaming) copy_streaming(dst_ + at, src_ + at, n);
        else memcpy(dst_ + at, src_ + at, n);
    }
    void resize(int n)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
        quit_ = false;
        const unsigned long long born = generation_;           // (read here, by the caller: a worker that starts late must not miss the first job)
        for (int k = 0; k < n; k++)
            workers_.emplace_back([this, k, born] {
                unsigned long long seen = born;
                for (;;) {
                    std::unique_lock<std::mutex> g(m_);
                    wake_.wait(g, [&] { return quit_ || generation_ != seen; });
                    if (quit_) return;
                    seen = generation_;
                    g.unlock();
                    slice(k);
                    g.lock();
                    if (--pending_ == 0) done_.notify_one();
                }
            });
    }
    std::mutex m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> workers_;
    unsigned char* dst_ = nullptr; const unsigned char* src_ = nullptr;
    size_t bytes_ = 0, piece_ = 0;
    unsigned long long generation_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

struct fx_stream {
    fx_context* ctx = nullptr;
    FillPool fill;
    int hops = 0, slots = 0, fmt = FX_SAMPLE_F32;
    size_t in_bytes = 0, out_bytes = 0;
    // Large batches: three queues, so that PCIe runs in both directions while the kernels run -- `copy` carries batch k+1's samples
    // to the device, the context's stream analyses batch k, `back` returns batch k-1's vectors.  (Until round 4 the results went
    // back on `copy`: the next batch's samples then queued behind a copy that waits for the analysis before it, and nothing overlapped.)
    hipStream_t copy = nullptr, back = nullptr;
    // Small batches are launch-bound (one 4096-pt hop: four kernels, three copies and five events cost ~150 us of host
    // and dispatch time for ~40 us of GPU work): there the whole step -- input copy, per-call scalars, the four
    // kernels, result copies -- is captured once per ring slot and buffer parity into a hipGraph and replayed.
    bool use_graph = false;
    // One hop per call (BASELINE configs[4]) is all latency: there the whole step is ONE launch of fx_hop_kernel
    // (csrc/fx_hop_kernel.hip.h: three wavefronts per channel + the tail), which reads the hop from the pinned slot,
    // writes the 12-float vectors back to it and then stores the call's sequence number to the slot's flag; collect
    // polls that flag.  No graph, no event, no second kernel.
    bool use_hop_kernel = false;
    bool zero_copy = false;               // captured step: kernels read / write the pinned slot directly (a few KB per step)
    unsigned* d_arrivals = nullptr;       // workgroups of the running hop kernel that have finished (zero between calls)
    void*     d_stage = nullptr;          // [C][N/2] samples: the hop kernel's device copy of the hop it is analysing
    unsigned  next_seq = 0;
    fxk::FramePart* g_part = nullptr;     // scratch the captured kernels own (a graph keeps its addresses)
    float*          g_raw = nullptr;
    struct Slot {
        void*  h_in = nullptr;  void* d_in = nullptr;
        float* d_raw = nullptr; float* d_sm = nullptr;
        float* h_raw = nullptr; float* h_sm = nullptr;
        hipEvent_t copied = nullptr, done = nullptr, out = nullptr;
        unsigned* h_flag = nullptr;       // pinned, coherent: sequence number of the last hop-kernel call that completed in this slot
        const void* dev_in = nullptr; float* dev_raw = nullptr; float* dev_sm = nullptr; unsigned* dev_flag = nullptr;   // device views of the pinned buffers
        unsigned  seq = 0;                // sequence number of the call in flight in this slot
        int       frames = 0;             // analysis frames per channel of the batch in flight in this slot (hops_per_batch, or what fx_stream_submit_samples made of its block)
        bool      by_event = false;       // the batch in flight completes with the `out` event (every path but the one-launch hop kernel, which raises a flag)
        fxk::DynParams* h_dyn = nullptr;  // pinned: what changes from call to call
        fxk::DynParams* d_dyn = nullptr;
        hipGraphExec_t  exec[2] = {nullptr, nullptr};     // per parity of the context's ping-pong buffers
    };
    std::vector<Slot> ring;
    int head = 0;        // next slot to acquire
    int tail = 0;        // oldest slot in flight
    int in_flight = 0;
    bool acquired = false;
    int since_release = 0;   // batches of the three-queue path collected since the streams were last synchronised (fx_stream_collect_samples)
};

static fx_status submit_large(fx_stream* s, fx_stream::Slot& sl, size_t in_bytes, int num_samples);
#define HIP_TRY_OR(expr, after) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { after; \
        return fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)

extern "C" {

fx_status fx_stream_destroy(fx_stream* s)
{
    if (!s) return FX_OK;
    if (s->ctx) (void) hipSetDevice(s->ctx->device);
    if (s->copy) (void) hipStreamSynchronize(s->copy);
    if (s->ctx && s->ctx->stream) (void) hipStreamSynchronize(s->ctx->stream);
    if (s->back) (void) hipStreamSynchronize(s->back);
    for (auto& sl : s->ring) {
        if (sl.h_in) (void) hipHostFree(sl.h_in);
        if (sl.h_raw) (void) hipHostFree(sl.h_raw);
        if (sl.h_sm) (void) hipHostFree(sl.h_sm);
        if (sl.d_in) (void) hipFree(sl.d_in);
        if (sl.d_raw) (void) hipFree(sl.d_raw);
        if (sl.d_sm) (void) hipFree(sl.d_sm);
        for (int q = 0; q < 2; q++) if (sl.exec[q]) (void) hipGraphExecDestroy(sl.exec[q]);
        if (sl.h_dyn) (void) hipHostFree(sl.h_dyn);
        if (sl.h_flag) (void) hipHostFree(sl.h_flag);
        if (sl.d_dyn) (void) hipFree(sl.d_dyn);
        if (sl.copied) (void) hipEventDestroy(sl.copied);
        if (sl.done) (void) hipEventDestroy(sl.done);
        if (sl.out) (void) hipEventDestroy(sl.out);
    }
    if (s->d_arrivals) (void) hipFree(s->d_arrivals);
    if (s->d_stage) (void) hipFree(s->d_stage);
    if (s->g_part) (void) hipFree(s->g_part);
    if (s->g_raw) (void) hipFree(s->g_raw);
    if (s->copy) (void) hipStreamDestroy(s->copy);
    if (s->back) (void) hipStreamDestroy(s->back);
    delete s;
    return FX_OK;
}

fx_status fx_stream_create(fx_context* c, int hops_per_batch, int slots, int sample_format, fx_stream** out)
{
    if (!c || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (hops_per_batch < 1 || slots < 1 || slots > 64) return fx_fail(FX_ERR_INVALID_ARGUMENT, "hops_per_batch >= 1 and 1 <= slots <= 64 required");
    if (!known_format(sample_format)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown sample format %d", sample_format);
    HIP_TRY(hipSetDevice(c->device));
    fx_stream* s = new (std::nothrow) fx_stream();
    if (!s) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->ctx = c; s->hops = hops_per_batch; s->slots = slots; s->fmt = sample_format;
    s->fill.streaming = c->tuning.stream_fill_streaming != 0;
    s->in_bytes = (size_t) c->C * hops_per_batch * (c->N / 2) * sample_size(sample_format);
    s->out_bytes = (size_t) c->C * hops_per_batch * FX_NUM_FEATURES * sizeof(float);
    s->ring.resize((size_t) slots);
#define S_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fx_status st_ = fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); fx_stream_destroy(s); return st_; } } while (0)
    S_TRY(hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking));
    S_TRY(hipStreamCreateWithFlags(&s->back, hipStreamNonBlocking));
    // which of the equivalent paths runs: by batch size, unless the context's tuning forces one (experiments, tests)
    s->use_graph = c->tuning.stream_graph >= 0 ? c->tuning.stream_graph != 0 : (size_t) c->C * hops_per_batch <= 4096;
    // (up to 1 MiB of hops per call: the kernel reads each hop out of the pinned slot exactly once, 16 bytes per lane)
    s->use_hop_kernel = hops_per_batch == 1 && s->in_bytes <= 1024 * 1024 && fxk::hop_kernel_available(c->N)
                        && !(c->flags & (FX_SPECTRAL_ONLY | FX_HARMONIC_ONLY)) && c->tuning.stream_hop_kernel != 0 && s->use_graph;
    if (s->use_hop_kernel) s->use_graph = false;
    s->zero_copy = c->tuning.stream_zero_copy >= 0 ? c->tuning.stream_zero_copy != 0 : s->in_bytes <= 64 * 1024;
    if (s->use_hop_kernel) {
        S_TRY(hipMalloc((void**) &s->d_arrivals, sizeof(unsigned)));
        S_TRY(hipMemsetAsync(s->d_arrivals, 0, sizeof(unsigned), c->stream));
        S_TRY(hipMalloc(&s->d_stage, s->in_bytes));
    }
    // the hop kernel's results and flag are read by the host while the kernel may still be running: coherent (fine-grained) memory
    const unsigned host_flags = s->use_hop_kernel ? hipHostMallocCoherent : hipHostMallocDefault;
    if (s->use_graph) {
        S_TRY(hipMalloc((void**) &s->g_part, (size_t) c->C * hops_per_batch * sizeof(fxk::FramePart)));
        S_TRY(hipMalloc((void**) &s->g_raw, s->out_bytes));
    }
    for (auto& sl : s->ring) {
        if (s->use_graph) {
            S_TRY(hipHostMalloc((void**) &sl.h_dyn, sizeof(fxk::DynParams), hipHostMallocDefault));
            S_TRY(hipMalloc((void**) &sl.d_dyn, sizeof(fxk::DynParams)));
        }
        if (s->use_hop_kernel) {
            S_TRY(hipHostMalloc((void**) &sl.h_flag, 64, hipHostMallocCoherent));
            *sl.h_flag = 0;
        }
        S_TRY(hipHostMalloc(&sl.h_in, s->in_bytes, host_flags));
        S_TRY(hipHostMalloc((void**) &sl.h_raw, s->out_bytes, host_flags));
        S_TRY(hipHostMalloc((void**) &sl.h_sm, s->out_bytes, host_flags));
        S_TRY(hipMalloc(&sl.d_in, s->in_bytes));
        S_TRY(hipMalloc((void**) &sl.d_raw, s->out_bytes));
        S_TRY(hipMalloc((void**) &sl.d_sm, s->out_bytes));
        S_TRY(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        S_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        S_TRY(hipEventCreateWithFlags(&sl.out, hipEventDisableTiming));
    }
#undef S_TRY
    *out = s;
    return FX_OK;
}

int fx_stream_in_flight(fx_stream* s) { return s ? s->in_flight : 0; }

fx_status fx_stream_acquire(fx_stream* s, void** host_slot)
{
    if (!s || !host_slot) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (s->acquired) return fx_fail(FX_ERR_INVALID_ARGUMENT, "a slot is already acquired; submit it first");
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (s->in_flight == s->slots)
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "all %d slots are in flight; collect a batch first", s->slots);
    *host_slot = s->ring[(size_t) s->head].h_in;
    s->acquired = true;
    return FX_OK;
}

fx_status fx_stream_submit(fx_stream* s)
{
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null stream");
    if (!s->acquired) return fx_fail(FX_ERR_INVALID_ARGUMENT, "no slot acquired");
    fx_context* c = s->ctx;
    HIP_TRY_OR(hipSetDevice(c->device), s->acquired = false);
    fx_stream::Slot& sl = s->ring[(size_t) s->head];
    { const fx_status es = fx_check_device_error(c); if (es != FX_OK) { s->acquired = false; return es; } }
    if (c->carry_count > 0) {
        s->acquired = false;
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "%d samples per channel are pending from fx_stream_submit_samples / fx_push_samples; whole hops would overtake them", c->carry_count);
    }
    // Any failure below hands the slot back (the caller may fill and submit it again): the ring never wedges on
    // "a slot is already acquired".
#define SUB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { s->acquired = false; \
        return fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    if (s->use_hop_kernel) {
        if (!sl.dev_flag) {
            void* q = nullptr;
            SUB_TRY(hipHostGetDevicePointer(&q, sl.h_in, 0));   sl.dev_in = q;
            SUB_TRY(hipHostGetDevicePointer(&q, sl.h_raw, 0));  sl.dev_raw = static_cast<float*>(q);
            SUB_TRY(hipHostGetDevicePointer(&q, sl.h_sm, 0));   sl.dev_sm = static_cast<float*>(q);
            SUB_TRY(hipHostGetDevicePointer(&q, sl.h_flag, 0)); sl.dev_flag = static_cast<unsigned*>(q);
        }
        unsigned* flag_dev = sl.dev_flag;
        Step step;
        fx_status st0 = prepare_step(c, sl.dev_in, 1, s->fmt, 1, sl.dev_raw, sl.dev_sm, nullptr, nullptr, nullptr, &step);
        if (st0 != FX_OK) { s->acquired = false; return st0; }
        if (++s->next_seq == 0) s->next_seq = 1;            // 0 = "nothing completed yet"
        sl.seq = s->next_seq;
        const fxk::HopSignal sig = {s->d_arrivals, flag_dev, sl.seq, 0u, s->d_stage};
        const hipError_t e = fxk::launch_hop_kernel(c->N, step.fp, step.ep, sig, c->stream, step.hop_pairs);
        if (e != hipSuccess) { s->acquired = false; return fx_fail(FX_ERR_HIP, "launching the hop kernel failed: %s", hipGetErrorString(e)); }
        c->ev_valid = false;
        advance(c, 1);
        sl.frames = 1; sl.by_event = false;
        s->head = (s->head + 1) % s->slots;
        s->in_flight++;
        s->acquired = false;
        return FX_OK;
    }
    if (s->use_graph) {
        const int par = c->cur;
        fill_dyn(c, sl.h_dyn);
        if (!sl.exec[par]) {
            // capture the step once for this slot and parity: everything below is recorded, not executed
            // A few KB per step: the kernels read the hop and the per-call scalars straight from the pinned host slot and
            // write the 12-float vectors straight back (zero copy), so the graph is two kernel nodes and no copy nodes;
            // larger batches keep explicit copies (PCIe is read best in bulk).
            const bool zero_copy = s->zero_copy;
            const void* in_dev = sl.d_in;
            float* raw_dev = sl.d_raw; float* sm_dev = sl.d_sm;
            const fxk::DynParams* dyn_dev = sl.d_dyn;
            if (zero_copy) {
                void* q = nullptr;
                SUB_TRY(hipHostGetDevicePointer(&q, sl.h_in, 0));  in_dev = q;
                SUB_TRY(hipHostGetDevicePointer(&q, sl.h_raw, 0)); raw_dev = static_cast<float*>(q);
                SUB_TRY(hipHostGetDevicePointer(&q, sl.h_sm, 0));  sm_dev = static_cast<float*>(q);
                SUB_TRY(hipHostGetDevicePointer(&q, sl.h_dyn, 0)); dyn_dev = static_cast<const fxk::DynParams*>(q);
            }
            Step step;
            fx_status st0 = prepare_step(c, in_dev, s->hops, s->fmt, 1, raw_dev, sm_dev, s->g_part, s->g_raw, dyn_dev, &step);
            if (st0 != FX_OK) { s->acquired = false; return st0; }
            hipGraph_t graph = nullptr;
            SUB_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
            hipError_t e = hipSuccess;
            if (!zero_copy) {
                e = hipMemcpyAsync(sl.d_in, sl.h_in, s->in_bytes, hipMemcpyHostToDevice, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(sl.d_dyn, sl.h_dyn, sizeof(fxk::DynParams), hipMemcpyHostToDevice, c->stream);
            }
            if (e == hipSuccess) e = launch_frames(c, step);
            if (e == hipSuccess) e = fxk::launch_epilogue_kernels(step.ep, c->stream);
            if (!zero_copy) {
                if (e == hipSuccess) e = hipMemcpyAsync(sl.h_raw, sl.d_raw, s->out_bytes, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(sl.h_sm, sl.d_sm, s->out_bytes, hipMemcpyDeviceToHost, c->stream);
            }
            const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);       // (always: a failed step must not leave the capture open)
            if (e == hipSuccess) e = e2;
            if (e == hipSuccess) e = hipGraphInstantiate(&sl.exec[par], graph, nullptr, nullptr, 0);
            if (graph) (void) hipGraphDestroy(graph);
            if (e != hipSuccess) {
                sl.exec[par] = nullptr;
                s->acquired = false;
                return fx_fail(FX_ERR_HIP, "capturing the streaming step failed: %s", hipGetErrorString(e));
            }
        }
        SUB_TRY(hipGraphLaunch(sl.exec[par], c->stream));
        // the step is enqueued: the context has moved on whatever happens to the bookkeeping event below
        c->ev_valid = false;
        advance(c, s->hops);
        const hipError_t er = hipEventRecord(sl.out, c->stream);
        sl.frames = s->hops; sl.by_event = true;
        s->head = (s->head + 1) % s->slots;
        s->in_flight++;
        s->acquired = false;
        if (er != hipSuccess) return fx_fail(FX_ERR_HIP, "hipEventRecord failed: %s", hipGetErrorString(er));
        return FX_OK;
    }
    return submit_large(s, sl, s->in_bytes, -1);
}

fx_status fx_stream_submit_samples(fx_stream* s, int num_samples)
{
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null stream");
    if (!s->acquired) return fx_fail(FX_ERR_INVALID_ARGUMENT, "no slot acquired");
    fx_context* c = s->ctx;
    if (num_samples < 0 || (long long) num_samples > (long long) s->hops * (c->N / 2)) {
        s->acquired = false;
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "a slot holds 0 .. %lld samples per channel, got %d", (long long) s->hops * (c->N / 2), num_samples);
    }
    HIP_TRY_OR(hipSetDevice(c->device), s->acquired = false);
    { const fx_status es = fx_check_device_error(c); if (es != FX_OK) { s->acquired = false; return es; } }
    return submit_large(s, s->ring[(size_t) s->head], (size_t) c->C * (size_t) num_samples * sample_size(s->fmt), num_samples);
}

} // extern "C"

// The large-batch form of a submit: samples in on `copy`, analysis on the context's stream behind an event, vectors back on `back`.
// num_samples < 0: the slot holds hops_per_batch whole hops per channel; else a block of num_samples samples per channel
// ([C][num_samples], rows back to back), which fx_push_samples cuts into hops with the context's pending samples.
// Any failure hands the slot back (acquired = false) so the ring never wedges on "a slot is already acquired": before the
// analysis is enqueued nothing has happened; after it the context has moved on and the batch's results are lost with the error.
static fx_status submit_large(fx_stream* s, fx_stream::Slot& sl, size_t in_bytes, int num_samples)
{
    fx_context* c = s->ctx;
#define LG_TRY(expr, after) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { after; s->acquired = false; \
        return fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    if (in_bytes) LG_TRY(hipMemcpyAsync(sl.d_in, sl.h_in, in_bytes, hipMemcpyHostToDevice, s->copy), (void) 0);
    LG_TRY(hipEventRecord(sl.copied, s->copy), (void) hipStreamSynchronize(s->copy));
    LG_TRY(hipStreamWaitEvent(c->stream, sl.copied, 0), (void) hipStreamSynchronize(s->copy));
    int frames = s->hops;
    const fx_status st = num_samples < 0 ? run(c, sl.d_in, s->hops, s->fmt, FX_MEM_DEVICE, FX_MEM_DEVICE, 1, sl.d_raw, sl.d_sm)
                                         : fx_push_samples(c, sl.d_in, num_samples, s->fmt, FX_MEM_DEVICE, sl.d_raw, sl.d_sm, &frames);
    if (st != FX_OK) {
        // The copy is already enqueued, and fx_push_samples may have launched its re-blocking kernel on the context's stream before it failed: both
        // read the slot, so let both finish before the slot is handed back.  The ring stays usable; the STREAM of a samples submit does not:
        // the pending samples have moved on without the hops this block completed (fx_push_samples' contract) -- fx_reset_state, not a
        // re-submit of the same block.
        (void) hipStreamSynchronize(s->copy);
        if (num_samples >= 0) (void) hipStreamSynchronize(c->stream);
        s->acquired = false;
        return st;
    }
    const size_t out_bytes = (size_t) c->C * (size_t) frames * FX_NUM_FEATURES * sizeof(float);
    // from here on the analysis is enqueued: a failure loses this batch's results, not the ring (wait for the kernels that read the slot)
    LG_TRY(hipEventRecord(sl.done, c->stream), (void) hipStreamSynchronize(c->stream));
    LG_TRY(hipStreamWaitEvent(s->back, sl.done, 0), (void) hipStreamSynchronize(c->stream));
    if (out_bytes) {
        LG_TRY(hipMemcpyAsync(sl.h_raw, sl.d_raw, out_bytes, hipMemcpyDeviceToHost, s->back), (void) hipStreamSynchronize(c->stream));
        LG_TRY(hipMemcpyAsync(sl.h_sm, sl.d_sm, out_bytes, hipMemcpyDeviceToHost, s->back), ((void) hipStreamSynchronize(c->stream), (void) hipStreamSynchronize(s->back)));
    }
    LG_TRY(hipEventRecord(sl.out, s->back), ((void) hipStreamSynchronize(c->stream), (void) hipStreamSynchronize(s->back)));
#undef LG_TRY
    sl.frames = frames; sl.by_event = true;
    s->head = (s->head + 1) % s->slots;
    s->in_flight++;
    s->acquired = false;
    return FX_OK;
}

extern "C" {

fx_status fx_stream_push(fx_stream* s, const void* hops, int fill_threads)
{
    if (!s || !hops) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (fill_threads < 1 || fill_threads > 64) return fx_fail(FX_ERR_INVALID_ARGUMENT, "fill_threads must be in [1, 64]");
    void* slot = nullptr;
    const fx_status st = fx_stream_acquire(s, &slot);
    if (st != FX_OK) return st;
    s->fill.copy(slot, hops, s->in_bytes, fill_threads);
    return fx_stream_submit(s);
}

fx_status fx_stream_push_samples(fx_stream* s, const void* samples, int num_samples, int fill_threads)
{
    if (!s || (!samples && num_samples > 0)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (fill_threads < 1 || fill_threads > 64) return fx_fail(FX_ERR_INVALID_ARGUMENT, "fill_threads must be in [1, 64]");
    if (num_samples < 0 || (long long) num_samples > (long long) s->hops * (s->ctx->N / 2))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "a slot holds 0 .. %lld samples per channel, got %d", (long long) s->hops * (s->ctx->N / 2), num_samples);
    void* slot = nullptr;
    const fx_status st = fx_stream_acquire(s, &slot);
    if (st != FX_OK) return st;
    if (num_samples > 0) s->fill.copy(slot, samples, (size_t) s->ctx->C * (size_t) num_samples * sample_size(s->fmt), fill_threads);
    return fx_stream_submit_samples(s, num_samples);
}

fx_status fx_stream_collect(fx_stream* s, float* out_raw, float* out_smoothed)
{
    return fx_stream_collect_samples(s, out_raw, out_smoothed, nullptr);
}

fx_status fx_stream_collect_samples(fx_stream* s, float* out_raw, float* out_smoothed, int* frames_out)
{
    if (frames_out) *frames_out = 0;
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null stream");
    if (s->in_flight == 0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "nothing in flight");
    HIP_TRY(hipSetDevice(s->ctx->device));
    fx_stream::Slot& sl = s->ring[(size_t) s->tail];
    if (!sl.by_event) {
        // the kernel stores the call's sequence number after its results: poll it (a hop takes tens of microseconds,
        // an event wait costs as much again); if it does not show up soon -- a large grid, a busy device -- wait for the stream
        volatile unsigned* flag = sl.h_flag;
        bool seen = false;
        for (int spin = 0; spin < 200000; spin++) {
            if (*flag == sl.seq) { seen = true; break; }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
        HIP_TRY(hipEventSynchronize(sl.out));
    }
    const size_t got_bytes = (size_t) s->ctx->C * (size_t) sl.frames * FX_NUM_FEATURES * sizeof(float);
    if (out_raw && got_bytes) memcpy(out_raw, sl.h_raw, got_bytes);
    if (out_smoothed && got_bytes) memcpy(out_smoothed, sl.h_sm, got_bytes);
    if (frames_out) *frames_out = sl.frames;
    s->tail = (s->tail + 1) % s->slots;
    s->in_flight--;
    if (sl.by_event && ++s->since_release >= 64 && (s->in_flight == 0 || s->since_release >= 4096)) {
        // The three-queue path orders its work with events alone and never synchronises a stream; the HIP runtime keeps what it has
        // submitted to a stream on record until somebody does (measured: 1.9 KB of host memory per batch, 37 MB per 20 000 blocks of a live
        // stream -- tools/rss_probe.py).  With the ring drained every queue is idle and the three calls return at once; a ring that never
        // drains gets them every 4096 batches, where they wait for the batches still in flight.
        s->since_release = 0;
        HIP_TRY(hipStreamSynchronize(s->copy));
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        HIP_TRY(hipStreamSynchronize(s->back));
    }
    return fx_check_device_error(s->ctx);
}

// ---- OSC sink helpers (ref OSCFeatureAnalysisOutput.h:107, README.md:57) ----
static const int k_osc12[12] = {FX_ONSET, FX_RMS, FX_F0, FX_CENTROID, FX_SLOPE, FX_SPREAD,
                                FX_FLATNESS, FX_LER, FX_FLUX, FX_HER, FX_OER, FX_INHARM};
static const int k_osc10[10] = {FX_ONSET, FX_RMS, FX_F0, FX_CENTROID, FX_SLOPE, FX_SPREAD,
                                FX_FLATNESS, FX_FLUX, FX_HER, FX_INHARM};

void fx_pack_osc12(const float* f, float* out) { for (int i = 0; i < 12; i++) out[i] = f[k_osc12[i]]; }
void fx_pack_osc10(const float* f, float* out) { for (int i = 0; i < 10; i++) out[i] = f[k_osc10[i]]; }

int fx_osc_encode(const char* address, const float* f, unsigned char* out, int cap)
{
    if (!address || !f || !out) return -1;
    const int alen = (int) strlen(address);
    const int apad = (alen + 4) & ~3;           // NUL-terminated, padded to a multiple of 4
    const int tpad = 16;                        // ",ffffffffffff" + NUL -> 16
    const int total = apad + tpad + 48;
    if (total > cap) return -1;
    memset(out, 0, (size_t) total);
    memcpy(out, address, (size_t) alen);
    memcpy(out + apad, ",ffffffffffff", 13);
    unsigned char* p = out + apad + tpad;
    for (int i = 0; i < 12; i++) {
        unsigned int bits;
        memcpy(&bits, &f[k_osc12[i]], 4);
        p[0] = (unsigned char) (bits >> 24); p[1] = (unsigned char) (bits >> 16);
        p[2] = (unsigned char) (bits >> 8);  p[3] = (unsigned char) bits;
        p += 4;
    }
    return total;
}

int fx_osc_message_bytes(const char* prefix, int channel)
{
    if (!prefix || channel < 0) return -1;
    const size_t plen = strlen(prefix);
    if (plen > (size_t) fxk::FX_OSC_PREFIX_MAX) return -1;
    int digits = 1;
    for (int t = channel; t >= 10; t /= 10) digits++;
    return (((int) plen + digits + 4) & ~3) + 16 + 48;
}

int fx_osc_encode_batch(const char* prefix, int first_channel, int num_channels, const float* smoothed, unsigned char* out, int stride, int* lengths)
{
    if (!prefix || !smoothed || !out || num_channels < 0 || first_channel < 0 || (stride & 3)) return -1;
    if (num_channels == 0) return 0;
    if (first_channel > 0x7fffffff - num_channels) return -1;
    const int longest = fx_osc_message_bytes(prefix, first_channel + num_channels - 1);
    if (longest < 0 || stride < longest) return -1;
    const size_t plen = strlen(prefix);
    char address[fxk::FX_OSC_PREFIX_MAX + 16];
    memcpy(address, prefix, plen);
    for (int c = 0; c < num_channels; c++) {
        snprintf(address + plen, sizeof address - plen, "%d", first_channel + c);
        unsigned char* slot = out + (size_t) c * (size_t) stride;
        const int n = fx_osc_encode(address, smoothed + (size_t) c * FX_NUM_FEATURES, slot, stride);
        if (n < 0) return -1;
        memset(slot + n, 0, (size_t) (stride - n));
        if (lengths) lengths[c] = n;
    }
    return num_channels;
}

} // extern "C"
