// fx_kernels.h -- launch interface between the C-ABI shim (fx_capi.cpp) and the gfx950 kernels.
#ifndef FX_KERNELS_H
#define FX_KERNELS_H

#include <hip/hip_runtime.h>
#include "../../include/fx.h"

namespace fxk {

// Number of past frames of raw feature values kept per channel so that smoothing (10-deep
// ValueHistory) and onset detection (<= 32-deep histories of a 10-push RMS mean) of frame t can be
// evaluated without any sequential state: 32 + 9 rounded up.
constexpr int HLEN = 48;
constexpr int MAX_ONSET_WINDOW = 32;
// calls of at most this many frames per channel run finalise + smoothing/onset + history as one kernel (a quarter wavefront per channel)
constexpr int FUSED_TAIL_MAX_FRAMES = 8;

// What the frame kernel leaves per frame: every reduction over samples / bins / lags is done, the
// scalar tail (pow, log10, sqrt, a few divisions) is not.  One thread per frame finishes it in
// fx_finalise_kernel, where 64 frames share an instruction instead of one.
struct FramePart {
    double mag_sum;     // sum of re^2 over the M bins of the windowed spectrum
    double var;         // sum ((fc/nyq) - (centroid/nyq))^2 * mag (ref SpectralCharacteristics.h:137), summed as the reference writes it:
                        // valid only where `refined` is set; otherwise the finalise kernel forms it from the moments mag_sum, b1, b2
    double lhr;         // magnitudeSum at bin M/5 (inclusive)               (ref :86-87)
    double flux;        // sum of rectified differences, not yet / maxFlux   (ref :76-79)
    double flat_sum;    // flatnessMagnitudeSum                              (ref :91)
    double prod;        // magnitudeProduct with the serial IEEE semantics   (ref :92)
    double max_e;       // maxFFTMagnitude of the slope                      (ref :153-163)
    double b1;          // sum m * mag over the bins m: weightedMagnitudeSum (ref :95) = frpb * (b1 + mag_sum / 2), formed by the finalise kernel
    double vsum;        // sum (mag - mag_sum/M)^2
    double inh;         // inharmonicity before the log                      (ref HarmonicCharacteristics.h:239)
    double her_score;   // sum of the 18 probe maxima                        (ref :157-184)
    double sum_normed;  // sum of mag / max over all bins                    (ref :77); her = score / sum_normed
    double sum_sq;      // sum of the frame's squared samples: rms = (float) sqrt(sum_sq / N)  (ref RealTimeAnalyser.h:207)
    double b2;          // sum m^2 * mag: with mag_sum and b1 the moments the centroid (:127) and the spread (:135-141) are formed from
    float  cnt;         // numMagnitudesUsedInFlatnessCalculation
    float  lag;         // ref PitchAnalyser.h:188-189
    int    flags;       // bit 0: harmonic analyser ran past the 0.005 gate  (ref HarmonicCharacteristics.h:88)
    int    refined;     // `var` holds the reference's own sum (the moment form was not trustworthy for this frame, or the pair kernel summed it)
};
static_assert(sizeof(FramePart) == 128, "one 128-byte line per frame");

// Per-call scalars of a captured (hipGraph) analysis step.  A graph's kernel arguments are frozen at capture, so what
// changes from call to call lives here, in device memory, refreshed by a copy node at the head of the graph.
struct DynParams {
    double    nyquist;
    long long frames_before;
    long long onset_reset_frame;
    float     gain;
    float     onset_multiplier;
    int       onset_window;
    int       onset_type;
    int       hist_base;            // frames_before mod HLEN (EpilogueParams::hist_base)
};

constexpr int FX_MAX_CHUNKS = FX_MAX_UNITS;

struct FrameParams {
    const void*  in;            // frames [C][T][N] or hops [C][T][N/2]
    int          sample_format; // FX_SAMPLE_F32 / FX_SAMPLE_F16 / FX_SAMPLE_S16 / FX_SAMPLE_S24
    int          hop_mode;      // 1: `in` holds hops, windows are assembled from tail + hops
    int          T;             // frames (= hops) per channel in this call
    int          C;
    int          ch_per_wg;     // channels per workgroup (they share the twiddle table in LDS)
    int          waves_per_ch;  // wavefronts per channel = frames of a channel in flight
    int          direct_state;  // 1: a call of ONE frame per channel with both analysers -- the flux state is read and replaced in global
                                // memory (no LDS copy, no hand-over): T == 1, waves_per_ch == 1, num_chunks == 1
    // A long call over few channels is cut in time as well: num_chunks workgroups per channel group, each analysing
    // frames_per_chunk consecutive frames (many small work units keep every CU busy to the end of the launch; one
    // workgroup per channel is two rounds of 512 at the bench shape, and 8 % slower).  The one thing a chunk needs from
    // the one before it is the channel's flux state: workgroups take their (chunk, channel group) from a ticket counter,
    // chunk-major, so the predecessor's ticket is always lower -- held by a workgroup that is running or done -- and wait
    // for that channel's count of finished chunks.  queue: [0] the ticket counter, [1 + c] finished chunks of channel c;
    // zeroed before the launch.  num_chunks == 1: one workgroup per channel group, no queue.  Chunks need not be equal: long
    // ones first (little overhead), short ones last (the launch's tail is one short unit deep).
    int          num_chunks;
    unsigned*    queue;
    // A work unit that gives up waiting for its predecessor (after spin_limit polls) stores 1 here -- pinned host memory,
    // system scope -- and the host turns that into FX_ERR_HIP at the next synchronisation: a stale flux state is never
    // handed on silently.  debug_flags bit 0 (FX_HOOK_NO_HANDOVER, tests): units do not publish their hand-over, which forces the time-out.
    unsigned*    err;
    unsigned     spin_limit;
    unsigned     debug_flags;
    int          chunk_begin[FX_MAX_CHUNKS + 1];   // chunk k analyses frames [chunk_begin[k], chunk_begin[k + 1]); the last entry used is T
    float        gain;          // hop mode only (ref AudioDataCollector.h:88)
    const float* tail_in;       // [C][N/2] second half of the previous window (already gained)
    float*       tail_out;      // [C][N/2]
    float*       prev_re;       // [C][N/2] real parts of the last accepted spectral frame (flux state)
    const float* tw;            // [N][2] forward twiddles, (float)cos/sin of a double phase, in pass order (build_pass_twiddles)
    FramePart*   part;          // [C][T] per-frame partial results
    double       nyquist;
    double       bin_var;       // sum_i (i/M - 0.5)^2 / M, summed serially on the host (ref SpectralCharacteristics.h:182-189)
    float        lpf_a, lpf_b;  // ref RealTimeAudioAnalysis.h:122
    const DynParams* dyn;       // non-null in a captured step: overrides gain and nyquist
    const float* tw_image;      // 4096 points: the frame kernel's compact LDS image of `tw` (build_twiddle_image), copied as it is
    int          tw_quarter_turn;   // 4096 points: canonical[j + N/4] == (canonical[j].y, -canonical[j].x) bit for bit for the entries the
                                // frame kernel's compact twiddle image derives that way (twiddles_have_quarter_turn); 0: it reads them from `tw`
    float        tw_at_quarter[2];  // canonical[N/4] = ((float) cos(pi/2), -1): the one entry of those that is never a quarter turn of another
    float        first_tw[18];  // the <= 9 twiddles of the first FFT pass (re, im pairs): wave-uniform, so they travel as
                                // kernel arguments (SGPRs) instead of LDS reads; filled by fill_first_pass_twiddles
    // One-frame launches over a buffer of SEVERAL hops per channel (a call of two hops runs as two one-frame launches: measured faster than
    // the batch form at every size): hop-mode input is read as [C][in_hop_stride][N/2] at hop in_hop0 (+ t).  in_hop_stride == 0: [C][T][N/2].
    int          in_hop_stride, in_hop0;
    // fx_push_samples without the re-blocking pass (block_mode = 1; one-frame forms of windows >= 1024 points, both analysers: T == 1,
    // hop_mode == 1): `in` holds every channel's new BLOCK, rows of blk_in_row_bytes back to back; the hop a channel analyses is the
    // first N/2 samples of [its pending samples: blk_carry_in row, blk_carry_bytes valid | its block], and the kernel writes what is left
    // over to the channel's row of blk_carry_out (fx_blocks.hip.h).  Last in the struct: the other forms never look at these.
    int          block_mode;
    int          blk_hop0;      // the hop analysed is hop blk_hop0 of the stream (a call that completes two hops is two one-frame launches)
    int          blk_keep_rest; // this launch writes what the block leaves over (the call's last launch)
    int          blk_carry_bytes, blk_carry_row_bytes;
    long long    blk_in_row_bytes;
    const unsigned char* blk_carry_in;
    unsigned char*       blk_carry_out;
};

struct EpilogueParams {
    const FramePart* part;      // [C][T] from the frame kernel
    float*       raw;           // [C][T][12] raw values: written by fx_finalise_kernel, read by the smoothing kernel
    double       nyquist;
    double       bin_var;
    int          window;
    // what every frame's scalar tail would otherwise divide out for itself, all of it a function of the call's nyquist and window alone:
    // IEEE divisions and a square root done once on the host (the same correctly rounded values), see epilogue_constants()
    double       frpb;          // nyquist / (window / 2)                       (ref SpectralCharacteristics.h:64,105)
    double       inv_nyquist;   // 1.0 / nyquist
    double       inv_bins;      // 1.0 / (window / 2): a power of two, so x * inv_bins == x / (window / 2) exactly
    double       bin_std;       // sqrt(bin_var)                                 (ref :191)
    float*       hist;          // [C][HLEN][12] raw values of the newest HLEN frames, a RING per channel: the row of the frame with
                                // global index g (frames since the last state reset) is g mod HLEN, so a call only writes the rows of
                                // its own frames -- one row per channel for a one-hop call, not the whole table
    int          hist_base;     // frames_before mod HLEN: the ring row this call's first frame goes to
    float*       out_raw;       // [C][T][12] or nullptr
    float*       out_smoothed;  // [C][T][12] or nullptr
    int          out_stride;    // one-frame launches: the output buffers are [C][out_stride][12] and this launch's frame is number out_t0 of its
    int          out_t0;        // channel (a call of two hops = two one-frame launches); out_stride == 0: [C][T][12], frame t
    float*       latest;        // [C][12] smoothed values after the last frame
    int          C, T;
    long long    frames_before; // frames analysed since the last state reset, before this call
    long long    onset_reset_frame; // global index of the first frame after the last onset-window reset
    int          onset_window;  // history length of the OnsetDetector (default 5)
    int          onset_type;
    float        onset_multiplier;
    int          order_mode;    // FX_ORDER_*
    int          analysers;     // bit 0: spectral analyser runs, bit 1: harmonic analyser runs
    const DynParams* dyn;       // non-null in a captured step: overrides nyquist, frames_before, hist_base, onset_*
    unsigned*    clear_queue;   // a call cut in time (FrameParams::num_chunks > 1): its ticket counter and hand-over counts, which the step's
    int          clear_count;   // LAST kernel zeroes for the next call (a memset per call was two fill kernels and their gaps); else null
};

// Completion signal of a one-hop call (fx_hop_kernel): workgroups count themselves in `arrivals` (device memory, zero
// between calls); the last one resets it and stores `seq` to `host_flag` (pinned host memory, system scope), after the
// results -- which the kernel writes to pinned host memory as well -- are visible there.
// (All pointers null: a one-hop call on device-resident input through fx_push_hops -- the stream orders it, nothing to signal.)
struct HopSignal {
    unsigned* arrivals;
    unsigned* host_flag;
    unsigned  seq;
    unsigned  pad_;
    void*     stage;        // device memory, one hop per channel: the kernel copies the hop out of the pinned slot once
                            // (coalesced) and its three wavefronts read it from there (the slot is un-cached memory across PCIe)
};

// fx_push_samples (fx_reblock.hip): per channel, the byte stream [carry_in row: carry_bytes][in row: in_row_bytes] is split into
// out_row_bytes of whole hops (a multiple of 16) for `hops_out` and the remainder (< carry_row_bytes) for `carry_out`.
struct ReblockParams {
    const unsigned char* in;        // [C][in_row_bytes]: the new block, rows back to back (any alignment from the second row on)
    const unsigned char* carry_in;  // [C][carry_row_bytes]: the pending samples of each channel, carry_bytes valid
    unsigned char*       hops_out;  // [C][out_row_bytes]
    unsigned char*       carry_out; // [C][carry_row_bytes]: the other of the context's two carry buffers
    long long            in_row_bytes, out_row_bytes;
    int                  carry_bytes, carry_row_bytes;
    int                  C;
};
hipError_t launch_reblock_kernel(const ReblockParams& p, hipStream_t stream);
hipError_t clear_carry(unsigned char* carry, size_t bytes, hipStream_t stream);

// fx_get_osc_datagrams (fx_osc.hip): message c = OSC message "<prefix><first_channel + c>" with the twelve values of latest[c] in wire
// order (ref OSCFeatureAnalysisOutput.h:107), written at out + c * stride
constexpr int FX_OSC_PREFIX_MAX = 64;
struct OscParams {
    const float*   latest;        // [C][12]
    unsigned char* out;           // [C][stride], 4-byte aligned
    int            C, stride;     // stride: a multiple of 4, >= the longest message
    int            first_channel; // >= 0
    int            prefix_len;    // <= FX_OSC_PREFIX_MAX
    unsigned char  prefix[FX_OSC_PREFIX_MAX];
};
hipError_t launch_osc_kernel(const OscParams& p, hipStream_t stream);

// Re-order the reference's N-entry twiddle table (canonical[i] = (re, im) of e^{-2*pi*i/N} as floats)
// into the order the FFT passes read it; `out` has room for window_size complex entries.
void build_pass_twiddles(int window_size, const float* canonical, float* out);
// the first pass's constants, taken from the same pass-ordered table
void fill_first_pass_twiddles(int window_size, const float* pass_ordered, float* out18);
bool first_pass_twiddles_hermitian(int window_size, const float* first18);   // must hold before any launch
// FrameParams::tw_image: the frame kernel's LDS twiddle image where it is not the pass-ordered table itself (4096 points: CompactTw,
// fx_fft.hip.h).  Returns the number of complex entries written to `out` (room for window_size), 0 where the kernel copies `tw`.
int build_twiddle_image(int window_size, const float* pass_ordered, float* out);
// FrameParams::tw_quarter_turn for this table (the reference's N-entry table, canonical order); true where the kernel does not ask
bool twiddles_have_quarter_turn(int window_size, const float* canonical);

size_t frame_kernel_lds_bytes(int window_size, int channels_per_wg, int waves_per_channel, bool direct_state = false);
int frame_kernel_max_waves(int window_size);     // the frame kernel's launch bound, in wavefronts per workgroup
// measured-best workgroup shape for a window size: channels per workgroup x wavefronts per channel
void frame_kernel_preferred_shape(int window_size, int* channels_per_wg, int* waves_per_channel);
// Launches ceil(C / p.ch_per_wg) workgroups of p.ch_per_wg * p.waves_per_ch wavefronts; returns hipSuccess or the launch error.
hipError_t launch_frame_kernel(int window_size, const FrameParams& p, int analysers, hipStream_t stream);
// fills EpilogueParams::frpb .. bin_std from nyquist, window and bin_var (host and device: a captured step re-derives them from its DynParams)
__host__ __device__ inline void epilogue_constants(EpilogueParams& p)
{
    const double bins = (double) (p.window / 2);
    p.frpb = p.nyquist / bins;
    p.inv_nyquist = 1.0 / p.nyquist;
    p.inv_bins = 1.0 / bins;
    p.bin_std = sqrt(p.bin_var);
}
hipError_t launch_epilogue_kernels(const EpilogueParams& p, hipStream_t stream);
// launch_frame_kernel (p.direct_state: one frame per channel, one wavefront each) and launch_epilogue_kernels in ONE launch: the
// workgroup's first wavefronts finish its channels' hops when the frames are done (fx_frame_tail_kernel)
hipError_t launch_frame_tail_kernel(int window_size, const FrameParams& p, const EpilogueParams& ep, hipStream_t stream);
bool frame_tail_kernel_available(int window_size);   // windows of 1024 points and more (below that it is not built: it loses to two launches)
hipError_t prepare_kernels(int window_size);   // raises the kernels' dynamic-LDS limit on the CURRENT device (every fx_create: the attribute is per device)
hipError_t prepare_hop_kernel(int window_size);   // the same for fx_hop_kernel; hipSuccess where there is none for this size
// One frame across a PAIR of wavefronts (windows of 2048 / 4096 points, both analysers): p.waves_per_ch is then the number
// of pairs (= frames of a channel in flight) and a workgroup has p.ch_per_wg * p.waves_per_ch * 128 threads.
bool pair_kernel_available(int window_size);
int pair_kernel_max_pairs(int window_size);
size_t pair_kernel_lds_bytes(int window_size, int channels_per_wg, int pairs_per_channel);
hipError_t prepare_pair_kernel(int window_size);
hipError_t launch_pair_kernel(int window_size, const FrameParams& p, hipStream_t stream);
// One hop per channel, whole step in one launch (three wavefronts per channel + the tail), results and completion flag
// written by the kernel itself; p.T must be 1, p.hop_mode 1, both analysers on.  Window sizes: hop_kernel_available().
bool hop_kernel_available(int window_size);
// pairs: windows of 2048 / 4096 points with every analyser on a pair of wavefronts (six per channel; fx_hop_pair_kernel)
hipError_t launch_hop_kernel(int window_size, const FrameParams& p, const EpilogueParams& ep, const HopSignal& sig, hipStream_t stream, bool pairs = false);

} // namespace fxk

// Test hooks, NOT part of the public ABI (include/fx.h does not declare this; the library exports it for tests/ only).  Bits:
//   0  work units of a cut launch do not publish their hand-over (forces the time-out that FX_ERR_HIP reports)
//   1  the 4096-point kernel takes no twiddle from a quarter turn of another (the path of a host whose cos / sin lack that
//      symmetry: same values, read from the whole table)
//   2 / 3  one-frame calls through the batch kernels never / always finish the hop's tail in the frame kernel (default: while the
//      chip holds the call's workgroups at once)
//   4  fx_push_samples never feeds a block to the one-frame kernels directly (FrameParams::block_mode): every call re-blocks first, as
//      calls that complete several hops do
//   5  a call of two hops per channel runs the batch kernels' two-frame form instead of two one-frame launches
// None changes a result bit (bit 0 makes the call fail, as it must).
#define FX_HOOK_NO_HANDOVER      1u
#define FX_HOOK_NO_QUARTER_TURN  2u
#define FX_HOOK_TAIL_NEVER_FUSED 4u
#define FX_HOOK_TAIL_ALWAYS_FUSED 8u
#define FX_HOOK_NO_BLOCK_FEED    16u
#define FX_HOOK_NO_TWO_LAUNCHES  32u
extern "C" fx_status fx_set_tuning_internal(fx_context* ctx, unsigned test_hooks);
#endif
