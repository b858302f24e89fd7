/*
 * fx.h -- C ABI of libfx_hip.so: the MI355X (gfx950) implementation of the
 * reference's per-frame audio feature path
 *     RealTimeAnalyser -> SpectralCharacteristics / HarmonicCharacteristics /
 *     PitchAnalyser  ->  AudioFeatures (12 floats)  ->  OSC feature message.
 *
 * The reference (SeanSoraghan/Feature-Extractor) has no FFI layer; the seam this
 * library replaces is "analysis window in -> AudioFeatures::updateFeature out".
 * Every entry point cites the reference interface it stands in for; paths are
 * relative to the reference's Source/ directory.
 *
 * Conventions
 *   - plain C types only; every function returns an fx_status (0 = FX_OK) and
 *     fx_last_error() returns a description of the last failure on this thread.
 *   - a context analyses `num_channels` independent mono channels (one
 *     AnalyserTrackController each, AnalyserTrackController.h:199-206) on ONE
 *     GPU.  Calls on one context must be externally serialised.
 *   - all work is enqueued on the context's HIP stream; fx_sync() waits for it.
 *   - feature vectors use the AudioFeatures slot order (RealTimeAnalyser.h:19-30).
 *   - numeric behaviour (NaN, inf, values > 1) is the reference's; nothing is
 *     clamped or "fixed".
 *   - there is NO CPU fallback: if no gfx950 device is usable every call fails
 *     with FX_ERR_NO_DEVICE.
 */
#ifndef FX_H
#define FX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_ABI_VERSION 6

typedef int fx_status;
enum {
    FX_OK = 0,
    FX_ERR_INVALID_ARGUMENT = 1,
    FX_ERR_NO_DEVICE = 2,
    FX_ERR_HIP = 3,
    FX_ERR_OUT_OF_MEMORY = 4,
    FX_ERR_UNSUPPORTED = 5
};

/* AudioFeatures::eAudioFeature, RealTimeAnalyser.h:17-32 */
enum {
    FX_ONSET = 0, FX_RMS, FX_F0, FX_CENTROID, FX_SPREAD, FX_FLATNESS, FX_LER,
    FX_FLUX, FX_SLOPE, FX_HER, FX_OER, FX_INHARM, FX_NUM_FEATURES
};

/* OnsetDetector::eOnsetDetectionType, SpectralCharacteristics.h:213-219 */
enum { FX_ONSET_SPECTRAL = 0, FX_ONSET_AMPLITUDE = 1, FX_ONSET_COMBINATION = 2 };

/* Where a caller buffer lives. */
enum { FX_MEM_HOST = 0, FX_MEM_DEVICE = 1 };

/* Sample formats accepted for audio input.  FX_SAMPLE_S16 is 16-bit signed PCM, little endian as the host has it: the load
 * stage of the kernels turns a sample v into the float v / 32768 -- exactly what JUCE's WAV reader (and include/fx_wav.hpp)
 * makes of a 16-bit file before AudioDataCollector sees it (AudioFilePlayer.h:41-61, AudioDataCollector.h:42-64) -- so an
 * integer source crosses PCIe at two bytes per sample and every result is bit for bit that of the decoded floats.
 * FX_SAMPLE_S24 is 24-bit signed PCM PACKED in three bytes per sample, little endian (the layout of a 24-bit WAV file's data
 * chunk): v / 8388608, again the reader's float exactly; a hop of window_size/2 samples is window_size/2 * 3 bytes. */
enum { FX_SAMPLE_F32 = 0, FX_SAMPLE_F16 = 1, FX_SAMPLE_S16 = 2, FX_SAMPLE_S24 = 3 };

/* fx_create flags.  The first three fix the order in which the reference's two
 * analysis threads write the ONE shared AudioFeatures object per hop
 * (AnalyserTrackController.h:20-21; a data race in the reference):            */
#define FX_ORDER_SPECTRAL_THEN_HARMONIC 0u   /* default */
#define FX_ORDER_HARMONIC_THEN_SPECTRAL 1u
#define FX_ORDER_ISOLATED               2u   /* one AudioFeatures per analyser */
#define FX_ORDER_MASK                   3u
/* Construct only one of the two analysers (the reference builds both, AnalyserTrackController.h:20-21).
 * Slots the absent analyser would write keep their initial state: raw 0, getValue NaN (0.0f / 0, as
 * AudioFeatures::getValue returns before any insert, RealTimeAnalyser.h:84-88). */
#define FX_SPECTRAL_ONLY                4u   /* RealTimeSpectralAnalyser only: RMS, centroid..slope, onset */
#define FX_HARMONIC_ONLY                8u   /* RealTimeHarmonicAnalyser only: RMS, F0, HER, OER, inharmonicity */
/* The low-latency kernel family, for hosts that run the reference's own cadence -- one analysis per hop as it arrives
 * (AudioDataCollector.h:66-94) -- and care about the round trip of that one hop rather than about frames per second.
 * Windows of 2048 and 4096 points (both analysers): every analysis frame is spread over a PAIR of wavefronts, which halves the
 * arithmetic on a hop's critical path (4096-point hop 35 -> 31 us) and costs throughput on long calls (2048 points: -25 %).
 * Results: every discrete decision (onset, pitch lag, peaks, gates) is identical to the default family's; the continuous slots
 * may differ from it in the last bit (sums over bins are added in another order).  One family per context, from fx_create to
 * fx_destroy, so a channel's smoothing history never mixes the two.  No effect on other window sizes or with one analyser. */
#define FX_LOW_LATENCY                  16u

typedef struct fx_context fx_context;

/* Replaces the RealTimeSpectralAnalyser + RealTimeHarmonicAnalyser constructor
 * pair, (AudioDataCollector&, AudioFeatures&, int windowSize, double sampleRate)
 * -- RealTimeAnalyser.h:100,136,196 -- for `num_channels` channels at once.
 * window_size: power of two in [256, 4096] (the collector ring is 4096,
 * AudioDataCollector.h:24; the app uses 2048, AnalyserTrackController.h:20-21). */
fx_status fx_create(fx_context** out, int device_id, int num_channels,
                    int window_size, double sample_rate, unsigned flags);
fx_status fx_destroy(fx_context* ctx);

/* Zero overlap buffers, flux state and every ValueHistory (a freshly
 * constructed AnalyserTrackController); settings are kept. */
fx_status fx_reset_state(fx_context* ctx);

/* RealTimeAnalyser::sampleRateChanged, RealTimeAnalyser.h:111-114 */
fx_status fx_set_sample_rate(fx_context* ctx, double sample_rate);
/* RealTimeSpectralAnalyser::setOnsetDetectionSensitivity, RealTimeAnalyser.h:244-248 */
fx_status fx_set_onset_sensitivity(fx_context* ctx, float sensitivity);
/* RealTimeSpectralAnalyser::setOnsetWindowLength, RealTimeAnalyser.h:250-254
 * (both onset histories are emptied, as ValueHistory::setHistoryLength does);
 * 1 <= length <= 32 (the GUI offers 3..21, AudioFeaturesListComponent.h:142-156) */
fx_status fx_set_onset_window(fx_context* ctx, int length);
/* RealTimeSpectralAnalyser::setOnsetDetectionType, RealTimeAnalyser.h:258 */
fx_status fx_set_onset_type(fx_context* ctx, int type);
/* AudioDataCollector::setGain, AudioDataCollector.h:124 (applied to hops only,
 * as AudioDataCollector::getAnalysisBuffer does at :88) */
fx_status fx_set_gain(fx_context* ctx, float gain);

/* Replaces RealTimeAudioDataOverlapper::getNextBuffer (RealTimeAudioAnalysis.h:
 * 205-228) + both run() loops (RealTimeAnalyser.h:141-177, :201-234) for
 * `num_hops` consecutive hops of window_size/2 samples per channel.
 *   hops         [num_channels][num_hops][window_size/2]  samples (sample_format)
 *   out_raw      [num_channels][num_hops][12]  values passed to updateFeature,
 *                or NULL
 *   out_smoothed [num_channels][num_hops][12]  AudioFeatures::getValue of every
 *                slot after each hop (RealTimeAnalyser.h:84-88), or NULL
 * Buffers are caller-owned and, for FX_MEM_DEVICE, must stay valid until the
 * stream reaches this call's work (fx_sync); a FX_MEM_DEVICE input must start on a
 * 16-byte boundary in every sample format (FX_ERR_INVALID_ARGUMENT otherwise: the
 * kernels read 16 bytes per lane, and a packed 24-bit sample may sit at any byte of
 * them).  FX_MEM_HOST buffers are copied synchronously. */
fx_status fx_push_hops(fx_context* ctx, const void* hops, int num_hops, int sample_format,
                       int mem_kind, float* out_raw, float* out_smoothed);

/* The collector's own interface: a device block of ANY length.  Replaces AudioDataCollector::audioDeviceIOCallback
 * (AudioDataCollector.h:36-70: whatever `numberOfSamples` the audio device delivers -- 441, 480, 512 ... -- goes into the ring)
 * together with the reader that pulls window_size/2 samples whenever they are there (getAnalysisBuffer, :72-94, called by
 * RealTimeAudioDataOverlapper::getNextBuffer, RealTimeAudioAnalysis.h:205-219) for all channels at once.
 *   samples      [num_channels][num_samples]  the block of every channel, rows back to back; num_samples >= 0
 * The context keeps what a block leaves over -- fewer than window_size/2 samples per channel, un-gained, in device memory --
 * analyses floor((pending + num_samples) / (window_size/2)) hops exactly as fx_push_hops would analyse the same samples cut
 * into hops (bit for bit), and keeps the rest.  The gain (fx_set_gain) is applied when a hop is analysed, as getAnalysisBuffer
 * applies it at read time (:88): a change reaches samples that are still pending.
 *   out_raw / out_smoothed  [num_channels][frames][12] with frames = (fx_pending_samples() + num_samples) / (window_size/2),
 *                           or NULL;  *frames_out (may be NULL) receives that count (0 is not an error)
 * FX_MEM_DEVICE blocks must start on a 4-byte boundary (on a 16-byte boundary with whole hops and nothing pending they are
 * analysed in place).  A block that completes one or two hops -- an audio device's 441 / 480 / 512 / 960 / 1024 samples against
 * windows of 1024 points and more -- is read by the analysis kernels directly, and so is a block of any length up to 4096 hops
 * against a 1024-point window; other calls pass through a re-blocking kernel first (same results, one more pass over the
 * samples).  Every channel's block has the same length (one device callback feeds all collectors,
 * AnalyserTrackController.h:31-41).
 * If the analysis of the completed hops fails (an allocation, a launch), the error is returned and those hops are lost -- the pending
 * samples have already moved on -- so the stream is no longer the caller's: fx_reset_state.  The sample format may change only while nothing is pending.  While samples are pending, fx_push_hops,
 * fx_process_frames and fx_stream_submit refuse (whole hops would overtake them); fx_reset_state drops them. */
fx_status fx_push_samples(fx_context* ctx, const void* samples, int num_samples, int sample_format, int mem_kind,
                          float* out_raw, float* out_smoothed, int* frames_out);
/* samples per channel held back by fx_push_samples / fx_stream_submit_samples, 0 <= n < window_size/2 */
int fx_pending_samples(fx_context* ctx);
/* AudioDataCollector::clearBuffer (AudioDataCollector.h:122; the transport buttons, AnalyserTrackController.h:135-137,167-171):
 * the pending samples become zeros; their count, like the ring's indices, stays. */
fx_status fx_clear_pending(fx_context* ctx);

/* Same analysis on already assembled windows (the `audioWindow` each run()
 * loop sees, RealTimeAnalyser.h:147,206): frames [num_channels][num_frames]
 * [window_size].  Gain is not applied.  The overlap state is left holding the
 * second half of each channel's last frame. */
fx_status fx_process_frames(fx_context* ctx, const void* frames, int num_frames, int sample_format,
                            int mem_kind, float* out_raw, float* out_smoothed);

/* Latest AudioFeatures::getValue of every slot, [num_channels][12]
 * (what OSCFeatureAnalysisOutput::sendSpectralFeaturesViaOSC samples,
 * OSCFeatureAnalysisOutput.h:91-104). */
fx_status fx_get_smoothed(fx_context* ctx, float* out, int mem_kind);

/* Page-locked host memory for buffers a host hands to the FX_MEM_HOST entry points again and again (a live engine's block FIFO, its result
 * buffers): copies to and from it run at the link's rate, where ordinary memory goes through the runtime's staging copy first.  Needs a
 * device (FX_ERR_NO_DEVICE without one); any thread; free with fx_host_free.  Using it is optional: every entry accepts ordinary memory. */
fx_status fx_host_alloc(void** out, size_t bytes);
fx_status fx_host_free(void* p);

/* Wait for all enqueued work of this context.  FX_ERR_HIP if a kernel of this context reported a failed
 * hand-over between work units (the results of that call and of the calls after it are not valid;
 * fx_reset_state clears the condition). */
fx_status fx_sync(fx_context* ctx);

/* The context's hipStream_t (as void*), so callers can order their own copies. */
fx_status fx_get_stream(fx_context* ctx, void** stream);

/* Device time of the analysis kernels of the most recent fx_push_hops /
 * fx_process_frames call, measured with HIP events on the context's stream
 * (milliseconds; synchronises). kernel 0 = frame kernel, 1 = smoothing/onset.
 * FX_ERR_INVALID_ARGUMENT if that call recorded none: by default one-frame calls
 * do not (fx_tuning::call_timing). */
fx_status fx_last_kernel_ms(fx_context* ctx, float* frame_kernel_ms, float* epilogue_kernel_ms);

/* ---- streaming ingest: replaces AudioDataCollector's ring + busy-wait reader ----
 * (AudioDataCollector.h:24,36-94: audio thread writes a 4096-sample ring, the analysis thread spins
 * until a hop is available.)  Here the producer owns a ring of `slots` PINNED host batches, each
 * [num_channels][hops_per_batch][window_size/2] samples of `sample_format` (any of FX_SAMPLE_*).
 * fx_stream_submit() enqueues the H2D copy of the filled slot on a side HIP stream, its analysis on the
 * context's stream behind an event and the copy of the vectors back on a third stream, so the samples
 * of batch k+1, the kernels of batch k and the results of batch k-1 travel at the same time (PCIe in
 * both directions while the kernels run: 94-98 % of the pinned-memcpy rate on MI355X for batches of
 * tens of megabytes); results return in submission order.
 * With hops_per_batch == 1 -- the reference's own cadence, one analysis per hop as it arrives -- a
 * submit is ONE kernel launch that reads the hop from the pinned slot, writes the vectors back to it
 * and raises a flag that fx_stream_collect() polls (32-33 us per 4096-sample window on MI355X; 29 us
 * with FX_LOW_LATENCY). */
typedef struct fx_stream fx_stream;
fx_status fx_stream_create(fx_context* ctx, int hops_per_batch, int slots, int sample_format, fx_stream** out);
fx_status fx_stream_destroy(fx_stream* s);
/* Next free slot to fill (pinned host memory).  Never blocks: FX_ERR_INVALID_ARGUMENT while every slot is still in flight
 * (fx_stream_collect frees the oldest) or while an acquired slot has not been submitted. */
fx_status fx_stream_acquire(fx_stream* s, void** host_slot);
/* Hand the acquired slot to the GPU (asynchronous). */
fx_status fx_stream_submit(fx_stream* s);
/* acquire + copy + submit for a producer whose batch sits in ordinary host memory: `hops` ([num_channels][hops_per_batch]
 * [window_size/2] samples of the stream's format) is copied into the next pinned slot by `fill_threads` host threads (1..64; a
 * persistent pool inside the stream: one memcpy thread moves ~12 GB/s, PCIe Gen5 takes 55) and the slot is submitted.  A
 * producer that can write its samples into the slot directly (fx_stream_acquire) saves that copy altogether. */
fx_status fx_stream_push(fx_stream* s, const void* hops, int fill_threads);
/* Wait for the OLDEST submitted batch and copy its results out: raw / smoothed
 * [num_channels][hops_per_batch][12] host floats (either may be NULL).  FX_ERR_INVALID_ARGUMENT
 * if nothing is in flight. */
fx_status fx_stream_collect(fx_stream* s, float* out_raw, float* out_smoothed);
/* The ring's form of fx_push_samples: the acquired slot holds a block of num_samples samples per channel, [num_channels]
 * [num_samples] with rows back to back, 0 <= num_samples <= hops_per_batch * window_size/2 (a device callback's block, or several
 * of them); the context's pending samples and the block are cut into hops on the device, analysed, and the rest is kept.  The
 * batch yields frames = (pending + num_samples) / (window_size/2) vectors per channel, which fx_stream_collect_samples reports.
 * Always the three-queue path (samples in, kernels, vectors back).  If it fails the slot is handed back and the ring stays usable, but -- as
 * with fx_push_samples -- the pending samples may have moved on without the hops this block completed: fx_reset_state, do not submit the
 * same block again. */
fx_status fx_stream_submit_samples(fx_stream* s, int num_samples);
/* acquire + copy (fill_threads host threads) + fx_stream_submit_samples for a block in ordinary host memory */
fx_status fx_stream_push_samples(fx_stream* s, const void* samples, int num_samples, int fill_threads);
/* fx_stream_collect that also says how many frames per channel the batch produced: raw / smoothed [num_channels][frames][12]
 * (room for hops_per_batch frames per channel is always enough) */
fx_status fx_stream_collect_samples(fx_stream* s, float* out_raw, float* out_smoothed, int* frames_out);
/* Number of submitted batches not yet collected. */
int fx_stream_in_flight(fx_stream* s);

/* ---- launch-shape knobs (experiments, tests) ----
 * Within a kernel family none of these changes a result bit (tests/test_gpu_parity.py pins that); they choose workgroup
 * shapes, how long calls are cut into work units, and which of the equivalent streaming paths runs.  The one knob that
 * selects a FAMILY is waves_per_frame (see FX_LOW_LATENCY): fx_set_tuning refuses to change it once the context has
 * analysed frames (FX_ERR_INVALID_ARGUMENT until fx_reset_state), so the two are never mixed in one history.  A context
 * takes its knobs ONCE, in fx_create, from the FX_* environment variables named below (later changes of
 * the environment have no effect: no entry point on the analysis path reads the environment);
 * fx_set_tuning replaces them explicitly.  A stream takes the stream_* knobs in fx_stream_create. */
#define FX_MAX_UNITS 24
typedef struct fx_tuning {
    int waves_per_channel;       /* FX_WAVES: frames of one channel in flight in a workgroup; 0 = measured best */
    int channels_per_workgroup;  /* FX_CHANNELS_PER_WG; 0 = measured best */
    int waves_per_frame;         /* FX_WAVES_PER_FRAME: kernel family -- 1 = a frame lives in one wavefront, 2 = in a pair of wavefronts
                                    (windows >= 2048 only); 0 = as the create flags say (pairs with FX_LOW_LATENCY, else one) */
    int frames_per_unit;         /* FX_FRAMES_PER_CHUNK: work-unit length of a call cut in time; 0 = never cut; -1 = measured best */
    int unit_plan_len;           /* FX_CHUNK_PLAN=a,b,...: explicit unit lengths (used when they add up to the call's frames) */
    int unit_plan[FX_MAX_UNITS];
    int stream_graph;            /* FX_STREAM_GRAPH: 1 / 0 force / forbid the captured hipGraph step; -1 = by batch size */
    int stream_hop_kernel;       /* FX_STREAM_HOP_KERNEL: 0 forbids the one-launch hop kernel in the ring; -1 = when it applies */
    int stream_zero_copy;        /* FX_STREAM_ZEROCOPY: 1 / 0 force / forbid zero-copy slots in the captured step; -1 = by size */
    int one_hop_kernel;          /* FX_ONE_HOP_KERNEL: 0 / 1 = fx_push_hops / fx_process_frames of ONE frame per channel never / always run
                                    the one-launch hop kernel; -1 = where it is the faster of the two (channels x window <= 2^20 samples; 2^22 for 4096-point windows on the default family) */
    int call_timing;             /* FX_CALL_TIMING: whether an analysis call records the three events fx_last_kernel_ms() reads: 1 / 0 = every /
                                    no call; -1 = calls of more than one frame per channel (a one-frame call is the live, latency-critical
                                    case, and the events cost it 13 of its 27 us back to back: each is a barrier packet between launches).
                                    Calls between fx_profile_begin / fx_profile_end are timed regardless */
    int handover_spin_limit;     /* FX_HANDOVER_SPINS: polls a work unit spends waiting for its predecessor's flux state
                                    before it gives up and the call is reported failed (FX_ERR_HIP); 0 = default (1 << 22) */
    int stream_fill_streaming;   /* FX_STREAM_FILL_STREAMING: fx_stream_push's fill threads write the pinned slot with non-temporal stores
                                    (1, and -1 = default) or with memcpy (0) */
} fx_tuning;
void fx_tuning_defaults(fx_tuning* t);     /* every knob "measured best" */
void fx_tuning_from_env(fx_tuning* t);     /* defaults overridden by the FX_* variables set right now */
fx_status fx_get_tuning(fx_context* ctx, fx_tuning* out);
fx_status fx_set_tuning(fx_context* ctx, const fx_tuning* t);

/* Host-only arithmetic, exposed for testing: how a call of `num_frames` frames per channel is cut into
 * work units for the frame kernel (several workgroups per channel, each analysing a run of consecutive
 * frames and handing the channel's flux state -- previousBinMagnitudes, SpectralCharacteristics.h:203 --
 * to the next through device memory).  Writes the unit lengths to sizes[0..n) and returns n (1 = one
 * workgroup per channel); the lengths are positive and add up to num_frames.  `tuning` may be NULL
 * (defaults).  Pure: reads neither the environment nor a device. */
int fx_plan_units(int window_size, unsigned flags, int waves_per_channel, int num_frames,
                  const fx_tuning* tuning, int* sizes, int cap);

/* Host-only, exposed for testing: which symmetries THIS host's cos / sin give the reference's float twiddle table
 * (juce::FFT's table: phase in double, entries rounded to float; ref SURVEY.md App. A.1) for a window size.
 * bit 0: the mirror symmetry of the 16-point first pass's constants -- the kernels rely on it and fx_create refuses a
 *        host without it (FX_ERR_UNSUPPORTED);
 * bit 1: table[j + N/4] == (table[j].im, -table[j].re) for the entries the 4096-point kernel's compact twiddle image
 *        forms that way -- a host without it gets them read from the whole table instead (slower, same values).
 * Sizes that use neither report the bit as set.  Pure: no device. */
int fx_twiddle_symmetry(int window_size);

/* Kernel-time accounting over a region of calls: fx_profile_begin() starts recording a HIP event
 * triple per analysis call on the context's stream (no synchronisation, at most 4096 calls);
 * fx_profile_end() synchronises and returns the summed device time of the frame kernel and of the
 * smoothing/onset kernels and the number of calls recorded. */
fx_status fx_profile_begin(fx_context* ctx);
fx_status fx_profile_end(fx_context* ctx, double* frame_kernel_ms, double* epilogue_kernel_ms, int* calls);

/* ---- multi-GPU: one process per GPU, channels sharded by contiguous blocks ----
 * Channels are independent (one AnalyserTrackController each, AnalyserTrackController.h:199-210), so
 * the analysis itself needs no exchange.  The only collective on the path is the gather of every
 * rank's latest smoothed vectors to the rank that owns the OSC sink -- the two OSCFeatureAnalysisOutput
 * senders of a track (AnalyserTrackController.h:22-23) sample AudioFeatures::getValue at 60 Hz and
 * send it (OSCFeatureAnalysisOutput.h:89-136).  It runs over RCCL (xGMI between the GPUs of a node).
 *
 *   rank 0:     fx_comm_unique_id(id)  -> hand the FX_COMM_ID_BYTES bytes to every rank (any channel)
 *   every rank: fx_comm_create(ctx, rank, world, id, FX_COMM_ID_BYTES)       (collective)
 *   per step:   fx_gather_smoothed(ctx, dst_rank, out, mem_kind)             (collective, asynchronous)
 *               fx_comm_sync(ctx) before reading `out`
 */
#define FX_COMM_ID_BYTES 128
/* ncclGetUniqueId.  librccl is loaded on first use of an fx_comm_* entry (single-GPU users never map it);
 * FX_ERR_UNSUPPORTED if it cannot be loaded. */
fx_status fx_comm_unique_id(void* id_out, int id_bytes);
/* Join this context to a communicator of `world_size` ranks (ncclCommInitRank on the context's
 * device; every rank calls it with the same id).  Ranks may own different channel counts; the
 * counts are exchanged here.  One communicator per context. */
fx_status fx_comm_create(fx_context* ctx, int rank, int world_size, const void* unique_id, int id_bytes);
fx_status fx_comm_destroy(fx_context* ctx);
/* Channels of all ranks (sum of their num_channels) and the channel offset of `rank`'s block. */
fx_status fx_comm_layout(fx_context* ctx, int* total_channels, int* first_channel_of_rank /*[world_size] or NULL*/);
/* Snapshot this rank's latest smoothed vectors [num_channels][12] (ordered after the analysis
 * calls made so far) and gather all ranks' blocks, in rank order, to `dst_rank`:
 *   out [total_channels][12] floats on dst_rank (FX_MEM_DEVICE or FX_MEM_HOST), ignored elsewhere.
 * The exchange runs on a side stream behind an event, double-buffered, so it overlaps the kernels of
 * the next analysis call; nothing blocks the host.  `out` is valid after fx_comm_sync().  */
fx_status fx_gather_smoothed(fx_context* ctx, int dst_rank, float* out, int mem_kind);
/* Wait for every gather issued so far on this context. */
fx_status fx_comm_sync(fx_context* ctx);
/* What a scaling run wants to see of the exchange: RCCL's own rank count of the communicator (ncclCommCount), the gathers issued, how many
 * of them were timed (HIP events around the send / receive group on the side stream; a gather still running when its staging slot comes
 * round again is not counted), and their summed and longest device time in milliseconds.  Any pointer may be NULL. */
fx_status fx_comm_stats(fx_context* ctx, int* rccl_ranks, int* gathers, int* gathers_timed, double* total_ms, double* max_ms);

/* ---- the reference's LEGACY offline analyser: struct AudioAnalyser (AudioAnalysis.h), SURVEY.md 8(f) rank 4 ----
 * Never instantiated by the reference application (AudioAnalysis.h:221-246 is commented out); provided for hosts that
 * want its features.  An fx_offline stands for `num_channels` AudioAnalyser objects (one per channel: the histogram-F0
 * estimate keeps `previousF0`, AudioAnalysis.h:109,273-293,697) on one GPU.  Buffers are [num_channels][...] row-major,
 * FX_MEM_HOST (copied synchronously) or FX_MEM_DEVICE (asynchronous on the object's stream; fx_offline_sync waits). */
typedef struct fx_offline fx_offline;
/* AudioAnalyser (windowSize, numChannels, nyquistFrequency, ...), AudioAnalysis.h:107-125 */
fx_status fx_offline_create(fx_offline** out, int device_id, int num_channels, double nyquist);
fx_status fx_offline_destroy(fx_offline* o);
fx_status fx_offline_reset(fx_offline* o);                         /* previousF0 := 0, as constructed (AudioAnalysis.h:109) */
fx_status fx_offline_sync(fx_offline* o);
fx_status fx_offline_get_previous_f0(fx_offline* o, double* out /*[num_channels]*/);
/* AudioAnalyser::analyseNormalisedZeroCrosses, AudioAnalysis.h:517-541: audio [C][num_samples] ->
 * out [C][num_downsamples] (the ZeroCrosses feature row, AudioAnalysis.h:538) */
fx_status fx_offline_zero_crosses(fx_offline* o, const float* audio, int num_samples, int num_downsamples, float* out, int mem_kind);
/* AudioAnalyser::setLogAttackTime, AudioAnalysis.h:611-622: envelope [n] (channel 0 of the energy envelope), the length of the
 * analysed audio, its number of downsamples and the (integer, AudioFeatures.h:303) sample rate -> estimatedLogAttackTime */
fx_status fx_offline_log_attack_time(fx_offline* o, const float* envelope, int n, int num_input_samples, int num_downsamples,
                                     int sample_rate, float* out, int mem_kind);
/* AudioAnalyser::calculateFFTLBP, AudioAnalysis.h:543-564 (which only prints): cur, prev [C][num_bins] magnitude frames ->
 * bits [C][num_bins] (|cur - prev| > 0.1f), highest_ratio [C] (last bin over the threshold / num_bins), activity_ratio [C] */
fx_status fx_offline_fft_lbp(fx_offline* o, const float* cur, const float* prev, int num_bins, unsigned char* bits,
                             float* highest_ratio, float* activity_ratio, int mem_kind);
/* AudioAnalyser::calculateHarmonicCharacteristics, AudioAnalysis.h:253-303 (peaks :350-393, frequency histogram :395-417,
 * estimateF0AndHERFromFrequencyHistogram :419-441, the octave rule :273-291, calculateInharmonicity :305-336):
 * magnitudes [C][num_bins] of one frame -> out3 [C][3] = f0, harmonicEnergyRatio, inharmonicity; 4 <= num_bins <= 4097 */
fx_status fx_offline_harmonic_characteristics(fx_offline* o, const float* magnitudes, int num_bins, float* out3, int mem_kind);
/* AudioAnalyser::calculateSpectralCharacteristics, AudioAnalysis.h:463-515 (the legacy full-spectrum form): magnitudes [C][num_bins] of one
 * frame -> out4 [C][4] = centroid / nyquist, spread, flatness, flux (AudioAnalysis.h:17-29).  Each channel's previousBinMagnitudes
 * (:120-121, :510, :700) is kept across calls and replaced only by frames that pass the 0.001 gate (:498-500); its length is fixed by
 * the first call (an analyser's window size) until fx_offline_reset.  1 <= num_bins <= 4097 */
fx_status fx_offline_spectral_characteristics(fx_offline* o, const float* magnitudes, int num_bins, float* out4, int mem_kind);
fx_status fx_offline_get_previous_bins(fx_offline* o, double* out /*[num_channels][num_bins]*/, int num_bins);
/* AudioAnalyser::calculateNormalisedSpectralSlope, AudioAnalysis.h:566-609: magnitudes [C][num_bins] -> out [C] */
fx_status fx_offline_spectral_slope(fx_offline* o, const float* magnitudes, int num_bins, float* out, int mem_kind);
/* AudioAnalyser::getConjugateComplexMultiplicationInPlace + analyseAutoCorrelation (+ getMaxIndex), AudioAnalysis.h:623-665:
 * data [C][num_items][2] interleaved (r, i) is replaced IN PLACE by each item times its conjugate; peak_bin [C] = the first item holding
 * the largest real part of the products, frequency [C] (doubles) = the estimate the reference prints for it,
 * peak_bin * (nyquist / num_items) + half a bin */
fx_status fx_offline_auto_correlation(fx_offline* o, float* data, int num_items, int* peak_bin, double* frequency, int mem_kind);

/* ---- OSC sink helpers (host side, no GPU) ---- */
/* Re-order one 12-slot vector into the wire order of
 * sender.send(bundleAddress, onset, rms, f0, centroid, slope, spread, flatness,
 * ler, flux, her, oer, inharm)  -- OSCFeatureAnalysisOutput.h:107 */
void fx_pack_osc12(const float* features12, float* out12);
/* README.md:57 order: onset, rms, f0, centroid, slope, spread, flatness, flux, her, inharm */
void fx_pack_osc10(const float* features12, float* out10);
/* Encode the OSC 1.0 message OSCSender::send(address, 12 floats) emits.
 * Returns the byte count (76 for "/Audio/A0") or -1 if cap is too small. */
int fx_osc_encode(const char* address, const float* features12, unsigned char* out, int cap);


/* ---- the sink at scale: every track's message in one call, every tick's datagrams in a few system calls ----
 * The reference sends one message per track per 60 Hz tick (OSCFeatureAnalysisOutput.h:84-113,133), from two senders per track
 * (AnalyserTrackController.h:22-23), addressed "/Audio/A<row>" (MainComponent.cpp:170).  At configs[3] that is 65 536 x 60 = 3.9e6
 * datagrams a second: they are formatted in bulk (on the GPU, straight from the context's latest vectors, so that the copy to the host
 * is already the datagrams) and handed to the kernel in batches (sendmmsg), by a few threads, paced at 60 Hz. */
/* Bytes of the message for address "<prefix><channel>" (76 for "/Audio/A", 0): address + NUL padded to 4, 16 bytes of type tags,
 * 48 of floats.  -1 if prefix is NULL, longer than 64 bytes, or channel < 0. */
int fx_osc_message_bytes(const char* prefix, int channel);
/* fx_osc_encode for num_channels tracks at once, on the host: message c = "<prefix><first_channel + c>" with smoothed[c][12]
 * (AudioFeatures slot order) is written at out + c * stride and its length to lengths[c] (lengths may be NULL); the rest of each
 * slot is zeroed.  stride: a multiple of 4, >= fx_osc_message_bytes(prefix, first_channel + num_channels - 1).
 * Returns num_channels, or -1 on a bad argument. */
int fx_osc_encode_batch(const char* prefix, int first_channel, int num_channels, const float* smoothed12,
                        unsigned char* out, int stride, int* lengths);
/* The same bytes for every channel of the context, formed ON THE DEVICE from the latest AudioFeatures::getValue vectors (what
 * fx_get_smoothed returns), ordered after the analysis calls made so far.  out: [num_channels][stride] bytes, FX_MEM_HOST (the call
 * returns when they are there) or FX_MEM_DEVICE (4-byte aligned; asynchronous on the context's stream -- pinned host memory mapped
 * into the device works the same way); lengths: host array [num_channels] or NULL. */
fx_status fx_get_osc_datagrams(fx_context* ctx, const char* prefix, int first_channel, unsigned char* out, int stride,
                               int* lengths, int mem_kind);

/* The sender: one UDP socket per target and thread, sendmmsg in chunks, `threads` sender threads each owning a slice of the tracks,
 * a 60 Hz timer (OSCFeatureAnalysisOutput::startTimerHz (60), :133) and a primary plus an optional secondary target
 * (AnalyserTrackController.h:22-23); targets are "ip[:port]", port 9000 by default, parsed as connectToAddress does (:115-123).
 * No GPU involved.  Like the reference's timer, a tick sends whatever was published last: a vector may go out twice, or never. */
typedef struct fx_osc_sender fx_osc_sender;
#define FX_OSC_SENDER_GSO 1u   /* runs of equal-length messages go to the kernel as ONE segmented send (UDP_SEGMENT) where the host's
                                  kernel supports it; the datagrams on the wire are the same */
fx_status fx_osc_sender_create(fx_osc_sender** out, const char* primary, const char* secondary /* or NULL */, int threads, unsigned flags);
fx_status fx_osc_sender_destroy(fx_osc_sender* s);
/* Publish the datagrams the next ticks send: `count` messages, message i at datagrams + i * stride, lengths[i] bytes.  Copied. */
fx_status fx_osc_sender_update(fx_osc_sender* s, const unsigned char* datagrams, int stride, const int* lengths, int count);
/* One tick, now, on the caller's thread + the sender's threads: every published message to every target; *sent (may be NULL) =
 * datagrams the kernel accepted. */
fx_status fx_osc_sender_send(fx_osc_sender* s, long long* sent);
/* startTimerHz / stopTimer */
fx_status fx_osc_sender_start(fx_osc_sender* s, double rate_hz);
fx_status fx_osc_sender_stop(fx_osc_sender* s);
typedef struct fx_osc_sender_stats {
    long long ticks;            /* timer ticks and fx_osc_sender_send calls so far */
    long long late_ticks;       /* timer ticks that began after the next one was due (the sender does not keep up with the rate) */
    long long datagrams;        /* accepted by the kernel */
    long long dropped;          /* refused by the kernel (full socket buffer: EAGAIN / ENOBUFS) or failed */
    long long syscalls;         /* sendmmsg / sendmsg calls */
    double    last_tick_ms, max_tick_ms, total_tick_ms;   /* time to hand a tick's datagrams to the kernel */
} fx_osc_sender_stats;
fx_status fx_osc_sender_get_stats(fx_osc_sender* s, fx_osc_sender_stats* out);

/* A counting receiver for tests, benchmarks and soak runs of the sender (loopback diagnostics, not part of the analysis path):
 * `threads` sockets bound to bind_address ("127.0.0.1:0" = any free port) with SO_REUSEPORT, drained by recvmmsg.  With
 * keep_channels > 0 and a prefix it also keeps the newest message of every address "<prefix><n>", n < keep_channels. */
typedef struct fx_osc_receiver fx_osc_receiver;
#define FX_OSC_RECEIVER_NO_GRO 1u   /* do not ask for UDP_GRO: every datagram makes its own way through the receiving stack, as it would from a
                                       remote sender (with it, a segmented send over loopback arrives whole and is split in user space) */
fx_status fx_osc_receiver_create(fx_osc_receiver** out, const char* bind_address, int threads, const char* prefix, int keep_channels, unsigned flags);
fx_status fx_osc_receiver_destroy(fx_osc_receiver* r);
int fx_osc_receiver_port(fx_osc_receiver* r);
/* datagrams and bytes received so far, and those that were not an OSC message of twelve floats; any pointer may be NULL */
fx_status fx_osc_receiver_get_stats(fx_osc_receiver* r, long long* datagrams, long long* bytes, long long* malformed);
/* newest message kept for `channel`: copied to out (cap bytes), its length to *len (0 = none seen yet) */
fx_status fx_osc_receiver_last(fx_osc_receiver* r, int channel, unsigned char* out, int cap, int* len);

const char* fx_last_error(void);
int fx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FX_H */
