// fx_context.h -- the context behind the C ABI (include/fx.h), shared by fx_capi.cpp and fx_comm.cpp.  Internal.
#ifndef FX_CONTEXT_H
#define FX_CONTEXT_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "fx_kernels.h"

fx_status fx_fail(fx_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP,       \
                           "%s failed: %s", #expr, hipGetErrorString(e_));                     \
    } while (0)

struct fx_comm;     // fx_comm.cpp
void fx_comm_release(fx_context* ctx);   // called by fx_destroy
fx_status fx_check_device_error(fx_context* ctx);   // after a synchronisation: FX_ERR_HIP if a kernel reported a failed hand-over

struct fx_context {
    int      device = 0;
    int      C = 0, N = 0;
    double   sample_rate = 48000.0;
    unsigned flags = 0;
    // settings (ref RealTimeAnalyser.h:244-258, SpectralCharacteristics.h:237-241,311, AudioDataCollector.h:129)
    float    gain = 1.0f;
    int      onset_window = 5;
    int      onset_type = FX_ONSET_AMPLITUDE;
    float    onset_multiplier = 1.7f;
    long long frames_seen = 0;
    long long onset_reset_frame = 0;

    hipStream_t stream = nullptr;
    hipEvent_t  ev[3] = {nullptr, nullptr, nullptr};
    bool        ev_valid = false;
    bool        profiling = false;
    std::vector<hipEvent_t> prof_events;     // 3 per recorded call
    size_t      prof_used = 0;

    float* d_tw = nullptr;        // [N][2]
    float* d_prev = nullptr;      // [C][N/2]
    float* d_tail[2] = {nullptr, nullptr};   // [C][N/2], ping-pong
    float* d_hist = nullptr;      // [C][HLEN][12]: per channel a ring of the newest HLEN frames' raw values (row = frame index mod HLEN)
    float* d_latest = nullptr;    // [C][12]
    int    cur = 0;
    unsigned test_hooks = 0;      // fx_set_tuning_internal (fx_kernels.h): tests only
    fx_tuning tuning;             // launch-shape knobs: taken from the environment ONCE, in fx_create (fx_set_tuning replaces them)
    unsigned* h_err = nullptr;    // pinned, coherent: a kernel stores 1 here when a work unit gave up waiting for its predecessor (sticky)
    unsigned* d_err = nullptr;    // device view of h_err
    unsigned* d_queue = nullptr;  // [1 + C]: ticket counter and per-channel chunk counts of a frame-kernel launch cut in time (FrameParams::queue)

    // fx_push_samples: what a channel's device blocks have left over, < N/2 samples in `carry_format` (AudioDataCollector's ring holds
    // them un-gained, AudioDataCollector.h:42-64,88); two buffers, the re-blocking kernel reads one and writes the other
    unsigned char* d_carry[2] = {nullptr, nullptr};   // [C][N/2 * 4 bytes]
    int    carry_cur = 0;
    int    carry_count = 0;       // samples per channel pending (the same for every channel: blocks arrive for all channels at once)
    int    carry_format = FX_SAMPLE_F32;
    unsigned char* d_hops = nullptr;    // [C][hops][N/2] samples assembled for one fx_push_samples call
    size_t hops_cap = 0;

    float* d_raw = nullptr;       // [C][T_cap][12]
    fxk::FramePart* d_part = nullptr;   // [C][T_cap]
    size_t part_cap = 0;
    void*  d_in = nullptr;        // staging for host input
    float* d_out_raw = nullptr;   // staging for host output
    float* d_out_sm = nullptr;
    size_t raw_cap = 0, in_cap = 0, out_cap = 0;

    unsigned char* d_osc = nullptr;   // [C][stride]: fx_get_osc_datagrams' messages before they go to a host buffer
    size_t osc_cap = 0;

    double bin_var = 0.0;
    float  lpf_a = 0.0f, lpf_b = 0.0f;
    float  first_tw[18] = {0};
    bool   tw_quarter_turn = false;     // FrameParams::tw_quarter_turn
    float  tw_at_quarter[2] = {0.0f, -1.0f};
    int    compute_units = 256;         // of this context's device (MI355X: 256)

    fx_comm* comm = nullptr;      // fx_comm_create (fx_comm.cpp); null for a single-GPU context
};

#endif
