// host_sanitize.cpp -- the host shim (csrc/fx_capi.cpp, csrc/fx_comm.cpp: ring, captured steps, fill pool, re-blocking plumbing, RCCL
// gather) built for the CPU against the fake runtime of tests/cpp/fake_hip/ with -fsanitize=address,undefined (and, as a second
// binary, -fsanitize=thread), then walked through every HIP call site with an injected failure.
//
// For every scenario: once without failures (everything must succeed, nothing may be left allocated), then once per HIP call k with
// call k failing.  After the failure is reported the shim must (1) not crash, overrun or leak -- the sanitizers and the fake's
// allocation table say; (2) not wedge -- with the fault gone, fx_reset_state and the next call of the same kind must succeed: a ring
// must hand out a slot again, a context must analyse again; (3) give everything back at destroy.
// This replaces, for the host side, the busy-wait reader of the reference's collector (ref AudioDataCollector.h:72-94), whose failure
// mode is a spin that never ends.
//
// Built and run by tests/test_host_sanitized_cpu.py.  usage: host_sanitize [asan | tsan]
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>        // the FAKE one (tests/cpp/fake_hip/hip)

#include "fx.h"
#include "fx_realtime.hpp"        // fx::LiveAnalyser, fx::OSCBatchSender: the live engine's threads run under the sanitizers too

namespace {

int g_problems = 0;
std::string g_where;

void problem(const char* what, const char* detail = "")
{
    std::fprintf(stderr, "PROBLEM [%s]: %s %s (last error: %s)\n", g_where.c_str(), what, detail, fx_last_error());
    g_problems++;
}

struct World {
    fx_context* ctx = nullptr;
    fx_stream*  ring = nullptr;
    std::vector<float> hops, raw, sm;
    std::vector<void*> device;          // "device" buffers of the scenario (hipMalloc of the fake)
    int C = 0, N = 0;
    int slot_samples = 0;               // capacity of a ring slot, samples per channel
    int ring_fmt = FX_SAMPLE_F32;
};

// With the fault gone: the context analyses, the ring cycles, everything can be destroyed and nothing stays allocated.
void recover_and_destroy(World& w, bool after_failure)
{
    fake_hip_fail_at(0);
    if (w.ring) {
        // whatever is in flight can be collected; a slot can be acquired and submitted again
        while (fx_stream_in_flight(w.ring) > 0)
            if (fx_stream_collect_samples(w.ring, nullptr, nullptr, nullptr) != FX_OK) { problem("collect after the fault"); break; }
    }
    if (w.ctx) {
        if (fx_reset_state(w.ctx) != FX_OK) problem("fx_reset_state after the fault");
        if (fx_push_hops(w.ctx, w.hops.data(), 2, FX_SAMPLE_F32, FX_MEM_HOST, w.raw.data(), w.sm.data()) != FX_OK) problem("fx_push_hops after the fault");
        int frames = -1;
        if (fx_push_samples(w.ctx, w.hops.data(), w.N / 2 + 3, FX_SAMPLE_F32, FX_MEM_HOST, w.raw.data(), w.sm.data(), &frames) != FX_OK || frames != 1) problem("fx_push_samples after the fault");
        if (fx_reset_state(w.ctx) != FX_OK) problem("second fx_reset_state");
    }
    if (w.ring) {
        for (int round = 0; round < 3; round++) {
            void* slot = nullptr;
            if (fx_stream_acquire(w.ring, &slot) != FX_OK) { problem("the ring is wedged: acquire after the fault"); break; }
            if (round == 2 ? fx_stream_submit_samples(w.ring, w.slot_samples / 2 + 1) != FX_OK : fx_stream_submit(w.ring) != FX_OK) {
                // (whole hops after a partial block are refused by design; the last round is the partial one)
                problem("submit after the fault"); break;
            }
            if (fx_stream_collect_samples(w.ring, nullptr, nullptr, nullptr) != FX_OK) { problem("collect after the fault (second round)"); break; }
        }
        if (fx_stream_destroy(w.ring) != FX_OK) problem("fx_stream_destroy");
        w.ring = nullptr;
    }
    if (w.ctx) { if (fx_destroy(w.ctx) != FX_OK) problem("fx_destroy"); w.ctx = nullptr; }
    for (void* p : w.device) (void) hipFree(p);
    w.device.clear();
    if (fake_hip_live() != 0) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "%ld objects / %ld bytes", fake_hip_live(), fake_hip_live_bytes());
        problem(after_failure ? "leak after an injected failure:" : "leak on the clean path:", buf);
    }
}

#define STEP(expr) do { if ((expr) != FX_OK) return false; } while (0)

bool make_context(World& w, int C, int N, unsigned flags = 0)
{
    w.C = C; w.N = N;
    w.hops.assign((size_t) C * 700 * (N / 2), 0.0f);
    for (size_t i = 0; i < w.hops.size(); i++) w.hops[i] = 0.25f * (float) ((i * 2654435761u >> 8) & 1023) / 1023.0f - 0.125f;
    w.raw.assign((size_t) C * 700 * 12, 0.0f); w.sm = w.raw;
    STEP(fx_create(&w.ctx, 0, C, N, 48000.0, flags));
    return true;
}

// A: the batch entry points, host and device buffers, cut launches, timing, the collector's interface
bool scenario_batch(World& w)
{
    if (!make_context(w, 8, 1024)) return false;
    for (int T : {1, 3, 20, 300})
        STEP(fx_push_hops(w.ctx, w.hops.data(), T, FX_SAMPLE_F32, FX_MEM_HOST, w.raw.data(), w.sm.data()));
    STEP(fx_process_frames(w.ctx, w.hops.data(), 2, FX_SAMPLE_F32, FX_MEM_HOST, w.raw.data(), nullptr));
    float a = 0, b = 0;
    STEP(fx_last_kernel_ms(w.ctx, &a, &b));
    void *d_in = nullptr, *d_raw = nullptr, *d_sm = nullptr;
    if (hipMalloc(&d_in, (size_t) 8 * 40 * 512 * 4) != hipSuccess) return false;
    w.device.push_back(d_in);
    if (hipMalloc(&d_raw, (size_t) 8 * 40 * 12 * 4) != hipSuccess) return false;
    w.device.push_back(d_raw);
    if (hipMalloc(&d_sm, (size_t) 8 * 40 * 12 * 4) != hipSuccess) return false;
    w.device.push_back(d_sm);
    STEP(fx_push_hops(w.ctx, d_in, 40, FX_SAMPLE_F32, FX_MEM_DEVICE, (float*) d_raw, (float*) d_sm));
    STEP(fx_push_hops(w.ctx, d_in, 1, FX_SAMPLE_S16, FX_MEM_DEVICE, (float*) d_raw, nullptr));
    STEP(fx_sync(w.ctx));
    STEP(fx_get_smoothed(w.ctx, w.sm.data(), FX_MEM_HOST));
    STEP(fx_profile_begin(w.ctx));
    STEP(fx_push_hops(w.ctx, w.hops.data(), 5, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, w.sm.data()));
    double fa = 0, fb = 0; int calls = 0;
    STEP(fx_profile_end(w.ctx, &fa, &fb, &calls));
    // device blocks of any length: host and device, formats, the guard rails
    int frames = 0;
    for (int n : {100, 1000, 4097, 0, 1, 511}) {
        STEP(fx_push_samples(w.ctx, w.hops.data(), n, FX_SAMPLE_F32, FX_MEM_HOST, w.raw.data(), w.sm.data(), &frames));
    }
    if (fx_pending_samples(w.ctx) != (100 + 1000 + 4097 + 1 + 511) % 512) problem("pending sample count");
    if (fx_push_hops(w.ctx, w.hops.data(), 1, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, nullptr) == FX_OK) problem("whole hops accepted while samples are pending");
    STEP(fx_clear_pending(w.ctx));
    STEP(fx_push_samples(w.ctx, d_in, 2000, FX_SAMPLE_F32, FX_MEM_DEVICE, (float*) d_raw, (float*) d_sm, &frames));
    STEP(fx_reset_state(w.ctx));
    STEP(fx_push_samples(w.ctx, d_in, 1024, FX_SAMPLE_S24, FX_MEM_DEVICE, (float*) d_raw, nullptr, &frames));      // whole hops, nothing pending: analysed in place
    STEP(fx_push_samples(w.ctx, w.hops.data(), 777, FX_SAMPLE_S24, FX_MEM_HOST, w.raw.data(), nullptr, &frames));
    STEP(fx_set_onset_window(w.ctx, 9));
    STEP(fx_set_gain(w.ctx, 0.5f));
    STEP(fx_sync(w.ctx));
    return true;
}

// B: one hop per submit -- the one-launch hop kernel path of the ring (flag polling), slots reused several times
bool scenario_ring_hop_kernel(World& w)
{
    if (!make_context(w, 2, 4096, FX_LOW_LATENCY)) return false;
    w.slot_samples = 2048; w.ring_fmt = FX_SAMPLE_F16;
    STEP(fx_stream_create(w.ctx, 1, 3, FX_SAMPLE_F16, &w.ring));
    for (int i = 0; i < 7; i++) {
        if (fx_stream_in_flight(w.ring) == 3) STEP(fx_stream_collect(w.ring, w.raw.data(), w.sm.data()));
        void* slot = nullptr;
        STEP(fx_stream_acquire(w.ring, &slot));
        std::memset(slot, 0, (size_t) 2 * 2048 * 2);
        STEP(fx_stream_submit(w.ring));
    }
    STEP(fx_stream_collect(w.ring, w.raw.data(), w.sm.data()));
    void* slot = nullptr;
    STEP(fx_stream_acquire(w.ring, &slot));
    if (fx_stream_acquire(w.ring, &slot) == FX_OK) problem("a second acquire without a submit was accepted");
    STEP(fx_stream_submit(w.ring));
    while (fx_stream_in_flight(w.ring)) STEP(fx_stream_collect(w.ring, w.raw.data(), nullptr));
    return true;
}

// C: small batches -- the captured (hipGraph) step, both parities of the context's ping-pong buffers, zero-copy and copy forms
bool scenario_ring_graph(World& w, bool zero_copy)
{
    if (!make_context(w, 4, 1024)) return false;
    fx_tuning t;
    STEP(fx_get_tuning(w.ctx, &t));
    t.stream_zero_copy = zero_copy ? 1 : 0;
    STEP(fx_set_tuning(w.ctx, &t));
    w.slot_samples = 3 * 512;
    STEP(fx_stream_create(w.ctx, 3, 2, FX_SAMPLE_F32, &w.ring));
    for (int i = 0; i < 6; i++) {
        if (fx_stream_in_flight(w.ring) == 2) STEP(fx_stream_collect(w.ring, w.raw.data(), w.sm.data()));
        STEP(fx_stream_push(w.ring, w.hops.data(), 1));
    }
    while (fx_stream_in_flight(w.ring)) STEP(fx_stream_collect(w.ring, nullptr, w.sm.data()));
    return true;
}

// D: large batches -- three queues, the fill pool, and the ring's form of fx_push_samples
bool scenario_ring_large(World& w, int max_threads)
{
    // (64 channels x 65 hops: more than 4096 frames per batch, so the ring takes the plain three-queue path; 256-point windows keep a
    // batch at 2 MiB -- enough for the fill pool to split -- and the walk short)
    if (!make_context(w, 64, 256)) return false;
    w.slot_samples = 65 * 128;
    STEP(fx_stream_create(w.ctx, 65, 3, FX_SAMPLE_F32, &w.ring));
    std::vector<float> batch((size_t) 64 * 65 * 128, 0.01f), raw((size_t) 64 * 65 * 12), sm(raw.size());
    const int threads[] = {1, 4, 2, 8, 64, 3, 1, 16, 5};
    for (int i = 0; i < 9; i++) {
        if (fx_stream_in_flight(w.ring) == 3) STEP(fx_stream_collect(w.ring, raw.data(), sm.data()));
        const int th = threads[i] < max_threads ? threads[i] : max_threads;
        STEP(fx_stream_push(w.ring, batch.data(), th));
    }
    while (fx_stream_in_flight(w.ring)) STEP(fx_stream_collect(w.ring, raw.data(), sm.data()));
    int frames = 0, total = 0;
    for (int n : {480, 8000, 0, 8320, 1}) {
        STEP(fx_stream_push_samples(w.ring, batch.data(), n, n > 5000 ? (max_threads < 6 ? max_threads : 6) : 1));
        STEP(fx_stream_collect_samples(w.ring, raw.data(), sm.data(), &frames));
        total += frames;
    }
    if (total != (480 + 8000 + 8320 + 1) / 128) problem("frames out of the ring's sample blocks");
    void* slot = nullptr;
    STEP(fx_stream_acquire(w.ring, &slot));
    if (fx_stream_submit(w.ring) == FX_OK) problem("whole hops accepted by the ring while samples are pending");
    if (fx_stream_acquire(w.ring, &slot) != FX_OK) {
        if (!fake_hip_failed()) problem("the refused submit did not hand its slot back");
        return false;
    }
    STEP(fx_stream_submit_samples(w.ring, 128 - fx_pending_samples(w.ctx)));
    STEP(fx_stream_collect_samples(w.ring, raw.data(), sm.data(), &frames));
    if (frames != 1 || fx_pending_samples(w.ctx) != 0) problem("the block that completes a hop");
    return true;
}

// E: the RCCL gather (a communicator of one rank, librccl.so.1 = tests/cpp/fake_hip/fake_rccl.cpp)
bool scenario_comm(World& w)
{
    if (!make_context(w, 8, 1024)) return false;
    unsigned char id[FX_COMM_ID_BYTES];
    STEP(fx_comm_unique_id(id, FX_COMM_ID_BYTES));
    STEP(fx_comm_create(w.ctx, 0, 1, id, FX_COMM_ID_BYTES));
    STEP(fx_push_hops(w.ctx, w.hops.data(), 4, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, nullptr));
    std::vector<float> out((size_t) 8 * 12);
    for (int i = 0; i < 3; i++) STEP(fx_gather_smoothed(w.ctx, 0, out.data(), FX_MEM_HOST));
    STEP(fx_comm_sync(w.ctx));
    void* d_out = nullptr;
    if (hipMalloc(&d_out, out.size() * 4) != hipSuccess) return false;
    w.device.push_back(d_out);
    STEP(fx_gather_smoothed(w.ctx, 0, (float*) d_out, FX_MEM_DEVICE));
    STEP(fx_comm_sync(w.ctx));
    int ranks = 0, gathers = 0;
    STEP(fx_comm_stats(w.ctx, &ranks, &gathers, nullptr, nullptr, nullptr));
    if (ranks != 1 || gathers != 4) problem("communicator statistics");
    STEP(fx_comm_destroy(w.ctx));
    return true;
}

// F: the arithmetic of fx_push_samples / fx_stream_submit_samples, on the host: random block lengths, every sample format, host and "device"
// blocks, a long block, empty blocks -- the hops the shim hands to the analysis kernels (the fake launchers log them, the fake re-blocking
// launcher really moves the bytes), put end to end per channel, must be the stream that went in, cut at whole hops.  ASan checks every bound
// the shim computed on the way.
bool scenario_block_arithmetic(World& w)
{
    unsigned r = 12345;
    auto rnd = [&] (unsigned m) { r = r * 1664525u + 1013904223u; return (r >> 8) % m; };
    const int formats[4] = { FX_SAMPLE_F32, FX_SAMPLE_S16, FX_SAMPLE_S24, FX_SAMPLE_F16 };
    const size_t widths[4] = { 4, 2, 3, 2 };
    for (int round = 0; round < 12; round++) {
        const int fi = round % 4, N = round % 3 == 0 ? 256 : (round % 3 == 1 ? 1024 : 4096), C = 1 + (int) rnd(5), H = N / 2;
        const size_t esz = widths[fi];
        const int total = 5 * H + (int) rnd((unsigned) (3 * H));
        World v;
        if (!make_context(v, C, N, round == 7 ? FX_LOW_LATENCY : 0u)) { recover_and_destroy(v, false); return false; }
        std::vector<unsigned char> stream ((size_t) C * (size_t) total * esz);
        for (size_t i = 0; i < stream.size(); i++) stream[i] = (unsigned char) (rnd(251) + 1);
        std::vector<float> raw ((size_t) C * (size_t) (total / H + 1) * 12), sm (raw.size());
        fake_hop_log_clear();
        fake_hop_log_enable(1);
        int at = 0, frames_total = 0;
        bool ok = true;
        while (at < total && ok) {
            static const int lengths[] = { 0, 1, 2, 3, 63, 441, 480, 512, 1000, 4097 };
            int n = lengths[rnd(10)];
            if (rnd(7) == 0) n = (int) rnd((unsigned) (2 * N));
            if (n > total - at) n = total - at;
            std::vector<unsigned char> block ((size_t) C * (size_t) n * esz + 16);
            for (int c = 0; c < C; c++) std::memcpy (&block[(size_t) c * (size_t) n * esz], &stream[((size_t) c * (size_t) total + (size_t) at) * esz], (size_t) n * esz);
            int frames = -1;
            const bool device = rnd(3) == 0;
            void* d_block = nullptr;
            const void* src = block.data();
            if (device) {
                if (hipMalloc (&d_block, block.size()) != hipSuccess) { ok = false; break; }
                std::memcpy (d_block, block.data(), block.size());
                src = d_block;
            }
            const fx_status st = fx_push_samples (v.ctx, src, n, formats[fi], device ? FX_MEM_DEVICE : FX_MEM_HOST, device ? nullptr : raw.data(), nullptr, &frames);
            if (d_block) (void) hipFree (d_block);
            if (st != FX_OK) { ok = false; break; }
            frames_total += frames;
            at += n;
            if (fx_pending_samples (v.ctx) != at % H) problem ("pending sample count after a block");
        }
        fake_hop_log_enable(0);
        if (ok) {
            if (frames_total != total / H) problem ("frames out of the blocks");
            for (int c = 0; c < C; c++) {
                const size_t want = (size_t) (total / H) * (size_t) H * esz;
                if (fake_hop_log_size (c) != want || std::memcmp (fake_hop_log_bytes (c), &stream[(size_t) c * (size_t) total * esz], want) != 0) {
                    char msg[128];
                    std::snprintf (msg, sizeof msg, "round %d: window %d, %d channels, format %d, channel %d", round, N, C, formats[fi], c);
                    problem ("the hops handed to the kernels are not the stream cut at whole hops:", msg);
                    break;
                }
            }
        }
        fake_hop_log_clear();
        const bool failed_here = !ok;
        recover_and_destroy (v, fake_hip_failed() != 0);
        if (failed_here) return false;
    }
    // the scenario's own world stays empty: every round built and destroyed its context
    (void) w;
    return true;
}

typedef bool (*Scenario)(World&);
bool ring_graph_zero(World& w) { return scenario_ring_graph(w, true); }
bool ring_graph_copy(World& w) { return scenario_ring_graph(w, false); }
bool ring_large(World& w) { return scenario_ring_large(w, 8); }
bool ring_large_many_threads(World& w) { return scenario_ring_large(w, 64); }

void walk(const char* name, Scenario run, bool inject)
{
    g_where = name;
    World w;
    fake_hip_reset();
    const bool ok = run(w);
    const long calls = fake_hip_calls();
    if (!ok) problem("the scenario fails without any injected failure");
    recover_and_destroy(w, false);
    std::printf("%-28s clean run: %ld HIP calls\n", name, calls);
    if (!inject) return;
    int reported = 0;
    for (long k = 1; k <= calls; k++) {
        char tag[128];
        World v;
        fake_hip_reset();
        fake_hip_fail_at(k);
        const bool fine = run(v);
        std::snprintf(tag, sizeof tag, "%s, HIP call %ld (%s) failing", name, k, fake_hip_failed() ? fake_hip_failed_name() : "not reached");
        g_where = tag;
        // a failure inside a destroy / free path, or one the shim may absorb (a timing event it can do without), is allowed to pass silently;
        // everything else must have been reported
        if (fake_hip_failed() && !fine) reported++;
        else if (getenv("HOST_SANITIZE_VERBOSE")) std::printf("    not reported: %s%s\n", tag, fake_hip_failed() ? "" : " (the scenario ended before that call)");
        recover_and_destroy(v, true);
    }
    std::printf("%-28s %ld injected failures walked, %d reported to the caller, problems so far: %d\n", name, calls, reported, g_problems);
}

// G: the batch sender and the counting receiver (csrc/fx_osc_sender.cpp: sender threads, the timer thread, publications swapped under
// them, the receiver's threads) -- no HIP in it, so nothing to inject: the sanitizers watch the threads.  Publications race the timer on
// purpose; sender and receiver are destroyed while the other is still running.
void osc_threads()
{
    g_where = "osc sender / receiver";
    const int C = 3000;
    std::vector<float> v((size_t) C * 12);
    for (size_t i = 0; i < v.size(); i++) v[i] = (float) i * 0.25f;
    const int stride = fx_osc_message_bytes("/Audio/A", C - 1);
    std::vector<unsigned char> d((size_t) C * (size_t) stride);
    std::vector<int> len((size_t) C);
    if (fx_osc_encode_batch("/Audio/A", 0, C, v.data(), d.data(), stride, len.data()) != C) { problem("fx_osc_encode_batch"); return; }
    for (unsigned flags : {0u, FX_OSC_SENDER_GSO}) {
        fx_osc_receiver* rx = nullptr;
        fx_osc_sender* tx = nullptr;
        if (fx_osc_receiver_create(&rx, "127.0.0.1:0", 3, "/Audio/A", C, flags ? 0u : FX_OSC_RECEIVER_NO_GRO) != FX_OK) { problem("fx_osc_receiver_create"); return; }
        char target[64];
        std::snprintf(target, sizeof target, "127.0.0.1:%d", fx_osc_receiver_port(rx));
        if (fx_osc_sender_create(&tx, target, target, 4, flags) != FX_OK) { problem("fx_osc_sender_create"); fx_osc_receiver_destroy(rx); return; }
        long long sent = -1;
        if (fx_osc_sender_send(tx, &sent) != FX_OK || sent != 0) problem("a tick before anything was published");
        if (fx_osc_sender_update(tx, d.data(), stride, len.data(), C) != FX_OK) problem("fx_osc_sender_update");
        if (fx_osc_sender_send(tx, &sent) != FX_OK || sent != 2ll * C) problem("one tick by hand");
        if (fx_osc_sender_start(tx, 500.0) != FX_OK) problem("fx_osc_sender_start");
        for (int k = 0; k < 60; k++) {                         // publications of changing size under a running timer, ticks by hand against it
            const int n = C - (k % 7) * 100;
            if (fx_osc_sender_update(tx, d.data(), stride, len.data(), n) != FX_OK) problem("fx_osc_sender_update under the timer");
            if (k % 5 == 0 && fx_osc_sender_send(tx, nullptr) != FX_OK) problem("fx_osc_sender_send under the timer");
            unsigned char got[160]; int n_got = 0;
            if (fx_osc_receiver_last(rx, k, got, (int) sizeof got, &n_got) != FX_OK) problem("fx_osc_receiver_last");
            if (n_got != 0 && (n_got != len[(size_t) k] || std::memcmp(got, d.data() + (size_t) k * (size_t) stride, (size_t) n_got) != 0)) problem("a received message differs from the one sent");
            usleep(2000);
        }
        fx_osc_sender_stats st;
        if (fx_osc_sender_get_stats(tx, &st) != FX_OK || st.ticks < 10 || st.datagrams < 10ll * (C - 600)) problem("sender statistics");
        long long n = 0, bad = 0;
        if (fx_osc_receiver_get_stats(rx, &n, nullptr, &bad) != FX_OK || n <= 0 || bad != 0) problem("receiver statistics");
        if (flags) { fx_osc_receiver_destroy(rx); usleep(20000); fx_osc_sender_destroy(tx); }      // the timer still running in both orders
        else { fx_osc_sender_destroy(tx); fx_osc_receiver_destroy(rx); }
    }
    std::printf("%-28s clean run (threads only)\n", "osc sender / receiver");
}

// H: the live engine (include/fx_realtime.hpp: fx::LiveAnalyser) over the fake runtime -- an "audio thread" pushing blocks of changing length
// into the FIFO as fast as it can (a full FIFO drops and counts), the worker feeding the ring and publishing to a callback and an
// OSCBatchSender with its timer running, getStats / latestSmoothed read from a third thread meanwhile, drain, stop, destruction in the
// documented order.  Every block is accounted for: analysed or counted as dropped; the frames published are those of the blocks analysed.
void live_engine()
{
    g_where = "live engine";
    fake_hip_reset();
    const int C = 6, N = 1024, H = N / 2, blocks = 400;
    try {
        fx_osc_receiver* rx = nullptr;
        if (fx_osc_receiver_create(&rx, "127.0.0.1:0", 1, "/Audio/A", C, 0u) != FX_OK) { problem("fx_osc_receiver_create"); return; }
        {
            fx::RealTimeBatchAnalyser analyser(C, N);
            fx::OSCBatchSender sender("127.0.0.1:" + std::to_string(fx_osc_receiver_port(rx)), "", 2, false);
            sender.startTimerHz(500);
            long long framesSeen = 0;
            {
                fx::LiveAnalyser live(analyser, 700, 4);
                live.attachOSCSender(&sender, "/Audio/A", 0);
                live.setFramesAnalysedCallback([&](int frames, const float*, const float*) { framesSeen += frames; });
                std::atomic<bool> watching{true};
                std::atomic<int> settersMade{0};
                std::thread watcher([&] {                        // a "message thread": reads statistics, and changes settings THROUGH the worker
                    int k = 0;
                    while (watching.load()) {
                        (void) live.getStats(); (void) live.latestSmoothed();
                        const float g = 0.5f + 0.01f * (float) (k++ % 50);
                        live.callOnWorker([g, &settersMade](fx::RealTimeBatchAnalyser& a) { a.setGain(g); a.setOnsetDetectionSensitivity(g); settersMade++; });
                        usleep(300);
                    }
                });
                std::vector<float> block((size_t) C * 700, 0.25f);
                long long samplesIn = 0, pushedOk = 0;
                std::thread audio([&] {
                    unsigned r = 99;
                    for (int b = 0; b < blocks; b++) {
                        r = r * 1664525u + 1013904223u;
                        const int n = 1 + (int) ((r >> 8) % 700u);
                        if (live.pushBlock(block.data(), n)) { samplesIn += n; pushedOk++; }
                        if (b % 7 == 0) usleep(200);
                    }
                });
                audio.join();
                live.drain();
                watching = false;
                watcher.join();
                const fx::LiveAnalyser::Stats st = live.getStats();
                if (st.errors != 0) problem("the live engine reported an analysis error", live.lastError().c_str());
                if (settersMade.load() < 1) problem("no queued setter call was made by the worker");
                if (st.blocksIn != pushedOk || st.blocksAnalysed != pushedOk || st.blocksIn + st.blocksDropped != blocks) problem("blocks not accounted for");
                if (st.framesPerChannel != samplesIn / H || framesSeen != st.framesPerChannel) problem("frames published differ from the frames the accepted blocks completed");
                live.stop();
            }
            sender.stopTimer();
        }
        fx_osc_receiver_destroy(rx);
    } catch (const std::exception& e) { problem("exception", e.what()); }
    if (fake_hip_live() != 0) problem("the live engine left device objects behind");
    std::printf("%-28s clean run (threads only)\n", "live engine");
}

} // namespace

int main(int argc, char** argv)
{
    const std::string mode = argc > 1 ? argv[1] : "asan";
    if (mode == "tsan") {
        // the fill pool: resize up and down between 1 and 64 threads, jobs back to back, destroyed while idle
        walk("ring, large batches x64", ring_large_many_threads, false);
        walk("ring, large batches", ring_large, false);
        osc_threads();
        live_engine();
    } else {
        walk("batch calls", scenario_batch, true);
        walk("ring, one-launch hop kernel", scenario_ring_hop_kernel, true);
        walk("ring, captured step (0-copy)", ring_graph_zero, true);
        walk("ring, captured step (copies)", ring_graph_copy, true);
        walk("ring, large batches", ring_large, true);
        walk("rccl gather", scenario_comm, true);
        walk("block arithmetic", scenario_block_arithmetic, false);
        osc_threads();
        live_engine();
        // failures of RCCL itself
        void (*reset)(void) = (void (*)(void)) dlsym(RTLD_DEFAULT, "fake_rccl_reset");
        if (!reset) problem("the fake librccl is not the one loaded");
        for (int k = 1; reset && k <= 14; k++) {
            char v[16], tag[64];
            std::snprintf(v, sizeof v, "%d", k);
            std::snprintf(tag, sizeof tag, "rccl gather, RCCL call %d failing", k);
            g_where = tag;
            setenv("FAKE_RCCL_FAIL_AT", v, 1);
            reset();
            World w;
            fake_hip_reset();
            (void) scenario_comm(w);
            unsetenv("FAKE_RCCL_FAIL_AT");
            recover_and_destroy(w, true);
        }
    }
    std::printf("host_sanitize: %d problem(s)\n", g_problems);
    return g_problems ? 1 : 0;
}
