// fake_hip.cpp -- the malloc-backed runtime behind tests/cpp/fake_hip/hip/hip_runtime.h, and stand-ins for the kernel launchers of
// csrc/fx_kernels.h (which only count as calls).  Everything is synchronous; every entry point is a countable, failable call.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "fx_kernels.h"

struct fake_stream { int capturing = 0; };
struct fake_event { int recorded = 0; };
struct fake_graph { int x = 0; };
struct fake_graph_exec { int x = 0; };

namespace {
std::mutex g_m;
std::map<void*, size_t> g_blocks;          // device and host allocations
long g_objects = 0;                        // streams, events, graphs
std::atomic<long> g_calls{0};
long g_fail_at = 0;
int g_failed = 0;
const char* g_failed_name = "";
hipError_t g_last = hipSuccess;

// one countable call; `oom`: what an allocation reports when it is the one to fail
hipError_t tick(const char* name, bool oom = false)
{
    const long k = ++g_calls;
    if (g_fail_at && k == g_fail_at) { g_failed = 1; g_failed_name = name; g_last = oom ? hipErrorOutOfMemory : hipErrorUnknown; return g_last; }
    return hipSuccess;
}
}

extern "C" {
void fake_hip_reset(void) { g_calls = 0; g_fail_at = 0; g_failed = 0; g_failed_name = ""; g_last = hipSuccess; }
void fake_hip_fail_at(long call) { g_fail_at = call; g_failed = 0; }
long fake_hip_calls(void) { return g_calls; }
long fake_hip_live(void) { std::lock_guard<std::mutex> g(g_m); return (long) g_blocks.size() + g_objects; }
long fake_hip_live_bytes(void) { std::lock_guard<std::mutex> g(g_m); long b = 0; for (auto& kv : g_blocks) b += (long) kv.second; return b; }
int fake_hip_failed(void) { return g_failed; }
const char* fake_hip_failed_name(void) { return g_failed_name; }
hipError_t fake_hip_count(const char* name) { return tick(name); }

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (fake)" : "injected failure (fake)"); }
hipError_t hipGetLastError(void) { const hipError_t e = g_last; g_last = hipSuccess; return e; }
// FAKE_HIP_DEVICES: how many "GPUs" there are (default 1; the multi-rank gather tests give every rank process its own)
hipError_t hipGetDeviceCount(int* n) { if (tick("hipGetDeviceCount") != hipSuccess) return g_last; const char* e = getenv("FAKE_HIP_DEVICES"); *n = e && atoi(e) > 0 ? atoi(e) : 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { if (tick("hipGetDeviceProperties") != hipSuccess) return g_last; memset(p, 0, sizeof *p); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipSetDevice(int) { return tick("hipSetDevice"); }

static hipError_t alloc(const char* name, void** out, size_t bytes)
{
    *out = nullptr;
    if (tick(name, true) != hipSuccess) return g_last;
    void* p = malloc(bytes ? bytes : 1);
    if (!p) return hipErrorOutOfMemory;
    memset(p, 0xA5, bytes);               // (uninitialised "device" memory is not zero)
    std::lock_guard<std::mutex> g(g_m);
    g_blocks[p] = bytes;
    *out = p;
    return hipSuccess;
}
static hipError_t release(const char* name, void* p)
{
    // a failing free still gives the memory back: what the leak check asks is whether the SHIM let go of everything
    const hipError_t e = tick(name);
    if (p) {
        std::lock_guard<std::mutex> g(g_m);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end()) { fprintf(stderr, "fake hip: %s of a pointer that is not allocated: %p (call %ld, armed failure fired in %s)\n", name, p, (long) g_calls, g_failed_name); int* boom = nullptr; *boom = 1; }
        g_blocks.erase(it);
        free(p);
    }
    return e;
}
hipError_t hipMalloc(void** p, size_t n) { return alloc("hipMalloc", p, n); }
hipError_t hipFree(void* p) { return release("hipFree", p); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return alloc("hipHostMalloc", p, n); }
hipError_t hipHostFree(void* p) { return release("hipHostFree", p); }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { if (tick("hipHostGetDevicePointer") != hipSuccess) return g_last; *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (tick("hipMemcpy") != hipSuccess) return g_last; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (tick("hipMemcpyAsync") != hipSuccess) return g_last; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (tick("hipMemsetAsync") != hipSuccess) return g_last; memset(d, v, n); return hipSuccess; }

} // extern "C"
template <typename T> static hipError_t make(const char* name, T** out)
{
    *out = nullptr;
    if (tick(name) != hipSuccess) return g_last;
    *out = new T();
    std::lock_guard<std::mutex> g(g_m);
    g_objects++;
    return hipSuccess;
}
template <typename T> static hipError_t unmake(const char* name, T* p)
{
    const hipError_t e = tick(name);
    if (p) { delete p; std::lock_guard<std::mutex> g(g_m); g_objects--; }
    return e;
}
extern "C" {
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return make("hipStreamCreateWithFlags", s); }
hipError_t hipStreamDestroy(hipStream_t s) { return unmake("hipStreamDestroy", s); }
hipError_t hipStreamSynchronize(hipStream_t) { return tick("hipStreamSynchronize"); }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return tick("hipStreamWaitEvent"); }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { if (tick("hipStreamBeginCapture") != hipSuccess) return g_last; if (s) s->capturing = 1; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g)
{
    // (ends the capture even when it is the call that fails, as the real one does)
    *g = nullptr;
    if (s) s->capturing = 0;
    return make("hipStreamEndCapture", g);
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { return make("hipGraphInstantiate", e); }
hipError_t hipGraphDestroy(hipGraph_t g) { return unmake("hipGraphDestroy", g); }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { return unmake("hipGraphExecDestroy", e); }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return tick("hipGraphLaunch"); }
hipError_t hipEventCreate(hipEvent_t* e) { return make("hipEventCreate", e); }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return make("hipEventCreateWithFlags", e); }
hipError_t hipEventDestroy(hipEvent_t e) { return unmake("hipEventDestroy", e); }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (tick("hipEventRecord") != hipSuccess) return g_last; if (e) e->recorded = 1; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return tick("hipEventSynchronize"); }
hipError_t hipEventQuery(hipEvent_t) { return tick("hipEventQuery"); }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { if (tick("hipEventElapsedTime") != hipSuccess) return g_last; *ms = 0.001f; return hipSuccess; }
} // extern "C"

// ---- the hops the shim hands to the analysis kernels (hop mode), per channel ----
#include <vector>
namespace {
std::vector<std::vector<unsigned char>> g_hops;
int g_hop_log = 0;
void log_hops(const fxk::FrameParams& p, int window)
{
    if (!g_hop_log || !p.hop_mode || !p.in) return;
    const size_t esz = p.sample_format == FX_SAMPLE_F32 ? 4 : (p.sample_format == FX_SAMPLE_S24 ? 3 : 2);
    const size_t row = (size_t) p.T * (size_t) (window / 2) * esz;
    if ((int) g_hops.size() < p.C) g_hops.resize((size_t) p.C);
    const unsigned char* in = static_cast<const unsigned char*>(p.in);
    for (int c = 0; c < p.C; c++) g_hops[(size_t) c].insert(g_hops[(size_t) c].end(), in + (size_t) c * row, in + (size_t) (c + 1) * row);
}
}
// What the fake "analysis" leaves in `latest` ([C][12]): a hash of each channel's input in the call and of the calls so far, as a small
// whole number -- so that every channel of every rank holds something of its own for the gather tests (tests/cpp/comm_ranks.cpp), and a
// block that lands at the wrong offset, or a stale snapshot, shows.
namespace {
std::vector<unsigned> g_input_hash;
unsigned g_latest_calls = 0;
void hash_input(const fxk::FrameParams& p, int window)
{
    if ((int) g_input_hash.size() < p.C) g_input_hash.resize((size_t) p.C, 0u);
    if (!p.in) return;
    const size_t esz = p.sample_format == FX_SAMPLE_F32 ? 4 : (p.sample_format == FX_SAMPLE_S24 ? 3 : 2);
    const size_t row = (size_t) p.T * (size_t) (p.hop_mode ? window / 2 : window) * esz;
    const unsigned char* in = static_cast<const unsigned char*>(p.in);
    for (int c = 0; c < p.C; c++) {
        unsigned h = 2166136261u;
        for (size_t i = 0; i < row; i++) h = (h ^ in[(size_t) c * row + i]) * 16777619u;
        g_input_hash[(size_t) c] = h;
    }
}
void fill_latest(const fxk::EpilogueParams& ep)
{
    if (!ep.latest) return;
    g_latest_calls++;
    for (int c = 0; c < ep.C; c++)
        for (int k = 0; k < FX_NUM_FEATURES; k++)
            ep.latest[(size_t) c * FX_NUM_FEATURES + k] = (float) (((c < (int) g_input_hash.size() ? g_input_hash[(size_t) c] : 0u) + 7919u * g_latest_calls + 31u * (unsigned) k) % 1000003u);
}
}
extern "C" {
void fake_hop_log_clear(void) { g_hops.clear(); }
void fake_hop_log_enable(int on) { g_hop_log = on; }
const unsigned char* fake_hop_log_bytes(int c) { return c < (int) g_hops.size() ? g_hops[(size_t) c].data() : nullptr; }
size_t fake_hop_log_size(int c) { return c < (int) g_hops.size() ? g_hops[(size_t) c].size() : 0; }
}

// A block-fed one-frame launch (FrameParams::block_mode), on the host: what the BLOCKS kernels do with the bytes -- the hop is the head of
// [pending | block], the rest goes to the other carry buffer -- so that the hop log and the next call's pending samples are what the real
// kernels would leave, and ASan checks every bound the shim handed over.  Returns the parameters of the equivalent hop-fed launch.
namespace {
std::vector<unsigned char> g_block_hops;
fxk::FrameParams from_blocks(const fxk::FrameParams& p, int window)
{
    if (!p.block_mode) {
        if (!p.in_hop_stride || !p.hop_mode || !p.in) return p;
        // a one-frame launch over a buffer of several hops per channel: hand the log the hop it reads
        const size_t esz = p.sample_format == FX_SAMPLE_F32 ? 4 : (p.sample_format == FX_SAMPLE_S24 ? 3 : 2);
        const size_t hop = (size_t) (window / 2) * esz;
        g_block_hops.assign((size_t) p.C * hop, 0);
        const unsigned char* in = static_cast<const unsigned char*>(p.in);
        for (int c = 0; c < p.C; c++) memcpy(&g_block_hops[(size_t) c * hop], in + ((size_t) c * (size_t) p.in_hop_stride + (size_t) p.in_hop0) * hop, hop);
        fxk::FrameParams q = p;
        q.in = g_block_hops.data(); q.in_hop_stride = 0; q.in_hop0 = 0;
        return q;
    }
    const size_t esz = p.sample_format == FX_SAMPLE_F32 ? 4 : (p.sample_format == FX_SAMPLE_S24 ? 3 : 2);
    const size_t hop = (size_t) (window / 2) * esz, hops = (size_t) p.T * hop;
    const size_t total = (size_t) p.blk_carry_bytes + (size_t) p.blk_in_row_bytes;
    const size_t from = (size_t) p.blk_hop0 * hop;
    if (p.T < 1 || !p.hop_mode || total < from + hops || (p.blk_keep_rest && (total >= from + hops + hop || total - from - hops > (size_t) p.blk_carry_row_bytes))) { fprintf(stderr, "fake hip: a block feed that does not hold its hops\n"); int* boom = nullptr; *boom = 1; }
    g_block_hops.assign((size_t) p.C * hops, 0);
    const unsigned char* in = static_cast<const unsigned char*>(p.in);
    for (int c = 0; c < p.C; c++)
        for (size_t d = from; d < (p.blk_keep_rest ? total : from + hops); d++) {
            const unsigned char v = d < (size_t) p.blk_carry_bytes ? p.blk_carry_in[(size_t) c * p.blk_carry_row_bytes + d] : in[(size_t) c * (size_t) p.blk_in_row_bytes + (d - p.blk_carry_bytes)];
            if (d < from + hops) g_block_hops[(size_t) c * hops + (d - from)] = v;
            else p.blk_carry_out[(size_t) c * p.blk_carry_row_bytes + (d - from - hops)] = v;
        }
    fxk::FrameParams q = p;
    q.in = g_block_hops.data();
    q.block_mode = 0;
    return q;
}
}

// ---- csrc/fx_kernels.h: launchers that launch nothing (each one a countable call), host helpers with plausible answers ----
namespace fxk {
void build_pass_twiddles(int n, const float* canonical, float* out) { memcpy(out, canonical, sizeof(float) * 2 * (size_t) n); }
void fill_first_pass_twiddles(int, const float*, float* out18) { for (int i = 0; i < 18; i++) out18[i] = 0.0f; }
bool first_pass_twiddles_hermitian(int, const float*) { return true; }
int build_twiddle_image(int, const float*, float*) { return 0; }
bool twiddles_have_quarter_turn(int, const float*) { return true; }
size_t frame_kernel_lds_bytes(int n, int ch, int k, bool direct) { return (size_t) 8 * n + (direct ? 0 : (size_t) ch * 9 * n / 4) + (size_t) ch * k * (n <= 1024 ? 17 * n / 2 : 9 * n / 2); }
int frame_kernel_max_waves(int n) { return n <= 512 ? 16 : 8; }
void frame_kernel_preferred_shape(int n, int* ch, int* k) { *ch = 1; *k = n == 2048 ? 4 : 8; }
hipError_t launch_frame_kernel(int n, const FrameParams& p0, int, hipStream_t) { if (p0.C <= 0 || p0.T <= 0) return hipSuccess; const hipError_t e = fake_hip_count("launch_frame_kernel"); if (e == hipSuccess) { const FrameParams p = from_blocks(p0, n); log_hops(p, n); hash_input(p, n); } return e; }
hipError_t launch_epilogue_kernels(const EpilogueParams& p, hipStream_t) { if (p.C <= 0 || p.T <= 0) return hipSuccess; const hipError_t e = fake_hip_count("launch_epilogue_kernels"); if (e == hipSuccess) fill_latest(p); return e; }
bool frame_tail_kernel_available(int n) { return n >= 1024; }
hipError_t launch_frame_tail_kernel(int n, const FrameParams& p0, const EpilogueParams& ep, hipStream_t) { const hipError_t e = fake_hip_count("launch_frame_tail_kernel"); if (e == hipSuccess) { const FrameParams p = from_blocks(p0, n); log_hops(p, n); hash_input(p, n); fill_latest(ep); } return e; }
hipError_t prepare_kernels(int) { return fake_hip_count("prepare_kernels"); }
hipError_t prepare_hop_kernel(int) { return fake_hip_count("prepare_hop_kernel"); }
bool pair_kernel_available(int n) { return n == 2048 || n == 4096; }
int pair_kernel_max_pairs(int n) { return n == 2048 ? 8 : 6; }
size_t pair_kernel_lds_bytes(int n, int ch, int k) { return (size_t) 8 * n + (size_t) ch * k * 17408; }
hipError_t prepare_pair_kernel(int) { return fake_hip_count("prepare_pair_kernel"); }
hipError_t launch_pair_kernel(int n, const FrameParams& p, hipStream_t) { const hipError_t e = fake_hip_count("launch_pair_kernel"); if (e == hipSuccess) { log_hops(p, n); hash_input(p, n); } return e; }
bool hop_kernel_available(int n) { return n == 1024 || n == 2048 || n == 4096; }
hipError_t launch_hop_kernel(int n, const FrameParams& p0, const EpilogueParams& ep, const HopSignal& sig, hipStream_t, bool)
{
    const hipError_t e = fake_hip_count("launch_hop_kernel");
    if (e == hipSuccess) { const FrameParams p = from_blocks(p0, n); log_hops(p, n); hash_input(p, n); fill_latest(ep); }
    // the real kernel raises the slot's flag when the hop is done; fx_stream_collect polls it
    if (e == hipSuccess && sig.host_flag) *sig.host_flag = sig.seq;
    return e;
}
hipError_t launch_reblock_kernel(const ReblockParams& p, hipStream_t)
{
    if (p.C <= 0 || (long long) p.carry_bytes + p.in_row_bytes <= 0) return hipSuccess;
    const hipError_t e = fake_hip_count("launch_reblock_kernel");
    if (e != hipSuccess) return e;
    // the byte movement of fx_reblock.hip, on the host: ASan then checks every bound the shim handed over
    for (int c = 0; c < p.C; c++) {
        const long long total = (long long) p.carry_bytes + p.in_row_bytes;
        for (long long d = 0; d < total; d++) {
            const unsigned char v = d < p.carry_bytes ? p.carry_in[(size_t) c * p.carry_row_bytes + d] : p.in[(size_t) c * p.in_row_bytes + (d - p.carry_bytes)];
            if (d < p.out_row_bytes) p.hops_out[(size_t) c * p.out_row_bytes + d] = v;
            else p.carry_out[(size_t) c * p.carry_row_bytes + (d - p.out_row_bytes)] = v;
        }
    }
    return hipSuccess;
}
hipError_t launch_osc_kernel(const OscParams& p, hipStream_t)
{
    if (p.C <= 0) return hipSuccess;
    const hipError_t e = fake_hip_count("launch_osc_kernel");
    if (e != hipSuccess) return e;
    // the bytes of fx_osc.hip through the host encoder: ASan checks the bounds the shim handed over
    char prefix[FX_OSC_PREFIX_MAX + 1] = {0};
    memcpy(prefix, p.prefix, (size_t) p.prefix_len);
    return fx_osc_encode_batch(prefix, p.first_channel, p.C, p.latest, p.out, p.stride, nullptr) == p.C ? hipSuccess : hipErrorInvalidValue;
}
hipError_t clear_carry(unsigned char* carry, size_t bytes, hipStream_t s) { return bytes ? hipMemsetAsync(carry, 0, bytes, s) : hipSuccess; }
} // namespace fxk
