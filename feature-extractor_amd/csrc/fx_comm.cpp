// fx_comm.cpp -- the one collective on the path: the gather of every rank's latest smoothed feature
// vectors to the rank that owns the OSC sink (ref AnalyserTrackController.h:22-23: two
// OSCFeatureAnalysisOutput senders per track; OSCFeatureAnalysisOutput.h:89-136: they sample
// AudioFeatures::getValue at 60 Hz).  RCCL over xGMI, called through rccl.h directly; one process per
// GPU.  xGMI is point to point, and the payload is tiny (48 B per channel: 384 KiB per 8192-channel
// shard), so the gather is a group of ncclSend / ncclRecv pairs into the sink's buffer -- no ring, no
// all-gather traffic to ranks that do not need the data.
//
// Ordering is entirely on the device: the snapshot of `latest` is taken on the context's stream, an
// event hands it to the communicator's side stream, and the next snapshot into the same staging slot
// waits for the gather that read it -- the host never blocks and the exchange overlaps the next
// analysis call's kernels.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library itself is loaded on first use (below)

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fx_context.h"

// librccl is half a gigabyte and only the multi-GPU gather needs it: libfx_hip.so does not link it.  The first
// fx_comm_* call loads it (the copy already in the process if the host -- PyTorch, say -- has mapped one: same SONAME,
// one RCCL per process), and a host without it gets FX_ERR_UNSUPPORTED from these entries while every single-GPU entry
// works.
namespace {
struct Rccl {
    decltype(&::ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&::ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&::ncclCommCount)      CommCount = nullptr;
    decltype(&::ncclAllGather)      AllGather = nullptr;
    decltype(&::ncclGroupStart)     GroupStart = nullptr;
    decltype(&::ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&::ncclSend)           Send = nullptr;
    decltype(&::ncclRecv)           Recv = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    std::string why;                 // empty = loaded
};

const Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> names = {"librccl.so.1", "librccl.so"};
        for (const char* var : {"ROCM_PATH", "ROCM_HOME"})
            if (const char* root = getenv(var)) names.push_back(std::string(root) + "/lib/librccl.so.1");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        void* h = nullptr;
        for (const std::string& n : names) if ((h = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) { const char* e = dlerror(); r.why = std::string("librccl could not be loaded: ") + (e ? e : "not found"); return; }
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) { ok = false; r.why = std::string("librccl lacks ") + name; } return p; };
#define FX_RCCL_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(sym(name))
        FX_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");   FX_RCCL_SYM(CommInitRank, "ncclCommInitRank"); FX_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        FX_RCCL_SYM(CommCount, "ncclCommCount");       FX_RCCL_SYM(AllGather, "ncclAllGather");       FX_RCCL_SYM(GroupStart, "ncclGroupStart");
        FX_RCCL_SYM(GroupEnd, "ncclGroupEnd");         FX_RCCL_SYM(Send, "ncclSend");                 FX_RCCL_SYM(Recv, "ncclRecv");
        FX_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef FX_RCCL_SYM
        if (ok) r.why.clear();
    });
    return &r;
}
} // namespace
#define RCCL_OR_UNSUPPORTED() do { if (!rccl()->why.empty()) return fx_fail(FX_ERR_UNSUPPORTED, "%s", rccl()->why.c_str()); } while (0)

struct fx_comm {
    ncclComm_t  comm = nullptr;
    int         rank = 0, world = 1;
    std::vector<int> channels;          // per rank
    std::vector<int> first;             // channel offset of each rank's block
    int         total = 0;
    hipStream_t side = nullptr;
    float*      stage[2] = {nullptr, nullptr};   // [C][12] snapshots of `latest`
    hipEvent_t  snap[2] = {nullptr, nullptr};    // snapshot taken (context stream)
    hipEvent_t  sent[2] = {nullptr, nullptr};    // gather that read stage[i] finished (side stream)
    hipEvent_t  began[2] = {nullptr, nullptr};   // ... and started (side stream, behind the snapshot): began -> sent is the gather's device time
    bool        sent_valid[2] = {false, false};
    bool        timed_pending[2] = {false, false};   // a (began, sent) pair not yet read
    int         rccl_ranks = 0;                  // ncclCommCount at creation
    int         gathers = 0, gathers_timed = 0;
    double      gather_ms_total = 0.0, gather_ms_max = 0.0;
    float*      d_out = nullptr;                 // sink-side device buffer for FX_MEM_HOST destinations
    float*      h_out = nullptr;                 // pinned bounce buffer for FX_MEM_HOST destinations
    float*      h_dst = nullptr;                 // caller's host buffer of the gather in flight, copied at fx_comm_sync
    unsigned    slot = 0;
};

// Device time of the gather that used staging slot s (events on the side stream), if it has finished (`finished`: the caller has
// synchronised the stream; otherwise the event is queried and an unfinished gather is simply not counted).
static void harvest(fx_comm* m, int s, bool finished)
{
    if (!m->timed_pending[s]) return;
    if (!finished && hipEventQuery(m->sent[s]) != hipSuccess) { (void) hipGetLastError(); return; }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, m->began[s], m->sent[s]) == hipSuccess) {
        m->gathers_timed++;
        m->gather_ms_total += ms;
        if (ms > m->gather_ms_max) m->gather_ms_max = ms;
    } else {
        (void) hipGetLastError();
    }
    m->timed_pending[s] = false;
}

#define NCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fx_fail(FX_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_));  \
    } while (0)

void fx_comm_release(fx_context* c)
{
    if (!c || !c->comm) return;
    fx_comm* m = c->comm;
    (void) hipSetDevice(c->device);
    if (m->side) (void) hipStreamSynchronize(m->side);
    if (m->comm) (void) rccl()->CommDestroy(m->comm);
    for (int i = 0; i < 2; i++) {
        if (m->stage[i]) (void) hipFree(m->stage[i]);
        if (m->snap[i]) (void) hipEventDestroy(m->snap[i]);
        if (m->sent[i]) (void) hipEventDestroy(m->sent[i]);
        if (m->began[i]) (void) hipEventDestroy(m->began[i]);
    }
    if (m->d_out) (void) hipFree(m->d_out);
    if (m->h_out) (void) hipHostFree(m->h_out);
    if (m->side) (void) hipStreamDestroy(m->side);
    delete m;
    c->comm = nullptr;
}

extern "C" {

fx_status fx_comm_unique_id(void* id_out, int id_bytes)
{
    if (!id_out || id_bytes < (int) sizeof(ncclUniqueId))
        return fx_fail(FX_ERR_INVALID_ARGUMENT, "id buffer must hold FX_COMM_ID_BYTES (%d) bytes", (int) sizeof(ncclUniqueId));
    static_assert(sizeof(ncclUniqueId) == FX_COMM_ID_BYTES, "FX_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    RCCL_OR_UNSUPPORTED();
    ncclUniqueId id;
    NCCL_TRY(rccl()->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return FX_OK;
}

fx_status fx_comm_create(fx_context* c, int rank, int world, const void* unique_id, int id_bytes)
{
    if (!c || !unique_id) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (c->comm) return fx_fail(FX_ERR_INVALID_ARGUMENT, "this context already belongs to a communicator");
    if (world < 1 || rank < 0 || rank >= world) return fx_fail(FX_ERR_INVALID_ARGUMENT, "rank %d outside world of %d", rank, world);
    if (id_bytes != (int) sizeof(ncclUniqueId)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unique id must be %d bytes", (int) sizeof(ncclUniqueId));
    RCCL_OR_UNSUPPORTED();
    HIP_TRY(hipSetDevice(c->device));
    fx_comm* m = new (std::nothrow) fx_comm();
    if (!m) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->comm = m;
    m->rank = rank;
    m->world = world;
    auto bail = [&](fx_status s) { fx_comm_release(c); return s; };
#define M_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fx_fail(e_ == hipErrorOutOfMemory ? FX_ERR_OUT_OF_MEMORY : FX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
#define M_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return bail(fx_fail(FX_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_))); } while (0)
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    M_NCCL(rccl()->CommInitRank(&m->comm, world, id, rank));
    {
        // what RCCL itself thinks the communicator is (a launch that started fewer ranks than it claims shows up here)
        int count = -1;
        M_NCCL(rccl()->CommCount(m->comm, &count));
        fprintf(stderr, "[fx_comm] rank %d of %d: RCCL communicator of %d rank(s) on device %d\n", rank, world, count, c->device);
        m->rccl_ranks = count;
        if (count != world) return bail(fx_fail(FX_ERR_HIP, "RCCL reports %d ranks in the communicator, expected %d", count, world));
    }
    M_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
    const size_t bytes = (size_t) c->C * FX_NUM_FEATURES * sizeof(float);
    for (int i = 0; i < 2; i++) {
        M_HIP(hipMalloc((void**) &m->stage[i], bytes));
        M_HIP(hipEventCreateWithFlags(&m->snap[i], hipEventDisableTiming));
        M_HIP(hipEventCreate(&m->sent[i]));
        M_HIP(hipEventCreate(&m->began[i]));
    }
    // every rank's channel count (shards may differ by a remainder block)
    {
        int* d_counts = nullptr;
        M_HIP(hipMalloc((void**) &d_counts, sizeof(int) * (size_t) world));
        hipError_t e = hipMemcpyAsync(d_counts + rank, &c->C, sizeof(int), hipMemcpyHostToDevice, m->side);
        ncclResult_t r = ncclSuccess;
        if (e == hipSuccess) r = rccl()->AllGather(d_counts + rank, d_counts, 1, ncclInt32, m->comm, m->side);
        m->channels.assign((size_t) world, 0);
        if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(m->channels.data(), d_counts, sizeof(int) * (size_t) world, hipMemcpyDeviceToHost, m->side);
        if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(m->side);
        (void) hipFree(d_counts);
        if (r != ncclSuccess) return bail(fx_fail(FX_ERR_HIP, "ncclAllGather of the channel counts failed: %s", rccl()->GetErrorString(r)));
        if (e != hipSuccess) return bail(fx_fail(FX_ERR_HIP, "exchange of the channel counts failed: %s", hipGetErrorString(e)));
    }
    m->first.assign((size_t) world, 0);
    m->total = 0;
    for (int r = 0; r < world; r++) { m->first[(size_t) r] = m->total; m->total += m->channels[(size_t) r]; }
    if (m->channels[(size_t) rank] != c->C) return bail(fx_fail(FX_ERR_HIP, "channel-count exchange returned %d for this rank, expected %d", m->channels[(size_t) rank], c->C));
#undef M_HIP
#undef M_NCCL
    return FX_OK;
}

fx_status fx_comm_destroy(fx_context* c)
{
    if (!c) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null context");
    fx_comm_release(c);
    return FX_OK;
}

fx_status fx_comm_layout(fx_context* c, int* total_channels, int* first_of_rank)
{
    if (!c || !c->comm) return fx_fail(FX_ERR_INVALID_ARGUMENT, "context has no communicator");
    if (total_channels) *total_channels = c->comm->total;
    if (first_of_rank) for (int r = 0; r < c->comm->world; r++) first_of_rank[r] = c->comm->first[(size_t) r];
    return FX_OK;
}

fx_status fx_gather_smoothed(fx_context* c, int dst, float* out, int mem_kind)
{
    if (!c || !c->comm) return fx_fail(FX_ERR_INVALID_ARGUMENT, "context has no communicator (fx_comm_create)");
    fx_comm* m = c->comm;
    if (dst < 0 || dst >= m->world) return fx_fail(FX_ERR_INVALID_ARGUMENT, "destination rank %d outside world of %d", dst, m->world);
    if (mem_kind != FX_MEM_HOST && mem_kind != FX_MEM_DEVICE) return fx_fail(FX_ERR_INVALID_ARGUMENT, "unknown memory kind %d", mem_kind);
    const bool sink = m->rank == dst;
    if (sink && !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "the destination rank needs an output buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t total_bytes = (size_t) m->total * FX_NUM_FEATURES * sizeof(float);
    float* d_dst = out;
    if (sink && mem_kind == FX_MEM_HOST) {
        if (m->h_dst) {                       // one host-destination gather in flight at a time
            HIP_TRY(hipStreamSynchronize(m->side));
            memcpy(m->h_dst, m->h_out, total_bytes);
            m->h_dst = nullptr;
        }
        if (!m->d_out) HIP_TRY(hipMalloc((void**) &m->d_out, total_bytes));
        if (!m->h_out) HIP_TRY(hipHostMalloc((void**) &m->h_out, total_bytes, hipHostMallocDefault));
        d_dst = m->d_out;
    }
    const int s = (int) (m->slot & 1u);
    m->slot++;
    // the gather that last read this staging slot must be over before the snapshot overwrites it
    if (m->sent_valid[s]) HIP_TRY(hipStreamWaitEvent(c->stream, m->sent[s], 0));
    harvest(m, s, false);                        // (its timing, if it has finished: the events are about to be recorded again)
    const size_t bytes = (size_t) c->C * FX_NUM_FEATURES * sizeof(float);
    HIP_TRY(hipMemcpyAsync(m->stage[s], c->d_latest, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipEventRecord(m->snap[s], c->stream));
    HIP_TRY(hipStreamWaitEvent(m->side, m->snap[s], 0));
    HIP_TRY(hipEventRecord(m->began[s], m->side));
    NCCL_TRY(rccl()->GroupStart());
    ncclResult_t r = rccl()->Send(m->stage[s], (size_t) c->C * FX_NUM_FEATURES, ncclFloat32, dst, m->comm, m->side);
    if (r == ncclSuccess && sink) {
        for (int src = 0; src < m->world && r == ncclSuccess; src++)
            r = rccl()->Recv(d_dst + (size_t) m->first[(size_t) src] * FX_NUM_FEATURES, (size_t) m->channels[(size_t) src] * FX_NUM_FEATURES,
                         ncclFloat32, src, m->comm, m->side);
    }
    const ncclResult_t re = rccl()->GroupEnd();
    if (r != ncclSuccess) return fx_fail(FX_ERR_HIP, "ncclSend / ncclRecv failed: %s", rccl()->GetErrorString(r));
    if (re != ncclSuccess) return fx_fail(FX_ERR_HIP, "ncclGroupEnd failed: %s", rccl()->GetErrorString(re));
    HIP_TRY(hipEventRecord(m->sent[s], m->side));
    m->sent_valid[s] = true;
    m->timed_pending[s] = true;
    m->gathers++;
    if (sink && mem_kind == FX_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(m->h_out, m->d_out, total_bytes, hipMemcpyDeviceToHost, m->side));
        m->h_dst = out;
    }
    return FX_OK;
}

fx_status fx_comm_sync(fx_context* c)
{
    if (!c || !c->comm) return fx_fail(FX_ERR_INVALID_ARGUMENT, "context has no communicator");
    fx_comm* m = c->comm;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(m->side));
    harvest(m, 0, true);
    harvest(m, 1, true);
    if (m->h_dst) {
        memcpy(m->h_dst, m->h_out, (size_t) m->total * FX_NUM_FEATURES * sizeof(float));
        m->h_dst = nullptr;
    }
    return FX_OK;
}

fx_status fx_comm_stats(fx_context* c, int* rccl_ranks, int* gathers, int* gathers_timed, double* total_ms, double* max_ms)
{
    if (!c || !c->comm) return fx_fail(FX_ERR_INVALID_ARGUMENT, "context has no communicator");
    const fx_comm* m = c->comm;
    if (rccl_ranks) *rccl_ranks = m->rccl_ranks;
    if (gathers) *gathers = m->gathers;
    if (gathers_timed) *gathers_timed = m->gathers_timed;
    if (total_ms) *total_ms = m->gather_ms_total;
    if (max_ms) *max_ms = m->gather_ms_max;
    return FX_OK;
}

} // extern "C"
