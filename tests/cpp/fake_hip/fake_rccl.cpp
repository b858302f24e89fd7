// fake_rccl.cpp -> librccl.so.1 for the sanitizer builds of the host shim: the ten RCCL entry points csrc/fx_comm.cpp names, for
// 1 .. 8 ranks that are SEPARATE PROCESSES on one machine (one process per GPU, as the product runs), with no GPU and no RCCL.
//
// A communicator is a POSIX shared-memory segment named by the unique id.  Every ordered pair of ranks (src, dst) owns a one-message
// mailbox in it; ncclSend / ncclRecv queue operations and ncclGroupEnd drives all of them to completion (chunk by chunk, whichever can
// move), so a group of one send and `world` receives on the sink and one send on every other rank -- fx_gather_smoothed's shape --
// matches across processes exactly as RCCL matches it: by (peer, order of issue).  ncclAllGather is a pair of barriers around a
// per-rank scratch row.  The fake HIP runtime is synchronous, so "stream order" is call order.
//
// What it checks that the real library would not tell you: a receive whose length differs from the matching send is an ERROR here
// (ncclInvalidArgument: with real RCCL it is silent corruption or a hang); ranks that never arrive are a TIMEOUT (ncclSystemError
// after FAKE_RCCL_TIMEOUT_MS, default 20 s) instead of a hang.
//
// Failure injection: FAKE_RCCL_FAIL_AT=<k> makes the k-th RCCL call OF THIS PROCESS fail (ncclInternalError).  A rank that fails --
// injected or timed out -- marks the communicator aborted in shared memory, so its peers' waits end with ncclSystemError rather than
// running into the timeout (what ncclCommAbort / the async-error watchdog do for a real job).  An aborted communicator stays dead.
//
// The segment is unlinked by the last rank that leaves (ncclCommDestroy, or a failed ncclCommInitRank); tests/test_host_sanitized_cpu.py
// checks /dev/shm afterwards.
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
constexpr int      kMaxRanks = 8;
constexpr size_t   kChunk = 256 * 1024;          // mailbox payload; longer messages go through in pieces
constexpr size_t   kGatherRow = 4096;            // ncclAllGather: bytes per rank

struct Mailbox {
    std::atomic<uint64_t> posted;                // pieces written by the sender
    std::atomic<uint64_t> taken;                 // pieces read by the receiver; full while posted != taken
    uint64_t total;                              // length of the whole message this piece belongs to
    uint64_t offset;                             // where in the message the piece starts
    uint64_t bytes;                              // length of the piece
    unsigned char data[kChunk];
};

// All-zero is the valid initial state (a fresh segment is zero-filled by ftruncate).
struct Shared {
    std::atomic<int> world;                      // 0 until the first rank arrives
    std::atomic<int> joined, left, aborted;
    std::atomic<int> claimed[kMaxRanks];
    std::atomic<int> gone[kMaxRanks];            // the rank has destroyed its communicator
    std::atomic<int> bar_count, bar_generation;
    unsigned char gather[kMaxRanks][kGatherRow];
    Mailbox box[kMaxRanks][kMaxRanks];           // [src][dst]
};

long g_calls = 0;
std::mutex g_m;
std::vector<struct fake_nccl_comm*> g_comms;    // of this process, for the abort-on-failure rule
std::atomic<unsigned> g_id_counter{0};

long timeout_ms()
{
    const char* e = getenv("FAKE_RCCL_TIMEOUT_MS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? v : 20000;
}
double now_ms() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
size_t width(ncclDataType_t t) { return t == ncclInt8 ? 1 : 4; }
} // namespace

struct fake_nccl_comm {
    int world = 0, rank = 0;
    Shared* sh = nullptr;
    char name[64] = {0};
};

namespace {
void abort_all()
{
    std::lock_guard<std::mutex> g(g_m);
    for (fake_nccl_comm* c : g_comms) if (c->sh) c->sh->aborted.store(1, std::memory_order_release);
}
// one countable call
ncclResult_t tick()
{
    const char* e = getenv("FAKE_RCCL_FAIL_AT");
    ++g_calls;
    if (e && atol(e) == g_calls) { abort_all(); return ncclInternalError; }
    return ncclSuccess;
}

struct Op { bool send; int peer; unsigned char* ptr; size_t bytes, done; bool started; };
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
thread_local fake_nccl_comm* t_group_comm = nullptr;

// Waits for `ready()`; false on abort, on timeout, or when `hopeless()` says that what is awaited can no longer happen (the rank it
// must come from has destroyed its communicator: a real job would hang there until its watchdog ends it).  hopeless() is asked right
// after a ready() that returned false and may use what that call saw.  The communicator is marked aborted on every false return.
template <typename F, typename H> bool wait_for(Shared* sh, F ready, H hopeless)
{
    const double t0 = now_ms();
    const long limit = timeout_ms();
    for (unsigned spins = 0; !ready(); spins++) {
        if (sh->aborted.load(std::memory_order_acquire)) return false;
        if (hopeless() || ((spins & 255u) == 255u && now_ms() - t0 > limit)) { sh->aborted.store(1, std::memory_order_release); return false; }
        if (spins < 64) sched_yield(); else usleep(50);
    }
    return true;
}

bool barrier(fake_nccl_comm* c)
{
    Shared* sh = c->sh;
    const int gen = sh->bar_generation.load(std::memory_order_acquire);
    if (sh->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
        sh->bar_count.store(0, std::memory_order_relaxed);
        sh->bar_generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    bool somebody_left = false;      // (read BEFORE the generation: a rank that arrived and then left has already moved it on)
    return wait_for(sh, [&] { somebody_left = sh->left.load(std::memory_order_acquire) > 0; return sh->bar_generation.load(std::memory_order_acquire) != gen; },
                    [&] { return somebody_left; });
}

// One step of one operation; returns whether anything moved.  *bad: a length mismatch between a send and its receive.
bool progress(fake_nccl_comm* c, Op& op, bool* bad)
{
    Shared* sh = c->sh;
    if (op.send) {
        Mailbox& b = sh->box[c->rank][op.peer];
        const uint64_t p = b.posted.load(std::memory_order_relaxed);
        if (b.taken.load(std::memory_order_acquire) != p) return false;           // the last piece is still there
        const size_t n = op.bytes - op.done < kChunk ? op.bytes - op.done : kChunk;
        b.total = op.bytes; b.offset = op.done; b.bytes = n;
        if (n) memcpy(b.data, op.ptr + op.done, n);
        b.posted.store(p + 1, std::memory_order_release);
        op.done += n; op.started = true;
        return true;
    }
    Mailbox& b = sh->box[op.peer][c->rank];
    const uint64_t t = b.taken.load(std::memory_order_relaxed);
    if (b.posted.load(std::memory_order_acquire) == t) return false;              // nothing has arrived
    if (b.total != op.bytes || b.offset != op.done || b.bytes > op.bytes - op.done) { *bad = true; return false; }
    if (b.bytes) memcpy(op.ptr + op.done, b.data, b.bytes);
    op.done += b.bytes; op.started = true;
    b.taken.store(t + 1, std::memory_order_release);
    return true;
}
bool finished(const Op& op) { return op.started && op.done == op.bytes; }

ncclResult_t run_group(fake_nccl_comm* c, std::vector<Op>& ops)
{
    if (!c) { ops.clear(); return ncclSuccess; }                                  // an empty group
    Shared* sh = c->sh;
    if (sh->aborted.load(std::memory_order_acquire)) { ops.clear(); return ncclSystemError; }
    bool bad = false, orphaned = false;
    const bool ok = wait_for(sh, [&] {
        bool all = true;
        orphaned = false;
        // operations on one (direction, peer) complete in the order they were issued: only the first unfinished one may move
        bool busy_send[kMaxRanks] = {false}, busy_recv[kMaxRanks] = {false};
        for (Op& op : ops) {
            if (finished(op)) continue;
            bool* busy = op.send ? busy_send : busy_recv;
            // (the peer's departure is read BEFORE the attempt: if it had left by then and the attempt still finds nothing, nothing will come)
            const bool peer_gone = sh->gone[op.peer].load(std::memory_order_acquire) != 0;
            if (!busy[op.peer]) while (!finished(op) && progress(c, op, &bad)) {}
            if (bad) return true;
            if (!finished(op)) { busy[op.peer] = true; all = false; if (peer_gone) orphaned = true; }
        }
        return all;
    }, [&] { return orphaned; });
    ops.clear();
    if (bad) { sh->aborted.store(1, std::memory_order_release); return ncclInvalidArgument; }
    return ok ? ncclSuccess : ncclSystemError;
}

void leave(fake_nccl_comm* c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> g(g_m);
        for (size_t i = 0; i < g_comms.size(); i++) if (g_comms[i] == c) { g_comms.erase(g_comms.begin() + (long) i); break; }
    }
    if (c->sh) {
        // whoever leaves last removes the name (ranks that never arrived are not waited for: joined is what there is)
        if (c->rank >= 0) c->sh->gone[c->rank].store(1, std::memory_order_release);
        const int gone = c->sh->left.fetch_add(1, std::memory_order_acq_rel) + 1;
        if (gone >= c->sh->joined.load(std::memory_order_acquire)) shm_unlink(c->name);
        munmap(c->sh, sizeof(Shared));
    }
    delete c;
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    if (tick() != ncclSuccess) return ncclInternalError;
    memset(id, 0, sizeof *id);
    timespec t; clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/fxfakerccl-%ld-%u-%lx", (long) getpid(), g_id_counter.fetch_add(1), (unsigned long) t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank)
{
    *out = nullptr;
    const ncclResult_t injected = tick();
    if (world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return injected != ncclSuccess ? injected : ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (id.internal[0] != '/' || strlen(id.internal) >= sizeof(fake_nccl_comm::name)) return injected != ncclSuccess ? injected : ncclInvalidArgument;
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, (off_t) sizeof(Shared)) != 0) { close(fd); shm_unlink(id.internal); return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(id.internal); return ncclSystemError; }
    fake_nccl_comm* c = new fake_nccl_comm();
    c->world = world; c->rank = rank; c->sh = static_cast<Shared*>(p);
    strcpy(c->name, id.internal);
    Shared* sh = c->sh;
    sh->joined.fetch_add(1, std::memory_order_acq_rel);
    // an injected failure of this very call: the rank has shown up far enough for the others to learn that it failed (the bootstrap's job)
    if (injected != ncclSuccess) { sh->aborted.store(1, std::memory_order_release); c->rank = -1; leave(c); return injected; }
    { std::lock_guard<std::mutex> g(g_m); g_comms.push_back(c); }
    int expected = 0;
    const bool same_world = sh->world.compare_exchange_strong(expected, world) || expected == world;
    expected = 0;
    const bool rank_free = sh->claimed[rank].compare_exchange_strong(expected, 1);
    if (!same_world || !rank_free) { sh->aborted.store(1, std::memory_order_release); c->rank = -1; leave(c); return ncclInvalidArgument; }
    bool somebody_left = false;
    if (!wait_for(sh, [&] { somebody_left = sh->left.load(std::memory_order_acquire) > 0; return sh->joined.load(std::memory_order_acquire) >= world; },
                  [&] { return somebody_left; })) { leave(c); return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) { leave(c); return ncclSuccess; }

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { if (tick() != ncclSuccess) return ncclInternalError; *n = c->world; return ncclSuccess; }

ncclResult_t ncclAllGather(const void* s, void* d, size_t n, ncclDataType_t t, ncclComm_t c, hipStream_t)
{
    if (tick() != ncclSuccess) return ncclInternalError;
    const size_t bytes = n * width(t);
    if (bytes > kGatherRow) return ncclInvalidArgument;
    Shared* sh = c->sh;
    if (sh->aborted.load(std::memory_order_acquire)) return ncclSystemError;
    memcpy(sh->gather[c->rank], s, bytes);
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; r++) memcpy(static_cast<unsigned char*>(d) + (size_t) r * bytes, sh->gather[r], bytes);
    if (!barrier(c)) return ncclSystemError;                                     // nobody overwrites a row somebody still reads
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    if (tick() != ncclSuccess) return ncclInternalError;
    if (t_depth++ == 0) { t_ops.clear(); t_group_comm = nullptr; }
    return ncclSuccess;
}

static ncclResult_t queue(bool send, void* p, size_t n, ncclDataType_t t, int peer, ncclComm_t c)
{
    if (tick() != ncclSuccess) return ncclInternalError;
    if (!c || peer < 0 || peer >= c->world) return ncclInvalidArgument;
    if (t_group_comm && t_group_comm != c) return ncclInvalidArgument;          // (one communicator per group is all the shim does)
    t_group_comm = c;
    t_ops.push_back({send, peer, static_cast<unsigned char*>(p), n * width(t), 0, false});
    if (t_depth == 0) { const ncclResult_t r = run_group(c, t_ops); t_group_comm = nullptr; return r; }
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* s, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) { return queue(true, const_cast<void*>(s), n, t, peer, c); }
ncclResult_t ncclRecv(void* d, size_t n, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t) { return queue(false, d, n, t, peer, c); }

ncclResult_t ncclGroupEnd(void)
{
    const ncclResult_t injected = tick();
    if (t_depth > 0 && --t_depth > 0) return injected;
    fake_nccl_comm* c = t_group_comm;
    t_group_comm = nullptr;
    if (injected != ncclSuccess) { t_ops.clear(); return injected; }             // (tick() has marked the communicator aborted)
    return run_group(c, t_ops);
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "a peer failed, left or never arrived (fake rccl)";
    case ncclInvalidArgument: return "invalid argument or mismatched send / receive lengths (fake rccl)";
    default: return "injected failure (fake rccl)";
    }
}

// test handles
void fake_rccl_reset(void) { g_calls = 0; t_ops.clear(); t_depth = 0; t_group_comm = nullptr; }
long fake_rccl_calls(void) { return g_calls; }
}
