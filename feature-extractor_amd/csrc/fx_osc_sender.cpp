// fx_osc_sender.cpp -- the OSC sink at scale (host only, no GPU): the 60 Hz sender of every track's feature message, and a counting
// receiver for tests and soak runs.
//
// The reference builds two OSCFeatureAnalysisOutput objects per track (AnalyserTrackController.h:22-23), each a 60 Hz juce::Timer whose
// callback formats and sends ONE message (OSCFeatureAnalysisOutput.h:84-113,133).  At the channel counts this library analyses --
// 8192 per GPU, 65 536 per node -- that shape is 131 072 timers and 7.9e6 system calls a second.  Here the messages of a tick arrive
// already formatted ([count][stride] bytes: fx_get_osc_datagrams writes them on the GPU, fx_osc_encode_batch on the host), and a tick
// hands them to the kernel in batches: `threads` sender threads, each owning a contiguous slice of the tracks and one connected UDP
// socket per target, sendmmsg of up to 1024 messages per call -- or, with FX_OSC_SENDER_GSO, runs of equal-length messages as segmented
// sends (UDP_SEGMENT: one trip through the stack per 64 datagrams; the datagrams on the wire are the same).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/udp.h>
#include <netdb.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fx.h"

fx_status fx_fail(fx_status code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#ifndef UDP_SEGMENT
#define UDP_SEGMENT 103
#endif
#ifndef UDP_GRO
#define UDP_GRO 104
#endif

namespace {

// "ip[:port]", port 9000 by default: OSCFeatureAnalysisOutput::connectToAddress (ref OSCFeatureAnalysisOutput.h:115-123)
bool parse_target(const char* text, sockaddr_in* out, int default_port)
{
    std::string s = text ? text : "";
    int port = default_port;
    const size_t sep = s.rfind(':');
    if (sep != std::string::npos) { port = atoi(s.c_str() + sep + 1); s = s.substr(0, s.find(':')); }
    if (port < 0 || port > 65535) return false;
    *out = sockaddr_in();
    out->sin_family = AF_INET;
    out->sin_port = htons((unsigned short) port);
    if (inet_pton(AF_INET, s.c_str(), &out->sin_addr) == 1) return true;
    // a host name ("localhost", "renderer.local"): juce::OSCSender::connect takes one too
    addrinfo hints = addrinfo(), *found = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_DGRAM;
    if (s.empty() || getaddrinfo(s.c_str(), nullptr, &hints, &found) != 0 || found == nullptr) return false;
    out->sin_addr = reinterpret_cast<const sockaddr_in*>(found->ai_addr)->sin_addr;
    freeaddrinfo(found);
    return true;
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Batch {
    std::vector<unsigned char> data;
    std::vector<int> len;
    int stride = 0, count = 0;
};

constexpr int kChunk = 1024;         // messages per sendmmsg (UIO_MAXIOV is the limit of one message's iov, not of this)
constexpr int kSegments = 64;        // UDP_MAX_SEGMENTS: datagrams per segmented send

} // namespace

struct fx_osc_sender {
    std::vector<sockaddr_in> targets;
    int threads = 1;
    unsigned flags = 0;
    std::vector<int> fds;                         // [thread][target]
    std::atomic<bool> gso{false};

    std::mutex pub;                               // `current` and the pool of spare batches
    std::shared_ptr<Batch> current;
    std::vector<std::unique_ptr<Batch>> spare;

    std::mutex tick_lock;                         // one tick at a time (the timer against fx_osc_sender_send)
    std::mutex wm;
    std::condition_variable wake, done;
    unsigned long long tick_seq = 0;
    int pending = 0;
    std::shared_ptr<Batch> tick_batch;
    bool quit = false;
    std::vector<std::thread> workers;             // threads - 1 of them: the thread that runs the tick sends slice 0

    std::thread timer;
    std::atomic<bool> running{false};

    std::atomic<long long> ticks{0}, late{0}, datagrams{0}, dropped{0}, syscalls{0};
    std::mutex stat_lock;
    double last_ms = 0.0, max_ms = 0.0, total_ms = 0.0;
};

namespace {

// messages [lo, hi) of the batch to one socket, plain: sendmmsg, one message per datagram
void send_plain(fx_osc_sender* s, int fd, const Batch& b, int lo, int hi)
{
    static thread_local std::vector<mmsghdr> msgs(kChunk);
    static thread_local std::vector<iovec> iov(kChunk);
    long long sent = 0, dropped = 0, calls = 0;
    for (int at = lo; at < hi;) {
        const int n = hi - at < kChunk ? hi - at : kChunk;
        for (int i = 0; i < n; i++) {
            iov[(size_t) i].iov_base = const_cast<unsigned char*>(b.data.data()) + (size_t) (at + i) * (size_t) b.stride;
            iov[(size_t) i].iov_len = (size_t) b.len[(size_t) (at + i)];
            msgs[(size_t) i] = mmsghdr();
            msgs[(size_t) i].msg_hdr.msg_iov = &iov[(size_t) i];
            msgs[(size_t) i].msg_hdr.msg_iovlen = 1;
        }
        int done = 0;
        while (done < n) {
            const int r = sendmmsg(fd, msgs.data() + done, (unsigned) (n - done), 0);
            calls++;
            if (r > 0) { done += r; sent += r; continue; }
            if (r < 0 && errno == EINTR) continue;
            // ECONNREFUSED (nobody listens: the ICMP answer to an earlier datagram), ENOBUFS / EAGAIN (no room): that datagram is lost, go on
            done++; dropped++;
        }
        at += n;
    }
    s->datagrams += sent; s->dropped += dropped; s->syscalls += calls;
}

// the same with UDP_SEGMENT: a run of equal-length messages (at most 64, at most 65 507 bytes) is ONE send whose payload the
// stack cuts back into the datagrams; up to 16 runs per sendmmsg.  false: the kernel does not do it (the caller falls back for good).
bool send_segmented(fx_osc_sender* s, int fd, const Batch& b, int lo, int hi)
{
    constexpr int kRuns = 16;
    static thread_local std::vector<mmsghdr> msgs(kRuns);
    static thread_local std::vector<iovec> iov((size_t) kRuns * kSegments);
    struct Control { alignas(cmsghdr) char buf[CMSG_SPACE(sizeof(uint16_t))]; };
    static thread_local std::vector<Control> control(kRuns);
    long long sent = 0, dropped = 0, calls = 0;
    int at = lo;
    while (at < hi) {
        int runs = 0, first_of_run[kRuns + 1];
        int k = at;
        while (k < hi && runs < kRuns) {
            const int len = b.len[(size_t) k];
            int n = 1;
            while (k + n < hi && n < kSegments && b.len[(size_t) (k + n)] == len && (n + 1) * len <= 65507) n++;
            first_of_run[runs] = k;
            mmsghdr& m = msgs[(size_t) runs];
            m = mmsghdr();
            for (int i = 0; i < n; i++) {
                iovec& v = iov[(size_t) runs * kSegments + (size_t) i];
                v.iov_base = const_cast<unsigned char*>(b.data.data()) + (size_t) (k + i) * (size_t) b.stride;
                v.iov_len = (size_t) len;
            }
            m.msg_hdr.msg_iov = &iov[(size_t) runs * kSegments];
            m.msg_hdr.msg_iovlen = (size_t) n;
            if (n > 1) {
                m.msg_hdr.msg_control = control[(size_t) runs].buf;
                m.msg_hdr.msg_controllen = sizeof control[(size_t) runs].buf;
                cmsghdr* cm = CMSG_FIRSTHDR(&m.msg_hdr);
                cm->cmsg_level = SOL_UDP;
                cm->cmsg_type = UDP_SEGMENT;
                cm->cmsg_len = CMSG_LEN(sizeof(uint16_t));
                const uint16_t seg = (uint16_t) len;
                memcpy(CMSG_DATA(cm), &seg, sizeof seg);
            }
            k += n;
            runs++;
        }
        first_of_run[runs] = k;
        int done = 0;
        while (done < runs) {
            const int r = sendmmsg(fd, msgs.data() + done, (unsigned) (runs - done), 0);
            calls++;
            if (r > 0) { for (int i = done; i < done + r; i++) sent += first_of_run[i + 1] - first_of_run[i]; done += r; continue; }
            if (r < 0 && errno == EINTR) continue;
            if (r < 0 && (errno == EINVAL || errno == EIO || errno == ENOPROTOOPT || errno == EOPNOTSUPP) && sent == 0 && dropped == 0 && done == 0 && at == lo) {
                s->syscalls += calls;
                return false;                   // segmentation refused on the first send: not a kernel (or a route) that does it
            }
            dropped += first_of_run[done + 1] - first_of_run[done];
            done++;
        }
        at = k;
    }
    s->datagrams += sent; s->dropped += dropped; s->syscalls += calls;
    return true;
}

void send_slice(fx_osc_sender* s, int t, const Batch& b)
{
    const int lo = (int) ((long long) b.count * t / s->threads), hi = (int) ((long long) b.count * (t + 1) / s->threads);
    if (lo >= hi) return;
    for (size_t k = 0; k < s->targets.size(); k++) {
        const int fd = s->fds[(size_t) t * s->targets.size() + k];
        if (s->gso.load(std::memory_order_relaxed)) {
            if (send_segmented(s, fd, b, lo, hi)) continue;
            s->gso.store(false, std::memory_order_relaxed);
        }
        send_plain(s, fd, b, lo, hi);
    }
}

void worker(fx_osc_sender* s, int t)
{
    unsigned long long seen = 0;
    for (;;) {
        std::shared_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> g(s->wm);
            s->wake.wait(g, [&] { return s->quit || s->tick_seq != seen; });
            if (s->quit) return;
            seen = s->tick_seq;
            b = s->tick_batch;
        }
        if (b) send_slice(s, t, *b);
        {
            std::lock_guard<std::mutex> g(s->wm);
            if (--s->pending == 0) s->done.notify_all();
        }
    }
}

// one tick on the calling thread (slice 0) and the workers; returns the datagrams the kernel accepted
long long run_tick(fx_osc_sender* s)
{
    std::lock_guard<std::mutex> one(s->tick_lock);
    std::shared_ptr<Batch> b;
    { std::lock_guard<std::mutex> g(s->pub); b = s->current; }
    s->ticks++;
    if (!b || b->count == 0) return 0;
    const long long before = s->datagrams.load();
    const double t0 = now_ms();
    if (s->threads > 1) {
        std::lock_guard<std::mutex> g(s->wm);
        s->tick_batch = b;
        s->pending = s->threads - 1;
        s->tick_seq++;
        s->wake.notify_all();
    }
    send_slice(s, 0, *b);
    if (s->threads > 1) {
        std::unique_lock<std::mutex> g(s->wm);
        s->done.wait(g, [&] { return s->pending == 0; });
        s->tick_batch.reset();
    }
    const double ms = now_ms() - t0;
    {
        std::lock_guard<std::mutex> g(s->stat_lock);
        s->last_ms = ms; s->total_ms += ms;
        if (ms > s->max_ms) s->max_ms = ms;
    }
    return s->datagrams.load() - before;
}

void stop_timer(fx_osc_sender* s)
{
    s->running = false;
    if (s->timer.joinable()) s->timer.join();
}

} // namespace

extern "C" {

fx_status fx_osc_sender_create(fx_osc_sender** out, const char* primary, const char* secondary, int threads, unsigned flags)
{
    if (!out || !primary) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (threads < 1 || threads > 64) return fx_fail(FX_ERR_INVALID_ARGUMENT, "1 .. 64 sender threads, not %d", threads);
    fx_osc_sender* s = new (std::nothrow) fx_osc_sender();
    if (!s) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    s->threads = threads;
    s->flags = flags;
    s->gso = (flags & FX_OSC_SENDER_GSO) != 0;
    sockaddr_in a;
    if (!parse_target(primary, &a, 9000)) { delete s; return fx_fail(FX_ERR_INVALID_ARGUMENT, "cannot parse the target '%s' (ip[:port])", primary); }
    s->targets.push_back(a);
    if (secondary && *secondary) {
        if (!parse_target(secondary, &a, 9000)) { delete s; return fx_fail(FX_ERR_INVALID_ARGUMENT, "cannot parse the secondary target '%s' (ip[:port])", secondary); }
        s->targets.push_back(a);
    }
    for (int t = 0; t < threads; t++)
        for (const sockaddr_in& target : s->targets) {
            const int fd = socket(AF_INET, SOCK_DGRAM, 0);
            int sndbuf = 8 << 20;
            if (fd >= 0) (void) setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &sndbuf, sizeof sndbuf);
            if (fd < 0 || connect(fd, reinterpret_cast<const sockaddr*>(&target), sizeof target) != 0) {
                const int e = errno;
                if (fd >= 0) close(fd);
                for (int f : s->fds) close(f);
                delete s;
                return fx_fail(FX_ERR_INVALID_ARGUMENT, "UDP socket for the sender: %s", strerror(e));
            }
            s->fds.push_back(fd);
        }
    for (int t = 1; t < threads; t++) s->workers.emplace_back(worker, s, t);
    *out = s;
    return FX_OK;
}

fx_status fx_osc_sender_destroy(fx_osc_sender* s)
{
    if (!s) return FX_OK;
    stop_timer(s);
    { std::lock_guard<std::mutex> g(s->wm); s->quit = true; s->wake.notify_all(); }
    for (std::thread& w : s->workers) w.join();
    for (int f : s->fds) close(f);
    s->current.reset();            // (its deleter files the batch in `spare`, which goes with the object)
    delete s;
    return FX_OK;
}

fx_status fx_osc_sender_update(fx_osc_sender* s, const unsigned char* datagrams, int stride, const int* lengths, int count)
{
    if (!s || count < 0 || (count > 0 && (!datagrams || !lengths))) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (count > 0 && stride < 1) return fx_fail(FX_ERR_INVALID_ARGUMENT, "stride %d", stride);
    for (int i = 0; i < count; i++)
        if (lengths[i] < 1 || lengths[i] > stride || lengths[i] > 65507) return fx_fail(FX_ERR_INVALID_ARGUMENT, "message %d is %d bytes in a slot of %d", i, lengths[i], stride);
    std::unique_ptr<Batch> b;
    {
        std::lock_guard<std::mutex> g(s->pub);
        if (!s->spare.empty()) { b = std::move(s->spare.back()); s->spare.pop_back(); }
    }
    if (!b) b.reset(new (std::nothrow) Batch());
    if (!b) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    try {
        b->data.assign(datagrams, datagrams + (size_t) count * (size_t) stride);
        b->len.assign(lengths, lengths + count);
    } catch (const std::bad_alloc&) { return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    b->stride = stride;
    b->count = count;
    // a batch goes back to the pool when the last tick that sends it lets go of it
    std::shared_ptr<Batch> shared(b.release(), [s](Batch* p) {
        std::lock_guard<std::mutex> g(s->pub);
        if (s->spare.size() < 3) s->spare.emplace_back(p); else delete p;
    });
    std::shared_ptr<Batch> old;
    {
        std::lock_guard<std::mutex> g(s->pub);
        old.swap(s->current);
        s->current = std::move(shared);
    }
    return FX_OK;              // (`old` is released here, outside the lock its deleter takes)
}

fx_status fx_osc_sender_send(fx_osc_sender* s, long long* sent)
{
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null sender");
    const long long n = run_tick(s);
    if (sent) *sent = n;
    return FX_OK;
}

fx_status fx_osc_sender_start(fx_osc_sender* s, double rate_hz)
{
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null sender");
    if (!(rate_hz > 0.0) || rate_hz > 100000.0) return fx_fail(FX_ERR_INVALID_ARGUMENT, "timer rate %g Hz", rate_hz);
    stop_timer(s);
    s->running = true;
    s->timer = std::thread([s, rate_hz] {
        using clock = std::chrono::steady_clock;
        const auto period = std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(1.0 / rate_hz));
        clock::time_point next = clock::now() + period;
        while (s->running.load()) {
            std::this_thread::sleep_until(next);
            if (!s->running.load()) break;
            const clock::time_point began = clock::now();
            if (began >= next + period) {          // the tick before this one took longer than a period: this one is late, the missed ones are skipped
                s->late++;
                next = began;
            }
            (void) run_tick(s);                    // timerCallback -> sendSpectralFeaturesViaOSC, ref OSCFeatureAnalysisOutput.h:84-87
            next += period;
        }
    });
    return FX_OK;
}

fx_status fx_osc_sender_stop(fx_osc_sender* s)
{
    if (!s) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null sender");
    stop_timer(s);
    return FX_OK;
}

fx_status fx_osc_sender_get_stats(fx_osc_sender* s, fx_osc_sender_stats* out)
{
    if (!s || !out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    out->ticks = s->ticks.load(); out->late_ticks = s->late.load(); out->datagrams = s->datagrams.load();
    out->dropped = s->dropped.load(); out->syscalls = s->syscalls.load();
    std::lock_guard<std::mutex> g(s->stat_lock);
    out->last_tick_ms = s->last_ms; out->max_tick_ms = s->max_ms; out->total_tick_ms = s->total_ms;
    return FX_OK;
}

} // extern "C"

// ---------------------------------------------------------------- the counting receiver ----------------------------------------------------------------
struct fx_osc_receiver {
    std::vector<int> fds;
    std::vector<std::thread> threads;
    std::atomic<bool> quit{false};
    std::atomic<long long> datagrams{0}, bytes{0}, malformed{0};
    int port = 0;
    bool gro = false;                                  // UDP_GRO taken by the sockets (see receive())
    std::string prefix;
    int keep = 0;
    static constexpr int kSlot = 160;                  // 4 bytes of length + a message of up to 156
    std::vector<unsigned char> last;                   // [keep][kSlot]
    std::unique_ptr<std::atomic_flag[]> busy;          // [keep]
};

namespace {

// an OSC message of twelve floats: address, padded; ",ffffffffffff", padded; 48 bytes
bool well_formed(const unsigned char* m, int len, int* address_len)
{
    if (len < 76 || (len & 3) || m[0] != '/') return false;
    const int apad = len - 64;
    int alen = 0;
    while (alen < apad && m[alen]) alen++;
    if (alen == apad || ((alen + 4) & ~3) != apad) return false;
    *address_len = alen;
    return memcmp(m + apad, ",ffffffffffff\0\0\0", 16) == 0;
}

void take(fx_osc_receiver* r, const unsigned char* m, int len, long long* bad)
{
    int alen = 0;
    if (!well_formed(m, len, &alen)) { ++*bad; return; }
    if (r->keep > 0 && alen > (int) r->prefix.size() && len + 4 <= fx_osc_receiver::kSlot && memcmp(m, r->prefix.data(), r->prefix.size()) == 0) {
        long ch = 0;
        bool digits = true;
        for (int k = (int) r->prefix.size(); k < alen; k++) { if (m[k] < '0' || m[k] > '9' || ch > 100000000) { digits = false; break; } ch = ch * 10 + (m[k] - '0'); }
        if (digits && ch < r->keep) {
            std::atomic_flag& f = r->busy[(size_t) ch];
            while (f.test_and_set(std::memory_order_acquire)) {}
            unsigned char* slot = r->last.data() + (size_t) ch * fx_osc_receiver::kSlot;
            memcpy(slot, &len, 4);
            memcpy(slot + 4, m, (size_t) len);
            f.clear(std::memory_order_release);
        }
    }
}

// With UDP_GRO on the socket a segmented send that never left the host (loopback) arrives as it was sent: one buffer of up to 64
// datagrams and a control message with their length -- the receiver then costs one trip through the stack per 64 datagrams as well.
void receive(fx_osc_receiver* r, int fd, bool gro)
{
    const int kBatch = gro ? 32 : 256, kRoom = gro ? 65536 : 256;
    std::vector<unsigned char> room((size_t) kBatch * (size_t) kRoom);
    std::vector<mmsghdr> msgs((size_t) kBatch);
    std::vector<iovec> iov((size_t) kBatch);
    struct Control { alignas(cmsghdr) char buf[CMSG_SPACE(sizeof(int)) + 64]; };
    std::vector<Control> control((size_t) kBatch);
    while (!r->quit.load(std::memory_order_relaxed)) {
        for (int i = 0; i < kBatch; i++) {
            iov[(size_t) i].iov_base = room.data() + (size_t) i * (size_t) kRoom;
            iov[(size_t) i].iov_len = (size_t) kRoom;
            msgs[(size_t) i] = mmsghdr();
            msgs[(size_t) i].msg_hdr.msg_iov = &iov[(size_t) i];
            msgs[(size_t) i].msg_hdr.msg_iovlen = 1;
            if (gro) { msgs[(size_t) i].msg_hdr.msg_control = control[(size_t) i].buf; msgs[(size_t) i].msg_hdr.msg_controllen = sizeof control[(size_t) i].buf; }
        }
        const int n = recvmmsg(fd, msgs.data(), (unsigned) kBatch, MSG_WAITFORONE, nullptr);
        if (n <= 0) continue;                          // (the socket's receive time-out: look at `quit` again)
        long long count = 0, b = 0, bad = 0;
        for (int i = 0; i < n; i++) {
            const unsigned char* m = room.data() + (size_t) i * (size_t) kRoom;
            const int len = (int) msgs[(size_t) i].msg_len;
            b += len;
            if (msgs[(size_t) i].msg_hdr.msg_flags & MSG_TRUNC) { bad++; count++; continue; }
            int segment = len;
            if (gro)
                for (cmsghdr* cm = CMSG_FIRSTHDR(&msgs[(size_t) i].msg_hdr); cm; cm = CMSG_NXTHDR(&msgs[(size_t) i].msg_hdr, cm))
                    if (cm->cmsg_level == SOL_UDP && cm->cmsg_type == UDP_GRO) { int v = 0; memcpy(&v, CMSG_DATA(cm), sizeof v); if (v > 0) segment = v; }
            for (int at = 0; at < len; at += segment) { take(r, m + at, len - at < segment ? len - at : segment, &bad); count++; }
            if (len == 0) { bad++; count++; }
        }
        r->datagrams += count; r->bytes += b; r->malformed += bad;
    }
}

} // namespace

extern "C" {

fx_status fx_osc_receiver_create(fx_osc_receiver** out, const char* bind_address, int threads, const char* prefix, int keep_channels, unsigned flags)
{
    if (!out) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (threads < 1 || threads > 64) return fx_fail(FX_ERR_INVALID_ARGUMENT, "1 .. 64 receiver threads, not %d", threads);
    if (keep_channels < 0 || keep_channels > (1 << 24) || (keep_channels > 0 && !prefix)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "keep_channels %d needs a prefix", keep_channels);
    sockaddr_in a;
    if (!parse_target(bind_address ? bind_address : "127.0.0.1:0", &a, 0)) return fx_fail(FX_ERR_INVALID_ARGUMENT, "cannot parse the address '%s'", bind_address);
    fx_osc_receiver* r = new (std::nothrow) fx_osc_receiver();
    if (!r) return fx_fail(FX_ERR_OUT_OF_MEMORY, "host allocation failed");
    r->prefix = prefix ? prefix : "";
    r->keep = keep_channels;
    if (keep_channels > 0) {
        r->last.assign((size_t) keep_channels * fx_osc_receiver::kSlot, 0);
        r->busy.reset(new std::atomic_flag[(size_t) keep_channels]);
        for (int i = 0; i < keep_channels; i++) r->busy[(size_t) i].clear();
    }
    for (int t = 0; t < threads; t++) {
        const int fd = socket(AF_INET, SOCK_DGRAM, 0);
        int one = 1, rcvbuf = 64 << 20;
        timeval tv{0, 100000};
        bool ok = fd >= 0 && setsockopt(fd, SOL_SOCKET, SO_REUSEPORT, &one, sizeof one) == 0 && setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv) == 0;
        if (ok) (void) setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &rcvbuf, sizeof rcvbuf);       // (as much of it as net.core.rmem_max allows)
        if (ok && t == 0) r->gro = !(flags & FX_OSC_RECEIVER_NO_GRO) && setsockopt(fd, IPPROTO_UDP, UDP_GRO, &one, sizeof one) == 0;
        else if (ok && r->gro) ok = setsockopt(fd, IPPROTO_UDP, UDP_GRO, &one, sizeof one) == 0;
        ok = ok && bind(fd, reinterpret_cast<const sockaddr*>(&a), sizeof a) == 0;
        if (ok && t == 0) {
            socklen_t len = sizeof a;
            ok = getsockname(fd, reinterpret_cast<sockaddr*>(&a), &len) == 0;      // the port the kernel chose: the other sockets join it
            r->port = ntohs(a.sin_port);
        }
        if (!ok) {
            const int e = errno;
            if (fd >= 0) close(fd);
            for (int f : r->fds) close(f);
            delete r;
            return fx_fail(FX_ERR_INVALID_ARGUMENT, "UDP socket for the receiver: %s", strerror(e));
        }
        r->fds.push_back(fd);
    }
    for (int fd : r->fds) r->threads.emplace_back(receive, r, fd, r->gro);
    *out = r;
    return FX_OK;
}

fx_status fx_osc_receiver_destroy(fx_osc_receiver* r)
{
    if (!r) return FX_OK;
    r->quit = true;
    for (std::thread& t : r->threads) t.join();
    for (int f : r->fds) close(f);
    delete r;
    return FX_OK;
}

int fx_osc_receiver_port(fx_osc_receiver* r) { return r ? r->port : -1; }

fx_status fx_osc_receiver_get_stats(fx_osc_receiver* r, long long* datagrams, long long* bytes, long long* malformed)
{
    if (!r) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null receiver");
    if (datagrams) *datagrams = r->datagrams.load();
    if (bytes) *bytes = r->bytes.load();
    if (malformed) *malformed = r->malformed.load();
    return FX_OK;
}

fx_status fx_osc_receiver_last(fx_osc_receiver* r, int channel, unsigned char* out, int cap, int* len)
{
    if (!r || !out || !len) return fx_fail(FX_ERR_INVALID_ARGUMENT, "null argument");
    if (channel < 0 || channel >= r->keep) return fx_fail(FX_ERR_INVALID_ARGUMENT, "channel %d is not one of the %d kept", channel, r->keep);
    std::atomic_flag& f = r->busy[(size_t) channel];
    while (f.test_and_set(std::memory_order_acquire)) {}
    const unsigned char* slot = r->last.data() + (size_t) channel * fx_osc_receiver::kSlot;
    int n = 0;
    memcpy(&n, slot, 4);
    const bool fits = n <= cap;
    if (fits) memcpy(out, slot + 4, (size_t) n);
    f.clear(std::memory_order_release);
    if (!fits) return fx_fail(FX_ERR_INVALID_ARGUMENT, "the message is %d bytes, the buffer %d", n, cap);
    *len = n;
    return FX_OK;
}

} // extern "C"
