// Multi-GPU gather through the C ABI, from C++ with no torch / Python in the process: one process per rank (fork), the communicator
// id travels through pipes, every rank analyses its own channel shard and the latest smoothed vectors are gathered to the sink rank
// (fx_gather_smoothed: ref AnalyserTrackController.h:199-210 -- one controller per channel, nothing shared -- and
// OSCFeatureAnalysisOutput.h:89-113 -- the sender samples every track's AudioFeatures).  The PARENT, which makes no GPU call, receives
// what every rank held locally after every round and what every sink gathered, and checks each rank's block at its offset, bit for bit.
//
// Two builds: against libfx_hip.so + RCCL on a box with >= world GPUs (tests/test_gpu_sharded.py), and against the fake HIP runtime and
// the multi-process fake RCCL of tests/cpp/fake_hip/ under ASan / UBSan on the CPU (tests/test_host_sanitized_cpu.py), where this is
// the only place the world > 1 branches of csrc/fx_comm.cpp run without an 8-GPU node.
//
//   comm_ranks <world> [shards=8192,8191,1,37] [sinks=0,0,0,3,1] [dst=host|device|mixed] [fail=<rank>:<k>] [hipfail=<rank>:<k>]
//     shards  channels per rank (default 5 + rank: ragged on purpose)
//     sinks   destination rank of each round (default 0,0,0); no fx_comm_sync between rounds: they are all in flight together
//     dst     where the sink's table lives; mixed = host on even rounds, device on odd ones
//     fail    (fake RCCL only) the k-th RCCL call of rank <rank> fails: every rank must come back with an error or a result, none may
//             hang, crash or leak, and the failing rank's context must still analyse afterwards
//     hipfail (fake HIP only) the same for the k-th HIP call that rank makes after fx_create
//   exit code 77 = not enough GPUs (skip)
#include <dlfcn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "fx.h"

namespace {
constexpr int N = 1024, kHops = 3;

struct Plan {
    int world = 2;
    std::vector<int> shards, sinks;
    std::string dst = "mixed";
    int fail_rank = -1, fail_call = 0;
    bool fail_hip = false;
    bool device_round(int round) const { return dst == "device" || (dst == "mixed" && (round & 1)); }
    int first(int rank) const { int f = 0; for (int r = 0; r < rank; r++) f += shards[(size_t) r]; return f; }
    int total() const { return first(world); }
};

std::vector<int> ints(const char* s) { std::vector<int> v; while (*s) { v.push_back(atoi(s)); while (*s && *s != ',') s++; if (*s) s++; } return v; }

void fill_hops(std::vector<float>& h, int first_channel, int channels, int round)
{
    // any deterministic per-channel signal: a tone whose pitch depends on the global channel id, another stretch of it every round
    const int half = N / 2;
    for (int c = 0; c < channels; c++) {
        const double f = 110.0 * std::pow(2.0, ((first_channel + c) % 24) / 12.0) * (1.0 + (first_channel + c) / 65536.0);
        for (int n = 0; n < kHops * half; n++) {
            const int t = n + round * kHops * half;
            h[(size_t) c * kHops * half + n] = (float) (0.5 * std::sin(2.0 * 3.14159265358979323846 * f * t / 48000.0) + 0.01 * ((t * 7 + (first_channel + c) * 13) % 17 - 8) / 8.0);
        }
    }
}

bool read_all(int fd, void* p, size_t n) { char* b = (char*) p; while (n) { ssize_t r = read(fd, b, n); if (r <= 0) return false; b += r; n -= (size_t) r; } return true; }
bool write_all(int fd, const void* p, size_t n) { const char* b = (const char*) p; while (n) { ssize_t r = write(fd, b, n); if (r <= 0) return false; b += r; n -= (size_t) r; } return true; }

// What a rank reports to the parent: status (0 = all rounds done, 1 = an fx call failed and the rank recovered, 2 = broken), then, when
// status is 0, its own [channels][12] after every round and, for the rounds it was the sink of, the gathered [total][12].
struct Report { int status = 2; std::vector<std::vector<float>> mine, table; };

int run_rank(const Plan& plan, int rank, int id_fd, int out_fd)
{
    const int world = plan.world, channels = plan.shards[(size_t) rank], total = plan.total(), rounds = (int) plan.sinks.size();
    const bool may_fail = plan.fail_rank >= 0;
    unsigned char id[FX_COMM_ID_BYTES];
    fx_context* ctx = nullptr;
    std::vector<void*> device;
    int status = 0;
    const char* failed_in = "";
#define TRY(call) do { if (status == 0 && (call) != FX_OK) { status = 1; failed_in = #call; fprintf(stderr, "rank %d: %s -> %s\n", rank, #call, fx_last_error()); } } while (0)
    TRY(fx_create(&ctx, rank, channels, N, 48000.0, 0));
    if (status) return 2;
    void (*hip_fail_at)(long) = (void (*)(long)) dlsym(RTLD_DEFAULT, "fake_hip_fail_at");
    void (*hip_reset)(void) = (void (*)(void)) dlsym(RTLD_DEFAULT, "fake_hip_reset");
    if (plan.fail_hip && plan.fail_rank == rank) {
        if (!hip_fail_at || !hip_reset) { fprintf(stderr, "hipfail= needs the fake HIP runtime\n"); fx_destroy(ctx); return 2; }
        hip_reset();
        hip_fail_at(plan.fail_call);
    }
    if (!read_all(id_fd, id, sizeof id)) { fprintf(stderr, "rank %d: no id\n", rank); fx_destroy(ctx); return 2; }
    bool have_id = false;
    for (unsigned char b : id) have_id = have_id || b != 0;
    if (!have_id) { status = 1; failed_in = "(rank 0 could not make the id)"; }
    TRY(fx_comm_create(ctx, rank, world, id, FX_COMM_ID_BYTES));
    std::vector<int> first((size_t) world, -1);
    int got_total = -1;
    TRY(fx_comm_layout(ctx, &got_total, first.data()));
    if (status == 0) {
        bool ok = got_total == total;
        for (int r = 0; r < world; r++) ok = ok && first[(size_t) r] == plan.first(r);
        if (!ok) { fprintf(stderr, "rank %d: layout is total %d, first of this rank %d; expected %d, %d\n", rank, got_total, first[(size_t) rank], total, plan.first(rank)); status = 2; }
    }
    Report rep;
    rep.mine.resize((size_t) rounds);
    rep.table.resize((size_t) rounds);
    std::vector<float*> d_table((size_t) rounds, nullptr);
    std::vector<float> h((size_t) channels * kHops * (N / 2));
    for (int round = 0; round < rounds && status == 0; round++) {
        const int sink = plan.sinks[(size_t) round];
        fill_hops(h, plan.first(rank), channels, round);
        TRY(fx_push_hops(ctx, h.data(), kHops, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, nullptr));
        float* out = nullptr;
        int kind = FX_MEM_HOST;
        if (rank == sink) {
            rep.table[(size_t) round].assign((size_t) total * 12, -1.0f);
            out = rep.table[(size_t) round].data();
            if (plan.device_round(round)) {
                void* p = nullptr;
                if (hipMalloc(&p, (size_t) total * 12 * sizeof(float)) != hipSuccess) { status = 1; failed_in = "hipMalloc (this program's own table)"; break; }
                device.push_back(p);
                d_table[(size_t) round] = out = (float*) p;
                kind = FX_MEM_DEVICE;
            }
        }
        TRY(fx_gather_smoothed(ctx, sink, out, kind));
        rep.mine[(size_t) round].resize((size_t) channels * 12);
        TRY(fx_get_smoothed(ctx, rep.mine[(size_t) round].data(), FX_MEM_HOST));
    }
    TRY(fx_comm_sync(ctx));
    for (int round = 0; round < rounds && status == 0; round++)
        if (d_table[(size_t) round] && hipMemcpy(rep.table[(size_t) round].data(), d_table[(size_t) round], (size_t) total * 12 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { status = 1; failed_in = "hipMemcpy (this program's own table)"; }
    if (status == 0) {
        int ranks = 0, gathers = 0;
        TRY(fx_comm_stats(ctx, &ranks, &gathers, nullptr, nullptr, nullptr));
        if (status == 0 && (ranks != world || gathers != rounds)) { fprintf(stderr, "rank %d: statistics say %d ranks, %d gathers\n", rank, ranks, gathers); status = 2; }
    }
    if (status == 1 && !may_fail) status = 2;
    if (status == 1) fprintf(stderr, "rank %d: failure reported by %s; recovering\n", rank, failed_in);
    // whatever happened: the communicator can be destroyed, the context analyses again, everything is given back
    unsetenv("FAKE_RCCL_FAIL_AT");
    if (hip_fail_at) hip_fail_at(0);
    if (fx_comm_destroy(ctx) != FX_OK) { fprintf(stderr, "rank %d: fx_comm_destroy: %s\n", rank, fx_last_error()); status = 2; }
    if (fx_push_hops(ctx, h.data(), kHops, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, nullptr) != FX_OK || fx_sync(ctx) != FX_OK) { fprintf(stderr, "rank %d: the context does not analyse after the communicator is gone: %s\n", rank, fx_last_error()); status = 2; }
    if (fx_destroy(ctx) != FX_OK) { fprintf(stderr, "rank %d: fx_destroy: %s\n", rank, fx_last_error()); status = 2; }
    for (void* p : device) (void) hipFree(p);
    if (long (*live)(void) = (long (*)(void)) dlsym(RTLD_DEFAULT, "fake_hip_live"))
        if (live() != 0) { fprintf(stderr, "rank %d: %ld device objects still allocated\n", rank, live()); status = 2; }
    rep.status = status;
    bool sent = write_all(out_fd, &rep.status, sizeof rep.status);
    for (int round = 0; round < rounds && sent && status == 0; round++) {
        sent = write_all(out_fd, rep.mine[(size_t) round].data(), rep.mine[(size_t) round].size() * sizeof(float));
        if (sent && rank == plan.sinks[(size_t) round]) sent = write_all(out_fd, rep.table[(size_t) round].data(), rep.table[(size_t) round].size() * sizeof(float));
    }
    return sent && status != 2 ? 0 : 1;
}
} // namespace

int main(int argc, char** argv)
{
    Plan plan;
    plan.world = argc > 1 ? atoi(argv[1]) : 2;
    if (plan.world < 1 || plan.world > 8) return 2;
    for (int r = 0; r < plan.world; r++) plan.shards.push_back(5 + r);
    plan.sinks = {0, 0, 0};
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a.rfind("shards=", 0) == 0) plan.shards = ints(a.c_str() + 7);
        else if (a.rfind("sinks=", 0) == 0) plan.sinks = ints(a.c_str() + 6);
        else if (a.rfind("dst=", 0) == 0) plan.dst = a.substr(4);
        else if (a.rfind("hipfail=", 0) == 0) { plan.fail_hip = true; plan.fail_rank = atoi(a.c_str() + 8); const char* c = strchr(a.c_str(), ':'); plan.fail_call = c ? atoi(c + 1) : 1; }
        else if (a.rfind("fail=", 0) == 0) { plan.fail_rank = atoi(a.c_str() + 5); const char* c = strchr(a.c_str(), ':'); plan.fail_call = c ? atoi(c + 1) : 1; }
        else { fprintf(stderr, "comm_ranks: unknown argument %s\n", a.c_str()); return 2; }
    }
    if ((int) plan.shards.size() != plan.world) { fprintf(stderr, "comm_ranks: %d shards for %d ranks\n", (int) plan.shards.size(), plan.world); return 2; }
    for (int s : plan.sinks) if (s < 0 || s >= plan.world) return 2;
    for (int s : plan.shards) if (s < 1) return 2;
    const int world = plan.world, rounds = (int) plan.sinks.size();
    fflush(nullptr);
    // the parent makes no GPU call; the device count comes from a short-lived child
    {
        int pf[2]; if (pipe(pf)) return 2;
        pid_t p = fork();
        if (p == 0) {
            fx_context* probe = nullptr; int n = 0;
            for (; n < world; n++) { if (fx_create(&probe, n, 1, 1024, 48000.0, 0) != FX_OK) break; fx_destroy(probe); }
            write_all(pf[1], &n, sizeof n); _exit(0);
        }
        int n = 0; close(pf[1]); read_all(pf[0], &n, sizeof n); close(pf[0]); waitpid(p, nullptr, 0);
        if (n < world) { fprintf(stderr, "comm_ranks: %d GPU(s) visible, %d needed -- skipped\n", n, world); return 77; }
    }
    std::vector<int> id_r((size_t) world), id_w((size_t) world), res_r((size_t) world), res_w((size_t) world);
    for (int r = 0; r < world; r++) { int pf[2]; if (pipe(pf)) return 2; id_r[(size_t) r] = pf[0]; id_w[(size_t) r] = pf[1]; }
    for (int r = 0; r < world; r++) { int pf[2]; if (pipe(pf)) return 2; res_r[(size_t) r] = pf[0]; res_w[(size_t) r] = pf[1]; }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        pid_t p = fork();
        if (p == 0) {
            for (int q = 0; q < world; q++) { close(res_r[(size_t) q]); if (q != r) { close(res_w[(size_t) q]); close(id_r[(size_t) q]); } }
            if (plan.fail_rank == r && !plan.fail_hip) { char v[16]; snprintf(v, sizeof v, "%d", plan.fail_call); setenv("FAKE_RCCL_FAIL_AT", v, 1); }
            if (r == 0) {                                        // rank 0 makes the id and hands it to everyone (itself included); all zeros = it could not
                unsigned char id[FX_COMM_ID_BYTES];
                if (fx_comm_unique_id(id, sizeof id) != FX_OK) { fprintf(stderr, "fx_comm_unique_id: %s\n", fx_last_error()); memset(id, 0, sizeof id); }
                for (int q = 0; q < world; q++) write_all(id_w[(size_t) q], id, sizeof id);
            }
            for (int q = 0; q < world; q++) close(id_w[(size_t) q]);
            exit(run_rank(plan, r, id_r[(size_t) r], res_w[(size_t) r]));     // exit, not _exit: the leak check of a sanitizer build runs at exit
        }
        kids.push_back(p);
    }
    for (int r = 0; r < world; r++) { close(id_r[(size_t) r]); close(id_w[(size_t) r]); close(res_w[(size_t) r]); }
    // every rank's report, in rank order (a rank blocked on its pipe has already destroyed its communicator: nobody waits for it)
    std::vector<Report> reps((size_t) world);
    for (int r = 0; r < world; r++) {
        Report& rep = reps[(size_t) r];
        if (!read_all(res_r[(size_t) r], &rep.status, sizeof rep.status)) { rep.status = 2; continue; }
        rep.mine.resize((size_t) rounds); rep.table.resize((size_t) rounds);
        for (int round = 0; round < rounds && rep.status == 0; round++) {
            rep.mine[(size_t) round].resize((size_t) plan.shards[(size_t) r] * 12);
            if (!read_all(res_r[(size_t) r], rep.mine[(size_t) round].data(), rep.mine[(size_t) round].size() * sizeof(float))) { rep.status = 2; break; }
            if (r == plan.sinks[(size_t) round]) {
                rep.table[(size_t) round].resize((size_t) plan.total() * 12);
                if (!read_all(res_r[(size_t) r], rep.table[(size_t) round].data(), rep.table[(size_t) round].size() * sizeof(float))) { rep.status = 2; break; }
            }
        }
    }
    int rc = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "comm_ranks: a rank ended with status 0x%x\n", st); rc = 1; } }
    int failed = 0, broken = 0;
    for (const Report& rep : reps) { failed += rep.status == 1; broken += rep.status == 2; }
    if (broken) rc = 1;
    if (plan.fail_rank >= 0) {
        if (rc == 0) printf("comm_ranks injected: %d ranks, call %d of rank %d failing: %d rank(s) reported a failure, all recovered\n", world, plan.fail_call, plan.fail_rank, failed);
        return rc;
    }
    if (failed) rc = 1;
    int distinct = 0;
    for (int round = 0; round < rounds && rc == 0; round++) {
        const std::vector<float>& table = reps[(size_t) plan.sinks[(size_t) round]].table[(size_t) round];
        for (int r = 0; r < world; r++) {
            const std::vector<float>& mine = reps[(size_t) r].mine[(size_t) round];
            if (memcmp(mine.data(), table.data() + (size_t) plan.first(r) * 12, mine.size() * sizeof(float)) != 0) {
                fprintf(stderr, "round %d: gathered block of rank %d (offset %d) differs from that rank's own features\n", round, r, plan.first(r));
                rc = 1;
            }
            // not vacuous: this rank's vectors are its own (they differ from its neighbour's, and from its own of the round before)
            if (r > 0 && memcmp(mine.data(), reps[(size_t) r - 1].mine[(size_t) round].data(), 12 * sizeof(float)) != 0) distinct++;
            if (round > 0 && memcmp(mine.data(), reps[(size_t) r].mine[(size_t) round - 1].data(), mine.size() * sizeof(float)) != 0) distinct++;
        }
    }
    if (rc == 0 && world > 1 && distinct == 0) { fprintf(stderr, "every rank holds the same vectors: the check shows nothing\n"); rc = 1; }
    if (rc == 0) printf("comm_ranks ok: %d ranks, %d channels, %d rounds gathered (sinks", world, plan.total(), rounds);
    if (rc == 0) { for (int s : plan.sinks) printf(" %d", s); printf("; dst %s)\n", plan.dst.c_str()); }
    return rc;
}
