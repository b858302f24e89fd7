// Multi-GPU gather through the C ABI, from C++ with no torch / Python in the process:
// one process per GPU (fork), the communicator id travels through a pipe, every rank analyses its own
// channel shard and the latest smoothed vectors are gathered to rank 0 over RCCL (fx_gather_smoothed).
// Rank 0 checks its gathered table against what each rank holds locally (sent back through pipes).
//   comm_ranks <world>     world <= number of visible GPUs; exit code 77 = not enough GPUs (skip)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

#include "fx.h"

static void fill_hops(std::vector<float>& h, int first_channel, int channels, int hops, int half)
{
    // any deterministic per-channel signal: a tone whose pitch depends on the global channel id
    for (int c = 0; c < channels; c++) {
        const double f = 110.0 * std::pow(2.0, ((first_channel + c) % 24) / 12.0);
        for (int n = 0; n < hops * half; n++)
            h[(size_t) c * hops * half + n] = (float) (0.5 * std::sin(2.0 * 3.14159265358979323846 * f * n / 48000.0) + 0.01 * ((n * 7 + c * 13) % 17 - 8) / 8.0);
    }
}

static bool read_all(int fd, void* p, size_t n) { char* b = (char*) p; while (n) { ssize_t r = read(fd, b, n); if (r <= 0) return false; b += r; n -= (size_t) r; } return true; }
static bool write_all(int fd, const void* p, size_t n) { const char* b = (const char*) p; while (n) { ssize_t r = write(fd, b, n); if (r <= 0) return false; b += r; n -= (size_t) r; } return true; }

#define CHECK(call) do { fx_status s_ = (call); if (s_ != FX_OK) { fprintf(stderr, "rank %d: %s -> %d: %s\n", rank, #call, s_, fx_last_error()); return 1; } } while (0)

static int run_rank(int rank, int world, int id_fd_in, int result_fd_out, const std::vector<int>& result_fd_in)
{
    const int N = 1024, hops = 14;
    const int channels = 5 + rank;                      // ragged shards on purpose
    unsigned char id[FX_COMM_ID_BYTES];
    fx_context* ctx = nullptr;
    CHECK(fx_create(&ctx, rank, channels, N, 48000.0, 0));
    if (!read_all(id_fd_in, id, sizeof id)) { fprintf(stderr, "rank %d: no id\n", rank); return 1; }
    CHECK(fx_comm_create(ctx, rank, world, id, FX_COMM_ID_BYTES));
    int total = 0; std::vector<int> first((size_t) world);
    CHECK(fx_comm_layout(ctx, &total, first.data()));
    int first_channel = 0; for (int r = 0; r < rank; r++) first_channel += 5 + r;
    if (first[(size_t) rank] != first_channel) { fprintf(stderr, "rank %d: layout says first channel %d, expected %d\n", rank, first[(size_t) rank], first_channel); return 1; }
    std::vector<float> h((size_t) channels * hops * (N / 2));
    fill_hops(h, first_channel, channels, hops, N / 2);
    std::vector<float> table((size_t) total * 12, -1.0f);
    // three rounds: the gather of round i overlaps the analysis of round i+1; the last one is checked
    for (int round = 0; round < 3; round++) {
        CHECK(fx_push_hops(ctx, h.data(), hops, FX_SAMPLE_F32, FX_MEM_HOST, nullptr, nullptr));
        CHECK(fx_gather_smoothed(ctx, 0, rank == 0 ? table.data() : nullptr, FX_MEM_HOST));
    }
    CHECK(fx_comm_sync(ctx));
    std::vector<float> mine((size_t) channels * 12);
    CHECK(fx_get_smoothed(ctx, mine.data(), FX_MEM_HOST));
    int rc = 0;
    if (rank != 0) {
        if (!write_all(result_fd_out, mine.data(), mine.size() * sizeof(float))) rc = 1;
    } else {
        for (int r = 0; r < world && rc == 0; r++) {
            const int cr = 5 + r;
            std::vector<float> theirs((size_t) cr * 12);
            if (r == 0) theirs = mine;
            else if (!read_all(result_fd_in[(size_t) r], theirs.data(), theirs.size() * sizeof(float))) { fprintf(stderr, "no result from rank %d\n", r); rc = 1; break; }
            if (memcmp(theirs.data(), table.data() + (size_t) first[(size_t) r] * 12, theirs.size() * sizeof(float)) != 0) {
                fprintf(stderr, "gathered block of rank %d differs from that rank's own features\n", r);
                rc = 1;
            }
        }
        if (rc == 0) printf("comm_ranks ok: %d ranks, %d channels gathered over RCCL\n", world, total);
    }
    CHECK(fx_comm_destroy(ctx));
    CHECK(fx_destroy(ctx));
    return rc;
}

int main(int argc, char** argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 2;
    if (world < 1 || world > 8) return 2;
    // the parent makes no GPU call; the device count comes from a short-lived child
    {
        int pf[2]; if (pipe(pf)) return 2;
        pid_t p = fork();
        if (p == 0) {
            fx_context* probe = nullptr; int n = 0;
            for (; n < 8; n++) { if (fx_create(&probe, n, 1, 1024, 48000.0, 0) != FX_OK) break; fx_destroy(probe); }
            write_all(pf[1], &n, sizeof n); _exit(0);
        }
        int n = 0; close(pf[1]); read_all(pf[0], &n, sizeof n); close(pf[0]); waitpid(p, nullptr, 0);
        if (n < world) { fprintf(stderr, "comm_ranks: %d GPU(s) visible, %d needed -- skipped\n", n, world); return 77; }
    }
    std::vector<int> id_r((size_t) world), id_w((size_t) world), res_r((size_t) world, -1), res_w((size_t) world, -1);
    for (int r = 0; r < world; r++) { int pf[2]; if (pipe(pf)) return 2; id_r[(size_t) r] = pf[0]; id_w[(size_t) r] = pf[1]; }
    for (int r = 1; r < world; r++) { int pf[2]; if (pipe(pf)) return 2; res_r[(size_t) r] = pf[0]; res_w[(size_t) r] = pf[1]; }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; r++) {
        pid_t p = fork();
        if (p == 0) {
            if (r == 0) {                                        // rank 0 makes the id and hands it to everyone (itself included)
                unsigned char id[FX_COMM_ID_BYTES];
                if (fx_comm_unique_id(id, sizeof id) != FX_OK) { fprintf(stderr, "fx_comm_unique_id: %s\n", fx_last_error()); _exit(1); }
                for (int q = 0; q < world; q++) write_all(id_w[(size_t) q], id, sizeof id);
            }
            _exit(run_rank(r, world, id_r[(size_t) r], res_w[(size_t) r], res_r));
        }
        kids.push_back(p);
    }
    int rc = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1; }
    return rc;
}
