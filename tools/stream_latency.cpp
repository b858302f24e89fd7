// Per-hop cost of the streaming ring (fx_stream_*) from C++, without an interpreter in the loop: BASELINE configs[4] shape
// (1 channel, 4096-pt windows, fp16 samples, one 2048-sample hop per call).
//   g++ -O2 -std=c++14 -I include tools/stream_latency.cpp -o stream_latency -L feature-extractor_amd/lib -lfx_hip -Wl,-rpath,$PWD/feature-extractor_amd/lib
//   ./stream_latency [window] [channels] [hops_per_call] [calls] [tone|ramp] [default|lowlat] [f16|s16|f32]
// (built by __graft_entry__.build() into tools/_bin/; bench.py's `streaming_hop` extra runs it as a child and reads the JSON line)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fx.h"

#define CHECK(x) do { fx_status s_ = (x); if (s_ != FX_OK) { fprintf(stderr, "%s: %s\n", #x, fx_last_error()); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 4096, C = argc > 2 ? atoi(argv[2]) : 1, H = argc > 3 ? atoi(argv[3]) : 1, calls = argc > 4 ? atoi(argv[4]) : 4000;
    const bool lowlat = argc > 6 && !strcmp(argv[6], "lowlat");          // FX_LOW_LATENCY: the pair family (windows >= 2048)
    const int fmt = argc > 7 && !strcmp(argv[7], "s16") ? FX_SAMPLE_S16 : (argc > 7 && !strcmp(argv[7], "f32") ? FX_SAMPLE_F32 : FX_SAMPLE_F16);
    fx_context* ctx = nullptr;
    CHECK(fx_create(&ctx, 0, C, N, 48000.0, lowlat ? FX_LOW_LATENCY : 0u));
    fx_stream* st = nullptr;
    CHECK(fx_stream_create(ctx, H, 3, fmt, &st));
    const size_t esz = fmt == FX_SAMPLE_F32 ? 4 : 2;
    const size_t hop_bytes = (size_t) C * H * (N / 2) * esz;
    // the bench's synthetic signal (a tone with two harmonics and a little noise, 55 * 2^(c/12) Hz per channel
    // starting at A3 here), `calls` hops of it, as fp16; argv[5] = "ramp" gives the round-1 test pattern instead
    // (positive values around 0.1: a DC-like input, the lag search's longest path)
    const bool ramp = argc > 5 && !strcmp(argv[5], "ramp");
    const int H2 = N / 2, nhops = 64;
    std::vector<std::vector<unsigned short>> hops(nhops, std::vector<unsigned short>(hop_bytes / 2));      // (f32: two shorts a sample)
    {
        auto to_half = [](float f) -> unsigned short {          // round to nearest even, normal range only
            unsigned u; memcpy(&u, &f, 4);
            const unsigned sign = (u >> 16) & 0x8000u; const int e = (int) ((u >> 23) & 0xff) - 127 + 15; unsigned m = u & 0x7fffffu;
            if (e <= 0) return (unsigned short) sign;
            unsigned h = sign | ((unsigned) e << 10) | (m >> 13);
            if ((m & 0x1fffu) > 0x1000u || ((m & 0x1fffu) == 0x1000u && (h & 1u))) h++;
            return (unsigned short) h;
        };
        unsigned long long rng = 0x5EEDull;
        for (int k = 0; k < nhops; k++)
            for (int c = 0; c < C; c++)
                for (int h = 0; h < H; h++)
                    for (int i = 0; i < H2; i++) {
                        const size_t at = ((size_t) c * H + h) * H2 + i;
                        float v;
                        if (ramp) v = 0.09375f + (float) ((at * 37) % 0x400) * (1.0f / 16384.0f);          // positive values around 0.1
                        else {
                            const double f = 220.0 * pow(2.0, (c % 36) / 12.0), ph = 2.0 * 3.14159265358979323846 * f * ((double) (k * H + h) * H2 + i) / 48000.0;
                            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                            const double u = ((double) (rng >> 40) / 16777216.0 - 0.5) * 0.1;
                            v = (float) (0.4 * sin(ph) + 0.2 * sin(2 * ph) + 0.1 * sin(3 * ph) + u);
                        }
                        if (fmt == FX_SAMPLE_F16) hops[k][at] = to_half(v);
                        else if (fmt == FX_SAMPLE_S16) hops[k][at] = (unsigned short) (short) lrintf(v * 32767.0f);
                        else memcpy(&hops[k][2 * at], &v, 4);
                    }
    }
    std::vector<float> sm((size_t) C * H * 12);
    double result_us[2] = {0.0, 0.0};
    for (int mode = 0; mode < 2; mode++) {
        // mode 0: one hop in flight (submit, then wait for it: the round trip); mode 1: up to three in flight (throughput)
        const int depth = mode == 0 ? 1 : 3;
        for (int warm = 0; warm < 2; warm++) {
            const int n = warm == 0 ? 200 : calls;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; i++) {
                if (fx_stream_in_flight(st) == depth) CHECK(fx_stream_collect(st, nullptr, sm.data()));
                void* slot = nullptr;
                CHECK(fx_stream_acquire(st, &slot));
                memcpy(slot, hops[i % nhops].data(), hop_bytes);
                CHECK(fx_stream_submit(st));
            }
            while (fx_stream_in_flight(st)) CHECK(fx_stream_collect(st, nullptr, sm.data()));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
            if (warm) result_us[mode] = us;
            if (warm) printf("%s: %.1f us per call of %d hop(s) x %d channel(s), %d-pt  (%.0f hops/s per channel = %.0fx real time at 48 kHz)\n",
                             mode == 0 ? "round trip  " : "three in flight", us, H, C, N, 1e6 * H / us, 1e6 * H / us * (N / 2) / 48000.0);
        }
    }
    printf("JSON {\"round_trip_us\": %.2f, \"three_in_flight_us\": %.2f, \"window\": %d, \"channels\": %d, \"hops_per_call\": %d, \"calls\": %d, "
           "\"family\": \"%s\", \"samples\": \"%s\", \"signal\": \"%s\"}\n", result_us[0], result_us[1], N, C, H, calls, lowlat ? "low_latency" : "default",
           fmt == FX_SAMPLE_F16 ? "f16" : (fmt == FX_SAMPLE_S16 ? "s16" : "f32"), ramp ? "ramp" : "tone");
    fx_stream_destroy(st);
    fx_destroy(ctx);
    return 0;
}
