// live_soak.cpp -- the whole live path under a clock, end to end, for minutes:
//   an "audio device" thread delivers a block of every channel every block/48000 s (480 samples = 10 ms by default)
//     -> fx::LiveAnalyser::pushBlock (a copy into the FIFO; ref AudioDataCollector.h:36-70)
//     -> worker: fx_push_samples straight from the page-locked FIFO slot -> GPU (blocks fed to the one-frame kernels) -> vectors   (ref :72-94, RealTimeAnalyser.h:141-234)
//     -> fx_get_osc_datagrams (messages formed on the GPU) -> fx::OSCBatchSender, 60 Hz timer, sendmmsg (ref OSCFeatureAnalysisOutput.h:84-136)
//     -> a local fx_osc_receiver that counts datagrams and keeps each channel's newest message.
// Reports: blocks dropped at the FIFO, ring / analysis errors, sender drops, late ticks, datagrams sent and received, block-arrival ->
// publication latency (p50 / p99 / max), the worker's busy share of real time, resident memory at the start and the end; at the end every
// channel's newest received datagram is compared with fx_osc_encode of the vector the analyser last published.
//
//   live_soak [channels=8192] [window=1024] [block=480] [seconds=60] [sender_threads=4] [gso=1] [dump=path]
// dump=path writes {int32 channels, window, block, blocks, pool_blocks; float32 smoothed[channels][12]} for tests/test_gpu_soak.py, which
// rebuilds the input (integer arithmetic below) and holds a sample of channels to the oracle.
// Build: g++ -std=c++14 -O2 -I include tools/live_soak.cpp -L feature-extractor_amd/lib -lfx_hip -Wl,-rpath,$PWD/feature-extractor_amd/lib -pthread
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fx_realtime.hpp"

namespace {
// The input, in integer arithmetic so that a test can rebuild it bit for bit: sample i of channel c is v / 32768 with
// v = triangle wave of the channel's own pitch / 2 + 12 bits of hashed noise.
inline int sample_value(unsigned c, unsigned i)
{
    const unsigned step = 200u + 37u * (c % 97u);
    const int phase = (int) ((i * step) & 0xFFFFu);
    const int tri = (phase < 32768 ? phase : 65535 - phase) - 16384;            // [-16384, 16383]
    unsigned h = c * 2654435761u + i * 40503u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return tri / 2 + (int) (h >> 20) - 2048;
}
long resident_kb()
{
    long size = 0, rss = 0;
    if (FILE* f = std::fopen("/proc/self/statm", "r")) { if (std::fscanf(f, "%ld %ld", &size, &rss) != 2) rss = 0; std::fclose(f); }
    return rss * (sysconf(_SC_PAGESIZE) / 1024);
}
} // namespace

int main(int argc, char** argv)
{
    int channels = 8192, window = 1024, block = 480, seconds = 60, senderThreads = 4, gso = 1, withOsc = 1;
    std::string dump;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const size_t eq = a.find('=');
        const std::string k = a.substr(0, eq), v = eq == std::string::npos ? "" : a.substr(eq + 1);
        if (k == "channels") channels = std::atoi(v.c_str()); else if (k == "window") window = std::atoi(v.c_str());
        else if (k == "block") block = std::atoi(v.c_str()); else if (k == "seconds") seconds = std::atoi(v.c_str());
        else if (k == "sender_threads") senderThreads = std::atoi(v.c_str()); else if (k == "gso") gso = std::atoi(v.c_str());
        else if (k == "dump") dump = v;
        else if (k == "osc") withOsc = std::atoi(v.c_str());            // 0: analysis only (diagnostics: nothing is published or sent)
        else { std::fprintf(stderr, "live_soak: unknown argument %s\n", a.c_str()); return 2; }
    }
    const double sampleRate = 48000.0;
    const int poolBlocks = 16;                                   // the device replays 16 blocks' worth of the signal
    try {
        std::vector<float> pool((size_t) poolBlocks * channels * block);
        for (int k = 0; k < poolBlocks; k++)
            for (int c = 0; c < channels; c++)
                for (int j = 0; j < block; j++)
                    pool[((size_t) k * channels + c) * block + j] = (float) sample_value((unsigned) c, (unsigned) (k * block + j)) / 32768.0f;

        fx::RealTimeBatchAnalyser analyser(channels, window, sampleRate);
        fx_osc_receiver* rx = nullptr;
        fx::check(fx_osc_receiver_create(&rx, "127.0.0.1:0", senderThreads, "/Audio/A", channels, 0u));
        fx::OSCBatchSender sender("127.0.0.1:" + std::to_string(fx_osc_receiver_port(rx)), "", senderThreads, gso != 0);
        fx::LiveAnalyser live(analyser, block, 8);
        if (withOsc) live.attachOSCSender(&sender, "/Audio/A", 0);
        sender.startTimerHz(60);

        // two seconds of warm-up (allocations, first touches, clocks), then the measured run
        const auto period = std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(block / sampleRate));
        long long pushed = 0, late = 0;
        long rss0 = 0;
        fx::LiveAnalyser::Stats warm {};
        fx_osc_sender_stats sw {};
        long long rxw = 0;
        auto next = std::chrono::steady_clock::now() + period;
        const long long warmBlocks = (long long) (2.0 * sampleRate / block), total = warmBlocks + (long long) (seconds * sampleRate / block);
        double lastNote = 0.0;
        const auto t0 = std::chrono::steady_clock::now();
        for (long long b = 0; b < total; b++) {
            std::this_thread::sleep_until(next);
            if (std::chrono::steady_clock::now() > next + period) late++;           // the "device" itself was late (this process was not scheduled)
            next += period;
            if (b == warmBlocks) { live.drain(); warm = live.getStats(); sw = sender.getStats(); fx_osc_receiver_get_stats(rx, &rxw, nullptr, nullptr); rss0 = resident_kb(); late = 0; }
            live.pushBlock(pool.data() + (size_t) (b % poolBlocks) * channels * block, block);
            pushed++;
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (t - lastNote > 20.0) { lastNote = t; std::printf("... %.0f s, %lld blocks, resident %ld KB\n", t, pushed, resident_kb()); std::fflush(stdout); }
        }
        live.drain();
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - 2.0;
        sender.stopTimer();
        const fx::LiveAnalyser::Stats st = live.getStats();
        const fx_osc_sender_stats ss = sender.getStats();
        const long rss1 = resident_kb();
        // one last tick with the final publication, then every channel's newest datagram against the vector it was formed from
        const long long lastSent = sender.sendNow();
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        long long rxn = 0, rxbad = 0;
        fx_osc_receiver_get_stats(rx, &rxn, nullptr, &rxbad);
        const std::vector<float> latest = live.latestSmoothed();
        long long wrong = 0;
        for (int c = 0; c < channels; c++) {
            unsigned char got[160], want[160];
            int n = 0;
            fx::check(fx_osc_receiver_last(rx, c, got, (int) sizeof got, &n));
            const int m = fx_osc_encode(("/Audio/A" + std::to_string(c)).c_str(), latest.data() + (size_t) c * 12, want, (int) sizeof want);
            if (n != m || std::memcmp(got, want, (size_t) m) != 0) wrong++;
        }
        const long long blocks = st.blocksIn - warm.blocksIn, frames = st.framesPerChannel - warm.framesPerChannel;
        const long long sent = ss.datagrams - sw.datagrams, ticks = ss.ticks - sw.ticks;
        std::printf("live_soak: %d channels x %d-pt, %d-sample blocks at 48 kHz, %d s after 2 s of warm-up; %d sender threads, segmented sends %s\n", channels, window, block, seconds, senderThreads, gso ? "on" : "off");
        std::printf("  blocks delivered %lld (device late %lld times), dropped at the FIFO %lld, analysis / ring errors %lld%s%s\n", blocks, late, st.blocksDropped - warm.blocksDropped, st.errors,
                    st.errors ? ": " : "", st.errors ? live.lastError().c_str() : "");
        std::printf("  frames per channel %lld = %.4g frames/s over all channels; worker busy %.1f %% of real time (real-time factor %.1f)\n", frames, (double) frames * channels / wall,
                    100.0 * (st.workerBusySeconds - warm.workerBusySeconds) / wall, wall / (st.workerBusySeconds - warm.workerBusySeconds));
        std::printf("  block arrival -> publication: p50 %.3f ms, p99 %.3f ms, max %.3f ms (over the whole run, warm-up included)\n", st.latencyMsP50, st.latencyMsP99, st.latencyMsMax);
        std::printf("  sender: %lld ticks (%.1f Hz), %lld late, %lld datagrams handed to the kernel (%.4g /s), %lld dropped, longest tick %.2f ms\n", ticks, ticks / wall, ss.late_ticks - sw.late_ticks, sent,
                    sent / wall, ss.dropped - sw.dropped, ss.max_tick_ms);
        std::printf("  receiver: %lld datagrams (%.4f of those sent), %lld malformed; after the last tick %lld of %d channels hold a datagram that is not fx_osc_encode of the last published vector\n",
                    rxn - rxw - lastSent, (double) (rxn - rxw - lastSent) / (double) (sent > 0 ? sent : 1), rxbad, wrong, channels);
        std::printf("  resident memory %ld KB after warm-up, %ld KB at the end (%+ld KB)\n", rss0, rss1, rss1 - rss0);
        const bool ok = st.errors == 0 && st.blocksDropped == warm.blocksDropped && ss.dropped == sw.dropped && (wrong == 0 || ! withOsc) && rxbad == 0 && frames > 0;
        // one machine-readable line (bench.py's `live_soak` extra reads it)
        std::printf("{\"live_soak\": {\"channels\": %d, \"window\": %d, \"block\": %d, \"seconds\": %d, \"blocks\": %lld, \"dropped_at_fifo\": %lld, \"errors\": %lld, "
                    "\"frames_per_s\": %.6g, \"worker_busy_share\": %.4f, \"real_time_factor\": %.3f, \"latency_ms_p50\": %.4f, \"latency_ms_p99\": %.4f, \"latency_ms_max\": %.4f, "
                    "\"sender_ticks\": %lld, \"sender_late_ticks\": %lld, \"datagrams_per_s\": %.6g, \"sender_dropped\": %lld, \"received_share\": %.6f, \"malformed\": %lld, "
                    "\"channels_with_wrong_datagram\": %lld, \"rss_kb_after_warmup\": %ld, \"rss_kb_end\": %ld, \"ok\": %s}}\n",
                    channels, window, block, seconds, blocks, st.blocksDropped - warm.blocksDropped, st.errors, (double) frames * channels / wall,
                    (st.workerBusySeconds - warm.workerBusySeconds) / wall, wall / (st.workerBusySeconds - warm.workerBusySeconds), st.latencyMsP50, st.latencyMsP99, st.latencyMsMax,
                    ticks, ss.late_ticks - sw.late_ticks, sent / wall, ss.dropped - sw.dropped, (double) (rxn - rxw - lastSent) / (double) (sent > 0 ? sent : 1), rxbad, wrong, rss0, rss1,
                    ok ? "true" : "false");
        std::printf("live_soak: %s\n", ok ? "ok" : "FAILED");
        if (! dump.empty()) {
            if (FILE* f = std::fopen(dump.c_str(), "wb")) {
                const int head[5] = { channels, window, block, (int) pushed, poolBlocks };
                std::fwrite(head, sizeof head, 1, f);
                std::fwrite(latest.data(), sizeof(float), latest.size(), f);
                std::fclose(f);
            }
        }
        live.stop();
        fx_osc_receiver_destroy(rx);
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "live_soak: %s\n", e.what());
        return 1;
    }
}
