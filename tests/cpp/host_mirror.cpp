// Exercises include/fx_realtime.hpp the way a C++ caller of the reference would use it.
//   host_mirror            : CPU-only checks (AudioFeatures mirror, OSC message, error without GPU)
//   host_mirror --gpu      : additionally runs hops through the GPU and replays the raw values
//                            through the AudioFeatures mirror; smoothed outputs must be identical.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "fx_realtime.hpp"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

int main (int argc, char** argv)
{
    const bool gpu = argc > 1 && std::strcmp (argv[1], "--gpu") == 0;

    AudioFeatures f;
    EXPECT (std::isnan (f.getValue (AudioFeatures::enRMS)));              // 0/0 before the first insert, as in the reference
    for (int i = 1; i <= 12; i++) f.updateFeature (AudioFeatures::enRMS, (float) i);
    EXPECT (f.getValue (AudioFeatures::enRMS) == (3.f + 4 + 5 + 6 + 7 + 8 + 9 + 10 + 11 + 12) / 10.f);
    f.updateFeature (AudioFeatures::enFlux, 0.25f);
    f.updateFeature (AudioFeatures::enFlux, 0.5f);
    EXPECT (f.getValue (AudioFeatures::enFlux) == 0.5f);                  // history length 1
    EXPECT (std::strcmp (AudioFeatures::getFeatureName (AudioFeatures::enHarmonicEnergyRatio), "H.E.R") == 0);
    EXPECT ((int) AudioFeatures::numFeatures == 12);

    float v[12];
    for (int i = 0; i < 12; i++) v[i] = (float) i;
    const std::string msg = fx::OSCFeatureMessage ("/Audio/A0", v);
    EXPECT (msg.size() == 76);
    EXPECT (std::memcmp (msg.data(), "/Audio/A0\0\0\0,ffffffffffff\0\0\0", 28) == 0);
    EXPECT ((unsigned char) msg[28 + 4 * 4] == 0x41 && (unsigned char) msg[28 + 4 * 4 + 1] == 0x00);   // 5th float is slope = 8.0f

    {
        // loopback: the sender's datagram is what OSCFeatureMessage builds
        int rx = ::socket (AF_INET, SOCK_DGRAM, 0);
        sockaddr_in a {};
        a.sin_family = AF_INET; a.sin_port = 0; a.sin_addr.s_addr = htonl (INADDR_LOOPBACK);
        EXPECT (::bind (rx, reinterpret_cast<sockaddr*> (&a), sizeof a) == 0);
        socklen_t len = sizeof a;
        ::getsockname (rx, reinterpret_cast<sockaddr*> (&a), &len);
        fx::OSCFeatureSender sender;
        EXPECT (sender.connectToAddress ("127.0.0.1:" + std::to_string ((int) ntohs (a.sin_port))));
        EXPECT (sender.send ("/Audio/A0", v));
        char buf[256];
        timeval tv { 2, 0 };
        ::setsockopt (rx, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        const ssize_t n = ::recv (rx, buf, sizeof buf, 0);
        EXPECT (n == 76 && std::memcmp (buf, msg.data(), 76) == 0);
        ::close (rx);
        fx::OSCFeatureSender dflt;
        EXPECT (dflt.connectToAddress ("127.0.0.1"));          // default port 9000
    }
    {
        // fx::OSCFeatureAnalysisOutput (ref OSCFeatureAnalysisOutput.h:23-145): connecting starts a 60 Hz timer that sends the latest snapshot;
        // nothing before the first frame; every datagram is the twelve values of ONE updateFeatures call
        int rx = ::socket (AF_INET, SOCK_DGRAM, 0);
        sockaddr_in a {};
        a.sin_family = AF_INET; a.sin_port = 0; a.sin_addr.s_addr = htonl (INADDR_LOOPBACK);
        EXPECT (::bind (rx, reinterpret_cast<sockaddr*> (&a), sizeof a) == 0);
        socklen_t len = sizeof a;
        ::getsockname (rx, reinterpret_cast<sockaddr*> (&a), &len);
        timeval tv { 0, 300000 };
        ::setsockopt (rx, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        char buf[256];
        {
            fx::OSCFeatureAnalysisOutput out ("127.0.0.1:" + std::to_string ((int) ntohs (a.sin_port)), "/Audio/A3");
            EXPECT (::recv (rx, buf, sizeof buf, 0) < 0);                        // no frame yet: the timer has nothing to read
            float w[12];
            for (int i = 0; i < 12; i++) w[i] = 0.5f + (float) i;
            out.updateFeatures (w);
            const std::string want = fx::OSCFeatureMessage ("/Audio/A3", w);
            int got = 0;
            for (int k = 0; k < 6; k++) { const ssize_t n = ::recv (rx, buf, sizeof buf, 0); if (n == (ssize_t) want.size() && std::memcmp (buf, want.data(), want.size()) == 0) got++; }
            EXPECT (got == 6);                                                    // 60 Hz: six datagrams well inside the time-outs
            for (int i = 0; i < 12; i++) w[i] = -1.0f - (float) i;
            out.updateFeatures (w);
            const std::string next = fx::OSCFeatureMessage ("/Audio/A3", w);
            bool seen = false;
            for (int k = 0; k < 6 && ! seen; k++) { const ssize_t n = ::recv (rx, buf, sizeof buf, 0); seen = n == (ssize_t) next.size() && std::memcmp (buf, next.data(), next.size()) == 0; }
            EXPECT (seen && out.getNumMessagesSent() >= 7);
        }                                                                         // (the destructor joins the timer thread)
        ::close (rx);
    }

    {
        // fx::OSCBatchSender: 1500 tracks to a primary and a secondary fx_osc_receiver, three ticks by hand, then the 60 Hz timer;
        // every message is fx::OSCFeatureMessage of its track ("/Audio/A999" is 76 bytes, "/Audio/A1000" 80)
        const int tracks = 1500;
        fx_osc_receiver* rx[2] = { nullptr, nullptr };
        for (int k = 0; k < 2; k++) EXPECT (fx_osc_receiver_create (&rx[k], "127.0.0.1:0", 1, "/Audio/A", tracks, 0u) == FX_OK);
        std::vector<float> v12 ((size_t) tracks * 12);
        for (size_t i = 0; i < v12.size(); i++) v12[i] = 0.001f * (float) i - 3.0f;
        {
            fx::OSCBatchSender out ("127.0.0.1:" + std::to_string (fx_osc_receiver_port (rx[0])), "127.0.0.1:" + std::to_string (fx_osc_receiver_port (rx[1])), 2, true);
            EXPECT (out.sendNow() == 0);                                          // nothing published yet
            out.updateFeatures ("/Audio/A", 0, v12.data(), tracks);
            long long sent = 0;
            for (int k = 0; k < 3; k++) sent += out.sendNow();
            EXPECT (sent == 3ll * 2 * tracks);
            out.startTimerHz (60);
            std::this_thread::sleep_for (std::chrono::milliseconds (250));
            out.stopTimer();
            const fx_osc_sender_stats st = out.getStats();
            EXPECT (st.ticks >= 3 + 8 && st.ticks <= 3 + 20 && st.dropped == 0 && st.datagrams == (st.ticks - 1) * 2 * tracks);
            std::this_thread::sleep_for (std::chrono::milliseconds (100));
            for (int k = 0; k < 2; k++)
            {
                long long n = 0, bad = 0;
                EXPECT (fx_osc_receiver_get_stats (rx[k], &n, nullptr, &bad) == FX_OK);
                EXPECT (n == (st.ticks - 1) * tracks && bad == 0);
                for (int c : { 0, 9, 10, 999, 1000, tracks - 1 })
                {
                    unsigned char got[160]; int len = 0;
                    EXPECT (fx_osc_receiver_last (rx[k], c, got, (int) sizeof got, &len) == FX_OK);
                    const std::string want = fx::OSCFeatureMessage ("/Audio/A" + std::to_string (c), v12.data() + (size_t) c * 12);
                    EXPECT (len == (int) want.size() && std::memcmp (got, want.data(), want.size()) == 0);
                    EXPECT (len == (c < 1000 ? 76 : 80));
                }
            }
        }
        for (int k = 0; k < 2; k++) fx_osc_receiver_destroy (rx[k]);
    }

    if (! gpu)
    {
        bool threw = false;
        try { fx::RealTimeBatchAnalyser a (2, 1000); } catch (const fx::Error& e) { threw = e.code == FX_ERR_INVALID_ARGUMENT; }
        EXPECT (threw);
        std::printf (failures ? "host_mirror: %d failure(s)\n" : "host_mirror: ok\n", failures);
        return failures ? 1 : 0;
    }

    const int C = 3, T = 30, N = 1024, H = N / 2;
    std::vector<float> hops ((size_t) C * T * H), raw ((size_t) C * T * 12), sm ((size_t) C * T * 12);
    unsigned s = 12345;
    for (int c = 0; c < C; c++)
        for (int i = 0; i < T * H; i++)
        {
            s = s * 1664525u + 1013904223u;
            const float gate = ((i / H) % 7 < 4) ? 1.0f : 0.0f;
            hops[(size_t) c * T * H + i] = gate * (0.5f * std::sin (0.02f * (c + 1) * i) + 0.05f * ((s >> 8) / 16777216.0f - 0.5f));
        }
    fx::RealTimeBatchAnalyser an (C, N);
    an.setOnsetDetectionType (fx::enAmplitude);
    an.pushHops (hops.data(), T, raw.data(), sm.data());
    // isolated-order replay is not what the default does; replay the default order by hand:
    // spectral writes (RMS, 6 spectral slots, onset) then harmonic writes (RMS, F0, HER, OER, inharm)
    for (int c = 0; c < C; c++)
    {
        AudioFeatures af;
        for (int t = 0; t < T; t++)
        {
            const float* r = &raw[((size_t) c * T + t) * 12];
            af.updateFeature (AudioFeatures::enRMS, r[FX_RMS]);
            af.updateFeature (AudioFeatures::enCentroid, r[FX_CENTROID]);
            af.updateFeature (AudioFeatures::enFlatness, r[FX_FLATNESS]);
            af.updateFeature (AudioFeatures::enLER, r[FX_LER]);
            af.updateFeature (AudioFeatures::enSpread, r[FX_SPREAD]);
            af.updateFeature (AudioFeatures::enFlux, r[FX_FLUX]);
            af.updateFeature (AudioFeatures::enSlope, r[FX_SLOPE]);
            af.updateFeature (AudioFeatures::enOnset, r[FX_ONSET]);
            af.updateFeature (AudioFeatures::enRMS, r[FX_RMS]);
            af.updateFeature (AudioFeatures::enF0, r[FX_F0]);
            af.updateFeature (AudioFeatures::enHarmonicEnergyRatio, r[FX_HER]);
            af.updateFeature (AudioFeatures::enOddEvenHarmonicRatio, r[FX_OER]);
            af.updateFeature (AudioFeatures::enInharmonicity, r[FX_INHARM]);
            for (int k = 0; k < 12; k++)
            {
                const float got = sm[((size_t) c * T + t) * 12 + k], want = af.getValue ((AudioFeatures::eAudioFeature) k);
                if (! (got == want || (std::isnan (got) && std::isnan (want))))
                {
                    std::printf ("smoothed mismatch c=%d t=%d slot=%d gpu=%g mirror=%g\n", c, t, k, got, want);
                    failures++;
                }
            }
        }
        std::vector<float> latest = an.getValues (c);
        for (int k = 0; k < 12; k++) EXPECT (latest[k] == sm[((size_t) c * T + T - 1) * 12 + k] || std::isnan (latest[k]));
    }
    // the tuning struct round-trips, and frames on wavefront pairs (2048-pt) give the same onsets as the default kernels
    {
        const int C2 = 2, T2 = 12, N2 = 2048, H2 = N2 / 2;
        std::vector<float> h2 ((size_t) C2 * T2 * H2), ra ((size_t) C2 * T2 * 12), sa (ra.size()), rb (ra.size()), sb (ra.size());
        for (size_t i = 0; i < h2.size(); i++) h2[i] = ((i / H2) % 5 < 3 ? 0.4f : 0.0f) * std::sin (0.03f * (float) (i % 4096));
        fx::RealTimeBatchAnalyser one (C2, N2), two (C2, N2);
        fx_tuning t = two.getTuning();
        EXPECT (t.waves_per_frame == 0 && t.frames_per_unit == -1);
        t.waves_per_frame = 2;
        two.setTuning (t);
        EXPECT (two.getTuning().waves_per_frame == 2);
        one.pushHops (h2.data(), T2, ra.data(), sa.data());
        two.pushHops (h2.data(), T2, rb.data(), sb.data());
        for (size_t i = 0; i < ra.size(); i += 12) { EXPECT (ra[i] == rb[i]); EXPECT (ra[i + FX_F0] == rb[i + FX_F0]); }
    }
    // round 4: the low-latency family as a constructor flag, 16- and 24-bit PCM hops, the ring (fx::HopRing) one hop per call and in
    // batches filled by the library's thread pool -- each must reproduce pushHops on the decoded floats bit for bit
    {
        const int C4 = 2, T4 = 24, N4 = 2048, H4 = N4 / 2;
        std::vector<std::int16_t> pcm ((size_t) C4 * T4 * H4);
        std::vector<unsigned char> pcm24 (pcm.size() * 3);
        std::vector<float> dec (pcm.size()), dec24 (pcm.size());
        unsigned r = 777;
        for (size_t i = 0; i < pcm.size(); i++)
        {
            r = r * 1664525u + 1013904223u;
            const float x = ((i / H4) % 6 < 4 ? 0.6f : 0.01f) * std::sin (0.011f * (float) (i % 5000)) + 0.02f * ((r >> 8) / 16777216.0f - 0.5f);
            pcm[i] = (std::int16_t) std::lrint (x * 32767.0f);
            dec[i] = (float) pcm[i] / 32768.0f;
            const int v24 = (int) std::lrint ((double) x * 8388607.0);
            pcm24[3 * i] = (unsigned char) v24; pcm24[3 * i + 1] = (unsigned char) (v24 >> 8); pcm24[3 * i + 2] = (unsigned char) (v24 >> 16);
            dec24[i] = (float) v24 / 8388608.0f;
        }
        std::vector<float> wr ((size_t) C4 * T4 * 12), ws (wr.size()), gr (wr.size()), gs (wr.size());
        auto sameBits = [&] (const std::vector<float>& a, const std::vector<float>& b) { return std::memcmp (a.data(), b.data(), a.size() * sizeof (float)) == 0; };
        for (int low = 0; low < 2; low++)
        {
            const unsigned flags = low ? FX_LOW_LATENCY : 0u;
            fx::RealTimeBatchAnalyser ref (C4, N4, 48000.0, 0, flags), a16 (C4, N4, 48000.0, 0, flags), a24 (C4, N4, 48000.0, 0, flags), ring1 (C4, N4, 48000.0, 0, flags),
                                      ring8 (C4, N4, 48000.0, 0, flags);
            ref.pushHops (dec.data(), T4, wr.data(), ws.data());
            a16.pushHopsPCM16 (pcm.data(), T4, gr.data(), gs.data());
            EXPECT (sameBits (gr, wr) && sameBits (gs, ws));
            {   // 24-bit: its own decoded floats
                std::vector<float> w24r (wr.size()), w24s (wr.size());
                fx::RealTimeBatchAnalyser ref24 (C4, N4, 48000.0, 0, flags);
                ref24.pushHops (dec24.data(), T4, w24r.data(), w24s.data());
                a24.pushHopsPCM24 (pcm24.data(), T4, gr.data(), gs.data());
                EXPECT (sameBits (gr, w24r) && sameBits (gs, w24s));
            }
            // one hop per call through the ring, up to three in flight: the hop is [channels][1][H4] -- gather it from the [C][T][H] layout
            {
                fx::HopRing ring (ring1, 1, 3, FX_SAMPLE_S16);
                std::vector<float> one ((size_t) C4 * 12), onesm ((size_t) C4 * 12);
                int collected = 0;
                auto take = [&] {
                    ring.collect (one.data(), onesm.data());
                    for (int c = 0; c < C4; c++)
                        for (int k = 0; k < 12; k++) { gr[((size_t) c * T4 + collected) * 12 + k] = one[(size_t) c * 12 + k]; gs[((size_t) c * T4 + collected) * 12 + k] = onesm[(size_t) c * 12 + k]; }
                    collected++;
                };
                for (int t = 0; t < T4; t++)
                {
                    if (ring.inFlight() == 3) take();
                    std::int16_t* slot = static_cast<std::int16_t*> (ring.nextSlot());
                    for (int c = 0; c < C4; c++) std::memcpy (slot + (size_t) c * H4, pcm.data() + ((size_t) c * T4 + t) * H4, (size_t) H4 * sizeof (std::int16_t));
                    ring.submit();
                }
                while (ring.inFlight()) take();
                EXPECT (collected == T4 && sameBits (gr, wr) && sameBits (gs, ws));
            }
            // batches of 8 hops pushed from ordinary memory with two fill threads
            {
                const int B = 8;
                fx::HopRing ring (ring8, B, 2, FX_SAMPLE_S16);
                std::vector<std::int16_t> batch ((size_t) C4 * B * H4);
                std::vector<float> br (ring.valuesPerBatch()), bs (ring.valuesPerBatch());
                int done = 0;
                auto take = [&] {
                    ring.collect (br.data(), bs.data());
                    for (int c = 0; c < C4; c++)
                        std::memcpy (&gr[((size_t) c * T4 + (size_t) done * B) * 12], &br[(size_t) c * B * 12], (size_t) B * 12 * sizeof (float)),
                        std::memcpy (&gs[((size_t) c * T4 + (size_t) done * B) * 12], &bs[(size_t) c * B * 12], (size_t) B * 12 * sizeof (float));
                    done++;
                };
                for (int b = 0; b < T4 / B; b++)
                {
                    if (ring.inFlight() == 2) take();
                    for (int c = 0; c < C4; c++) std::memcpy (&batch[(size_t) c * B * H4], pcm.data() + ((size_t) c * T4 + (size_t) b * B) * H4, (size_t) B * H4 * sizeof (std::int16_t));
                    ring.push (batch.data(), 2);
                }
                while (ring.inFlight()) take();
                EXPECT (done == T4 / B && sameBits (gr, wr) && sameBits (gs, ws));
            }
        }
        // the family cannot change under a history
        fx::RealTimeBatchAnalyser fam (C4, N4);
        fam.pushHops (dec.data(), 2, gr.data(), gs.data());
        fx_tuning tf = fam.getTuning();
        tf.waves_per_frame = 2;
        bool refused = false;
        try { fam.setTuning (tf); } catch (const fx::Error& e) { refused = e.code == FX_ERR_INVALID_ARGUMENT; }
        EXPECT (refused && fam.getTuning().waves_per_frame == 0);
        fam.reset();
        fam.setTuning (tf);
        EXPECT (fam.getTuning().waves_per_frame == 2);
    }
    // round 5: fx::AudioDataCollector (ref AudioDataCollector.h:18-138) -- device blocks of any length for several channels through
    // audioDeviceIOCallback, the gain at read time, clearBuffer -- and the ring fed with sample blocks: the bits of pushHops on the same stream
    {
        const int C5 = 3, N5 = 1024, H5 = N5 / 2, T5 = 13;
        std::vector<float> stream ((size_t) C5 * T5 * H5);
        unsigned r = 4242;
        for (size_t i = 0; i < stream.size(); i++)
        {
            r = r * 1664525u + 1013904223u;
            stream[i] = 0.5f * std::sin (0.02f * (float) (i % 7001) * (1.0f + (float) (i / (T5 * H5)))) + 0.03f * ((r >> 8) / 16777216.0f - 0.5f);
        }
        std::vector<float> wr ((size_t) C5 * T5 * 12), ws (wr.size()), gr (wr.size(), -1.0f), gs (wr.size(), -1.0f);
        fx::RealTimeBatchAnalyser ref (C5, N5), blocks (C5, N5), ringed (C5, N5);
        ref.setGain (0.75f);
        ref.pushHops (stream.data(), T5, wr.data(), ws.data());
        auto sameBits = [&] (const std::vector<float>& a, const std::vector<float>& b) { return std::memcmp (a.data(), b.data(), a.size() * sizeof (float)) == 0; };
        {
            fx::AudioDataCollector collector (blocks);
            collector.setGain (0.75f);
            int notified = 0, frames = 0;
            collector.setNotifyAnalysisThreadCallback ([&] (int n) { notified += n; });
            const int lengths[] = { 480, 441, 1, 63, 512, 1000, 4097, 7 };
            size_t at = 0;
            for (int k = 0; at < (size_t) T5 * H5; k++)
            {
                size_t n = (size_t) lengths[k % 8];
                if (n > (size_t) T5 * H5 - at) n = (size_t) T5 * H5 - at;
                const float* channels[3] = { stream.data() + at, stream.data() + (size_t) T5 * H5 + at, stream.data() + 2 * (size_t) T5 * H5 + at };
                const int got = collector.audioDeviceIOCallback (channels, C5, (int) n);
                EXPECT (got == collector.getNumFrames());
                for (int c = 0; c < C5; c++)
                    for (int f = 0; f < got; f++)
                    {
                        std::memcpy (&gr[((size_t) c * T5 + frames + f) * 12], collector.raw() + ((size_t) c * got + f) * 12, 12 * sizeof (float));
                        std::memcpy (&gs[((size_t) c * T5 + frames + f) * 12], collector.smoothed() + ((size_t) c * got + f) * 12, 12 * sizeof (float));
                    }
                frames += got;
                at += n;
            }
            EXPECT (frames == T5 && notified == T5 && collector.getNumPendingSamples() == 0);
            EXPECT (sameBits (gr, wr) && sameBits (gs, ws));
            // clearBuffer: what is pending becomes zeros, the indices stay (ref :122)
            const float* head[3] = { stream.data(), stream.data() + (size_t) T5 * H5, stream.data() + 2 * (size_t) T5 * H5 };
            EXPECT (collector.audioDeviceIOCallback (head, C5, 100) == 0 && collector.getNumPendingSamples() == 100);
            collector.clearBuffer();
            EXPECT (collector.getNumPendingSamples() == 100);
            bool refused = false;
            try { blocks.pushHops (stream.data(), 1, gr.data(), gs.data()); } catch (const fx::Error& e) { refused = e.code == FX_ERR_INVALID_ARGUMENT; }
            EXPECT (refused);                       // whole hops would overtake the pending samples
        }
        {
            // the ring: blocks of 700 samples per channel into slots that hold up to two hops' worth
            fx::HopRing ring (ringed, 2, 3, FX_SAMPLE_F32);
            ringed.setGain (0.75f);
            std::fill (gr.begin(), gr.end(), -1.0f); std::fill (gs.begin(), gs.end(), -1.0f);
            std::vector<float> br ((size_t) C5 * 2 * 12), bs (br.size()), piece;
            int frames = 0;
            auto take = [&] {
                const int got = ring.collectSamples (br.data(), bs.data());
                for (int c = 0; c < C5; c++)
                    for (int f = 0; f < got; f++)
                    {
                        std::memcpy (&gr[((size_t) c * T5 + frames + f) * 12], &br[((size_t) c * got + f) * 12], 12 * sizeof (float));
                        std::memcpy (&gs[((size_t) c * T5 + frames + f) * 12], &bs[((size_t) c * got + f) * 12], 12 * sizeof (float));
                    }
                frames += got;
            };
            for (size_t at = 0; at < (size_t) T5 * H5; at += 700)
            {
                const size_t n = (size_t) T5 * H5 - at < 700 ? (size_t) T5 * H5 - at : 700;
                piece.resize ((size_t) C5 * n);
                for (int c = 0; c < C5; c++) std::memcpy (&piece[(size_t) c * n], stream.data() + (size_t) c * T5 * H5 + at, n * sizeof (float));
                if (ring.inFlight() == 3) take();
                ring.pushSamples (piece.data(), (int) n);
            }
            while (ring.inFlight()) take();
            EXPECT (frames == T5 && sameBits (gr, wr) && sameBits (gs, ws));
        }
    }
    // round 6: the live engine as INTEGRATION.md section 2 binds it -- the audio thread only copies (fx::LiveAnalyser::audioDeviceIOCallback), a
    // worker owns the GPU ring and writes each track's AudioFeatures exactly as the two run() loops did; the sink's messages are formed on the
    // GPU and go out through fx::OSCBatchSender.  481-sample device blocks against a 1024-point window: every getValue equals pushHops' smoothed
    // vectors of the same stream bit for bit, and each track's newest datagram is OSCFeatureMessage of its AudioFeatures.
    {
        const int C6 = 5, T6 = 30, N6 = 1024, H6 = N6 / 2, B6 = 481;
        std::vector<float> stream ((size_t) C6 * T6 * H6), wr ((size_t) C6 * T6 * 12), ws (wr.size());
        unsigned r6 = 4242;
        for (size_t i = 0; i < stream.size(); i++) { r6 = r6 * 1664525u + 1013904223u; stream[i] = 0.5f * std::sin (0.017f * (float) (i % 7001)) * ((i / H6) % 4 < 3 ? 1.0f : 0.0f) + 0.03f * ((r6 >> 8) / 16777216.0f - 0.5f); }
        fx::RealTimeBatchAnalyser ref (C6, N6), liveAn (C6, N6);
        ref.pushHops (stream.data(), T6, wr.data(), ws.data());
        fx_osc_receiver* rx = nullptr;
        EXPECT (fx_osc_receiver_create (&rx, "127.0.0.1:0", 1, "/Audio/A", C6, 0u) == FX_OK);
        std::vector<AudioFeatures> features ((size_t) C6);
        int framesSeen = 0, mismatches = 0;
        {
            fx::OSCBatchSender sender ("127.0.0.1:" + std::to_string (fx_osc_receiver_port (rx)));
            fx::LiveAnalyser live (liveAn, 512);
            live.attachOSCSender (&sender, "/Audio/A", 0);
            live.setFramesAnalysedCallback ([&] (int frames, const float* raw, const float* smoothed) {       // the worker thread: where the run() loops were
                for (int c = 0; c < C6; c++)
                    for (int k = 0; k < frames; k++)
                    {
                        const float* v = raw + ((size_t) c * frames + k) * 12;
                        AudioFeatures& f = features[(size_t) c];
                        f.updateFeature (AudioFeatures::enRMS, v[FX_RMS]);           f.updateFeature (AudioFeatures::enCentroid, v[FX_CENTROID]);
                        f.updateFeature (AudioFeatures::enFlatness, v[FX_FLATNESS]); f.updateFeature (AudioFeatures::enLER, v[FX_LER]);
                        f.updateFeature (AudioFeatures::enSpread, v[FX_SPREAD]);     f.updateFeature (AudioFeatures::enFlux, v[FX_FLUX]);
                        f.updateFeature (AudioFeatures::enSlope, v[FX_SLOPE]);       f.updateFeature (AudioFeatures::enOnset, v[FX_ONSET]);
                        f.updateFeature (AudioFeatures::enRMS, v[FX_RMS]);           f.updateFeature (AudioFeatures::enF0, v[FX_F0]);
                        f.updateFeature (AudioFeatures::enHarmonicEnergyRatio, v[FX_HER]); f.updateFeature (AudioFeatures::enOddEvenHarmonicRatio, v[FX_OER]);
                        f.updateFeature (AudioFeatures::enInharmonicity, v[FX_INHARM]);
                        for (int q = 0; q < 12; q++)
                        {
                            const float got = f.getValue ((AudioFeatures::eAudioFeature) q), want = ws[((size_t) c * T6 + framesSeen + k) * 12 + q], lib = smoothed[((size_t) c * frames + k) * 12 + q];
                            if (! ((got == want && lib == want) || (std::isnan (got) && std::isnan (want) && std::isnan (lib)))) mismatches++;
                        }
                    }
                framesSeen += frames;
            });
            std::vector<const float*> in ((size_t) C6);
            for (size_t at = 0; at < (size_t) T6 * H6; at += B6)                       // the "audio thread"
            {
                const size_t n = (size_t) T6 * H6 - at < (size_t) B6 ? (size_t) T6 * H6 - at : (size_t) B6;
                for (int c = 0; c < C6; c++) in[(size_t) c] = stream.data() + (size_t) c * T6 * H6 + at;
                while (! live.audioDeviceIOCallback (in.data(), C6, (int) n)) std::this_thread::sleep_for (std::chrono::microseconds (200));   // (a device would not wait: here no block may be lost)
            }
            live.drain();
            const fx::LiveAnalyser::Stats st = live.getStats();
            EXPECT (framesSeen == T6 && mismatches == 0 && st.errors == 0 && st.framesPerChannel == T6 && st.blocksAnalysed == st.blocksIn);
            EXPECT (sender.sendNow() == C6);
            std::this_thread::sleep_for (std::chrono::milliseconds (100));
            for (int c = 0; c < C6; c++)
            {
                float v[12];
                for (int q = 0; q < 12; q++) v[q] = features[(size_t) c].getValue ((AudioFeatures::eAudioFeature) q);
                const std::string want = fx::OSCFeatureMessage ("/Audio/A" + std::to_string (c), v);
                unsigned char got[160]; int len = 0;
                EXPECT (fx_osc_receiver_last (rx, c, got, (int) sizeof got, &len) == FX_OK && len == (int) want.size() && std::memcmp (got, want.data(), want.size()) == 0);
            }
        }
        fx_osc_receiver_destroy (rx);
        // the same engine fed the device's own 16-bit integers: the vectors of pushHopsPCM16 on that stream, bit for bit
        {
            std::vector<std::int16_t> pcm (stream.size());
            for (size_t i = 0; i < pcm.size(); i++) pcm[i] = (std::int16_t) std::lrint (stream[i] * 32767.0f);
            std::vector<float> w16r (wr.size()), w16s (wr.size()), got ((size_t) C6 * T6 * 12, -1.0f);
            fx::RealTimeBatchAnalyser ref16 (C6, N6), live16 (C6, N6);
            ref16.pushHopsPCM16 (pcm.data(), T6, w16r.data(), w16s.data());
            int seen = 0;
            {
                fx::LiveAnalyser live (live16, 512, 8, FX_SAMPLE_S16);
                live.setFramesAnalysedCallback ([&] (int frames, const float*, const float* smoothed) {
                    for (int c = 0; c < C6; c++) std::memcpy (&got[((size_t) c * T6 + seen) * 12], &smoothed[(size_t) c * frames * 12], sizeof (float) * 12 * (size_t) frames);
                    seen += frames;
                });
                std::vector<std::int16_t> block ((size_t) C6 * B6);
                for (size_t at = 0; at < (size_t) T6 * H6; at += B6)
                {
                    const size_t n = (size_t) T6 * H6 - at < (size_t) B6 ? (size_t) T6 * H6 - at : (size_t) B6;
                    for (int c = 0; c < C6; c++) std::memcpy (&block[(size_t) c * n], &pcm[(size_t) c * T6 * H6 + at], n * sizeof (std::int16_t));
                    while (! live.pushBlock (block.data(), (int) n)) std::this_thread::sleep_for (std::chrono::microseconds (200));
                }
                live.drain();
                const float* none[1] = { nullptr };
                EXPECT (! live.audioDeviceIOCallback (none, C6, 0));               // float callbacks are refused by an integer engine (and counted)
            }
            EXPECT (seen == T6 && std::memcmp (got.data(), w16s.data(), got.size() * sizeof (float)) == 0);
        }
    }
    // the legacy offline analyser's mirror (ref AudioAnalysis.h)
    {
        const int C3 = 2, S = 4000, B = 513;
        std::vector<float> audio ((size_t) C3 * S), mags ((size_t) C3 * B, 0.01f);
        for (size_t i = 0; i < audio.size(); i++) audio[i] = std::sin (0.5f * (float) i);
        for (int c = 0; c < C3; c++) for (int b = 25; b < B; b += 25) mags[(size_t) c * B + b] = 5.0f;
        fx::AudioAnalyser legacy (C3, 24000.0);
        const std::vector<float> zc = legacy.analyseNormalisedZeroCrosses (audio.data(), S, 4);
        EXPECT (zc.size() == 8 && zc[0] > 0.28f && zc[0] < 0.36f);                   // a sine of 0.5 rad per sample changes sign every ~6.3 samples: 2 / 6.3
        const std::vector<fx::AudioAnalyser::HarmonicCharacteristics> hc = legacy.calculateHarmonicCharacteristics (mags.data(), B);
        EXPECT (hc.size() == 2 && hc[0].f0 == hc[1].f0 && std::fabs (hc[0].f0 - 25.0f * 24000.0f / 513.0f) < 1.0f);   // the comb's spacing
        // round 4: full-spectrum characteristics (state kept: the second call's flux is 0), slope, auto-correlation peak
        const std::vector<fx::AudioAnalyser::SpectralCharacteristics> sc1 = legacy.calculateSpectralCharacteristics (mags.data(), B);
        const std::vector<fx::AudioAnalyser::SpectralCharacteristics> sc2 = legacy.calculateSpectralCharacteristics (mags.data(), B);
        EXPECT (sc1.size() == 2 && sc1[0].flux > 0.0f && sc2[0].flux == 0.0f && sc1[0].centroid == sc2[0].centroid && sc1[0].centroid > 0.4f && sc1[0].centroid < 0.6f);
        const std::vector<float> slope = legacy.calculateNormalisedSpectralSlope (mags.data(), B);
        EXPECT (slope.size() == 2 && slope[0] == slope[1] && std::isfinite (slope[0]));
        std::vector<float> items ((size_t) C3 * 64 * 2, 0.1f);
        items[2 * 9] = 3.0f; items[2 * 9 + 1] = 4.0f;                                 // channel 0: item 9 -> 25
        items[(size_t) 64 * 2 + 2 * 30] = -2.0f;                                      // channel 1: item 30 -> 4.01
        std::vector<int> peaks;
        const std::vector<double> freq = legacy.analyseAutoCorrelation (items.data(), 64, &peaks);
        EXPECT (peaks[0] == 9 && peaks[1] == 30 && items[2 * 9] == 25.0f && items[2 * 9 + 1] == 0.0f && freq[0] == 9 * (24000.0 / 64) + 24000.0 / 128);
    }
    std::printf (failures ? "host_mirror --gpu: %d failure(s)\n" : "host_mirror --gpu: ok\n", failures);
    return failures ? 1 : 0;
}
