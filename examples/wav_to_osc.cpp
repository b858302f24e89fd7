// wav_to_osc -- BASELINE configs[0]: one mono channel of a WAV file, windowSize-sample frames with 50 %
// overlap through the analysers, smoothed AudioFeatures out as OSC feature messages.
//
// The reference does this with AudioFilePlayer -> AudioDataCollector -> the two analyser threads ->
// OSCFeatureAnalysisOutput's 60 Hz timer (ref Source/AudioFilePlayer.h:41-61, AudioDataCollector.h:36-94,
// RealTimeAnalyser.h:141-234, OSCFeatureAnalysisOutput.h:84-113).  Here the file is decoded by
// include/fx_wav.hpp, analysed on the GPU through include/fx_realtime.hpp (C ABI underneath) and the
// datagrams are written to a UDP target and/or a dump file (each record: u32 little-endian length + bytes).
//
//   wav_to_osc in.wav [--window 1024] [--channel 0] [--gain 1.0] [--address /Audio/A0]
//                     [--target 127.0.0.1:9000] [--dump out.bin] [--rate 0] [--batch 64] [--device 0] [--pcm16-direct | --pcm24-direct]
//                     [--device-block n]
//
// --device-block n: the file is played to the analysers the way an audio device would deliver it -- in blocks of n samples (441, 480, 512,
// anything) through fx::AudioDataCollector::audioDeviceIOCallback (ref AudioDataCollector.h:36-70), which analyses the hops as they complete
// (fx_push_samples).  The same datagrams, byte for byte, as with whole hops.
//
// --pcm16-direct / --pcm24-direct (16- / 24-bit PCM files): the samples go to the GPU as the file holds them (FX_SAMPLE_S16 / _S24, two /
// three bytes each) and are widened to v / 2^15 / v / 2^23 in the kernels' load stage -- the same datagrams, byte for byte, as with the decoded floats.
//
// --rate 0 (default) emits one message per hop.  --rate 60 emits what the reference's timer would read if
// the file played in real time: at t = k/60 s, the smoothed values after the last hop completed by t.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fx_realtime.hpp"
#include "fx_wav.hpp"

int main (int argc, char** argv)
{
    std::string path, address = "/Audio/A0", target, dump;
    int window = 1024, channel = 0, batch = 64, device = 0, deviceBlock = 0;
    double rate = 0.0;
    float gain = 1.0f;
    bool pcm16Direct = false, pcm24Direct = false;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        auto next = [&] () -> const char* { if (i + 1 >= argc) { std::fprintf (stderr, "%s needs a value\n", a.c_str()); std::exit (2); } return argv[++i]; };
        if      (a == "--window")  window = std::atoi (next());
        else if (a == "--channel") channel = std::atoi (next());
        else if (a == "--gain")    gain = (float) std::atof (next());
        else if (a == "--address") address = next();
        else if (a == "--target")  target = next();
        else if (a == "--dump")    dump = next();
        else if (a == "--rate")    rate = std::atof (next());
        else if (a == "--batch")   batch = std::atoi (next());
        else if (a == "--device")  device = std::atoi (next());
        else if (a == "--device-block") deviceBlock = std::atoi (next());
        else if (a == "--pcm16-direct") pcm16Direct = true;
        else if (a == "--pcm24-direct") pcm24Direct = true;
        else if (a[0] != '-')      path = a;
        else { std::fprintf (stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (path.empty()) { std::fprintf (stderr, "usage: wav_to_osc in.wav [--window N] [--channel c] [--gain g] [--address a] [--target ip[:port]] [--dump file] [--rate hz] [--batch hops]\n"); return 2; }

    fx::WavData wav;
    std::string error;
    if (! fx::readWav (path, wav, error)) { std::fprintf (stderr, "%s\n", error.c_str()); return 1; }
    if (channel < 0 || channel >= wav.numChannels) { std::fprintf (stderr, "channel %d not in file (%d channels)\n", channel, wav.numChannels); return 1; }
    if (batch < 1) batch = 1;

    int numHops = 0;
    const std::vector<float> hops = fx::hopsOfChannel (wav, channel, window, numHops);
    const std::size_t hop = (std::size_t) window / 2;
    std::vector<std::int16_t> hops16;
    if (pcm16Direct)
    {
        int n16 = 0;
        hops16 = fx::hopsOfChannelPCM16 (wav, channel, window, n16);
        if (wav.pcm16.empty() || n16 != numHops) { std::fprintf (stderr, "--pcm16-direct needs a 16-bit PCM file (this one: %d-bit%s)\n", wav.bitsPerSample, wav.isFloat ? " float" : ""); return 1; }
    }
    std::vector<unsigned char> hops24;
    if (pcm24Direct)
    {
        int n24 = 0;
        hops24 = fx::hopsOfChannelPCM24 (wav, channel, window, n24);
        if (wav.pcm24.empty() || n24 != numHops) { std::fprintf (stderr, "--pcm24-direct needs a 24-bit PCM file (this one: %d-bit%s)\n", wav.bitsPerSample, wav.isFloat ? " float" : ""); return 1; }
    }

    std::FILE* out = nullptr;
    if (! dump.empty() && (out = std::fopen (dump.c_str(), "wb")) == nullptr) { std::fprintf (stderr, "cannot write %s\n", dump.c_str()); return 1; }
    fx::OSCFeatureSender sender;
    if (! target.empty() && ! sender.connectToAddress (target)) { std::fprintf (stderr, "bad target %s\n", target.c_str()); return 1; }

    long sent = 0;
    auto emit = [&] (const float* smoothed12)
    {
        const std::string msg = fx::OSCFeatureMessage (address, smoothed12);
        if (out != nullptr)
        {
            const unsigned n = (unsigned) msg.size();
            const unsigned char len[4] = { (unsigned char) n, (unsigned char) (n >> 8), (unsigned char) (n >> 16), (unsigned char) (n >> 24) };
            std::fwrite (len, 1, 4, out);
            std::fwrite (msg.data(), 1, msg.size(), out);
        }
        if (! target.empty()) sender.send (address, smoothed12);
        ++sent;
    };

    try
    {
        fx::RealTimeBatchAnalyser analyser (1, window, (double) wav.sampleRate, device);
        analyser.setGain (gain);
        std::vector<float> raw ((std::size_t) batch * FX_NUM_FEATURES), smoothed ((std::size_t) batch * FX_NUM_FEATURES);
        long tick = 1;                                               // next timer tick (k / rate seconds)
        fx::AudioDataCollector collector (analyser);
        const std::vector<float> mono = deviceBlock > 0 ? wav.channel (channel) : std::vector<float>();
        const std::vector<std::int16_t> mono16 = (deviceBlock > 0 && pcm16Direct) ? fx::samplesOfChannelPCM16 (wav, channel) : std::vector<std::int16_t>();
        const std::vector<unsigned char> mono24 = (deviceBlock > 0 && pcm24Direct) ? fx::samplesOfChannelPCM24 (wav, channel) : std::vector<unsigned char>();
        std::size_t played = 0;                                      // --device-block: samples handed to the collector so far
        for (int done = 0; done < numHops; )
        {
            int n = numHops - done < batch ? numHops - done : batch;
            const float* values = smoothed.data();
            if (deviceBlock > 0)
            {
                // one device callback: the next deviceBlock samples of the file (the last block is what is left)
                if (played >= mono.size()) break;
                const int len = (int) (mono.size() - played < (std::size_t) deviceBlock ? mono.size() - played : (std::size_t) deviceBlock);
                if (pcm16Direct) n = collector.pushBlock (mono16.data() + played, len, FX_SAMPLE_S16);
                else if (pcm24Direct) n = collector.pushBlock (mono24.data() + 3 * played, len, FX_SAMPLE_S24);
                else { const float* one[1] = { mono.data() + played }; n = collector.audioDeviceIOCallback (one, 1, len); }
                played += (std::size_t) len;
                values = collector.smoothed();
            }
            else if (pcm16Direct) analyser.pushHopsPCM16 (hops16.data() + (std::size_t) done * hop, n, raw.data(), smoothed.data());
            else if (pcm24Direct) analyser.pushHopsPCM24 (hops24.data() + (std::size_t) done * hop * 3, n, raw.data(), smoothed.data());
            else                  analyser.pushHops (hops.data() + (std::size_t) done * hop, n, raw.data(), smoothed.data());
            for (int t = 0; t < n; ++t)
            {
                const float* v = values + (std::size_t) t * FX_NUM_FEATURES;
                if (rate <= 0.0) { emit (v); continue; }
                // hop (done + t) is complete at sample (done + t + 1) * hop; it is what every tick in
                // [that time, completion of the next hop) reads
                const double from = (double) (done + t + 1) * (double) hop / wav.sampleRate;
                const double to   = (double) (done + t + 2) * (double) hop / wav.sampleRate;
                while ((double) tick / rate < from) ++tick;          // ticks before the first result read nothing
                while ((double) tick / rate < to) { emit (v); ++tick; }
            }
            done += n;
        }
    }
    catch (const fx::Error& e)
    {
        std::fprintf (stderr, "analysis failed: %s\n", e.what());
        if (out != nullptr) std::fclose (out);
        return 1;
    }
    if (out != nullptr) std::fclose (out);
    std::printf ("%s: %d Hz, %d channel(s), %d-bit%s; %d hops of %zu samples; %ld OSC messages (%s)\n", path.c_str(), wav.sampleRate,
                 wav.numChannels, wav.bitsPerSample, wav.isFloat ? " float" : "", numHops, hop, sent, address.c_str());
    return 0;
}
