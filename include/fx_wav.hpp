// fx_wav.hpp -- RIFF/WAVE reader for the file-input case (BASELINE configs[0]).
//
// In the reference a file reaches the analysers through JUCE: AudioFormatManager::createReaderFor ->
// AudioFormatReaderSource -> AudioTransportSource -> AudioSourcePlayer (ref Source/AudioFilePlayer.h:41-61),
// and AudioDataCollector copies ONE device channel, `channelToCollect`, into its ring
// (ref Source/AudioDataCollector.h:42-64).  JUCE is out of scope; this header is the
// JUCE-free stand-in for that ingest: decode the file to float, pick a channel, cut it into hops.
//
// Sample conversion follows JUCE's WavAudioFormat reader: integer PCM is widened to a left-justified
// int32 and scaled by 1.0f / 0x7fffffff -- as a float that divisor is 2^31, so an n-bit sample v becomes
// v / 2^(n-1) exactly (8-bit is unsigned, offset 128).  32-bit int samples round to float first.  IEEE float
// files are passed through (64-bit narrowed).  This restates JUCE 4.2 behaviour from memory, like the
// rest of the JUCE boundary (parity unpinned).
//
// No resampling: AudioTransportSource would resample a file whose rate differs from the device's; here
// the caller analyses at the file's own rate (fx_set_sample_rate / the constructor argument).
#ifndef FX_WAV_HPP
#define FX_WAV_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace fx
{
struct WavData
{
    int    sampleRate = 0;
    int    numChannels = 0;
    int    bitsPerSample = 0;
    bool   isFloat = false;
    std::vector<float> interleaved;     // [frame][channel]
    std::vector<std::int16_t> pcm16;    // 16-bit PCM files only: the samples as the file holds them, [frame][channel] -- what
                                        // FX_SAMPLE_S16 ingests (the kernels make v / 2^15 of them, exactly as `interleaved` holds)
    std::vector<unsigned char> pcm24;   // 24-bit PCM files only: the data chunk as it is, three bytes per sample (FX_SAMPLE_S24: v / 2^23)

    std::size_t numFrames() const { return numChannels > 0 ? interleaved.size() / (std::size_t) numChannels : 0; }

    // what AudioDataCollector sees of this file: one channel (ref AudioDataCollector.h:53 channelToCollect)
    std::vector<float> channel (int c) const
    {
        std::vector<float> out (numFrames());
        for (std::size_t i = 0; i < out.size(); ++i) out[i] = interleaved[i * (std::size_t) numChannels + (std::size_t) c];
        return out;
    }
};

namespace wavdetail
{
    inline std::uint32_t u32 (const unsigned char* p) { return (std::uint32_t) p[0] | ((std::uint32_t) p[1] << 8) | ((std::uint32_t) p[2] << 16) | ((std::uint32_t) p[3] << 24); }
    inline std::uint16_t u16 (const unsigned char* p) { return (std::uint16_t) (p[0] | (p[1] << 8)); }
}

// Returns true and fills `out`, or false with a reason in `error`.
inline bool readWav (const std::string& path, WavData& out, std::string& error)
{
    using namespace wavdetail;
    std::FILE* f = std::fopen (path.c_str(), "rb");
    if (f == nullptr) { error = "cannot open " + path; return false; }
    std::vector<unsigned char> bytes;
    {
        unsigned char chunk[65536];
        std::size_t n;
        while ((n = std::fread (chunk, 1, sizeof chunk, f)) > 0) bytes.insert (bytes.end(), chunk, chunk + n);
        std::fclose (f);
    }
    if (bytes.size() < 12 || std::memcmp (bytes.data(), "RIFF", 4) != 0 || std::memcmp (bytes.data() + 8, "WAVE", 4) != 0)
    { error = "not a RIFF/WAVE file"; return false; }

    int format = 0, blockAlign = 0;
    bool haveFmt = false;
    const unsigned char* data = nullptr;
    std::size_t dataBytes = 0;
    std::size_t pos = 12;
    while (pos + 8 <= bytes.size())
    {
        const unsigned char* h = bytes.data() + pos;
        std::size_t len = u32 (h + 4);
        const std::size_t body = pos + 8;
        if (std::memcmp (h, "fmt ", 4) == 0)
        {
            if (len < 16 || body + 16 > bytes.size()) { error = "truncated fmt chunk"; return false; }
            const unsigned char* p = bytes.data() + body;
            format = u16 (p);
            out.numChannels = u16 (p + 2);
            out.sampleRate = (int) u32 (p + 4);
            blockAlign = u16 (p + 12);
            out.bitsPerSample = u16 (p + 14);
            if (format == 0xFFFE)                                   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
            {
                if (len < 40 || body + 40 > bytes.size()) { error = "truncated extensible fmt chunk"; return false; }
                format = u16 (p + 24);
            }
            haveFmt = true;
        }
        else if (std::memcmp (h, "data", 4) == 0)
        {
            if (body + len > bytes.size()) len = bytes.size() - body;  // streamed files leave the length short; read what is there
            data = bytes.data() + body;
            dataBytes = len;
            break;
        }
        pos = body + len + (len & 1);                               // chunks are word aligned
    }
    if (! haveFmt)        { error = "no fmt chunk";  return false; }
    if (data == nullptr)  { error = "no data chunk"; return false; }
    if (out.numChannels < 1) { error = "no channels"; return false; }
    out.isFloat = format == 3;
    const int bits = out.bitsPerSample;
    if (! ((format == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (format == 3 && (bits == 32 || bits == 64))))
    { error = "unsupported sample format (format tag " + std::to_string (format) + ", " + std::to_string (bits) + " bits)"; return false; }
    const int bytesPerSample = bits / 8;
    if (blockAlign != bytesPerSample * out.numChannels) { error = "inconsistent block alignment"; return false; }

    const std::size_t total = dataBytes / (std::size_t) blockAlign * (std::size_t) out.numChannels;
    out.interleaved.resize (total);
    out.pcm16.clear();
    out.pcm24.clear();
    if (format == 1 && bits == 16) out.pcm16.resize (total);
    if (format == 1 && bits == 24) out.pcm24.assign (data, data + total * 3);
    const float scale = 1.0f / (float) 0x7fffffff;                  // == 2^-31
    for (std::size_t i = 0; i < total; ++i)
    {
        const unsigned char* p = data + i * (std::size_t) bytesPerSample;
        float v;
        if (format == 3)
        {
            if (bits == 32) { std::memcpy (&v, p, 4); }
            else            { double d; std::memcpy (&d, p, 8); v = (float) d; }
        }
        else
        {
            std::int32_t wide;
            if (bits == 8)       wide = (std::int32_t) (((std::uint32_t) p[0] - 128u) << 24);
            else if (bits == 16) { wide = (std::int32_t) ((std::uint32_t) u16 (p) << 16); out.pcm16[i] = (std::int16_t) u16 (p); }
            else if (bits == 24) wide = (std::int32_t) (((std::uint32_t) p[0] << 8) | ((std::uint32_t) p[1] << 16) | ((std::uint32_t) p[2] << 24));
            else                 wide = (std::int32_t) u32 (p);
            v = (float) wide * scale;
        }
        out.interleaved[i] = v;
    }
    return true;
}

// The hop stream RealTimeAudioDataOverlapper::getNextBuffer would pull from the collector while the file
// plays once (ref RealTimeAudioAnalysis.h:205-219): consecutive blocks of windowSize/2 samples; a trailing
// partial block is dropped (the overlapper only ever reads whole hops).  Gain is applied by the analyser.
inline std::vector<float> hopsOfChannel (const WavData& wav, int channel, int windowSize, int& numHops)
{
    const std::size_t hop = (std::size_t) windowSize / 2;
    std::vector<float> mono = wav.channel (channel);
    numHops = (int) (mono.size() / hop);
    mono.resize ((std::size_t) numHops * hop);
    return mono;
}
// The same hop stream as 16-bit PCM, untouched (16-bit files only; empty otherwise): two bytes per sample across PCIe.
inline std::vector<std::int16_t> hopsOfChannelPCM16 (const WavData& wav, int channel, int windowSize, int& numHops)
{
    std::vector<std::int16_t> mono;
    numHops = 0;
    if (wav.pcm16.empty()) return mono;
    const std::size_t hop = (std::size_t) windowSize / 2, frames = wav.numFrames();
    numHops = (int) (frames / hop);
    mono.resize ((std::size_t) numHops * hop);
    for (std::size_t i = 0; i < mono.size(); ++i) mono[i] = wav.pcm16[i * (std::size_t) wav.numChannels + (std::size_t) channel];
    return mono;
}
// every sample of one channel of a 16-bit file, as the file holds it (what an audio device would deliver block by block)
inline std::vector<std::int16_t> samplesOfChannelPCM16 (const WavData& wav, int channel)
{
    std::vector<std::int16_t> mono;
    if (wav.pcm16.empty()) return mono;
    mono.resize (wav.numFrames());
    for (std::size_t i = 0; i < mono.size(); ++i) mono[i] = wav.pcm16[i * (std::size_t) wav.numChannels + (std::size_t) channel];
    return mono;
}
// every sample of one channel of a 24-bit file, packed in three bytes as the file holds it
inline std::vector<unsigned char> samplesOfChannelPCM24 (const WavData& wav, int channel)
{
    std::vector<unsigned char> mono;
    if (wav.pcm24.empty()) return mono;
    mono.resize (wav.numFrames() * 3);
    for (std::size_t i = 0; i < wav.numFrames(); ++i)
        for (int b = 0; b < 3; ++b) mono[3 * i + (std::size_t) b] = wav.pcm24[(i * (std::size_t) wav.numChannels + (std::size_t) channel) * 3 + (std::size_t) b];
    return mono;
}
// ... and as packed 24-bit PCM (24-bit files only): numHops * windowSize/2 * 3 bytes of the chosen channel.
inline std::vector<unsigned char> hopsOfChannelPCM24 (const WavData& wav, int channel, int windowSize, int& numHops)
{
    std::vector<unsigned char> mono;
    numHops = 0;
    if (wav.pcm24.empty()) return mono;
    const std::size_t hop = (std::size_t) windowSize / 2, frames = wav.numFrames();
    numHops = (int) (frames / hop);
    mono.resize ((std::size_t) numHops * hop * 3);
    for (std::size_t i = 0; i < (std::size_t) numHops * hop; ++i)
        for (int b = 0; b < 3; ++b) mono[3 * i + (std::size_t) b] = wav.pcm24[(i * (std::size_t) wav.numChannels + (std::size_t) channel) * 3 + (std::size_t) b];
    return mono;
}
} // namespace fx

#endif // FX_WAV_HPP
