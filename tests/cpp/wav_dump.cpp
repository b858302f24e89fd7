// wav_dump in.wav out.f32 -- decodes with include/fx_wav.hpp; prints "rate channels bits float frames",
// writes the interleaved float32 samples.  Exit 3 + reason on stderr when the file is rejected.
#include <cstdio>
#include "fx_wav.hpp"

int main (int argc, char** argv)
{
    if (argc < 3) return 2;
    fx::WavData wav;
    std::string error;
    if (! fx::readWav (argv[1], wav, error)) { std::fprintf (stderr, "%s\n", error.c_str()); return 3; }
    std::FILE* f = std::fopen (argv[2], "wb");
    if (f == nullptr) return 2;
    std::fwrite (wav.interleaved.data(), sizeof (float), wav.interleaved.size(), f);
    std::fclose (f);
    int hops = 0;
    const std::vector<float> h = fx::hopsOfChannel (wav, wav.numChannels - 1, 1024, hops);
    std::printf ("%d %d %d %d %zu %d %zu\n", wav.sampleRate, wav.numChannels, wav.bitsPerSample, wav.isFloat ? 1 : 0, wav.numFrames(), hops, h.size());
    return 0;
}
