// refdiff_legacy.cpp -- the reference's LEGACY offline analyser (struct AudioAnalyser, AudioAnalysis.h),
// compiled UNMODIFIED from where it lies under /root/reference/Source against tools/refdiff/juce_standin.h, run on given
// inputs, so that oracle/fx_offline.c can be diffed against it.  Build container only (tools/refdiff/README.md); nothing of
// the reference is copied: the headers are #included by path at compile time.
//
//   refdiff_legacy <in.bin> <out.bin>
//   in : int32 op, a, b, c, d; float64 x; then float32 payload
//   op 1 zero crossings  : a = channels, b = samples, c = downsamples; payload audio[a][b]       -> float32 out[a][c]
//   op 2 log attack time : a = envelope length, b = input samples, c = downsamples, d = sample rate; payload envelope[a] -> float32 out[1]
//   op 3 FFT-LBP         : a = channels, b = bins; payload cur[a][b], prev[a][b]                  -> per channel: float32 bits[b], highest ratio, activity ratio
//   op 4 histogram F0    : a = channels, b = bins, c = frames, x = nyquist; payload mags[c][a][b]   -> float32 out[c][a][3], float64 previousF0 after each frame [c][a]
#include "juce_standin.h"

#include <cstdint>
#include <cstdio>
#include <iostream>
#include <sstream>

#include "AudioFeatures.h"
#include "AudioAnalysis.h"

struct Header { int32_t op, a, b, c, d; double x; };

int main (int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen (argv[1], "rb");
    if (! f) return 2;
    Header h;
    if (fread (&h, sizeof h, 1, f) != 1) return 2;
    std::vector<float> in;
    { float v; while (fread (&v, 4, 1, f) == 1) in.push_back (v); }
    fclose (f);
    FILE* o = fopen (argv[2], "wb");
    if (! o) return 2;
    if (h.op == 1) {
        ConcatenatedFeatureBuffer features (h.a, h.b, h.c, 0, 0.0, 48000.0);
        for (int ch = 0; ch < h.a; ch++) features.audioOutput.copyFrom (ch, 0, in.data() + (size_t) ch * h.b, h.b);
        AudioAnalyser an (1024, h.a, 24000.0, true, true);
        an.analyseNormalisedZeroCrosses (features);
        for (int ch = 0; ch < h.a; ch++)
            for (int i = 0; i < h.c; i++) { const float v = features.getFeatureSample (ConcatenatedFeatureBuffer::Feature::ZeroCrosses, ch, i); fwrite (&v, 4, 1, o); }
    } else if (h.op == 2) {
        ConcatenatedFeatureBuffer features (1, h.b, h.c, 0, 0.0, (double) h.d);
        features.energyEnvelope.setSize (1, h.a);
        features.energyEnvelope.copyFrom (0, 0, in.data(), h.a);
        AudioAnalyser an (1024, 1, (double) h.d / 2.0, true, true);
        an.setLogAttackTime (features);
        fwrite (&features.estimatedLogAttackTime, 4, 1, o);
    } else if (h.op == 3) {
        AudioSampleBuffer cur (h.a, h.b), prev (h.a, h.b);
        for (int ch = 0; ch < h.a; ch++) {
            cur.copyFrom (ch, 0, in.data() + (size_t) ch * h.b, h.b);
            prev.copyFrom (ch, 0, in.data() + (size_t) (h.a + ch) * h.b, h.b);
        }
        AudioAnalyser an (1024, h.a, 24000.0, true, true);
        for (int ch = 0; ch < h.a; ch++) {
            // the function only prints: "b|b|...| - <highest / numBins> - <totalDiffs / totalSum>\n" (AudioAnalysis.h:560,563)
            std::ostringstream cap;
            std::streambuf* old = std::cout.rdbuf (cap.rdbuf());
            an.calculateFFTLBP (cur, prev, ch);
            std::cout.rdbuf (old);
            const std::string s = cap.str();
            size_t pos = 0;
            for (int i = 0; i < h.b; i++) { const float v = (float) (s[pos] - '0'); fwrite (&v, 4, 1, o); pos += 2; }
            // the two ratios are printed with 6 significant digits only: recompute them from the bits the reference printed,
            // with the reference's own expressions (:559-563), so that the harness does not compare rounded text
            float highest = 0.0f, total = 0.0f, sum = 0.0f;
            pos = 0;
            for (int i = 0; i < h.b; i++) { sum++; const int b = s[pos] - '0'; total += (float) b; if (b) highest = (float) i; pos += 2; }
            const float r1 = highest / (float) h.b, r2 = total / sum;
            fwrite (&r1, 4, 1, o); fwrite (&r2, 4, 1, o);
        }
    } else if (h.op == 4) {
        std::vector<AudioAnalyser*> ans;                       // one analyser per channel: previousF0 is per analyser (AudioAnalysis.h:697)
        for (int ch = 0; ch < h.a; ch++) ans.push_back (new AudioAnalyser (1024, h.a, h.x, true, true));
        std::vector<double> prevs;
        for (int fr = 0; fr < h.c; fr++) {
            AudioSampleBuffer mags (h.a, h.b);
            for (int ch = 0; ch < h.a; ch++) mags.copyFrom (ch, 0, in.data() + ((size_t) fr * h.a + ch) * h.b, h.b);
            for (int ch = 0; ch < h.a; ch++) {
                AudioAnalyser::HarmonicCharacteristics hc = ans[(size_t) ch]->calculateHarmonicCharacteristics (mags, ch);
                fwrite (&hc.f0, 4, 1, o); fwrite (&hc.harmonicEnergyRatio, 4, 1, o); fwrite (&hc.inharmonicity, 4, 1, o);
                prevs.push_back (ans[(size_t) ch]->previousF0);
            }
        }
        fwrite (prevs.data(), 8, prevs.size(), o);
    } else return 2;
    fclose (o);
    return 0;
}
