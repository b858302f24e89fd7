// refdiff_driver.cpp -- single-steps the reference's OWN hot-path headers (compiled unmodified, from where they
// lie under /root/reference/Source, against tools/refdiff/juce_standin.h) over hop streams and writes what they
// produce, so that oracle/fx_oracle.c can be diffed against the reference's feature arithmetic.
// Build container only; see tools/refdiff/README.md.  Nothing of the reference is copied: the headers are
// #included by path at compile time.
//
//   refdiff_driver <in.bin> <out.bin>
//   in : int32 N, C, T, order, onset_type, onset_window; float32 onset_sensitivity, gain; float64 sample_rate;
//        float32 hops[C][T][N/2]
//   out: float32 raw[C][T][12], smoothed[C][T][12], lag[C][T]
// order: 0 spectral analyser first, 1 harmonic first (one shared AudioFeatures, as AnalyserTrackController
// builds them, ref AnalyserTrackController.h:20-21), 2 one AudioFeatures per analyser.
#include "juce_standin.h"

// the raw value of a hop is the newest entry of each slot's ValueHistory, which AudioFeatures keeps private
#define private public
#include "AudioDataCollector.h"
#include "RealTimeAudioAnalysis.h"
#include "PitchAnalyser.h"
#include "SpectralCharacteristics.h"
#include "HarmonicCharacteristics.h"
#include "RealTimeAnalyser.h"
#undef private

#include <cstdint>
#include <cstdio>

struct Header { int32_t N, C, T, order, onset_type, onset_window; float onset_sensitivity, gain; double sample_rate; };

static void feed (AudioDataCollector& c, const float* hop, int n)
{
    const float* in[1] = { hop };
    c.audioDeviceIOCallback (in, 1, nullptr, 0, n);
}

int main (int argc, char** argv)
{
    if (argc != 3) return 2;
    FILE* f = fopen (argv[1], "rb");
    if (! f) return 2;
    Header h;
    if (fread (&h, sizeof h, 1, f) != 1) return 2;
    const int half = h.N / 2;
    std::vector<float> hops ((size_t) h.C * h.T * half);
    if (fread (hops.data(), sizeof (float), hops.size(), f) != hops.size()) return 2;
    fclose (f);
    std::vector<float> raw ((size_t) h.C * h.T * 12), sm (raw.size()), lag ((size_t) h.C * h.T);

    for (int c = 0; c < h.C; c++)
    {
        // one AnalyserTrackController's analysis half (ref AnalyserTrackController.h:199-206)
        AudioDataCollector specCollector (0), harmCollector (0);
        specCollector.setExpectedSamplesPerBlock (half);
        harmCollector.setExpectedSamplesPerBlock (half);
        specCollector.setGain (h.gain);
        harmCollector.setGain (h.gain);
        AudioFeatures shared, harmOwn;
        AudioFeatures& specFeatures = shared;
        AudioFeatures& harmFeatures = h.order == 2 ? harmOwn : shared;
        RealTimeSpectralAnalyser spectral (specCollector, specFeatures, h.N, h.sample_rate);
        RealTimeHarmonicAnalyser harmonic (harmCollector, harmFeatures, h.N, h.sample_rate);
        spectral.setOnsetDetectionType ((OnsetDetector::eOnsetDetectionType) h.onset_type);
        spectral.setOnsetDetectionSensitivity (h.onset_sensitivity);
        if (h.onset_window > 0) spectral.setOnsetWindowLength (h.onset_window);

        for (int t = 0; t < h.T; t++)
        {
            const float* hop = hops.data() + ((size_t) c * h.T + t) * half;
            feed (specCollector, hop, half);
            feed (harmCollector, hop, half);
            if (h.order == 1) { harmonic.step(); spectral.step(); }
            else              { spectral.step(); harmonic.step(); }
            float* r = raw.data() + ((size_t) c * h.T + t) * 12;
            float* s = sm.data() + ((size_t) c * h.T + t) * 12;
            for (int i = 0; i < 12; i++)
            {
                const bool harmSlot = i == AudioFeatures::enF0 || i == AudioFeatures::enHarmonicEnergyRatio
                                   || i == AudioFeatures::enOddEvenHarmonicRatio || i == AudioFeatures::enInharmonicity;
                AudioFeatures& a = harmSlot ? harmFeatures : specFeatures;
                r[i] = a.smoothedFeatures[(size_t) i].history.back();
                s[i] = a.getValue ((AudioFeatures::eAudioFeature) i);
            }
            lag[(size_t) c * h.T + t] = harmonic.getPitchAnalyser().getNormalisedLagPosition().getX() * (float) (2 * h.N);
        }
    }
    f = fopen (argv[2], "wb");
    if (! f) return 2;
    fwrite (raw.data(), sizeof (float), raw.size(), f);
    fwrite (sm.data(), sizeof (float), sm.size(), f);
    fwrite (lag.data(), sizeof (float), lag.size(), f);
    fclose (f);
    return 0;
}
