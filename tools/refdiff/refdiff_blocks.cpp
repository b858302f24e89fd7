// refdiff_blocks.cpp -- the reference's OWN collector + overlapper + analysers (headers compiled unmodified from /root/reference/Source
// against tools/refdiff/juce_standin.h) fed the way an audio device feeds them: AudioDataCollector::audioDeviceIOCallback with blocks of
// ANY length, the analysers stepped whenever half a window is waiting in the ring (ref AudioDataCollector.h:36-94,
// RealTimeAudioAnalysis.h:205-219).  What it pins for fx_push_samples: the ring is a FIFO of raw samples, the gain is applied when a hop is
// READ (:88), clearBuffer (:122) zeroes the ring's contents and leaves its indices.  Build container only (tools/refdiff/README.md).
//
//   refdiff_blocks <in.bin> <out.bin> [--notify-per-block]
// --notify-per-block (round 6): the APPLICATION's stepping instead of the canonical one.  The app's analysis threads run their loop once when they
// are started, before any audio (AnalyserTrackController.h:184-185, RealTimeAnalyser.h:141-177 / :201-234: getNextBuffer, analyse, wait (-1)), and
// once per notify() -- one per audio callback (AudioDataCollector.h:68-69; juce::Thread::notify is an auto-reset event: notifications that arrive
// while the thread is busy collapse into one).  Inside getNextBuffer the thread spins while indexesOverlap (N/2) (AudioDataCollector.h:77,96-105),
// and that test lets the reader run AHEAD of the writer whenever readIndex >= writeIndex + expectedSamplesPerBlock: hops of zeros at start-up,
// and -- where half a window is longer than a device block -- hops of audio from a lap ago.  Single-threaded model of that: a thread is
// either spinning in getNextBuffer (it reads as soon as the indices allow, which can only change at a callback) or waiting for a notification.
// The number of hops analysed is then whatever it comes to; out.bin's `frames` says.
//   in : int32 N, C, total, block, order, num_events; float64 sample_rate;
//        events[num_events]: int32 at_sample (an event takes effect before the block that STARTS at or after this sample), int32 kind
//                            (0 = AudioDataCollector::setGain, 1 = clearBuffer, 2 = setOnsetDetectionSensitivity, 3 = setOnsetWindowLength,
//                            4 = setOnsetDetectionType, 5 = sampleRateChanged on both analysers: ref RealTimeAnalyser.h:111-114,244-258), float32 value, int32 pad
//        float32 stream[C][total]
//   out: int32 frames; float32 raw[C][frames][12], smoothed[C][frames][12]
#include "juce_standin.h"

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
#include <cstdlib>
#include <string>

struct Header { int32_t N, C, total, block, order, num_events; double sample_rate; };
struct Event { int32_t at, kind; float value; int32_t pad; };

int main (int argc, char** argv)
{
    if (argc != 3 && argc != 4) return 2;
    const bool appStepping = argc == 4 && std::string (argv[3]) == "--notify-per-block";
    if (argc == 4 && ! appStepping) return 2;
    FILE* f = fopen (argv[1], "rb");
    if (! f) return 2;
    Header h;
    if (fread (&h, sizeof h, 1, f) != 1) return 2;
    std::vector<Event> events ((size_t) h.num_events);
    if (h.num_events && fread (events.data(), sizeof (Event), events.size(), f) != events.size()) return 2;
    std::vector<float> stream ((size_t) h.C * h.total);
    if (fread (stream.data(), sizeof (float), stream.size(), f) != stream.size()) return 2;
    fclose (f);
    const int half = h.N / 2;
    // (the app's stepping analyses one hop at thread start and at most one per callback and pending notification)
    const int frames = appStepping ? 2 * ((h.total + h.block - 1) / h.block) + 2 : h.total / half;
    std::vector<float> raw ((size_t) h.C * frames * 12), sm (raw.size());
    int framesDone = -1;

    for (int c = 0; c < h.C; c++)
    {
        AudioDataCollector specCollector (0), harmCollector (0);
        specCollector.setExpectedSamplesPerBlock (h.block);
        harmCollector.setExpectedSamplesPerBlock (h.block);
        AudioFeatures shared, harmOwn;
        AudioFeatures& specFeatures = shared;
        AudioFeatures& harmFeatures = h.order == 2 ? harmOwn : shared;
        RealTimeSpectralAnalyser spectral (specCollector, specFeatures, h.N, h.sample_rate);
        RealTimeHarmonicAnalyser harmonic (harmCollector, harmFeatures, h.N, h.sample_rate);
        int waiting = 0, done = 0;
        size_t next_event = 0;
        auto record = [&] {
            float* r = raw.data() + ((size_t) c * frames + done) * 12;
            float* s = sm.data() + ((size_t) c * frames + done) * 12;
            for (int i = 0; i < 12; i++)
            {
                const bool harmSlot = i == AudioFeatures::enF0 || i == AudioFeatures::enHarmonicEnergyRatio
                                   || i == AudioFeatures::enOddEvenHarmonicRatio || i == AudioFeatures::enInharmonicity;
                AudioFeatures& a = harmSlot ? harmFeatures : specFeatures;
                r[i] = a.smoothedFeatures[(size_t) i].history.back();
                s[i] = a.getValue ((AudioFeatures::eAudioFeature) i);
            }
            done++;
        };
        // the app's two threads: identical collectors, identical notifications, so they move in step; `spinning` = inside getNextBuffer
        struct ThreadModel { bool signalled = false, spinning = true; } threads;
        auto appRun = [&] {
            for (;;)
            {
                if (threads.spinning)
                {
                    const bool a = specCollector.indexesOverlap (half), b = harmCollector.indexesOverlap (half);
                    if (a != b) exit (5);
                    if (a) return;
                    if (done >= frames) exit (6);
                    if (h.order == 1) { harmonic.step(); spectral.step(); }
                    else              { spectral.step(); harmonic.step(); }
                    record();
                    threads.spinning = false;
                }
                if (! threads.signalled) return;
                threads.signalled = false;
                threads.spinning = true;
            }
        };
        if (appStepping) appRun();                               // startThread: the loop's first pass needs no notification
        for (int at = 0; at < h.total; at += h.block)
        {
            while (next_event < events.size() && events[next_event].at <= at)
            {
                const Event& e = events[next_event++];
                if (e.kind == 0)      { specCollector.setGain (e.value); harmCollector.setGain (e.value); }
                else if (e.kind == 1) { specCollector.clearBuffer(); harmCollector.clearBuffer(); }
                else if (e.kind == 2) spectral.setOnsetDetectionSensitivity (e.value);
                else if (e.kind == 3) spectral.setOnsetWindowLength ((int) e.value);
                else if (e.kind == 4) spectral.setOnsetDetectionType ((OnsetDetector::eOnsetDetectionType) (int) e.value);
                else if (e.kind == 5) { spectral.sampleRateChanged ((double) e.value); harmonic.sampleRateChanged ((double) e.value); }
            }
            const int n = h.total - at < h.block ? h.total - at : h.block;
            const float* in[1] = { stream.data() + (size_t) c * h.total + at };
            specCollector.audioDeviceIOCallback (in, 1, nullptr, 0, n);
            harmCollector.audioDeviceIOCallback (in, 1, nullptr, 0, n);
            if (appStepping) { threads.signalled = true; appRun(); continue; }
            waiting += n;
            // the reference's ring holds 4096 samples, and its reader spins (indexesOverlap, :96-105) while the writer is less than a block behind
            // it: single-threaded that would never end, so such a case is refused instead of run
            if (waiting > 4096 - h.block) return 3;
            // the analysis threads are notified by every block (:68-69) and read half a window whenever it is there
            while (waiting >= half)
            {
                if (h.order == 1) { harmonic.step(); spectral.step(); }
                else              { spectral.step(); harmonic.step(); }
                waiting -= half;
                record();
            }
        }
        if (! appStepping && done != frames) return 4;
        if (framesDone >= 0 && framesDone != done) return 7;      // every channel sees the same indices
        framesDone = done;
    }
    f = fopen (argv[2], "wb");
    if (! f) return 2;
    const int32_t nf = framesDone < 0 ? 0 : framesDone;
    fwrite (&nf, sizeof nf, 1, f);
    for (int c = 0; c < h.C; c++) fwrite (raw.data() + (size_t) c * frames * 12, sizeof (float), (size_t) nf * 12, f);
    for (int c = 0; c < h.C; c++) fwrite (sm.data() + (size_t) c * frames * 12, sizeof (float), (size_t) nf * 12, f);
    fclose (f);
    return 0;
}
