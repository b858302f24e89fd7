// refdiff_osc.cpp -- the reference's OWN OSCFeatureAnalysisOutput (ref OSCFeatureAnalysisOutput.h:23-145, compiled unmodified against
// tools/refdiff/juce_standin.h, whose OSCSender records what it is handed): which twelve values, in which order, a feature message carries,
// how "ip[:port]" is parsed, and at what rate the timer is started.  Build container only.
//
//   refdiff_osc <address-string>      prints:  host port timerHz | address | the 12 arguments of one sendSpectralFeaturesViaOSC(true)
// The AudioFeatures object is filled so that slot k's getValue() is 100 + k (slot order: RealTimeAnalyser.h:17-32): the printed arguments
// are then the slot indices in wire order.
#include "juce_standin.h"

#define private public
#include "AudioFeatures.h"
#include "AudioDataCollector.h"
#include "RealTimeAudioAnalysis.h"
#include "PitchAnalyser.h"
#include "SpectralCharacteristics.h"
#include "HarmonicCharacteristics.h"
#include "RealTimeAnalyser.h"
#include "OSCFeatureAnalysisOutput.h"
#undef private

#include <cstdio>

int main (int argc, char** argv)
{
    if (argc != 2) return 2;
    AudioFeatures features;
    for (int k = 0; k < (int) AudioFeatures::numFeatures; k++)
        for (int r = 0; r < 10; r++) features.updateFeature ((AudioFeatures::eAudioFeature) k, 100.0f + (float) k);
    OSCFeatureAnalysisOutput out (features, String (argv[1]), String ("/Audio/A7"));
    out.timerCallback();
    std::printf ("%s %d %d | %s |", out.sender.connectedHost.c_str(), out.sender.connectedPort, out.timerHz, out.sender.lastAddress.c_str());
    for (float v : out.sender.lastArguments) std::printf (" %g", (double) v);
    std::printf ("\n");
    return out.sender.messages == 1 ? 0 : 3;
}
