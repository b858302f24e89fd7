// fx_realtime.hpp -- header-only C++ host side above the C ABI (fx.h).
//
// Mirrors the reference's own interface for this path so that existing C++ callers read the same:
//   * AudioFeatures                 -- ref Source/RealTimeAnalyser.h:14-92 (same enum, updateFeature,
//                                      getValue, getFeatureName minus juce::String)
//   * fx::RealTimeBatchAnalyser     -- the pair RealTimeSpectralAnalyser + RealTimeHarmonicAnalyser
//                                      (ref RealTimeAnalyser.h:133-269) for many channels on one GPU;
//                                      setter names are the reference's.
//   * fx::AudioDataCollector        -- ref Source/AudioDataCollector.h:18-138: audioDeviceIOCallback takes device blocks of ANY
//                                      length for all channels; whole hops are analysed as they complete (fx_push_samples).
//   * fx::OSCFeatureMessage         -- ref Source/OSCFeatureAnalysisOutput.h:89-113 wire format.
//   * fx::OSCFeatureAnalysisOutput  -- ref Source/OSCFeatureAnalysisOutput.h:23-145: the 60 Hz timer that sends a track's latest values.
//   * fx::OSCBatchSender            -- the same sink for thousands of tracks: formed datagrams, sendmmsg, sender threads, one 60 Hz timer.
//   * fx::LiveAnalyser              -- the live engine: audioDeviceIOCallback on the audio thread NEVER blocks (a copy into a FIFO, as the
//                                      reference's collector copies into its ring, AudioDataCollector.h:36-70); a worker thread analyses
//                                      each block straight from its page-locked FIFO slot and publishes (callback, OSCBatchSender).
// No JUCE.  Errors are thrown as fx::Error (the reference only jasserts).
#ifndef FX_REALTIME_HPP
#define FX_REALTIME_HPP

#include <cstddef>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "fx.h"

// ---- ValueHistory, ref Source/RealTimeAudioAnalysis.h:40-96 -----------------------------------
struct ValueHistory
{
    explicit ValueHistory (int historyLength) : recordedHistory (0) { setHistoryLength (historyLength); }

    float getTotal() const
    {
        float total = 0.0f;
        for (std::size_t i = 0; i < history.size(); i++) total += history[i];
        return total;
    }
    void insertNewValueAndupdateHistory (float newValue)
    {
        for (std::size_t i = 0; i + 1 < history.size(); i++) history[i] = history[i + 1];
        if (! history.empty()) history[history.size() - 1] = newValue;
        if (recordedHistory < (int) history.size()) recordedHistory++;
    }
    void setHistoryLength (int historyLength)
    {
        recordedHistory = 0;
        history.assign ((std::size_t) (historyLength > 0 ? historyLength : 0), 0.0f);
    }
    std::vector<float> history;
    int recordedHistory;
};

// ---- AudioFeatures, ref Source/RealTimeAnalyser.h:14-92 ---------------------------------------
struct AudioFeatures
{
    enum eAudioFeature
    {
        enOnset = FX_ONSET, enRMS = FX_RMS, enF0 = FX_F0, enCentroid = FX_CENTROID, enSpread = FX_SPREAD,
        enFlatness = FX_FLATNESS, enLER = FX_LER, enFlux = FX_FLUX, enSlope = FX_SLOPE,
        enHarmonicEnergyRatio = FX_HER, enOddEvenHarmonicRatio = FX_OER, enInharmonicity = FX_INHARM,
        numFeatures = FX_NUM_FEATURES
    };

    static const char* getFeatureName (eAudioFeature f)          // ref :34-63
    {
        static const char* names[numFeatures] = { "Onset", "Amp.", "Pitch", "Centroid", "Spread", "Flatness",
                                                  "L.E.R", "Flux", "Slope", "H.E.R", "O.E.R", "Inharm." };
        return (f >= 0 && f < numFeatures) ? names[f] : "UNKNOWN";
    }
    static float getMaxValueForFeature (eAudioFeature) { return 1.0f; }     // ref :65-68

    AudioFeatures()                                               // ref :70-74
    {
        for (int f = 0; f < numFeatures; f++)
            smoothedFeatures.push_back (ValueHistory (f == enOnset || f == enFlux ? 1 : 10));
    }
    void updateFeature (eAudioFeature f, float v) { smoothedFeatures[(std::size_t) f].insertNewValueAndupdateHistory (v); }
    float getValue (eAudioFeature f) const                        // ref :84-88
    {
        const ValueHistory& h = smoothedFeatures[(std::size_t) f];
        return h.getTotal() / (float) h.recordedHistory;
    }

private:
    std::vector<ValueHistory> smoothedFeatures;
};

namespace fx
{
struct Error : std::runtime_error
{
    Error (fx_status c, const char* what) : std::runtime_error (what), code (c) {}
    fx_status code;
};

inline void check (fx_status s) { if (s != FX_OK) throw Error (s, fx_last_error()); }

// OnsetDetector::eOnsetDetectionType, ref Source/SpectralCharacteristics.h:213-219
enum eOnsetDetectionType { enSpectral = FX_ONSET_SPECTRAL, enAmplitude = FX_ONSET_AMPLITUDE, enCombination = FX_ONSET_COMBINATION };

class RealTimeBatchAnalyser
{
public:
    // ref RealTimeAnalyser.h:100 (AudioDataCollector&, AudioFeatures&, int windowSize, double sampleRate = 48000.0)
    RealTimeBatchAnalyser (int numChannels, int windowSize = 2048, double sampleRate = 48000.0, int deviceId = 0,
                           unsigned flags = FX_ORDER_SPECTRAL_THEN_HARMONIC)
        : channels (numChannels), window (windowSize)
    {
        check (fx_create (&ctx, deviceId, numChannels, windowSize, sampleRate, flags));
    }
    ~RealTimeBatchAnalyser() { fx_destroy (ctx); }
    RealTimeBatchAnalyser (const RealTimeBatchAnalyser&) = delete;
    RealTimeBatchAnalyser& operator= (const RealTimeBatchAnalyser&) = delete;

    void sampleRateChanged (double sr)                         { check (fx_set_sample_rate (ctx, sr)); }          // ref :111
    void setOnsetDetectionSensitivity (float s)                { check (fx_set_onset_sensitivity (ctx, s)); }     // ref :244
    void setOnsetWindowLength (int length)                     { check (fx_set_onset_window (ctx, length)); }     // ref :250
    void setOnsetDetectionType (eOnsetDetectionType t)         { check (fx_set_onset_type (ctx, (int) t)); }      // ref :258
    void setGain (float g)                                     { check (fx_set_gain (ctx, g)); }                  // AudioDataCollector.h:124
    void reset()                                               { check (fx_reset_state (ctx)); }

    // hops [channels][numHops][window/2] host floats -> raw / smoothed [channels][numHops][12]
    void pushHops (const float* hops, int numHops, float* raw, float* smoothed)
    {
        check (fx_push_hops (ctx, hops, numHops, FX_SAMPLE_F32, FX_MEM_HOST, raw, smoothed));
    }
    // the same hops as 16-bit PCM (FX_SAMPLE_S16: v / 32768 in the kernels' load stage -- what JUCE's WAV reader makes of a 16-bit
    // file before AudioDataCollector sees it, ref AudioFilePlayer.h:41-61): half the bytes across PCIe, the same bits out
    void pushHopsPCM16 (const std::int16_t* hops, int numHops, float* raw, float* smoothed)
    {
        check (fx_push_hops (ctx, hops, numHops, FX_SAMPLE_S16, FX_MEM_HOST, raw, smoothed));
    }
    // ... and as packed 24-bit PCM, three bytes per sample, little endian (FX_SAMPLE_S24: v / 8388608)
    void pushHopsPCM24 (const unsigned char* hops, int numHops, float* raw, float* smoothed)
    {
        check (fx_push_hops (ctx, hops, numHops, FX_SAMPLE_S24, FX_MEM_HOST, raw, smoothed));
    }
    void processFrames (const float* frames, int numFrames, float* raw, float* smoothed)
    {
        check (fx_process_frames (ctx, frames, numFrames, FX_SAMPLE_F32, FX_MEM_HOST, raw, smoothed));
    }
    // device-resident variants (pointers from hipMalloc); asynchronous until sync()
    void pushHopsDevice (const void* hops, int numHops, int sampleFormat, float* raw, float* smoothed)
    {
        check (fx_push_hops (ctx, hops, numHops, sampleFormat, FX_MEM_DEVICE, raw, smoothed));
    }
    void sync() { check (fx_sync (ctx)); }

    // latest AudioFeatures::getValue of every slot of one channel, as the OSC timer would read them
    std::vector<float> getValues (int channel)
    {
        latest.resize ((std::size_t) channels * FX_NUM_FEATURES);
        check (fx_get_smoothed (ctx, latest.data(), FX_MEM_HOST));
        return std::vector<float> (latest.begin() + (std::ptrdiff_t) channel * FX_NUM_FEATURES,
                                   latest.begin() + (std::ptrdiff_t) (channel + 1) * FX_NUM_FEATURES);
    }
    float getValue (int channel, AudioFeatures::eAudioFeature f) { return getValues (channel)[(std::size_t) f]; }

    // launch-shape knobs (struct fx_tuning of fx.h).  No knob changes a result bit within a kernel family; the family itself is a
    // constructor flag (FX_LOW_LATENCY: every frame on a pair of wavefronts, the lower one-hop latency at 2048 / 4096 points).
    fx_tuning getTuning()                  { fx_tuning t; check (fx_get_tuning (ctx, &t)); return t; }
    void setTuning (const fx_tuning& t)    { check (fx_set_tuning (ctx, &t)); }

    int getNumChannels() const { return channels; }
    int getWindowSize() const  { return window; }
    fx_context* handle()       { return ctx; }

private:
    fx_context* ctx = nullptr;
    int channels, window;
    std::vector<float> latest;
};

// AudioDataCollector, ref Source/AudioDataCollector.h:18-138, for every channel of a RealTimeBatchAnalyser at once.  The reference's
// collector copies each device block -- whatever length the audio device delivers -- into a 4096-sample ring and notifies the analysis
// thread, which reads window/2 samples (times the gain) whenever they are there (:36-94).  Here audioDeviceIOCallback hands the block of
// all channels to fx_push_samples: the library keeps what is left over (< window/2 samples per channel, in device memory), analyses the
// hops that completed and the frames' values are handed to the callback the analysis threads' updateFeature calls stood for.
class AudioDataCollector
{
public:
    explicit AudioDataCollector (RealTimeBatchAnalyser& analyserToFeed)
        : analyser (analyserToFeed), channels (analyserToFeed.getNumChannels()), hop (analyserToFeed.getWindowSize() / 2) {}

    // ref :36-70.  inputChannelData[c] points at `numberOfSamples` floats of channel c (JUCE's layout); returns the number of analysis
    // frames per channel this block completed (0 is normal for blocks shorter than a hop).  Their values: raw() / smoothed(),
    // [channel][frame][12], and the callback set below is called once per block that completed at least one frame.
    int audioDeviceIOCallback (const float* const* inputChannelData, int numInputChannels, int numberOfSamples)
    {
        if (numInputChannels < channels) throw Error (FX_ERR_INVALID_ARGUMENT, "fewer input channels than the analyser has");   // jassert, :44-47
        block.resize ((std::size_t) channels * (std::size_t) numberOfSamples);
        for (int c = 0; c < channels; ++c)
            if (numberOfSamples > 0) std::memcpy (block.data() + (std::size_t) c * (std::size_t) numberOfSamples, inputChannelData[c], sizeof (float) * (std::size_t) numberOfSamples);
        return pushBlock (block.data(), numberOfSamples, FX_SAMPLE_F32);
    }
    // the same for a block that is already [channel][numberOfSamples] in one piece, in any sample format of fx.h
    int pushBlock (const void* samples, int numberOfSamples, int sampleFormat = FX_SAMPLE_F32)
    {
        const std::size_t most = (std::size_t) ((fx_pending_samples (analyser.handle()) + numberOfSamples) / hop);
        rawValues.resize ((std::size_t) channels * most * FX_NUM_FEATURES);
        smoothedValues.resize (rawValues.size());
        int frames = 0;
        check (fx_push_samples (analyser.handle(), samples, numberOfSamples, sampleFormat, FX_MEM_HOST,
                                most ? rawValues.data() : nullptr, most ? smoothedValues.data() : nullptr, &frames));
        lastFrames = frames;
        if (frames > 0 && framesAnalysed) framesAnalysed (frames);      // where the reference calls notifyAnalysisThread(), :68-69
        return frames;
    }
    void setNotifyAnalysisThreadCallback (std::function<void (int)> f)  { framesAnalysed = f; }    // ref :107 (argument: frames per channel)
    void setGain (float g)       { analyser.setGain (g); }                                          // ref :124
    void clearBuffer()           { check (fx_clear_pending (analyser.handle())); }                  // ref :122
    int  getNumPendingSamples()  { return fx_pending_samples (analyser.handle()); }
    int  getNumFrames() const    { return lastFrames; }
    const float* raw() const      { return rawValues.data(); }        // [channel][getNumFrames()][12] of the last block
    const float* smoothed() const { return smoothedValues.data(); }

private:
    RealTimeBatchAnalyser& analyser;
    int channels, hop, lastFrames = 0;
    std::vector<float> block, rawValues, smoothedValues;
    std::function<void (int)> framesAnalysed;
};

// The stand-in for AudioDataCollector's ring (ref Source/AudioDataCollector.h:24,36-94: the audio thread writes a ring, the analysis thread
// spins until a hop is there) for hosts that batch: a ring of pinned host batches of `hopsPerBatch` hops per channel (fx_stream_*).  The producer writes into
// nextSlot() and submit()s it -- or push()es a batch from ordinary memory, copied by `fillThreads` host threads inside the library --
// and collect()s the results in submission order; samples in, kernels and results back overlap.  One hop per batch is the reference's
// own cadence: a submit is then ONE kernel launch and collect() polls a flag (32-33 us round trip for a 4096-point window on MI355X).
// Destroy the ring before its analyser.
class HopRing
{
public:
    HopRing (RealTimeBatchAnalyser& analyser, int hopsPerBatch, int slots = 3, int sampleFormat = FX_SAMPLE_F32)
        : values ((std::size_t) analyser.getNumChannels() * (std::size_t) hopsPerBatch * FX_NUM_FEATURES)
    {
        check (fx_stream_create (analyser.handle(), hopsPerBatch, slots, sampleFormat, &ring));
    }
    ~HopRing() { fx_stream_destroy (ring); }
    HopRing (const HopRing&) = delete;
    HopRing& operator= (const HopRing&) = delete;

    void* nextSlot()                                   { void* p = nullptr; check (fx_stream_acquire (ring, &p)); return p; }
    void submit()                                      { check (fx_stream_submit (ring)); }
    void push (const void* hops, int fillThreads = 1)  { check (fx_stream_push (ring, hops, fillThreads)); }
    // a block of ANY length up to a slot's capacity, [channels][numSamples] (fx_stream_submit_samples / fx_stream_push_samples)
    void submitSamples (int numSamples)                { check (fx_stream_submit_samples (ring, numSamples)); }
    void pushSamples (const void* samples, int numSamples, int fillThreads = 1) { check (fx_stream_push_samples (ring, samples, numSamples, fillThreads)); }
    int  collectSamples (float* raw, float* smoothed)  { int frames = 0; check (fx_stream_collect_samples (ring, raw, smoothed, &frames)); return frames; }
    int  inFlight() const                              { return fx_stream_in_flight (ring); }
    // the oldest batch's values: raw / smoothed [channels][hopsPerBatch][12], either may be null
    void collect (float* raw, float* smoothed)         { check (fx_stream_collect (ring, raw, smoothed)); }
    std::size_t valuesPerBatch() const                 { return values; }

private:
    fx_stream* ring = nullptr;
    std::size_t values;
};

// The reference's LEGACY offline analyser (struct AudioAnalyser, ref Source/AudioAnalysis.h), one per channel, on the GPU: the
// members a host would have called, with the reference's names.  Buffers are host memory, [numChannels][...] row-major.
class AudioAnalyser
{
public:
    struct HarmonicCharacteristics { float f0, harmonicEnergyRatio, inharmonicity; };     // ref AudioAnalysis.h:31-42

    AudioAnalyser (int numberOfChannels, double nyquistFrequency, int deviceId = 0) : channels (numberOfChannels)   // ref :107
    {
        check (fx_offline_create (&off, deviceId, numberOfChannels, nyquistFrequency));
    }
    ~AudioAnalyser() { fx_offline_destroy (off); }
    AudioAnalyser (const AudioAnalyser&) = delete;
    AudioAnalyser& operator= (const AudioAnalyser&) = delete;

    // ref :517-541: audio [channels][numSamples] -> the ZeroCrosses feature row [channels][numDownsamples]
    std::vector<float> analyseNormalisedZeroCrosses (const float* audio, int numSamples, int numDownsamples)
    {
        std::vector<float> out ((std::size_t) channels * numDownsamples);
        check (fx_offline_zero_crosses (off, audio, numSamples, numDownsamples, out.data(), FX_MEM_HOST));
        return out;
    }
    // ref :611-622: channel 0 of the energy envelope -> estimatedLogAttackTime
    float setLogAttackTime (const float* energyEnvelope, int numEnvelopeSamples, int numInputSamples, int numDownsamples, int sampleRate)
    {
        float v = 0.0f;
        check (fx_offline_log_attack_time (off, energyEnvelope, numEnvelopeSamples, numInputSamples, numDownsamples, sampleRate, &v, FX_MEM_HOST));
        return v;
    }
    // ref :543-564 (the reference prints these): bits [channels][numBins], the last bin over the threshold / numBins, the share of bins over it
    void calculateFFTLBP (const float* fftResults, const float* previousFFTFrame, int numBins, unsigned char* bits, float* highestRatio, float* activityRatio)
    {
        check (fx_offline_fft_lbp (off, fftResults, previousFFTFrame, numBins, bits, highestRatio, activityRatio, FX_MEM_HOST));
    }
    // ref :253-303: magnitudes [channels][numBins] of one frame; previousF0 is kept per channel across calls
    std::vector<HarmonicCharacteristics> calculateHarmonicCharacteristics (const float* fftResults, int numBins)
    {
        std::vector<float> raw ((std::size_t) channels * 3);
        check (fx_offline_harmonic_characteristics (off, fftResults, numBins, raw.data(), FX_MEM_HOST));
        std::vector<HarmonicCharacteristics> out ((std::size_t) channels);
        for (int c = 0; c < channels; c++) out[(std::size_t) c] = { raw[3 * (std::size_t) c], raw[3 * (std::size_t) c + 1], raw[3 * (std::size_t) c + 2] };
        return out;
    }
    // ref :463-515 (legacy full-spectrum form): magnitudes [channels][numBins] -> per channel centroid / nyquist, spread, flatness, flux;
    // previousBinMagnitudes is kept per channel across calls
    struct SpectralCharacteristics { float centroid, spread, flatness, flux; };             // ref AudioAnalysis.h:17-29
    std::vector<SpectralCharacteristics> calculateSpectralCharacteristics (const float* fftResults, int numBins)
    {
        std::vector<float> raw ((std::size_t) channels * 4);
        check (fx_offline_spectral_characteristics (off, fftResults, numBins, raw.data(), FX_MEM_HOST));
        std::vector<SpectralCharacteristics> out ((std::size_t) channels);
        for (int c = 0; c < channels; c++) out[(std::size_t) c] = { raw[4 * (std::size_t) c], raw[4 * (std::size_t) c + 1], raw[4 * (std::size_t) c + 2], raw[4 * (std::size_t) c + 3] };
        return out;
    }
    // ref :566-609
    std::vector<float> calculateNormalisedSpectralSlope (const float* fftResults, int numBins)
    {
        std::vector<float> out ((std::size_t) channels);
        check (fx_offline_spectral_slope (off, fftResults, numBins, out.data(), FX_MEM_HOST));
        return out;
    }
    // ref :623-665: data [channels][numItems] (r, i) pairs, replaced in place by item x conjugate; returns the frequency estimates the reference prints
    std::vector<double> analyseAutoCorrelation (float* data, int numItems, std::vector<int>* peakBins = nullptr)
    {
        std::vector<int> peaks ((std::size_t) channels);
        std::vector<double> freqs ((std::size_t) channels);
        check (fx_offline_auto_correlation (off, data, numItems, peaks.data(), freqs.data(), FX_MEM_HOST));
        if (peakBins != nullptr) *peakBins = peaks;
        return freqs;
    }
    fx_offline* handle() { return off; }

private:
    fx_offline* off = nullptr;
    int channels;
};

// The datagram OSCSender::send (bundleAddress, onset, rmsLevel, f0, centroid, slope, spread, flatness,
// ler, flux, her, oer, inharm) emits -- ref Source/OSCFeatureAnalysisOutput.h:107
inline std::string OSCFeatureMessage (const std::string& address, const float* features12)
{
    unsigned char buf[512];
    const int n = fx_osc_encode (address.c_str(), features12, buf, (int) sizeof buf);
    if (n < 0) throw Error (FX_ERR_INVALID_ARGUMENT, "OSC address too long");
    return std::string (reinterpret_cast<const char*> (buf), (std::size_t) n);
}
} // namespace fx

// ---- UDP sender for the feature message (POSIX sockets; ref OSCFeatureAnalysisOutput.h:115-136) ----
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

namespace fx
{
class OSCFeatureSender
{
public:
    OSCFeatureSender() = default;
    ~OSCFeatureSender() { if (fd >= 0) ::close (fd); }
    OSCFeatureSender (const OSCFeatureSender&) = delete;
    OSCFeatureSender& operator= (const OSCFeatureSender&) = delete;

    // "ip[:port]", port defaults to 9000 -- same parsing as connectToAddress (ref :115-123)
    bool connectToAddress (const std::string& newAddress)
    {
        int port = 9000;
        const std::size_t sep = newAddress.rfind (':');
        if (sep != std::string::npos) port = std::atoi (newAddress.c_str() + sep + 1);
        const std::string ip = newAddress.substr (0, newAddress.find (':'));
        if (fd < 0) fd = ::socket (AF_INET, SOCK_DGRAM, 0);
        if (fd < 0) return false;
        dest = sockaddr_in();
        dest.sin_family = AF_INET;
        dest.sin_port = htons ((unsigned short) port);
        connected = ::inet_pton (AF_INET, ip.c_str(), &dest.sin_addr) == 1;
        return connected;
    }

    // one sendSpectralFeaturesViaOSC (ref :89-113): features12 in AudioFeatures slot order
    bool send (const std::string& bundleAddress, const float* features12)
    {
        if (! connected) return false;
        const std::string msg = OSCFeatureMessage (bundleAddress, features12);
        return ::sendto (fd, msg.data(), msg.size(), 0, reinterpret_cast<const sockaddr*> (&dest), sizeof dest) == (ssize_t) msg.size();
    }

private:
    int fd = -1;
    bool connected = false;
    sockaddr_in dest {};
};
} // namespace fx

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace fx
{
// OSCFeatureAnalysisOutput, ref Source/OSCFeatureAnalysisOutput.h:23-145: a 60 Hz timer that reads the track's AudioFeatures::getValue of
// every slot and sends them as one OSC message to ip[:port] (:89-113, :133).  The reference's timer reads the AudioFeatures object the
// analysis threads are writing, unsynchronised; here the analysis side hands over a snapshot -- updateFeatures (smoothed12), the
// `smoothed` vector of the channel's newest frame -- and the timer thread sends the latest snapshot, so a datagram is always the twelve
// values of ONE frame.  One object per track and target, as the reference builds them (AnalyserTrackController.h:22-23: two per track).
// That shape -- a thread, a socket and a system call per track and tick -- is kept for source compatibility and is meant for the
// reference's own scale (up to ~100 tracks); a host with thousands of tracks uses fx::OSCBatchSender below.
class OSCFeatureAnalysisOutput
{
public:
    OSCFeatureAnalysisOutput (const std::string& ip, const std::string& bundle) : bundleAddress (bundle)
    {
        for (float& v : latest) v = 0.0f;
        if (! bundle.empty()) connectToAddress (ip);                                   // ref :79-80
    }
    ~OSCFeatureAnalysisOutput() { stopTimer(); }
    OSCFeatureAnalysisOutput (const OSCFeatureAnalysisOutput&) = delete;
    OSCFeatureAnalysisOutput& operator= (const OSCFeatureAnalysisOutput&) = delete;

    // the analysis side: the smoothed vector (AudioFeatures::getValue of every slot) after the channel's newest frame
    void updateFeatures (const float* smoothed12)
    {
        std::lock_guard<std::mutex> g (lock);
        for (int i = 0; i < FX_NUM_FEATURES; i++) latest[i] = smoothed12[i];
        have = true;
    }
    // ref :89-113: one message with what getValue returns right now (nothing before the first frame: the reference would send 0/0 = NaN)
    bool sendSpectralFeaturesViaOSC()
    {
        float v[FX_NUM_FEATURES];
        {
            std::lock_guard<std::mutex> g (lock);
            if (! have) return false;
            for (int i = 0; i < FX_NUM_FEATURES; i++) v[i] = latest[i];
        }
        if (! sender.send (bundleAddress, v)) return false;
        sent++;
        return true;
    }
    // ref :115-136: "ip[:port]" (port 9000 by default); a successful connect starts the 60 Hz timer
    bool connectToAddress (const std::string& newAddress)
    {
        stopTimer();
        if (! sender.connectToAddress (newAddress)) return false;
        startTimerHz (60);
        return true;
    }
    void startTimerHz (int hz)
    {
        stopTimer();
        running = true;
        timer = std::thread ([this, hz] {
            const std::chrono::nanoseconds period (1000000000ll / (hz > 0 ? hz : 60));
            std::chrono::steady_clock::time_point next = std::chrono::steady_clock::now() + period;
            while (running.load())
            {
                std::this_thread::sleep_until (next);
                next += period;
                if (running.load()) (void) sendSpectralFeaturesViaOSC();                  // timerCallback, ref :84-87
            }
        });
    }
    void stopTimer()
    {
        running = false;
        if (timer.joinable()) timer.join();
    }
    long getNumMessagesSent() const { return sent.load(); }

private:
    OSCFeatureSender sender;
    std::string bundleAddress { "/Audio/Features" };                                    // ref :144
    std::mutex lock;
    float latest[FX_NUM_FEATURES];
    bool have = false;
    std::atomic<bool> running { false };
    std::atomic<long> sent { 0 };
    std::thread timer;
};

// The sink at scale: ONE object for all tracks.  A tick's messages arrive formed -- fx_get_osc_datagrams writes them on the GPU from the
// context's latest vectors, fx_osc_encode_batch on the host -- and go out in sendmmsg batches from `threads` sender threads to a primary
// and an optional secondary target, paced by a 60 Hz timer (ref OSCFeatureAnalysisOutput.h:84-136, AnalyserTrackController.h:22-23).
// Thin wrapper of the fx_osc_sender_* entries of include/fx.h.
class OSCBatchSender
{
public:
    OSCBatchSender (const std::string& primary, const std::string& secondary = std::string(), int threads = 1, bool segmentedSends = false)
    {
        check (fx_osc_sender_create (&sender, primary.c_str(), secondary.empty() ? nullptr : secondary.c_str(), threads, segmentedSends ? FX_OSC_SENDER_GSO : 0u));
    }
    ~OSCBatchSender() { fx_osc_sender_destroy (sender); }
    OSCBatchSender (const OSCBatchSender&) = delete;
    OSCBatchSender& operator= (const OSCBatchSender&) = delete;

    // publish formed messages: message i at datagrams + i * stride, messageLengths[i] bytes (copied)
    void updateDatagrams (const unsigned char* datagrams, int stride, const int* messageLengths, int count) { check (fx_osc_sender_update (sender, datagrams, stride, messageLengths, count)); }
    // publish smoothed vectors [count][12] (AudioFeatures slot order) of tracks "<prefix><firstChannel + i>", formed on the host
    void updateFeatures (const std::string& prefix, int firstChannel, const float* smoothed12, int count)
    {
        const int stride = fx_osc_message_bytes (prefix.c_str(), firstChannel + (count > 0 ? count - 1 : 0));
        if (stride < 0) throw Error (FX_ERR_INVALID_ARGUMENT, "OSC prefix too long or a negative channel number");
        scratch.resize ((std::size_t) count * (std::size_t) stride);
        lengths.resize ((std::size_t) count);
        if (fx_osc_encode_batch (prefix.c_str(), firstChannel, count, smoothed12, scratch.data(), stride, lengths.data()) != count)
            throw Error (FX_ERR_INVALID_ARGUMENT, "fx_osc_encode_batch refused its arguments");
        updateDatagrams (scratch.data(), stride, lengths.data(), count);
    }
    // publish the context's latest vectors, formed on the device (the copy to the host is the datagrams)
    void updateFromContext (fx_context* ctx, int numChannels, const std::string& prefix, int firstChannel)
    {
        const int stride = fx_osc_message_bytes (prefix.c_str(), firstChannel + (numChannels > 0 ? numChannels - 1 : 0));
        if (stride < 0) throw Error (FX_ERR_INVALID_ARGUMENT, "OSC prefix too long or a negative channel number");
        scratch.resize ((std::size_t) numChannels * (std::size_t) stride);
        lengths.resize ((std::size_t) numChannels);
        check (fx_get_osc_datagrams (ctx, prefix.c_str(), firstChannel, scratch.data(), stride, lengths.data(), FX_MEM_HOST));
        updateDatagrams (scratch.data(), stride, lengths.data(), numChannels);
    }
    long long sendNow() { long long n = 0; check (fx_osc_sender_send (sender, &n)); return n; }          // one timerCallback for every track
    void startTimerHz (int hz) { check (fx_osc_sender_start (sender, (double) hz)); }                    // ref :133
    void stopTimer() { check (fx_osc_sender_stop (sender)); }
    fx_osc_sender_stats getStats() const { fx_osc_sender_stats st; check (fx_osc_sender_get_stats (sender, &st)); return st; }
    fx_osc_sender* handle() { return sender; }

private:
    fx_osc_sender* sender = nullptr;
    std::vector<unsigned char> scratch;
    std::vector<int> lengths;
};

// The live engine.  The reference's audio callback copies the device block into the collector's ring and notify()s the analysis thread
// (AudioDataCollector.h:36-70); the analysis thread takes window/2 samples when they are there, analyses and writes AudioFeatures
// (RealTimeAnalyser.h:141-177, :201-234).  Same split here, for all channels at once:
//   audio thread   audioDeviceIOCallback / pushBlock: ONE copy of the block into a preallocated FIFO slot and a notify -- no allocation, no
//                  HIP call, never waits for the GPU.  A full FIFO (the worker has fallen `fifoBlocks` blocks behind) drops the block and
//                  counts it, where the reference's writer would overrun the reader (AudioDataCollector.h:96-105).
//   worker thread  owns every call on the context (fx.h: calls on one context are serialised by the caller): fx_push_samples of the next
//                  block STRAIGHT from its FIFO slot -- the slots are page-locked (fx_host_alloc), so the copy to the GPU is the link's DMA
//                  and there is no second host copy; a block that completes one hop is read by the analysis kernels directly -- then the
//                  publication: the frames callback (the place of the run() loops' updateFeature calls) and, if an OSCBatchSender is
//                  attached, every channel's message formed on the GPU from the latest vectors.
// Latency is recorded per block that completed frames: arrival on the audio thread -> publication.
// Construct after the analyser, destroy before it.  Link with -pthread.
class LiveAnalyser
{
public:
    typedef std::function<void (int frames, const float* raw, const float* smoothed)> FramesCallback;   // [channels][frames][12], worker thread

    // sampleFormat: what pushBlock's blocks hold -- FX_SAMPLE_F32 (JUCE's float callbacks; audioDeviceIOCallback needs it), or the device's own
    // integers as they are (FX_SAMPLE_S16, packed FX_SAMPLE_S24) or FX_SAMPLE_F16: half / three quarters of the bytes across the link, the same
    // result bits as the floats JUCE would have made of them (fx.h, sample formats)
    LiveAnalyser (RealTimeBatchAnalyser& analyserToFeed, int maxBlockSamples, int fifoBlocks = 8, int sampleFormat = FX_SAMPLE_F32)
        : analyser (analyserToFeed), channels (analyserToFeed.getNumChannels()), hop (analyserToFeed.getWindowSize() / 2),
          maxBlock (maxBlockSamples), format (sampleFormat),
          sampleBytes (sampleFormat == FX_SAMPLE_F32 ? 4 : (sampleFormat == FX_SAMPLE_S24 ? 3 : 2)), slots ((std::size_t) (fifoBlocks > 1 ? fifoBlocks : 2))
    {
        if (maxBlockSamples < 1) throw Error (FX_ERR_INVALID_ARGUMENT, "maxBlockSamples must be positive");
        if (sampleFormat != FX_SAMPLE_F32 && sampleFormat != FX_SAMPLE_F16 && sampleFormat != FX_SAMPLE_S16 && sampleFormat != FX_SAMPLE_S24)
            throw Error (FX_ERR_INVALID_ARGUMENT, "unknown sample format");
        const std::size_t most = (std::size_t) ((maxBlockSamples + hop - 1) / hop + 1);            // frames a block plus the pending samples can complete
        const std::size_t values = (std::size_t) channels * most * FX_NUM_FEATURES * sizeof (float);
        try
        {
            for (Slot& s : slots) { void* p = nullptr; check (fx_host_alloc (&p, sampleBytes * (std::size_t) channels * (std::size_t) maxBlockSamples)); s.samples = static_cast<unsigned char*> (p); }
            void* p = nullptr;
            check (fx_host_alloc (&p, values)); rawValues = static_cast<float*> (p);
            check (fx_host_alloc (&p, values)); smoothedValues = static_cast<float*> (p);
            latencies.reserve (1 << 20);
            running = true;
            worker = std::thread ([this] { run(); });
        }
        catch (...) { running = false; release(); throw; }           // (the destructor of a half-built engine never runs: give the page-locked memory back here)
    }
    ~LiveAnalyser()
    {
        stop();
        release();
    }
    LiveAnalyser (const LiveAnalyser&) = delete;
    LiveAnalyser& operator= (const LiveAnalyser&) = delete;

    // AUDIO THREAD, ref AudioDataCollector.h:36-70: inputChannelData[c] -> numberOfSamples floats of channel c.  false: the block was dropped.
    bool audioDeviceIOCallback (const float* const* inputChannelData, int numInputChannels, int numberOfSamples)
    {
        if (format != FX_SAMPLE_F32 || numInputChannels < channels || numberOfSamples < 0 || numberOfSamples > maxBlock) { dropped++; return false; }
        Slot* s = claim();
        if (s == nullptr) return false;
        for (int c = 0; c < channels; ++c)
            std::memcpy (s->samples + sizeof (float) * (std::size_t) c * (std::size_t) numberOfSamples, inputChannelData[c], sizeof (float) * (std::size_t) numberOfSamples);
        publish (s, numberOfSamples);
        return true;
    }
    // the same for a block that is already [channels][numberOfSamples] in one piece, in the engine's sample format
    bool pushBlock (const void* samples, int numberOfSamples)
    {
        if (numberOfSamples < 0 || numberOfSamples > maxBlock) { dropped++; return false; }
        Slot* s = claim();
        if (s == nullptr) return false;
        std::memcpy (s->samples, samples, sampleBytes * (std::size_t) channels * (std::size_t) numberOfSamples);
        publish (s, numberOfSamples);
        return true;
    }

    // set before the first block (not synchronised against the worker)
    void setFramesAnalysedCallback (FramesCallback f)                               { framesAnalysed = f; }
    void attachOSCSender (OSCBatchSender* sender, const std::string& bundlePrefix = "/Audio/A", int firstChannel = 0)
    {
        osc = sender; oscPrefix = bundlePrefix; oscFirst = firstChannel;
    }

    // Anything else that touches the analyser while the engine runs -- setGain, setOnsetDetectionType, sampleRateChanged, clearBuffer ... from a
    // GUI / message thread -- goes through here: the call is queued and the WORKER makes it before it takes the next block, so that the
    // context still has one caller (fx.h).  Not for the audio thread (it takes a lock).
    void callOnWorker (std::function<void (RealTimeBatchAnalyser&)> f)
    {
        { std::lock_guard<std::mutex> g (wake); commands.push_back (std::move (f)); }
        ready.notify_one();
    }

    // wait until everything pushed so far has been analysed and published (not for the audio thread)
    void drain()
    {
        std::unique_lock<std::mutex> g (wake);
        idle.wait (g, [this] { return written.load() == consumed.load() && commands.empty() && ! busyNow; });
    }
    void stop()
    {
        if (! running.exchange (false)) return;
        { std::lock_guard<std::mutex> g (wake); }
        ready.notify_all();
        if (worker.joinable()) worker.join();
    }

    struct Stats
    {
        long long blocksIn, blocksDropped, blocksAnalysed, framesPerChannel, errors;
        double latencyMsP50, latencyMsP99, latencyMsMax;      // audio-thread arrival -> publication, over the blocks that completed frames
        double workerBusySeconds;                              // time the worker spent between taking a block and having published it
    };
    Stats getStats()
    {
        Stats st;
        st.blocksIn = written.load(); st.blocksDropped = dropped.load(); st.blocksAnalysed = consumed.load();
        std::lock_guard<std::mutex> g (statLock);
        st.framesPerChannel = frames; st.errors = errors; st.workerBusySeconds = busy;
        std::vector<float> v (latencies);
        std::sort (v.begin(), v.end());
        st.latencyMsP50 = v.empty() ? 0.0 : v[v.size() / 2];
        st.latencyMsP99 = v.empty() ? 0.0 : v[(std::size_t) ((double) (v.size() - 1) * 0.99)];
        st.latencyMsMax = v.empty() ? 0.0 : v.back();
        return st;
    }
    std::string lastError() { std::lock_guard<std::mutex> g (statLock); return errorText; }
    // the latest smoothed vectors [channels][12] as the worker last published them (a copy; any thread)
    std::vector<float> latestSmoothed() { std::lock_guard<std::mutex> g (statLock); return latest; }

private:
    typedef std::chrono::steady_clock Clock;
    struct Slot { unsigned char* samples = nullptr; int count = 0; Clock::time_point arrived; };

    void release()
    {
        for (Slot& s : slots) { if (s.samples != nullptr) fx_host_free (s.samples); s.samples = nullptr; }
        if (rawValues != nullptr) fx_host_free (rawValues);
        if (smoothedValues != nullptr) fx_host_free (smoothedValues);
        rawValues = smoothedValues = nullptr;
    }
    // single producer (the audio thread), single consumer (the worker): `written` / `consumed` count blocks, a slot is index mod size
    Slot* claim()
    {
        const long long w = written.load (std::memory_order_relaxed);
        if (w - consumed.load (std::memory_order_acquire) >= (long long) slots.size()) { dropped++; return nullptr; }
        return &slots[(std::size_t) (w % (long long) slots.size())];
    }
    void publish (Slot* s, int numberOfSamples)
    {
        s->count = numberOfSamples;
        s->arrived = Clock::now();
        written.fetch_add (1, std::memory_order_release);
        { std::lock_guard<std::mutex> g (wake); }                                   // (the worker holds it only around its look at the counters, never while it works)
        ready.notify_one();                                                         // notify(), ref AudioDataCollector.h:68-69
    }

    void fail (const char* what)
    {
        std::lock_guard<std::mutex> g (statLock);
        errors++;
        errorText = std::string (what) + ": " + fx_last_error();
    }
    // one block: analysed straight from its page-locked FIFO slot, then what it completed is published
    void analyse (Slot& s)
    {
        int got = 0;
        if (fx_push_samples (analyser.handle(), s.samples, s.count, format, FX_MEM_HOST, rawValues, smoothedValues, &got) != FX_OK) { fail ("fx_push_samples"); return; }
        if (got <= 0) return;
        if (framesAnalysed) framesAnalysed (got, rawValues, smoothedValues);
        if (osc != nullptr)
        {
            try { osc->updateFromContext (analyser.handle(), channels, oscPrefix, oscFirst); }
            catch (const Error&) { fail ("fx_get_osc_datagrams"); }
        }
        const double ms = std::chrono::duration<double, std::milli> (Clock::now() - s.arrived).count();
        std::lock_guard<std::mutex> g (statLock);
        frames += got;
        if (latencies.size() < latencies.capacity()) latencies.push_back ((float) ms);
        latest.resize ((std::size_t) channels * FX_NUM_FEATURES);
        for (int c = 0; c < channels; ++c)
            std::memcpy (&latest[(std::size_t) c * FX_NUM_FEATURES], &smoothedValues[((std::size_t) c * (std::size_t) got + (std::size_t) (got - 1)) * FX_NUM_FEATURES], sizeof (float) * FX_NUM_FEATURES);
    }
    void run()
    {
        for (;;)
        {
            std::vector<std::function<void (RealTimeBatchAnalyser&)>> todo;
            {
                std::unique_lock<std::mutex> g (wake);
                busyNow = false;
                if (written.load() == consumed.load() && commands.empty()) idle.notify_all();
                ready.wait (g, [this] { return ! running.load() || written.load() != consumed.load() || ! commands.empty(); });
                if (! running.load() && written.load() == consumed.load() && commands.empty()) return;
                todo.swap (commands);
                busyNow = true;
            }
            for (auto& f : todo)                                                    // the other threads' setter calls, in order, between blocks
            {
                try { f (analyser); } catch (const Error&) { fail ("a call queued with callOnWorker"); }
            }
            if (written.load (std::memory_order_acquire) == consumed.load()) continue;
            const Clock::time_point t0 = Clock::now();
            analyse (slots[(std::size_t) (consumed.load() % (long long) slots.size())]);
            consumed.fetch_add (1, std::memory_order_release);                      // the slot is the audio thread's again
            const double dt = std::chrono::duration<double> (Clock::now() - t0).count();
            std::lock_guard<std::mutex> g (statLock);
            busy += dt;
        }
    }

    RealTimeBatchAnalyser& analyser;
    int channels, hop, maxBlock, format;
    std::size_t sampleBytes;
    std::vector<Slot> slots;
    std::atomic<long long> written { 0 }, consumed { 0 }, dropped { 0 };
    std::mutex wake;
    std::condition_variable ready, idle;
    std::vector<std::function<void (RealTimeBatchAnalyser&)>> commands;   // guarded by `wake`
    bool busyNow = false;                                                  // guarded by `wake`: the worker is between taking work and having finished it
    std::atomic<bool> running { false };
    std::thread worker;
    FramesCallback framesAnalysed;
    OSCBatchSender* osc = nullptr;
    std::string oscPrefix;
    int oscFirst = 0;
    float* rawValues = nullptr;          // page-locked: fx_push_samples copies the vectors back without a staging copy
    float* smoothedValues = nullptr;
    std::mutex statLock;
    long long frames = 0, errors = 0;
    double busy = 0.0;
    std::vector<float> latencies, latest;
    std::string errorText;
};
} // namespace fx

#endif // FX_REALTIME_HPP
