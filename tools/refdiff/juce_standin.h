// juce_standin.h -- the handful of JUCE 4.2 types the reference's six hot-path headers touch, written
// from scratch (no JUCE or reference text) so that those headers can be COMPILED UNMODIFIED, from where
// they lie under /root/reference, and single-stepped as a differential check of oracle/fx_oracle.c.
//
// This is NOT a build of the reference (its JuceHeader.h / JUCE 4.2.3 are absent) and pins
// nothing by itself: the JUCE arithmetic on the path -- FFT, getRMSLevel, applyGainRamp, getMagnitude --
// is restated here from the published algorithm, as it is in the oracle.  What the
// harness does check is the ~1000 lines of feature arithmetic that ARE in the reference's headers.
//
// Build container only: nothing here travels to the GPU box except the vectors it produces.
#ifndef JUCE_STANDIN_H
#define JUCE_STANDIN_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

// ---- libm overloads as the reference's toolchains (VS2015 / Xcode) resolve them: unqualified
// log10 / exp / abs on a float or double argument pick the C++ overload set.
// REFDIFF_LOG10_CR: log10(float) is the correctly rounded float (the oracle's convention);
// otherwise it is this platform's log10f.
#ifdef REFDIFF_LOG10_CR
inline float log10 (float x) { return (float) ::log10 ((double) x); }
#else
inline float log10 (float x) { return ::log10f (x); }
#endif
inline float  exp  (float x)  { return ::expf (x); }
inline float  log  (float x)  { return ::logf (x); }
inline float  sqrt (float x)  { return ::sqrtf (x); }
inline float  pow  (float x, float y) { return ::powf (x, y); }
inline float  floor (float x) { return ::floorf (x); }
inline float  ceil (float x)  { return ::ceilf (x); }
inline double abs  (double x) { return ::fabs (x); }
inline float  abs  (float x)  { return ::fabsf (x); }
inline double log  (int x)    { return ::log ((double) x); }     // log (samplesPerWindow) / log (2)

const float  float_Pi  = 3.14159265358979323846f;
const double double_Pi = 3.14159265358979323846;

#define jassert(x)      ((void) 0)
#define jassertfalse    ((void) 0)
#ifndef DBG
#define DBG(x)          ((void) 0)
#endif
#define JUCE_DECLARE_NON_COPYABLE_WITH_LEAK_DETECTOR(T) T (const T&) = delete; T& operator= (const T&) = delete;
template <typename... A> void ignoreUnused (A&&...) {}
#define JUCE_LIVE_CONSTANT(x) (x)
template <typename T> T jmax (T a, T b) { return a < b ? b : a; }

// ---- what only the LEGACY headers (AudioFeatures.h, AudioAnalysis.h) touch ----
template <typename T> class Range
{
public:
    Range() : a(), b() {}
    Range (T s, T e) : a (s), b (e < s ? s : e) {}
    T getStart() const { return a; }
    T getEnd() const { return b; }
    void setStart (T s) { a = s; if (b < s) b = s; }
    void setEnd (T e) { b = e; if (e < a) a = e; }
private:
    T a, b;
};
class ReferenceCountedObject { public: virtual ~ReferenceCountedObject() {} };
template <typename T> class ReferenceCountedObjectPtr { public: ReferenceCountedObjectPtr() : p (nullptr) {} T* p; };
template <typename T> class HeapBlock
{
public:
    HeapBlock() {}
    explicit HeapBlock (size_t n) : d (n) {}
    template <typename I> T& operator[] (I i) { return d[(size_t) i]; }
    T* getData() { return d.data(); }
    void allocate (size_t n, bool) { d.assign (n, T()); }
private:
    std::vector<T> d;
};

class String
{
public:
    String() {}
    String (const char* t) : s (t) {}
    explicit String (float v)  : s (std::to_string (v)) {}
    explicit String (double v) : s (std::to_string (v)) {}
    explicit String (int v)    : s (std::to_string (v)) {}
    String operator+ (const String& o) const { String r; r.s = s + o.s; return r; }
    String operator+ (const char* o) const   { String r; r.s = s + o; return r; }
    String& operator<< (const String& o)     { s += o.s; return *this; }
    String& operator<< (const char* o)       { s += o; return *this; }
    // what only OSCFeatureAnalysisOutput.h touches (its address parsing, :115-123), by JUCE's documented behaviour
    bool operator!= (const String& o) const  { return s != o.s; }
    bool operator== (const String& o) const  { return s == o.s; }
    int lastIndexOfAnyOf (const String& chars) const { const size_t p = s.find_last_of (chars.s); return p == std::string::npos ? -1 : (int) p; }
    String fromLastOccurrenceOf (const String& sub, bool includeSub, bool) const
    {
        const size_t p = s.rfind (sub.s);
        String r;
        r.s = p == std::string::npos ? s : s.substr (includeSub ? p : p + sub.s.size());
        return r;
    }
    String upToFirstOccurrenceOf (const String& sub, bool includeSub, bool) const
    {
        const size_t p = s.find (sub.s);
        String r;
        r.s = p == std::string::npos ? s : s.substr (0, includeSub ? p + sub.s.size() : p);
        return r;
    }
    int getIntValue() const { return std::atoi (s.c_str()); }
    static const String empty;
    std::string s;
};
const String String::empty;

// ---- what only OSCFeatureAnalysisOutput.h touches: a timer that never fires by itself and a sender that records what it is given ----
class Timer
{
public:
    virtual ~Timer() {}
    virtual void timerCallback() = 0;
    void startTimerHz (int hz) { timerHz = hz; }
    void stopTimer() { timerHz = 0; }
    int timerHz = 0;
};
class OSCSender
{
public:
    bool connect (const String& host, int port) { connectedHost = host.s; connectedPort = port; return true; }
    template <typename... Args> bool send (const String& address, Args... args)
    {
        lastAddress = address.s;
        lastArguments = { (float) args... };
        ++messages;
        return true;
    }
    std::string connectedHost, lastAddress;
    int connectedPort = 0, messages = 0;
    std::vector<float> lastArguments;
};

template <typename T> class Atomic
{
public:
    Atomic() : v() {}
    Atomic (T x) : v (x) {}
    T get() const { return v; }
    void set (T x) { v = x; }
private:
    T v;
};

template <typename T> class Point
{
public:
    Point() : x(), y() {}
    Point (T a, T b) : x (a), y (b) {}
    T getX() const { return x; }
    T getY() const { return y; }
private:
    T x, y;
};

class BigInteger { public: int getHighestBit() const { return 63; } };
class AudioIODevice { public: BigInteger getActiveInputChannels() const { return BigInteger(); } };
class AudioIODeviceCallback
{
public:
    virtual ~AudioIODeviceCallback() {}
    virtual void audioDeviceIOCallback (const float**, int, float**, int, int) = 0;
    virtual void audioDeviceAboutToStart (AudioIODevice*) = 0;
    virtual void audioDeviceStopped() = 0;
};

// One pass of run() per call: wait() raises the exit flag, step() lowers it again (trap 2).
class Thread
{
public:
    explicit Thread (const String&) : stop (false) {}
    virtual ~Thread() {}
    virtual void run() = 0;
    bool threadShouldExit() const { return stop; }
    bool wait (int) { stop = true; return true; }
    void notify() {}
    void step() { stop = false; run(); }
private:
    bool stop;
};

// ---- AudioSampleBuffer: the members the path uses ----
class AudioSampleBuffer
{
public:
    AudioSampleBuffer() : nch (0), ns (0), cleared (false) {}
    AudioSampleBuffer (int channels, int samples) : nch (channels), ns (samples), d ((size_t) channels * samples), cleared (false) {}
    int getNumChannels() const { return nch; }
    int getNumSamples() const  { return ns; }
    const float* getReadPointer (int c) const { return d.data() + (size_t) c * ns; }
    float* getWritePointer (int c) { cleared = false; return d.data() + (size_t) c * ns; }
    float getSample (int c, int i) const { return d[(size_t) c * ns + i]; }
    void setSample (int c, int i, float v) { cleared = false; d[(size_t) c * ns + i] = v; }
    void clear() { std::fill (d.begin(), d.end(), 0.0f); cleared = true; }
    // keepExistingContent / clearExtraSpace as the path calls it (true, true): new zeroed block, old samples copied
    void setSize (int channels, int samples, bool keep = false, bool clearExtra = false, bool = false)
    {
        if (channels == nch && samples == ns) return;
        std::vector<float> nd ((size_t) channels * samples, 0.0f);
        if (keep)
            for (int c = 0; c < std::min (channels, nch); c++)
                std::memcpy (nd.data() + (size_t) c * samples, d.data() + (size_t) c * ns, sizeof (float) * (size_t) std::min (samples, ns));
        (void) clearExtra;
        d.swap (nd); nch = channels; ns = samples;
        if (! keep) cleared = false;
    }
    void copyFrom (int dc, int ds, const float* src, int n)
    {
        cleared = false;
        std::memcpy (d.data() + (size_t) dc * ns + ds, src, sizeof (float) * (size_t) n);
    }
    void copyFrom (int dc, int ds, const AudioSampleBuffer& src, int sc, int ss, int n)
    {
        if (src.cleared) { if (! cleared) std::memset (d.data() + (size_t) dc * ns + ds, 0, sizeof (float) * (size_t) n); return; }
        copyFrom (dc, ds, src.getReadPointer (sc) + ss, n);
    }
    // x *= g; g += (end - start) / n, all in float
    void applyGainRamp (int c, int start, int n, float g0, float g1)
    {
        if (cleared) return;
        if (g0 == g1) { float* p = d.data() + (size_t) c * ns + start; for (int i = 0; i < n; i++) p[i] *= g0; return; }
        const float inc = (g1 - g0) / (float) n;
        float* p = d.data() + (size_t) c * ns + start;
        float g = g0;
        for (int i = 0; i < n; i++) { p[i] *= g; g += inc; }
    }
    // sqrt of the double mean of float squares
    float getRMSLevel (int c, int start, int n) const
    {
        if (n <= 0 || cleared) return 0.0f;
        const float* p = getReadPointer (c) + start;
        double sum = 0.0;
        for (int i = 0; i < n; i++) { const float s = p[i]; sum += s * s; }
        return (float) std::sqrt (sum / n);
    }
    // (legacy headers) the range of a stretch of samples, a gain over a stretch of channel 0
    Range<float> findMinMax (int c, int start, int n) const
    {
        if (n <= 0 || cleared) return Range<float>();
        const float* p = getReadPointer (c) + start;
        float lo = p[0], hi = p[0];
        for (int i = 1; i < n; i++) { if (p[i] < lo) lo = p[i]; if (p[i] > hi) hi = p[i]; }
        return Range<float> (lo, hi);
    }
    void applyGain (int start, int n, float g) { for (int c = 0; c < nch; c++) { float* p = d.data() + (size_t) c * ns + start; for (int i = 0; i < n; i++) p[i] *= g; } }
    // max |x| (via the range's min and max)
    float getMagnitude (int c, int start, int n) const
    {
        if (cleared || n <= 0) return 0.0f;
        const float* p = getReadPointer (c) + start;
        float lo = p[0], hi = p[0];
        for (int i = 1; i < n; i++) { if (p[i] < lo) lo = p[i]; if (p[i] > hi) hi = p[i]; }
        return std::max (-lo, hi) > std::max (lo, -hi) ? std::max (-lo, hi) : std::max (lo, -hi);
    }
private:
    int nch, ns;
    std::vector<float> d;
    bool cleared;
};

// ---- FFT: mixed-radix decimation in time, factors 4...4[,2], table twiddles, fp32, no fused ops
// (SURVEY.md App. A.1).  Written as an explicit stage loop over a digit-reversed copy rather than
// JUCE's recursion: the butterfly DAG, and therefore every rounding, is the same.
class FFT
{
public:
    struct Complex { float r, i; };
    FFT (int order, bool isInverse) : n (1 << order), inverse (isInverse), tw ((size_t) n)
    {
        const double f = (isInverse ? 2.0 : -2.0) * double_Pi / n;
        for (int i = 0; i < n; i++) { const double ph = i * f; tw[(size_t) i].r = (float) std::cos (ph); tw[(size_t) i].i = (float) std::sin (ph); }
        for (int m = n; m > 1;) { const int r = (m % 4 == 0) ? 4 : 2; m /= r; radices.push_back (r); }   // outermost first
    }
    int getSize() const { return n; }

    void perform (const Complex* in, Complex* out) const
    {
        // decimation in time: output position of input sample i is its mixed-radix digit reversal
        for (int i = 0; i < n; i++) out[(size_t) reversed (i)] = in[i];
        int len = 1;                                          // length of the finished sub-transforms
        for (int s = (int) radices.size() - 1; s >= 0; s--) {
            const int r = radices[(size_t) s];
            const int stride = n / (len * r);                 // twiddle step of this stage
            for (int base = 0; base < n; base += len * r)
                for (int k = 0; k < len; k++)
                    r == 4 ? bfly4 (out + base + k, len, k * stride) : bfly2 (out + base + k, len, k * stride);
            len *= r;
        }
    }
    // (legacy headers; not exercised by the harness: AudioAnalysis.h:197 only) magnitudes of the first half of the spectrum
    void performFrequencyOnlyForwardTransform (float* d) const
    {
        std::vector<Complex> a ((size_t) n), b ((size_t) n);
        for (int i = 0; i < n; i++) { a[(size_t) i].r = d[i]; a[(size_t) i].i = 0.0f; }
        perform (a.data(), b.data());
        for (int i = 0; i < n; i++) d[i] = std::sqrt (b[(size_t) i].r * b[(size_t) i].r + b[(size_t) i].i * b[(size_t) i].i);
    }
    void performRealOnlyForwardTransform (float* d) const
    {
        std::vector<Complex> a ((size_t) n), b ((size_t) n);
        for (int i = 0; i < n; i++) { a[(size_t) i].r = d[i]; a[(size_t) i].i = 0.0f; }
        perform (a.data(), b.data());
        for (int i = 0; i < n; i++) { d[2 * i] = b[(size_t) i].r; d[2 * i + 1] = b[(size_t) i].i; }
    }
    void performRealOnlyInverseTransform (float* d) const
    {
        std::vector<Complex> a ((size_t) n), b ((size_t) n);
        for (int i = 0; i < n; i++) { a[(size_t) i].r = d[2 * i]; a[(size_t) i].i = d[2 * i + 1]; }
        perform (a.data(), b.data());
        const float scale = 1.0f / (float) n;
        for (int i = 0; i < n; i++) { d[i] = b[(size_t) i].r * scale; d[i + n] = b[(size_t) i].i * scale; }
    }
private:
    static Complex mul (Complex a, Complex b) { Complex c; c.r = a.r * b.r - a.i * b.i; c.i = a.r * b.i + a.i * b.r; return c; }
    static Complex add (Complex a, Complex b) { Complex c; c.r = a.r + b.r; c.i = a.i + b.i; return c; }
    static Complex sub (Complex a, Complex b) { Complex c; c.r = a.r - b.r; c.i = a.i - b.i; return c; }
    int reversed (int i) const
    {
        // sample i = sum_s digit_s * (product of the radices before s), outermost digit first;
        // its slot is the same digits read in the opposite order
        int slot = 0, span = n;
        for (size_t s = 0; s < radices.size(); s++) { const int r = radices[s]; span /= r; slot += (i % r) * span; i /= r; }
        return slot;
    }
    void bfly2 (Complex* d, int len, int t) const
    {
        const Complex s = mul (d[len], tw[(size_t) t]);
        d[len] = sub (d[0], s);
        d[0] = add (d[0], s);
    }
    void bfly4 (Complex* d, int len, int t) const
    {
        const Complex s0 = mul (d[len], tw[(size_t) t]), s1 = mul (d[2 * len], tw[(size_t) (2 * t)]), s2 = mul (d[3 * len], tw[(size_t) (3 * t)]);
        const Complex s3 = add (s0, s2), s4 = sub (s0, s2), s5 = sub (d[0], s1);
        d[0] = add (d[0], s1);
        d[2 * len] = sub (d[0], s3);
        d[0] = add (d[0], s3);
        if (inverse) { d[len].r = s5.r - s4.i; d[len].i = s5.i + s4.r; d[3 * len].r = s5.r + s4.i; d[3 * len].i = s5.i - s4.r; }
        else         { d[len].r = s5.r + s4.i; d[len].i = s5.i - s4.r; d[3 * len].r = s5.r - s4.i; d[3 * len].i = s5.i + s4.r; }
    }
    int n;
    bool inverse;
    std::vector<Complex> tw;
    std::vector<int> radices;
};

#endif
