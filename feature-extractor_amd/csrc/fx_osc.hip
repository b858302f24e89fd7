// fx_osc.hip -- the sink's messages written on the device: every channel's latest smoothed vector becomes the wire-ready OSC 1.0
// message OSCSender::send (bundleAddress, onset, rms, f0, centroid, slope, spread, flatness, ler, flux, her, oer, inharm) emits
// (ref OSCFeatureAnalysisOutput.h:91-107), address "<prefix><channel number>" (ref MainComponent.cpp:170: "/Audio/A" + row), so that
// the copy to the host IS the datagrams: nothing is left for the host to format at 65 536 tracks x 60 Hz.
//
// Layout of one message (fx_osc_encode, fx_capi.cpp, is the host form of the same bytes):
//   address, NUL, zero-padded to a multiple of 4 | ",ffffffffffff" NUL NUL NUL (16 bytes) | twelve big-endian float32 in wire order
// Message c sits at out + c * stride (stride a multiple of 4, >= the longest message); the bytes of a slot past its message are zero.
// Pure byte movement, HBM-bound and tiny (8192 channels: 48 B in, 80 B out per channel): one thread per 4-byte word of output, so a
// wavefront writes 256 consecutive bytes; the twelve floats of a channel are read once each, as they are in `latest`.
#include <hip/hip_runtime.h>

#include "fx_kernels.h"

namespace fxk {

namespace {

constexpr int OSC_THREADS = 256;

// wire position -> AudioFeatures slot (ref OSCFeatureAnalysisOutput.h:107)
__device__ __constant__ int k_wire_slot[12] = {FX_ONSET, FX_RMS, FX_F0, FX_CENTROID, FX_SLOPE, FX_SPREAD, FX_FLATNESS, FX_LER, FX_FLUX, FX_HER, FX_OER, FX_INHARM};

__global__ void __launch_bounds__(OSC_THREADS)
fx_osc_kernel(const OscParams p)
{
    const int words = p.stride >> 2;
    const long long g = (long long) blockIdx.x * OSC_THREADS + threadIdx.x;
    if (g >= (long long) p.C * words) return;
    const int c = (int) (g / words), w = (int) (g - (long long) c * words);
    const unsigned n = (unsigned) (p.first_channel + c);
    int digits = 1;
    for (unsigned t = n; t >= 10u; t /= 10u) digits++;
    const int alen = p.prefix_len + digits;
    const int apad = (alen + 4) & ~3;
    const int b0 = 4 * w;
    unsigned v = 0;
    if (b0 < apad) {
        // address bytes: the prefix, then the channel number in decimal, then zeros
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int b = b0 + j;
            unsigned ch = 0;
            if (b < p.prefix_len) ch = p.prefix[b];
            else if (b < alen) {
                unsigned t = n;
                for (int k = alen - 1 - b; k > 0; k--) t /= 10u;
                ch = '0' + t % 10u;
            }
            v |= ch << (8 * j);
        }
    } else if (b0 < apad + 16) {
        // ",ffffffffffff" and three NULs, as little-endian words
        const int t = (b0 - apad) >> 2;
        v = t == 0 ? 0x6666662Cu : (t == 3 ? 0x00000066u : 0x66666666u);
    } else if (b0 < apad + 64) {
        const int i = (b0 - apad - 16) >> 2;
        v = __builtin_bswap32(__float_as_uint(p.latest[(size_t) c * FX_NUM_FEATURES + k_wire_slot[i]]));
    }
    reinterpret_cast<unsigned*>(p.out + (size_t) c * (size_t) p.stride)[w] = v;
}

} // namespace

hipError_t launch_osc_kernel(const OscParams& p, hipStream_t stream)
{
    if (p.C <= 0) return hipSuccess;
    if (p.stride < 4 || (p.stride & 3) || p.prefix_len < 0 || p.prefix_len > FX_OSC_PREFIX_MAX || p.first_channel < 0 || !p.latest || !p.out ||
        (reinterpret_cast<uintptr_t>(p.out) & 3))
        return hipErrorInvalidValue;
    const long long total = (long long) p.C * (p.stride >> 2);
    hipLaunchKernelGGL(fx_osc_kernel, dim3((unsigned) ((total + OSC_THREADS - 1) / OSC_THREADS)), dim3(OSC_THREADS), 0, stream, p);
    return hipGetLastError();
}

} // namespace fxk
