// fx_reblock.hip -- the device side of fx_push_samples: device blocks of ANY length become whole hops.
//
// ref AudioDataCollector.h:36-70 takes whatever block the audio device delivers (441, 480, 512 ... samples) into a ring,
// and RealTimeAudioDataOverlapper::getNextBuffer (RealTimeAudioAnalysis.h:205-219) reads it back window/2 samples at a time.
// Here a channel's pending samples (< window/2, the "carry") and the new block form one byte stream per channel; its first
// hops x window/2 samples go to the hop buffer the frame kernels read ([C][hops][window/2], rows 16-byte aligned), the rest is
// the next carry.  Pure byte movement, HBM-bound: one thread per output dword, the source read as two aligned dwords and
// shifted into place (v_alignbyte_b32), so 16-bit and packed 24-bit samples move at dword rate whatever the block length
// makes of their alignment.  Samples are not converted and no gain is applied here: the frame kernels' load stage does both
// when the hop is analysed, as getAnalysisBuffer multiplies by the gain at read time (AudioDataCollector.h:88).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_kernels.h"

namespace fxk {

namespace {

constexpr int RB_THREADS = 256;

__global__ void __launch_bounds__(RB_THREADS)
fx_reblock_kernel(const ReblockParams p)
{
    const int c = blockIdx.y;
    const long long d0 = 4 * ((long long) blockIdx.x * RB_THREADS + threadIdx.x);       // byte offset in the channel's stream
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    if (d0 >= total) return;
    const unsigned char* in_row = p.in + (size_t) c * (size_t) p.in_row_bytes;
    const unsigned char* carry_row = p.carry_in + (size_t) c * (size_t) p.carry_row_bytes;
    unsigned v;
    const long long b0 = d0 - p.carry_bytes;
    if (b0 >= 0 && b0 + 8 <= p.in_row_bytes) {
        // wholly inside the new block, and so is the aligned pair of dwords around it
        const uintptr_t a = reinterpret_cast<uintptr_t>(in_row + b0);
        const unsigned* q = reinterpret_cast<const unsigned*>(a & ~(uintptr_t) 3);
        v = __builtin_amdgcn_alignbyte(q[1], q[0], (unsigned) (a & 3));
    } else {
        // across the carry / block boundary or at the end of the block: byte by byte, zeros past the end
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long s = d0 + k;
            if (s < total) {
                const unsigned b = s < p.carry_bytes ? carry_row[s] : in_row[s - p.carry_bytes];
                v |= b << (8 * k);
            }
        }
    }
    // (rows of whole hops are a multiple of 16 bytes, so a dword never straddles the two destinations; the last dword of the
    // carry may carry up to three bytes of zeros past the pending samples, inside the row)
    if (d0 < p.out_row_bytes) *reinterpret_cast<unsigned*>(p.hops_out + (size_t) c * (size_t) p.out_row_bytes + d0) = v;
    else *reinterpret_cast<unsigned*>(p.carry_out + (size_t) c * (size_t) p.carry_row_bytes + (d0 - p.out_row_bytes)) = v;
}

} // namespace

hipError_t launch_reblock_kernel(const ReblockParams& p, hipStream_t stream)
{
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    if (p.C <= 0 || total <= 0) return hipSuccess;
    if (p.carry_bytes < 0 || p.in_row_bytes < 0 || p.out_row_bytes < 0 || p.out_row_bytes > total || (p.out_row_bytes & 15) || (p.carry_row_bytes & 15) ||
        total - p.out_row_bytes > p.carry_row_bytes)
        return hipErrorInvalidValue;
    const long long dwords = (total + 3) / 4;
    // grid.y is limited to 65535: more channels than that go in slices (a context of 65 536 channels is configs[3])
    for (int c0 = 0; c0 < p.C; c0 += 65535) {
        ReblockParams q = p;
        const int cn = p.C - c0 < 65535 ? p.C - c0 : 65535;
        q.in += (size_t) c0 * (size_t) p.in_row_bytes;
        q.carry_in += (size_t) c0 * (size_t) p.carry_row_bytes;
        q.hops_out += (size_t) c0 * (size_t) p.out_row_bytes;
        q.carry_out += (size_t) c0 * (size_t) p.carry_row_bytes;
        hipLaunchKernelGGL(fx_reblock_kernel, dim3((unsigned) ((dwords + RB_THREADS - 1) / RB_THREADS), (unsigned) cn), dim3(RB_THREADS), 0, stream, q);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// AudioDataCollector::clearBuffer (AudioDataCollector.h:122): the ring's contents become zeros, its indices stay
hipError_t clear_carry(unsigned char* carry, size_t bytes, hipStream_t stream)
{
    return bytes ? hipMemsetAsync(carry, 0, bytes, stream) : hipSuccess;
}

} // namespace fxk
