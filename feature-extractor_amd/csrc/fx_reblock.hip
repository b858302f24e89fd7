// fx_reblock.hip -- the device side of fx_push_samples: device blocks of ANY length become whole hops.
//
// ref AudioDataCollector.h:36-70 takes whatever block the audio device delivers (441, 480, 512 ... samples) into a ring,
// and RealTimeAudioDataOverlapper::getNextBuffer (RealTimeAudioAnalysis.h:205-219) reads it back window/2 samples at a time.
// Here a channel's pending samples (< window/2, the "carry") and the new block form one byte stream per channel; its first
// hops x window/2 samples go to the hop buffer the frame kernels read ([C][hops][window/2], rows 16-byte aligned), the rest is
// the next carry.  Pure byte movement, HBM-bound: one thread per 16 output bytes (one global_store_dwordx4 per lane, 1 KB per wavefront),
// the source read as five aligned dwords and shifted into place (v_alignbyte_b32), so 16-bit and packed 24-bit samples move at full
// rate whatever the block length makes of their alignment.  Samples are not converted and no gain is applied here: the frame kernels' load stage does both
// when the hop is analysed, as getAnalysisBuffer multiplies by the gain at read time (AudioDataCollector.h:88).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_kernels.h"

namespace fxk {

namespace {

constexpr int RB_THREADS = 256;

// four dwords that need only dword alignment (the hardware's requirement for a 16-byte global access)
struct __attribute__((packed, aligned(4))) dwords4 { unsigned x, y, z, w; };

// One 16-byte piece of a channel's stream at byte offset d0 (see the file header), returned in v.
__device__ __forceinline__ void reblock_piece(const ReblockParams& p, const unsigned char* in_row, const unsigned char* carry_row, long long total, long long d0, unsigned (&v)[4])
{
    const long long b0 = d0 - p.carry_bytes;
    if (b0 >= 0 && b0 + 20 <= p.in_row_bytes) {
        // wholly inside the new block, and so are the five aligned dwords around it
        // (pointer arithmetic, not an integer round trip: the compiler keeps the global address space and emits global_load_dwordx4)
        const unsigned char* at = in_row + b0;
        const unsigned sh = (unsigned) (reinterpret_cast<uintptr_t>(at) & 3);
        const unsigned char* base = at - sh;
        const dwords4 q = *reinterpret_cast<const dwords4*>(base);
        const unsigned q4 = *reinterpret_cast<const unsigned*>(base + 16);
        v[0] = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
        v[1] = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
        v[2] = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
        v[3] = __builtin_amdgcn_alignbyte(q4, q.w, sh);
    } else if (d0 + 16 <= p.carry_bytes) {
        // wholly inside the pending samples: the carry row starts on a 16-byte boundary and so does this piece
        const uint4 q = *reinterpret_cast<const uint4*>(carry_row + d0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        // across the carry / block boundary or at the end of the block: byte by byte, zeros past the end
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long s = d0 + 4 * j + k;
                if (s < total) {
                    const unsigned b = s < p.carry_bytes ? carry_row[s] : in_row[s - p.carry_bytes];
                    v[j] |= b << (8 * k);
                }
            }
        }
    }
}

// PIECES pieces per thread, a workgroup's span apart (every access of a wavefront is 1 KB of consecutive bytes): the loads of all
// pieces are issued before the first store, so a lane keeps PIECES x 20 bytes in flight.
template <int PIECES>
__global__ void __launch_bounds__(RB_THREADS)
fx_reblock_kernel(const ReblockParams p)
{
    const int c = blockIdx.y;
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    const unsigned char* in_row = p.in + (size_t) c * (size_t) p.in_row_bytes;
    const unsigned char* carry_row = p.carry_in + (size_t) c * (size_t) p.carry_row_bytes;
    unsigned v[PIECES][4];
    long long at[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        at[k] = 16 * (((long long) blockIdx.x * PIECES + k) * RB_THREADS + threadIdx.x);      // byte offset in the channel's stream
        if (at[k] < total) reblock_piece(p, in_row, carry_row, total, at[k], v[k]);
    }
    // (rows of whole hops and carry rows are multiples of 16 bytes, so a piece never straddles the two destinations; the last piece of the
    // carry may hold up to fifteen bytes of zeros past the pending samples, inside the row)
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        if (at[k] >= total) continue;
        uint4* dst = at[k] < p.out_row_bytes ? reinterpret_cast<uint4*>(p.hops_out + (size_t) c * (size_t) p.out_row_bytes + at[k])
                                             : reinterpret_cast<uint4*>(p.carry_out + (size_t) c * (size_t) p.carry_row_bytes + (at[k] - p.out_row_bytes));
        *dst = uint4{v[k][0], v[k][1], v[k][2], v[k][3]};
    }
}

} // namespace

hipError_t launch_reblock_kernel(const ReblockParams& p, hipStream_t stream)
{
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    if (p.C <= 0 || total <= 0) return hipSuccess;
    if (p.carry_bytes < 0 || p.in_row_bytes < 0 || p.out_row_bytes < 0 || p.out_row_bytes > total || (p.out_row_bytes & 15) || (p.carry_row_bytes & 15) ||
        total - p.out_row_bytes > p.carry_row_bytes)
        return hipErrorInvalidValue;
    const long long pieces = (total + 15) / 16;         // (of 16 bytes)
    // rows shorter than a workgroup's span of four pieces per thread (16 KB) would leave most of such a workgroup idle
    const int per_thread = pieces >= 4 * RB_THREADS ? 4 : (pieces >= 2 * RB_THREADS ? 2 : 1);
    const unsigned gx = (unsigned) ((pieces + (long long) per_thread * RB_THREADS - 1) / ((long long) per_thread * RB_THREADS));
    // grid.y is limited to 65535: more channels than that go in slices (a context of 65 536 channels is configs[3])
    for (int c0 = 0; c0 < p.C; c0 += 65535) {
        ReblockParams q = p;
        const int cn = p.C - c0 < 65535 ? p.C - c0 : 65535;
        q.in += (size_t) c0 * (size_t) p.in_row_bytes;
        q.carry_in += (size_t) c0 * (size_t) p.carry_row_bytes;
        q.hops_out += (size_t) c0 * (size_t) p.out_row_bytes;
        q.carry_out += (size_t) c0 * (size_t) p.carry_row_bytes;
        if (per_thread == 4) hipLaunchKernelGGL(fx_reblock_kernel<4>, dim3(gx, (unsigned) cn), dim3(RB_THREADS), 0, stream, q);
        else if (per_thread == 2) hipLaunchKernelGGL(fx_reblock_kernel<2>, dim3(gx, (unsigned) cn), dim3(RB_THREADS), 0, stream, q);
        else hipLaunchKernelGGL(fx_reblock_kernel<1>, dim3(gx, (unsigned) cn), dim3(RB_THREADS), 0, stream, q);
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
