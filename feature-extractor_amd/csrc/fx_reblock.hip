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

#include "fx_blocks.hip.h"

namespace {

constexpr int RB_THREADS = 256;

// PIECES pieces per thread, a workgroup's span apart (every access of a wavefront is 1 KB of consecutive bytes): the loads of all
// pieces are issued before the first store, so a lane keeps PIECES x 20 bytes in flight.
template <int PIECES>
__global__ void __launch_bounds__(RB_THREADS)
fx_reblock_kernel(const ReblockParams p)
{
    const int c = blockIdx.y;
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    const BlockStream s{p.carry_in + (size_t) c * (size_t) p.carry_row_bytes, p.in + (size_t) c * (size_t) p.in_row_bytes, p.carry_bytes, p.in_row_bytes};
    unsigned v[PIECES][4];
    long long at[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; k++) {
        at[k] = 16 * (((long long) blockIdx.x * PIECES + k) * RB_THREADS + threadIdx.x);      // byte offset in the channel's stream
        if (at[k] < total) stream_piece16(s, total, at[k], v[k]);
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

// Short rows (fewer 16-byte pieces than a workgroup has threads: blocks of a few hundred samples, the live case): a workgroup per channel
// would leave most of its lanes idle, so the pieces of ALL channels are dealt out in one run -- thread g takes piece g % pieces of channel
// g / pieces -- and a workgroup spans as many channels as fill it.
__global__ void __launch_bounds__(RB_THREADS)
fx_reblock_rows_kernel(const ReblockParams p, const unsigned pieces)
{
    const unsigned long long g = (unsigned long long) blockIdx.x * RB_THREADS + threadIdx.x;
    const unsigned c = (unsigned) (g / pieces);
    if (c >= (unsigned) p.C) return;
    const long long at = 16ll * (long long) (g - (unsigned long long) c * pieces);
    const long long total = (long long) p.carry_bytes + p.in_row_bytes;
    const BlockStream s{p.carry_in + (size_t) c * (size_t) p.carry_row_bytes, p.in + (size_t) c * (size_t) p.in_row_bytes, p.carry_bytes, p.in_row_bytes};
    unsigned v[4];
    stream_piece16(s, total, at, v);
    uint4* dst = at < p.out_row_bytes ? reinterpret_cast<uint4*>(p.hops_out + (size_t) c * (size_t) p.out_row_bytes + at)
                                      : reinterpret_cast<uint4*>(p.carry_out + (size_t) c * (size_t) p.carry_row_bytes + (at - p.out_row_bytes));
    *dst = uint4{v[0], v[1], v[2], v[3]};
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
    if (pieces < RB_THREADS) {
        const unsigned long long all = (unsigned long long) p.C * (unsigned long long) pieces;
        hipLaunchKernelGGL(fx_reblock_rows_kernel, dim3((unsigned) ((all + RB_THREADS - 1) / RB_THREADS)), dim3(RB_THREADS), 0, stream, p, (unsigned) pieces);
        return hipGetLastError();
    }
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
