// fx_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for the per-frame audio feature path
//   RealTimeAnalyser -> SpectralCharacteristics / HarmonicCharacteristics / PitchAnalyser
// of SeanSoraghan/Feature-Extractor.  "ref:" citations are relative to the reference's Source/.
//
// Mapping:
//   * fx_frame_kernel: one workgroup per CHANNEL, K <= 8 waves (64 lanes each); wave w analyses frames
//     w, w+K, w+2K ... of its channel, so a whole analysis frame lives in ONE wavefront: no workgroup
//     barriers inside a frame, only wave-local LDS exchanges through one buffer per wave.
//   * the only frame-to-frame dependency on this path, the spectral-flux state
//     (previousBinMagnitudes, ref SpectralCharacteristics.h:203), is handed from the wave of frame
//     t-1 to the wave of frame t through LDS with a turn counter.
//   * FFTs follow the exact rounding DAG of the reference's FFT (JUCE 4.2 kiss-style radix-4/2
//     decimation in time, table twiddles, no fused multiply-add), executed as three register-resident
//     passes (radix 16/8/4 first pass on real input, then radix-16 / radix-4 passes) over a padded
//     LDS image with compile-time offsets and pass-ordered twiddles, so spectra are bit-identical to
//     the CPU path and every discrete decision downstream (pitch lag, peaks, onset) agrees.
//   * reductions over bins run in fp64 (the reference accumulates in double) and are combined with
//     DPP row operations; the few places where serial ORDER is semantics (flatness product, low-pass,
//     lag scan, flat-spectrum ties) are handled exactly.
//   * the kernel is bound by VALU issue first and the LDS pipe second, so values that only one
//     lane produces go straight to global memory and lane-to-lane hand-overs use DPP / ds_bpermute where the
//     data already sits in the right lane's registers.
//   * fx_finalise_kernel (thread = frame) runs the scalar tail (logRMS, pow, log10, sqrt, divisions);
//     fx_epilogue_kernel (thread = channel x frame) evaluates ValueHistory smoothing and onset
//     detection, which depend only on a bounded window of past raw values; fx_history_kernel carries
//     that window to the next call.
//
// No MFMA: this is <=4096-point FFTs and reductions, not a dense contraction.

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "fx_kernels.h"

#pragma clang fp contract(off)

namespace fxk {

#include "fx_wave.hip.h"
#include "fx_fft.hip.h"
#include "fx_blocks.hip.h"
#include "fx_frame_kernel.hip.h"

// The library builds this file twice (build.py): FX_PART=1 holds the frame kernels up to 1024 points, the tail kernels
// and every host-side helper; FX_PART=2 holds only the frame kernels for 2048 and 4096 points.  The two halves want
// different compiler options (the scheduler's alternative register-pressure tracker is worth +3.5 % at 1024 points and
// costs 7 % at 4096).  Without FX_PART everything is one object (tools/).
#ifndef FX_PART
#define FX_PART 0
#endif
// FX_PART=3 holds fx_hop_kernel (one hop per call in one launch: the frame sections and the tail's device functions).
#if FX_PART == 0 || FX_PART == 1
#define FX_WITH_TAIL_KERNELS
#endif
#if FX_PART == 0 || FX_PART == 2 || FX_PART == 3
#include "fx_pair_kernel.hip.h"
#endif
#include "fx_tail_kernels.hip.h"        // (device functions everywhere; the tail kernels themselves where FX_WITH_TAIL_KERNELS is defined)
#if FX_PART == 0 || FX_PART == 3
#include "fx_hop_kernel.hip.h"
#endif


#if FX_PART == 0 || FX_PART == 3
bool hop_kernel_available(int n) { return n == 1024 || n == 2048 || n == 4096; }
// hipFuncSetAttribute is per device: every context prepares the kernels it will launch on its own device (fx_create),
// so there is no process-wide "already prepared" state to go stale when a second device or thread comes along.
hipError_t prepare_hop_kernel(int n)
{
    hipError_t e = hipSuccess;
    switch (n) {
        case 1024: return hop_prepare_t<1024>();
        case 2048: e = hop_prepare_t<2048>(); return e != hipSuccess ? e : hop_pair_prepare_t<2048>();
        case 4096: e = hop_prepare_t<4096>(); return e != hipSuccess ? e : hop_pair_prepare_t<4096>();
        default:   return hipSuccess;
    }
}
hipError_t launch_hop_kernel(int n, const FrameParams& p, const EpilogueParams& ep, const HopSignal& sig, hipStream_t stream, bool pairs)
{
    if (p.C <= 0) return hipSuccess;
    if (p.T != 1 || ep.T != 1 || ep.analysers != 3) return hipErrorInvalidValue;
    if (!hop_kernel_available(n)) return hipErrorInvalidValue;
    if (pairs && p.block_mode) return hipErrorInvalidValue;      // (the pair family reads hops: fx_push_samples re-blocks for it)
    if (pairs && n == 2048) return hop_pair_launch_t<2048>(p, ep, sig, stream);
    if (pairs && n == 4096) return hop_pair_launch_t<4096>(p, ep, sig, stream);
    switch (n) {
        case 1024: return hop_launch_t<1024>(p, ep, sig, stream);
        case 2048: return hop_launch_t<2048>(p, ep, sig, stream);
        default:   return hop_launch_t<4096>(p, ep, sig, stream);
    }
}
#endif

#if FX_PART != 3
// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int N> static size_t lds_bytes_t(int ch, int k, bool direct = false)
{
    return FrameLds<N>::bytes(ch, k, direct);
}


// per window size: the only host functions that name the frame kernels (so that each half of the build instantiates
// its own sizes and nothing else)
template <int N> hipError_t prepare_t()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    // frames + tails in one launch exists from 1024 points on (frame_tail_kernel_available): below that a frame is no longer than the
    // tail and the fused form loses either way (profiles/r04_live_cadence.txt), so those sizes are not instantiated
    if constexpr (N >= 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_tail_kernel<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        // the block-fed one-frame forms (FrameParams::block_mode)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_tail_kernel<N, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if constexpr (N == 1024) {           // the batch kernel's block-fed form: 1024 points only (see launch_t)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    } else {
        return hipSuccess;
    }
}

template <int N> hipError_t launch_t(const FrameParams& p, int analysers, hipStream_t stream)
{
    const size_t lds = lds_bytes_t<N>(p.ch_per_wg, p.waves_per_ch, p.direct_state != 0);
    const dim3 grid((unsigned) ((p.C + p.ch_per_wg - 1) / p.ch_per_wg) * (unsigned) (p.num_chunks > 1 ? p.num_chunks : 1)),
               block((unsigned) (p.ch_per_wg * p.waves_per_ch) * 64);
    if (p.direct_state) {
        if (analysers != 3 || p.T != 1 || p.waves_per_ch != 1 || p.num_chunks > 1) return hipErrorInvalidValue;
        if (p.block_mode) {
            if constexpr (N >= 1024) {
                if (!block_feed_valid(N, p)) return hipErrorInvalidValue;
                hipLaunchKernelGGL((fx_frame_kernel<N, true, true, true, true>), grid, block, lds, stream, p);
            } else {
                return hipErrorInvalidValue;
            }
        } else {
            hipLaunchKernelGGL((fx_frame_kernel<N, true, true, true>), grid, block, lds, stream, p);
        }
    }
    else if (p.block_mode) {
        // Calls of several frames: the batch kernel's block-fed form (both analysers).  Built for 1024 points only: the split sizes read a
        // window three times, sample by sample, and the two-piece addressing costs them more than the re-blocking pass it saves (measured,
        // us per call, block-fed / re-blocked: 4096 channels x 2048 points x 4000-sample blocks 302 / 293, 1024 x 4096 x 10 000 296 / 275;
        // 8192 x 1024 points x 4097 samples 428 / 481, x 30 719 samples 2869 / 3193).
        if constexpr (N == 1024) {
            if (analysers != 3 || !block_feed_valid(N, p)) return hipErrorInvalidValue;
            hipLaunchKernelGGL((fx_frame_kernel<N, true, true, false, true>), grid, block, lds, stream, p);
        } else {
            return hipErrorInvalidValue;
        }
    }
    else if (analysers == 3) hipLaunchKernelGGL((fx_frame_kernel<N, true, true>), grid, block, lds, stream, p);
    else if (analysers == 1) hipLaunchKernelGGL((fx_frame_kernel<N, true, false>), grid, block, lds, stream, p);
    else if (analysers == 2) hipLaunchKernelGGL((fx_frame_kernel<N, false, true>), grid, block, lds, stream, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// one frame per channel and the hop's tail in one launch (fx_frame_tail_kernel): p as for the DIRECT frame kernel
template <int N> hipError_t launch_tail_t(const FrameParams& p, const EpilogueParams& ep, hipStream_t stream)
{
    if (!p.direct_state || p.T != 1 || ep.T != 1 || ep.analysers != 3 || p.waves_per_ch != 1 || p.num_chunks > 1) return hipErrorInvalidValue;
    size_t lds = lds_bytes_t<N>(p.ch_per_wg, 1, true);
    const size_t tail = sizeof(f2) * FrameLds<N>::TW_ENTRIES + (size_t) ((p.ch_per_wg + TAIL_CHANNELS - 1) / TAIL_CHANNELS) * ONE_HOP_TAIL_BYTES;
    if (tail > lds) lds = tail;                               // (small windows: the transform buffers are smaller than a tail's ring copies)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((unsigned) ((p.C + p.ch_per_wg - 1) / p.ch_per_wg)), block((unsigned) p.ch_per_wg * 64);
    if constexpr (N >= 1024) {
        if (p.block_mode) {
            if (!block_feed_valid(N, p)) return hipErrorInvalidValue;
            hipLaunchKernelGGL((fx_frame_tail_kernel<N, true>), grid, block, lds, stream, p, ep);
        } else {
            hipLaunchKernelGGL((fx_frame_tail_kernel<N, false>), grid, block, lds, stream, p, ep);
        }
        return hipGetLastError();
    } else {
        return hipErrorInvalidValue;
    }
}

#if FX_PART == 0 || FX_PART == 2
// ---- the pair kernel (one frame across two wavefronts; windows of 2048 and 4096 points) ----
template <int N> static size_t pair_lds_bytes_t(int ch, int k)
{
    typedef PGeo<N> PG;
    return sizeof(f2) * N + (size_t) ch * sizeof(float) * PG::PREV_FLOATS + (size_t) ch * k * (PG::BUF_BYTES + PG::PAIR_EXTRA);
}
template <int N> static hipError_t launch_pair_t(const FrameParams& p, hipStream_t stream)
{
    const size_t lds = pair_lds_bytes_t<N>(p.ch_per_wg, p.waves_per_ch);
    const dim3 grid((unsigned) ((p.C + p.ch_per_wg - 1) / p.ch_per_wg) * (unsigned) (p.num_chunks > 1 ? p.num_chunks : 1)),
               block((unsigned) (p.ch_per_wg * p.waves_per_ch) * 128);
    hipLaunchKernelGGL((fx_pair_kernel<N>), grid, block, lds, stream, p);
    return hipGetLastError();
}
bool pair_kernel_available(int n) { return n == 2048 || n == 4096; }
int pair_kernel_max_pairs(int n) { return n == 2048 ? POcc<2048>::MAX_PAIRS : (n == 4096 ? POcc<4096>::MAX_PAIRS : 0); }
size_t pair_kernel_lds_bytes(int n, int ch, int k) { return n == 2048 ? pair_lds_bytes_t<2048>(ch, k) : (n == 4096 ? pair_lds_bytes_t<4096>(ch, k) : 0); }
hipError_t prepare_pair_kernel(int n)
{
    if (n == 2048) return hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_pair_kernel<2048>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (n == 4096) return hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_pair_kernel<4096>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return hipSuccess;
}
hipError_t launch_pair_kernel(int n, const FrameParams& p, hipStream_t stream)
{
    if (p.C <= 0 || p.T <= 0) return hipSuccess;
    if (!pair_kernel_available(n) || p.ch_per_wg < 1 || p.waves_per_ch < 1 || p.ch_per_wg * p.waves_per_ch > pair_kernel_max_pairs(n) || p.block_mode) return hipErrorInvalidValue;
    if (p.num_chunks > 1) {
        if (!p.queue || p.num_chunks > FX_MAX_CHUNKS || p.chunk_begin[0] != 0 || p.chunk_begin[p.num_chunks] != p.T) return hipErrorInvalidValue;
        for (int k = 0; k < p.num_chunks; k++) if (p.chunk_begin[k + 1] <= p.chunk_begin[k]) return hipErrorInvalidValue;
    }
    return n == 2048 ? launch_pair_t<2048>(p, stream) : launch_pair_t<4096>(p, stream);
}
#endif

#if FX_PART == 1
extern template hipError_t prepare_t<2048>();
extern template hipError_t prepare_t<4096>();
extern template hipError_t launch_t<2048>(const FrameParams&, int, hipStream_t);
extern template hipError_t launch_t<4096>(const FrameParams&, int, hipStream_t);
extern template hipError_t launch_tail_t<2048>(const FrameParams&, const EpilogueParams&, hipStream_t);
extern template hipError_t launch_tail_t<4096>(const FrameParams&, const EpilogueParams&, hipStream_t);
#elif FX_PART == 2
template hipError_t prepare_t<2048>();
template hipError_t prepare_t<4096>();
template hipError_t launch_t<2048>(const FrameParams&, int, hipStream_t);
template hipError_t launch_t<4096>(const FrameParams&, int, hipStream_t);
template hipError_t launch_tail_t<2048>(const FrameParams&, const EpilogueParams&, hipStream_t);
template hipError_t launch_tail_t<4096>(const FrameParams&, const EpilogueParams&, hipStream_t);
#endif

#if FX_PART != 2
template <int N> static void build_tw_t(const float* canon, float* out)
{
    typedef Plan<N> PL;
    const f2* c = reinterpret_cast<const f2*>(canon);
    f2* o = reinterpret_cast<f2*>(out);
    for (int i = 0; i < N; i++) o[i] = f2{0.f, 0.f};
    const int Rs[2] = {PL::R1, PL::R2}, Ls[2] = {PL::L1, PL::L2}, offs[2] = {PL::OFF1, PL::OFF2};
    for (int ps = 0; ps < 2; ps++) {
        const int R = Rs[ps], L0 = Ls[ps], off = offs[ps];
        for (int q = 1; q <= 3; q++)
            for (int k = 0; k < L0; k++) o[off + (q - 1) * L0 + k] = c[k * (N / (4 * L0)) * q];
        if (R == 16)
            for (int jin = 0; jin < 4; jin++)
                for (int q = 1; q <= 3; q++)
                    for (int k = 0; k < L0; k++)
                        o[off + 3 * L0 + (jin * 3 + q - 1) * L0 + k] = c[(k + L0 * jin) * (N / (16 * L0)) * q];
    }
    typedef Geo<N> G;
    if (G::RA == 16) {
        for (int jin = 1; jin <= 3; jin++)
            for (int q = 1; q <= 3; q++) o[PL::OFFA + (jin - 1) * 3 + q - 1] = c[jin * (N / 16) * q];
    } else if (G::RA == 8) {
        for (int q = 1; q <= 3; q++) o[PL::OFFA + q - 1] = c[(N / 8) * q];
    }
    static_assert(PL::OFFA + (G::RA == 16 ? 9 : (G::RA == 8 ? 3 : 0)) <= N, "pass-ordered twiddles must fit the N-entry table");
}

template <int N> static void fill_first_t(const float* ordered, float* out18)
{
    for (int i = 0; i < 18; i++) out18[i] = 0.0f;
    constexpr int cnt = Geo<N>::RA == 16 ? 9 : (Geo<N>::RA == 8 ? 3 : 0);
    for (int i = 0; i < 2 * cnt; i++) out18[i] = ordered[2 * Plan<N>::OFFA + i];
}

// ta[0..2] = W^1, W^2, W^3; ta[6..8] = W^3, W^6, W^9 with W = e^{-2*pi*i/16} (re, im pairs): the symmetry the
// 16-input first pass relies on to skip its mirrored butterfly.  Any libm within 1e-9 of the true cosines
// produces these floats; a table without it is refused rather than silently mis-transformed.
bool first_pass_twiddles_hermitian(int n, const float* t)
{
    if (Geo<1024>::RA != 16 || (n != 1024 && n != 4096)) return true;     // other sizes have no 16-input first pass
    return t[12] == -t[1] && t[13] == -t[0]             // W^3 = -i * conj(W^1)
        && t[14] == -t[2] && t[15] == t[3]              // W^6 = -conj(W^2)
        && t[16] == t[5] && t[17] == t[4];              // W^9 = i * conj(W^3)
}

// The 4096-point frame kernel's compact twiddle image forms last-pass rows 10 and 13 -- tw[2*k + 1024] and tw[2*k + 1536], k < 256 --
// as quarter turns of rows 4 and 7, tw[2*k] and tw[2*k + 512] (fx_fft.hip.h, CompactTw).  Exact cosines have that symmetry; the
// float table has it wherever libm's cos and sin round consistently, which is checked here entry by entry, not assumed.
bool twiddles_have_quarter_turn(int n, const float* canonical)
{
    if (n != 4096) return true;
    const f2* c = reinterpret_cast<const f2*>(canonical);
    for (int j = 2; j < 1024; j += 2)             // (j = 0: tw[N/4] is ((float) cos(pi/2), -1), never the quarter turn (-0, -1) of tw[0]; the kernel is handed it)
        if (!(c[j + 1024].x == c[j].y && c[j + 1024].y == -c[j].x)) return false;
    return true;
}

int build_twiddle_image(int n, const float* ordered, float* out)
{
    if (n != 4096) return 0;
    typedef CompactTw<4096> CT;
    static_assert(CT::ENTRIES <= 4096, "the image fits the room of a table");
    const f2* t = reinterpret_cast<const f2*>(ordered);
    f2* o = reinterpret_cast<f2*>(out);
    for (int e = 0; e < CT::ENTRIES; e++) { const int from = CT::source(e); o[e] = from >= 0 ? t[from] : f2{0.0f, 0.0f}; }
    return CT::ENTRIES;
}

void fill_first_pass_twiddles(int n, const float* ordered, float* out18)
{
    switch (n) {
        case 256:  fill_first_t<256>(ordered, out18); break;
        case 512:  fill_first_t<512>(ordered, out18); break;
        case 1024: fill_first_t<1024>(ordered, out18); break;
        case 2048: fill_first_t<2048>(ordered, out18); break;
        case 4096: fill_first_t<4096>(ordered, out18); break;
        default: break;
    }
}

void build_pass_twiddles(int n, const float* canonical, float* out)
{
    switch (n) {
        case 256:  build_tw_t<256>(canonical, out); break;
        case 512:  build_tw_t<512>(canonical, out); break;
        case 1024: build_tw_t<1024>(canonical, out); break;
        case 2048: build_tw_t<2048>(canonical, out); break;
        case 4096: build_tw_t<4096>(canonical, out); break;
        default: break;
    }
}

template <int N> static int max_waves_t() { return Occ<N>::MAX_THREADS / 64; }
int frame_kernel_max_waves(int n)
{
    switch (n) {
        case 256:  return max_waves_t<256>();
        case 512:  return max_waves_t<512>();
        case 1024: return max_waves_t<1024>();
        case 2048: return max_waves_t<2048>();
        case 4096: return max_waves_t<4096>();
        default:   return 1;
    }
}

// Measured on MI355X (see Occ<N>): up to 1024 points 1 channel x 8 waves (two workgroups per CU); 2048: 1 x 4 (two
// workgroups, 8 waves per CU: 12 waves as 3 x 4, 2 x 6 or 4 x 3 in one workgroup measured within 4 % of it, the LDS
// pipe and VALU issue being co-limiting by then); 4096: 1 x 8 (FrameLds<4096>: all the LDS holds).
void frame_kernel_preferred_shape(int n, int* ch, int* k)
{
    if (n <= 1024)      { *ch = 1; *k = 8; }
    else if (n == 2048) { *ch = 1; *k = 4; }
    else                { *ch = 1; *k = 8; }
}

size_t frame_kernel_lds_bytes(int n, int ch, int k, bool direct_state)
{
    switch (n) {
        case 256:  return lds_bytes_t<256>(ch, k, direct_state);
        case 512:  return lds_bytes_t<512>(ch, k, direct_state);
        case 1024: return lds_bytes_t<1024>(ch, k, direct_state);
        case 2048: return lds_bytes_t<2048>(ch, k, direct_state);
        case 4096: return lds_bytes_t<4096>(ch, k, direct_state);
        default:   return 0;
    }
}

hipError_t prepare_kernels(int n)
{
    switch (n) {
        case 256:  return prepare_t<256>();
        case 512:  return prepare_t<512>();
        case 1024: return prepare_t<1024>();
        case 2048: return prepare_t<2048>();
        case 4096: return prepare_t<4096>();
        default:   return hipErrorInvalidValue;
    }
}

hipError_t launch_frame_kernel(int n, const FrameParams& p, int analysers, hipStream_t stream)
{
    if (p.C <= 0 || p.T <= 0) return hipSuccess;
    if (p.ch_per_wg < 1 || p.waves_per_ch < 1 || p.ch_per_wg * p.waves_per_ch > frame_kernel_max_waves(n)) return hipErrorInvalidValue;
    if (p.num_chunks > 1) {
        if (!p.queue || p.num_chunks > FX_MAX_CHUNKS || p.chunk_begin[0] != 0 || p.chunk_begin[p.num_chunks] != p.T) return hipErrorInvalidValue;
        for (int k = 0; k < p.num_chunks; k++) if (p.chunk_begin[k + 1] <= p.chunk_begin[k]) return hipErrorInvalidValue;
    }
    switch (n) {
        case 256:  return launch_t<256>(p, analysers, stream);
        case 512:  return launch_t<512>(p, analysers, stream);
        case 1024: return launch_t<1024>(p, analysers, stream);
        case 2048: return launch_t<2048>(p, analysers, stream);
        case 4096: return launch_t<4096>(p, analysers, stream);
        default:   return hipErrorInvalidValue;
    }
}

bool frame_tail_kernel_available(int n) { return n >= 1024; }
hipError_t launch_frame_tail_kernel(int n, const FrameParams& p, const EpilogueParams& ep, hipStream_t stream)
{
    if (p.C <= 0) return hipSuccess;
    if (!frame_tail_kernel_available(n)) return hipErrorInvalidValue;
    if (p.ch_per_wg < 1 || p.ch_per_wg > frame_kernel_max_waves(n)) return hipErrorInvalidValue;
    switch (n) {
        case 256:  return launch_tail_t<256>(p, ep, stream);
        case 512:  return launch_tail_t<512>(p, ep, stream);
        case 1024: return launch_tail_t<1024>(p, ep, stream);
        case 2048: return launch_tail_t<2048>(p, ep, stream);
        case 4096: return launch_tail_t<4096>(p, ep, stream);
        default:   return hipErrorInvalidValue;
    }
}

hipError_t launch_epilogue_kernels(const EpilogueParams& p, hipStream_t stream)
{
    if (p.C <= 0 || p.T <= 0) return hipSuccess;
    if (p.T <= FUSED_TAIL_MAX_FRAMES) {
        hipLaunchKernelGGL(fx_tail_fused_kernel, dim3((unsigned) ((p.C + TAIL_CHANNELS - 1) / TAIL_CHANNELS)), dim3(64), 0, stream, p);
        return hipGetLastError();
    }
    const long long n1 = (long long) p.C * p.T;
    hipLaunchKernelGGL(fx_finalise_kernel, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, stream, p);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(fx_epilogue_kernel, dim3((unsigned) (p.C * ((p.T + EPI_TILE - 1) / EPI_TILE))), dim3(EPI_TILE), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long n2 = (long long) p.C * (p.T < HLEN ? p.T : HLEN) * FX_NUM_FEATURES;      // the call's newest rows go to the ring
    hipLaunchKernelGGL(fx_history_kernel, dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

#endif // FX_PART != 2
#endif // FX_PART != 3

} // namespace fxk

