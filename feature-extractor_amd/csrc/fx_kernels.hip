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
//   * fx_finalise_kernel (thread = frame) runs the scalar tail (pow, log10, sqrt, divisions);
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

#define FX_MARK(name) asm volatile("; FXMARK " name)

// Diagnostic build only (-DFX_STAMPS): per-section wave-cycle shares, summed into p.debug[section].
// Never part of the shipped library; stamped builds are not timed (MI355X guide, "In-kernel stamps").
#ifdef FX_STAMPS
#define FX_STAMP(idx) do { unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
                           stamp_acc[idx] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define FX_STAMP(idx) do {} while (0)
#endif

typedef float  __attribute__((ext_vector_type(2))) f2;
typedef float  __attribute__((ext_vector_type(4))) f4;

__device__ __forceinline__ void wave_fence()
{
    // LDS operations of one wavefront execute in order; this only stops the compiler from moving
    // LDS accesses of different lanes' data across the exchange point.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Re-materialise a lane-derived value so that nothing computed from it is hoisted out of the frame
// loop (loop-invariant code motion would otherwise keep hundreds of addresses, window gains and
// shuffle indices live across the whole loop and spill them).
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// ---------------------------------------------------------------------------------------------
// wavefront reductions (all lanes receive the result, as a wave-uniform value).
// DPP row operations instead of ds_bpermute shuffles: VALU latency instead of an LDS round trip per
// step.  quad_perm xor1, xor2 -> row_half_mirror -> row_mirror give every lane its 16-lane row total;
// row_bcast:15 / row_bcast:31 carry row totals forward so lane 63 holds the wave total.
// ---------------------------------------------------------------------------------------------
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143 };

template <int CTRL, int ROW_MASK> __device__ __forceinline__ int dpp_i(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_f(float old, float v)
{
    return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(old), __float_as_int(v)));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_d(double old, double v)
{
    const int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(old), __double2loint(v));
    const int hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast63(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Full-mask DPP read with bound_ctrl: lanes whose source does not exist receive 0 and nothing depends on
// the destination's old contents, so the compiler needs no move to initialise it.  With every row
// enabled row_bcast:15 gives rows 1..3 the total of the row before them and row_bcast:31 gives rows
// 2..3 lane 31's value, so after both steps row 3 holds (r3 + r2) + (r1 + r0): lane 63 has the wave
// total although rows 0..2 do not -- which is all a reduction needs.
template <int CTRL> __device__ __forceinline__ int dppz_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dppz_f(float v) { return __int_as_float(dppz_i<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dppz_d(double v)
{
    const int lo = dppz_i<CTRL>(__double2loint(v));
    const int hi = dppz_i<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
    v += dppz_d<DPP_XOR1>(v);
    v += dppz_d<DPP_XOR2>(v);
    v += dppz_d<DPP_HALF_MIRROR>(v);
    v += dppz_d<DPP_MIRROR>(v);
    v += dppz_d<DPP_BCAST15>(v);
    v += dppz_d<DPP_BCAST31>(v);
    return bcast63(v);
}
// (Moving the exchange steps to ds_swizzle -- the LDS crossbar instead of VALU DPP moves -- was measured 3 % slower:
// the LDS pipe is the kernel's second limiter.)
// maximum of values that are >= 0 (or NaN, which never wins -- as in `if (x > max) max = x`)
__device__ __forceinline__ float wave_maxf(float v)
{
    v = fmaxf(v, dppz_f<DPP_XOR1>(v));
    v = fmaxf(v, dppz_f<DPP_XOR2>(v));
    v = fmaxf(v, dppz_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dppz_f<DPP_MIRROR>(v));
    v = fmaxf(v, dppz_f<DPP_BCAST15>(v));
    v = fmaxf(v, dppz_f<DPP_BCAST31>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i(int v)
{
    int t;
    t = dpp_i<DPP_XOR1, 0xF>(v, v);        v = t < v ? t : v;
    t = dpp_i<DPP_XOR2, 0xF>(v, v);        v = t < v ? t : v;
    t = dpp_i<DPP_HALF_MIRROR, 0xF>(v, v); v = t < v ? t : v;
    t = dpp_i<DPP_MIRROR, 0xF>(v, v);      v = t < v ? t : v;
    t = dpp_i<DPP_BCAST15, 0xA>(v, v);     v = t < v ? t : v;
    t = dpp_i<DPP_BCAST31, 0xC>(v, v);     v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_sum_i(int v)
{
    v += dpp_i<DPP_XOR1, 0xF>(0, v);
    v += dpp_i<DPP_XOR2, 0xF>(0, v);
    v += dpp_i<DPP_HALF_MIRROR, 0xF>(0, v);
    v += dpp_i<DPP_MIRROR, 0xF>(0, v);
    v += dpp_i<DPP_BCAST15, 0xA>(0, v);
    v += dpp_i<DPP_BCAST31, 0xC>(0, v);
    return __builtin_amdgcn_readlane(v, 63);
}

// value of a wave-uniform lane (readlane needs the index in an SGPR)
__device__ __forceinline__ int lane_get(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ float lane_get(float v, int l) { return __int_as_float(lane_get(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_get(double v, int l)
{
    return __hiloint2double(lane_get(__double2hiint(v), l), lane_get(__double2loint(v), l));
}
// lane i receives lane i-1's value (wave_shr:1); lane 0 receives `first`
enum { DPP_WAVE_SHR1 = 0x138, DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118 };
__device__ __forceinline__ int shift_up1(int v, int first) { return dpp_i<DPP_WAVE_SHR1, 0xF>(first, v); }
__device__ __forceinline__ float shift_up1(float v, float first) { return dpp_f<DPP_WAVE_SHR1, 0xF>(first, v); }
__device__ __forceinline__ double shift_up1(double v, double first) { return dpp_d<DPP_WAVE_SHR1, 0xF>(first, v); }

// inclusive prefix sum over lanes (Kogge-Stone inside each 16-lane row, row totals carried by row_bcast)
__device__ __forceinline__ int wave_scan_incl_i(int v)
{
    v += dpp_i<DPP_ROW_SHR1, 0xF>(0, v);
    v += dpp_i<DPP_ROW_SHR2, 0xF>(0, v);
    v += dpp_i<DPP_ROW_SHR4, 0xF>(0, v);
    v += dpp_i<DPP_ROW_SHR8, 0xF>(0, v);
    v += dpp_i<DPP_BCAST15, 0xA>(0, v);
    v += dpp_i<DPP_BCAST31, 0xC>(0, v);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LDS images
//   complex image: position p at p + (p >> 4)            (one float2 of padding per 16)
//   real image   : sample  n at n + 4 * (n >> 4)         (16 B of padding per 16 floats, keeps
//                                                          16-byte alignment of 4-sample groups)
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int cpad(int p) { return p + (p >> 4); }
__host__ __device__ constexpr int rpad(int n) { return n + ((n >> 4) << 2); }

template <int N> struct Geo {
    static constexpr int M      = N / 2;            // numMagnitudes (ref SpectralCharacteristics.h:104)
    static constexpr int P      = N / 64;           // samples per lane
    static constexpr int U      = M / 64;           // bins per lane
    // ONE LDS buffer per wavefront, reused as: real image of the frame (rpad layout), complex image
    // of each transform (cpad layout), v / running-sum arrays of the lag scan, harmonic scratch.
    static constexpr int CBUF   = cpad(N) + 2;      // float2 elements (>= rpad(N)+8 floats, >= 2N+8 floats)
    static constexpr int LOG2N  = (N == 256) ? 8 : (N == 512) ? 9 : (N == 1024) ? 10 : (N == 2048) ? 11 : 12;
    // first FFT pass: R inputs per item, G items per lane, R*G == P
    static constexpr int RA     = (N == 256) ? 4 : ((N == 512 || N == 2048) ? 8 : 16);
    static constexpr int LOG2RA = (RA == 4) ? 2 : (RA == 8) ? 3 : 4;
    static constexpr int ITEMS_A = N / RA;
    static constexpr int GA     = ITEMS_A / 64;
    static constexpr int IDIG   = (LOG2N - LOG2RA) / 2;   // base-4 digits of an item index
};

// U consecutive floats of a lane, moved with the widest LDS access the alignment allows (a stride-U
// ds_read_b32 pattern would be an 8-way bank conflict for U = 8)
template <int U> __device__ __forceinline__ void lds_load_block(const float* base, float (&out)[U])
{
    if (U % 4 == 0) {
#pragma unroll
        for (int j = 0; j < U; j += 4) {
            const f4 v = *reinterpret_cast<const f4*>(base + j);
            out[j] = v.x; out[j + 1] = v.y; out[j + 2] = v.z; out[j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < U; j += 2) {
            const f2 v = *reinterpret_cast<const f2*>(base + j);
            out[j] = v.x; out[j + 1] = v.y;
        }
    }
}
template <int U> __device__ __forceinline__ void lds_store_block(float* base, const float (&in)[U])
{
    if (U % 4 == 0) {
#pragma unroll
        for (int j = 0; j < U; j += 4) *reinterpret_cast<f4*>(base + j) = f4{in[j], in[j + 1], in[j + 2], in[j + 3]};
    } else {
#pragma unroll
        for (int j = 0; j < U; j += 2) *reinterpret_cast<f2*>(base + j) = f2{in[j], in[j + 1]};
    }
}

// ---------------------------------------------------------------------------------------------
// FFT: the butterfly DAG of juce::FFT (kiss-style decimation in time, factors 4,...,4[,2]; SURVEY.md
// App. A.1), regrouped into register-resident passes.  Arithmetic is plain fp32 multiply / add /
// subtract -- never fused -- on the same operands and the same table twiddles as the reference, so
// every output is bit-identical to the CPU transform (up to the sign of an exact zero).
// ---------------------------------------------------------------------------------------------
// a * t (forward) or a * conj(t) (inverse: the inverse table is the exact conjugate, cos being even
// and sin odd).  Reference form: (a.r*t.r - a.i*t.i, a.r*t.i + a.i*t.r).
template <bool INV> __device__ __forceinline__ f2 twmul(f2 a, f2 t)
{
    // Three packed instructions.  hipcc does not form the mixed per-half negation by itself (it emits
    // two v_pk_add + a v_mov instead), so the VOP3P modifiers are spelled out:
    //   forward: p = (a.r*t.r, a.r*t.i), q = (a.i*(-t.i), a.i*t.r), result = p + q
    //   inverse: p = (a.r*t.r, a.r*(-t.i)), q = (a.i*t.i, a.i*t.r), result = p + q
    // (-x)*y == -(x*y) and p + (-q) == p - q exactly, so the roundings are those of the reference form.
    f2 p, q, r;
    if (INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(a), "v"(t));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(q) : "v"(a), "v"(t));
    }
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(p), "v"(q));
    return r;
}
// the same for a purely real a = (r, 0): the products with 0 only contribute exact zeros
template <bool INV> __device__ __forceinline__ f2 twmul_real(float r, f2 t)
{
    const f2 p = f2{r, r} * t;
    return INV ? f2{p.x, -p.y} : p;
}

// (a.x + b.y, a.y - b.x) if NEG_HI, else (a.x - b.y, a.y + b.x): a -/+ i*b
template <bool NEG_HI> __device__ __forceinline__ f2 pk_add_rot(f2 a, f2 b)
{
    f2 r;
    if (NEG_HI) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// butterfly4 after the three twiddle products s0, s1, s2
template <bool INV>
__device__ __forceinline__ void bfly4_core(f2& d0, f2& d1, f2& d2, f2& d3, f2 s0, f2 s1, f2 s2)
{
    const f2 s3 = s0 + s2;
    const f2 s4 = s0 - s2;
    const f2 s5 = d0 - s1;
    const f2 a = d0 + s1;
    d2 = a - s3;
    d0 = a + s3;
    // d1 = s5 -/+ i*s4, d3 = s5 +/- i*s4: one packed add each with the second operand's halves swapped and
    // one of them negated (x - y == x + (-y) exactly).  Left to itself hipcc computes all four sums and
    // differences and re-pairs the halves with moves.
    if (INV) {
        d1 = pk_add_rot<false>(s5, s4);     // (s5.x - s4.y, s5.y + s4.x)
        d3 = pk_add_rot<true>(s5, s4);      // (s5.x + s4.y, s5.y - s4.x)
    } else {
        d1 = pk_add_rot<true>(s5, s4);
        d3 = pk_add_rot<false>(s5, s4);
    }
}
// butterfly4 on four REAL inputs with unit twiddles (first stage of a real-input transform)
template <bool INV>
__device__ __forceinline__ void bfly4_real(float a0, float a1, float a2, float a3, f2& d0, f2& d1, f2& d2, f2& d3)
{
    const float s3 = a1 + a3, s4 = a1 - a3, s5 = a0 - a2, a = a0 + a2;
    d2 = f2{a - s3, 0.0f};
    d0 = f2{a + s3, 0.0f};
    d1 = f2{s5, INV ? s4 : -s4};
    d3 = f2{s5, INV ? -s4 : s4};
}

// Twiddle storage.  The N-entry table of the reference, tw[i] = ((float)cos, (float)sin)(-2*pi*i/N),
// is re-ordered on the host into the order the passes read it (fx_kernels.h, build_pass_twiddles), so
// that the 64 lanes of a pass read consecutive entries (conflict-free) at compile-time offsets:
//   later pass (R, L0), element index i = jin + LREL*(q + 4*g):
//     stage 1 (LREL = 1): [OFF + (q-1)*L0 + k]                      = tw[k * N/(4*L0) * q]
//     stage 2 (LREL = 4): [OFF + 3*L0 + (jin*3 + q-1)*L0 + k]       = tw[(k + L0*jin) * N/(16*L0) * q]
//   first pass constants:  [OFFA + (jin-1)*3 + q-1]                 = tw[jin * N/(4*R1) * q]
template <int N> struct Plan {
    static constexpr int R1 = 16;
    static constexpr int L1 = (N == 256) ? 4 : ((N == 512 || N == 2048) ? 8 : 16);
    static constexpr int R2 = (N >= 2048) ? 16 : 4;
    static constexpr int L2 = N / R2;
    static constexpr int OFF1 = 0;
    static constexpr int OFF2 = 15 * L1;
    static constexpr int OFFA = OFF2 + (R2 == 16 ? 15 : 3) * L2;
};

// offset of element i of an item inside the padded complex image, relative to cpad(base):
// cpad(base + L0*i) - cpad(base) is a compile-time constant because base = blk*(R*L0) + k, k < L0
__host__ __device__ constexpr int item_off(int L0, int i) { return L0 * i + (L0 >= 16 ? (L0 / 16) * i : ((L0 * i) >> 4)); }

// A later pass: every item of R elements (stride L0) is loaded from the complex image, its 1 or 2
// radix-4 stages run in registers, and it is stored back to the same positions.
template <int N, int R, int L0, int TWOFF, bool INV>
__device__ __forceinline__ void fft_pass(f2* cbuf, const f2* tw, int lane)
{
    lane = opaque(lane);
    constexpr int ITEMS = N / R;
    for (int it = lane; it < ITEMS; it += 64) {
        f2 e[R];
        const int k = it % L0;
        const int base = (it / L0) * (R * L0) + k;
        f2* img = cbuf + cpad(base);
        const f2* t1 = tw + TWOFF + k;
#pragma unroll
        for (int i = 0; i < R; i++) e[i] = img[item_off(L0, i)];
        // stage 1: butterflies over i = q + 4*g
        {
            const f2 w1 = t1[0], w2 = t1[L0], w3 = t1[2 * L0];
#pragma unroll
            for (int g = 0; g < R / 4; g++)
                bfly4_core<INV>(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3],
                                twmul<INV>(e[4 * g + 1], w1), twmul<INV>(e[4 * g + 2], w2), twmul<INV>(e[4 * g + 3], w3));
        }
        if constexpr (R == 16) {
            // stage 2: butterflies over i = jin + 4*q
            const f2* t2 = t1 + 3 * L0;
#pragma unroll
            for (int jin = 0; jin < 4; jin++) {
                const f2 w1 = t2[(jin * 3 + 0) * L0], w2 = t2[(jin * 3 + 1) * L0], w3 = t2[(jin * 3 + 2) * L0];
                bfly4_core<INV>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12],
                                twmul<INV>(e[jin + 4], w1), twmul<INV>(e[jin + 8], w2), twmul<INV>(e[jin + 12], w3));
            }
        }
#pragma unroll
        for (int i = 0; i < R; i++) img[item_off(L0, i)] = e[i];
    }
    wave_fence();
}

// base-4 digit reversal of the low 2*DIGITS bits
template <int DIGITS> __device__ __forceinline__ int rev4(int x)
{
    unsigned r = __builtin_bitreverse32((unsigned) x) >> (32 - 2 * DIGITS);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return (int) r;
}

// Bartlett gain, ref RealTimeAudioAnalysis.h:141-151: two JUCE gain ramps 0->1 and 1->0 whose
// float accumulation is exact for power-of-two N: w[i] = 2i/N (i < N/2), 2 - 2i/N (i >= N/2).
template <int N> __device__ __forceinline__ float bartlett_gain(int n)
{
    const float inc = 2.0f / N;
    return n < N / 2 ? (float) n * inc : 1.0f - (float) (n - N / 2) * inc;
}

// Sample (or bin) index that feeds input j of this lane's g-th first-pass item: the mixed-radix
// digit reversal of juce::FFT's decimation in time.  For a fixed (g, j) the 64 lanes cover 64
// consecutive indices, so LDS / global accesses in this order are conflict-free / coalesced.
template <int N> __device__ __forceinline__ int first_pass_index(int lane, int g, int j)
{
    typedef Geo<N> G;
    const int revj = (G::RA == 4) ? j : (G::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3));
    return rev4<G::IDIG>(lane + 64 * g) + G::ITEMS_A * revj;
}

// The same position inside the padded real image, split into a per-lane base and a compile-time step:
// ITEMS_A is a multiple of 16, so rpad(low + ITEMS_A*r) = rpad(low) + (ITEMS_A + ITEMS_A/4)*r and the
// RA accesses of an item are one address register plus immediate offsets.
template <int N> __device__ __forceinline__ int first_pass_rbase(int lane, int g) { return rpad(rev4<Geo<N>::IDIG>(lane + 64 * g)); }
template <int N> __host__ __device__ constexpr int first_pass_rstep(int j)
{
    return (Geo<N>::ITEMS_A + Geo<N>::ITEMS_A / 4)
         * ((Geo<N>::RA == 4) ? j : (Geo<N>::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3)));
}
static_assert(Geo<256>::ITEMS_A % 16 == 0 && Geo<512>::ITEMS_A % 16 == 0 && Geo<1024>::ITEMS_A % 16 == 0, "rpad splits only at multiples of 16");

// First pass: the lane's P REAL inputs are already in registers in first_pass_index order (imag = 0,
// as in performRealOnlyForwardTransform and in PitchAnalyser's re*re spectrum).  Stages at length 1
// have unit twiddles and real operands; the stage after them sees real operands in half of its
// butterflies.  Results go to the complex image.
template <int N, bool INV>
__device__ __forceinline__ void fft_first_pass(const float (&xin)[Geo<N>::P], f2* cbuf, const float (&ftw)[18], int lane)
{
    typedef Geo<N> G;
    constexpr int R = G::RA;
    lane = opaque(lane);
    f2 ta[9];                      // wave-uniform: kernel arguments, not LDS
#pragma unroll
    for (int i = 0; i < 9; i++) ta[i] = f2{ftw[2 * i], ftw[2 * i + 1]};
#pragma unroll
    for (int g = 0; g < G::GA; g++) {
        const float* x = &xin[g * R];
        f2 e[R];
        if constexpr (R == 4) {
            bfly4_real<INV>(x[0], x[1], x[2], x[3], e[0], e[1], e[2], e[3]);
        } else if constexpr (R == 8) {
            // radix-2 at length 1 (unit twiddle, real): pairs (2g', 2g'+1)
            float r[8];
#pragma unroll
            for (int h = 0; h < 4; h++) { r[2 * h] = x[2 * h] + x[2 * h + 1]; r[2 * h + 1] = x[2 * h] - x[2 * h + 1]; }
            // radix-4 at length 2: legs i = jin + 2*q
            bfly4_real<INV>(r[0], r[2], r[4], r[6], e[0], e[2], e[4], e[6]);
            e[1] = f2{r[1], 0.0f};
            bfly4_core<INV>(e[1], e[3], e[5], e[7], twmul_real<INV>(r[3], ta[0]), twmul_real<INV>(r[5], ta[1]), twmul_real<INV>(r[7], ta[2]));
        } else {
            // radix-4 at length 1: groups (4g', .., 4g'+3)
#pragma unroll
            for (int h = 0; h < 4; h++)
                bfly4_real<INV>(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3], e[4 * h], e[4 * h + 1], e[4 * h + 2], e[4 * h + 3]);
            // radix-4 at length 4: legs i = jin + 4*q; jin = 0 and 2 have real operands
            {
                f2 o0, o1, o2, o3;
                bfly4_real<INV>(e[0].x, e[4].x, e[8].x, e[12].x, o0, o1, o2, o3);
                e[0] = o0; e[4] = o1; e[8] = o2; e[12] = o3;
            }
            bfly4_core<INV>(e[1], e[5], e[9], e[13], twmul<INV>(e[5], ta[0]), twmul<INV>(e[9], ta[1]), twmul<INV>(e[13], ta[2]));
            bfly4_core<INV>(e[2], e[6], e[10], e[14], twmul_real<INV>(e[6].x, ta[3]), twmul_real<INV>(e[10].x, ta[4]), twmul_real<INV>(e[14].x, ta[5]));
            bfly4_core<INV>(e[3], e[7], e[11], e[15], twmul<INV>(e[7], ta[6]), twmul<INV>(e[11], ta[7]), twmul<INV>(e[15], ta[8]));
        }
        f2* img = cbuf + cpad((lane + 64 * g) * R);
#pragma unroll
        for (int i = 0; i < R; i++) img[i] = e[i];         // R <= 16 contiguous positions: no pad inside
    }
    wave_fence();
}

// What the last pass leaves in the wave's LDS buffer.
enum { OUT_RE_LOW = 1,        // float re[M] (plain layout): all the harmonic analyser reads (ref HarmonicCharacteristics.h:63)
       OUT_RE_LOW_MAXABS = 2, // the same + max(|re|,|im|) over bins [0, M/2) returned per lane (ref SpectralCharacteristics.h:153)
       OUT_POWER = 3,         // float re*re of all N bins in the real image (rpad layout): ref PitchAnalyser.h:97-103
       OUT_LAG = 4 };         // float v[s] = (re_s/N)^2 * s, s in [0,N), and v[N] from imag[0] (plain layout): ref :119-123

// Last pass -- radix 4 at length N/4 (N <= 1024) or radix 16 at length N/16 -- with all of a lane's items in
// registers (128 VGPRs of them at N = 4096, which runs one wave per SIMD anyway), fused with the consumer of the spectrum, so the full complex image is never written back and
// re-read: the spectral / harmonic analysers only read re of bins < N/2, the pitch analyser only re*re, the
// lag search only the squared, lag-weighted real part.
template <int N, bool INV, int OUT>
__device__ __forceinline__ float fft_last_pass_fused(f2* cbuf, const f2* tw, int lane, float scale)
{
    typedef Plan<N> PL;
    constexpr int R = PL::R2, L0 = PL::L2, GI = (N / R) / 64, TWOFF = PL::OFF2, M = N / 2;
    lane = opaque(lane);
    f2 e[GI][R];
#pragma unroll
    for (int g = 0; g < GI; g++) {
        const f2* img = cbuf + cpad(lane + 64 * g);
#pragma unroll
        for (int i = 0; i < R; i++) e[g][i] = img[item_off(L0, i)];
    }
    wave_fence();                 // the wave has read the whole complex image; the buffer may be rewritten
    float* fbuf = reinterpret_cast<float*>(cbuf);
    float aux = 0.0f;
#pragma unroll
    for (int g = 0; g < GI; g++) {
        const int k = lane + 64 * g;
        const f2* t1 = tw + TWOFF + k;
        {
            const f2 w1 = t1[0], w2 = t1[L0], w3 = t1[2 * L0];
#pragma unroll
            for (int h = 0; h < R / 4; h++)
                bfly4_core<INV>(e[g][4 * h], e[g][4 * h + 1], e[g][4 * h + 2], e[g][4 * h + 3],
                                twmul<INV>(e[g][4 * h + 1], w1), twmul<INV>(e[g][4 * h + 2], w2), twmul<INV>(e[g][4 * h + 3], w3));
        }
        if constexpr (R == 16) {
            const f2* t2 = t1 + 3 * L0;
#pragma unroll
            for (int jin = 0; jin < 4; jin++) {
                const f2 w1 = t2[(jin * 3 + 0) * L0], w2 = t2[(jin * 3 + 1) * L0], w3 = t2[(jin * 3 + 2) * L0];
                bfly4_core<INV>(e[g][jin], e[g][jin + 4], e[g][jin + 8], e[g][jin + 12],
                                twmul<INV>(e[g][jin + 4], w1), twmul<INV>(e[g][jin + 8], w2), twmul<INV>(e[g][jin + 12], w3));
            }
        }
        // e[g][i] is bin k + L0*i
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int bin = k + L0 * i;
            if (OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS) {
                if (i < R / 2) fbuf[bin] = e[g][i].x;                          // bins >= N/2 are never read
                if (OUT == OUT_RE_LOW_MAXABS && i < R / 4) aux = fmaxf(aux, fmaxf(fabsf(e[g][i].x), fabsf(e[g][i].y)));
            } else if (OUT == OUT_POWER) {
                fbuf[rpad(bin)] = e[g][i].x * e[g][i].x;
            } else {
                const float d = e[g][i].x * scale;
                fbuf[bin] = d * d * (float) bin;
            }
        }
        if (OUT == OUT_LAG && g == 0 && lane == 0) { const float d = e[0][0].y * scale; fbuf[N] = d * d * (float) N; }
    }
    (void) M;
    wave_fence();
    return aux;
}

// Whole transform of one wavefront: P real inputs per lane (first-pass order) -> OUT (see above).
template <int N, bool INV, int OUT>
__device__ __forceinline__ float fft_from_regs(const float (&xin)[Geo<N>::P], f2* cbuf, const f2* tw, const float (&ftw)[18],
                                               int lane, float scale = 0.0f)
{
    typedef Plan<N> PL;
    fft_first_pass<N, INV>(xin, cbuf, ftw, lane);
    fft_pass<N, PL::R1, PL::L1, PL::OFF1, INV>(cbuf, tw, lane);
    return fft_last_pass_fused<N, INV, OUT>(cbuf, tw, lane, scale);
}

// ---------------------------------------------------------------------------------------------
// frame load: global -> real LDS image, 16 B per lane, coalesced
// ---------------------------------------------------------------------------------------------
template <int HALF>
__device__ __forceinline__ void load_half(const void* src, int sample_format, float gain, bool apply_gain,
                                          float* rbuf, int dst_off, float* tail_out, int lane)
{
    // HALF is a multiple of 128 samples; 4 samples per lane per step
    for (int i = lane * 4; i < HALF; i += 256) {
        f4 v;
        if (sample_format == FX_SAMPLE_F16) {
            const uint2 raw = *reinterpret_cast<const uint2*>(static_cast<const __half*>(src) + i);
            const __half2 a = *reinterpret_cast<const __half2*>(&raw.x);
            const __half2 b = *reinterpret_cast<const __half2*>(&raw.y);
            const float2 fa = __half22float2(a), fb = __half22float2(b);
            v = f4{fa.x, fa.y, fb.x, fb.y};
        } else {
            v = *reinterpret_cast<const f4*>(static_cast<const float*>(src) + i);
        }
        if (apply_gain) v *= gain;                       // ref AudioDataCollector.h:88
        *reinterpret_cast<f4*>(&rbuf[rpad(dst_off + i)]) = v;
        if (tail_out) *reinterpret_cast<f4*>(tail_out + i) = v;
    }
}

// Both halves of a window with every global load issued before the first one is consumed (one memory
// round trip per frame instead of one per 1 KB piece).  F16_A / F16_B: sample format of the source of
// the first / second half (the carried-over tail is always fp32).  N >= 512.
template <int N, bool F16_A, bool F16_B>
__device__ __forceinline__ void load_window(const void* src_a, const void* src_b, float gain_a, float gain_b,
                                            float* rbuf, float* tail_out, int lane)
{
    constexpr int HALF = N / 2, QH = HALF / 256;
    uint4 ra[QH], rb[QH];
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        if (F16_A) { const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const __half*>(src_a) + i); ra[q] = uint4{v.x, v.y, 0u, 0u}; }
        else       ra[q] = *reinterpret_cast<const uint4*>(static_cast<const float*>(src_a) + i);
    }
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        if (F16_B) { const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const __half*>(src_b) + i); rb[q] = uint4{v.x, v.y, 0u, 0u}; }
        else       rb[q] = *reinterpret_cast<const uint4*>(static_cast<const float*>(src_b) + i);
    }
    auto widen = [](uint4 r, bool f16) -> f4 {
        if (f16) {
            const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&r.x));
            const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
            return f4{a.x, a.y, b.x, b.y};
        }
        return f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
    };
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        const f4 v = widen(ra[q], F16_A) * gain_a;             // ref AudioDataCollector.h:88 (x * 1.0f is exact)
        *reinterpret_cast<f4*>(&rbuf[rpad(i)]) = v;
    }
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        const f4 v = widen(rb[q], F16_B) * gain_b;
        *reinterpret_cast<f4*>(&rbuf[rpad(HALF + i)]) = v;
        if (tail_out) *reinterpret_cast<f4*>(tail_out + i) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// the frame kernel
// ---------------------------------------------------------------------------------------------
struct FlatProd { double mant; int exp; };   // value = mant * 2^exp, mant in [0.5,1) (or 0)

__device__ __forceinline__ FlatProd fp_mul(FlatProd a, double m)
{
    // multiply and renormalise; exact up to one rounding of the mantissa product
    const double p = a.mant * m;
    FlatProd r;
    r.exp = a.exp + __builtin_amdgcn_frexp_exp(p);
    r.mant = __builtin_amdgcn_frexp_mant(p);
    return r;
}
__device__ __forceinline__ FlatProd fp_mul2(FlatProd a, FlatProd b)
{
    const double p = a.mant * b.mant;
    FlatProd r;
    r.exp = a.exp + b.exp + __builtin_amdgcn_frexp_exp(p);
    r.mant = __builtin_amdgcn_frexp_mant(p);
    return r;
}

// waves per SIMD the register allocator must leave room for (LDS bounds residency as well)
#ifndef FX_OCC_SMALL
#define FX_OCC_SMALL 4
#endif
template <int N> struct Occ {
    static constexpr int WAVES_PER_SIMD = N <= 1024 ? FX_OCC_SMALL : (N == 2048 ? 2 : 1);
    // N = 4096 fits at most 3 waves per workgroup in the LDS; a 256-thread bound lets it use the whole
    // register file at one wave per SIMD instead of spilling
    static constexpr int MAX_THREADS = N == 4096 ? 256 : 512;
};

// SPEC / HARM: which of the reference's two analysers run (RealTimeSpectralAnalyser,
// RealTimeHarmonicAnalyser -- both by default, as AnalyserTrackController constructs them)
template <int N, bool SPEC, bool HARM>
__global__ void __launch_bounds__(Occ<N>::MAX_THREADS, Occ<N>::WAVES_PER_SIMD)
fx_frame_kernel(const FrameParams p)
{
    typedef Geo<N> G;
    constexpr int M = G::M, P = G::P, U = G::U, HALF = N / 2;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2*    tw   = reinterpret_cast<f2*>(smem);                              // [N]
    float* prev = reinterpret_cast<float*>(tw + N);                         // [M]  re of the last accepted frame
    int*   turn = reinterpret_cast<int*>(prev + M);                         // [4]
    FramePart* parts = reinterpret_cast<FramePart*>(turn + 4);              // [waves] per-frame results, filled as they appear
    unsigned char* per_wave = reinterpret_cast<unsigned char*>(parts + (blockDim.x >> 6));
    constexpr size_t WAVE_BYTES = sizeof(f2) * G::CBUF;

    const int nwaves = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6;
    const int lane0 = threadIdx.x & 63;
    const int c = blockIdx.x;
    const int T = p.T;

    f2*    cbuf = reinterpret_cast<f2*>(per_wave + WAVE_BYTES * wave);
    float* rbuf = reinterpret_cast<float*>(cbuf);      // the same memory viewed as the real image

    // workgroup prologue: twiddle table + this channel's flux state into LDS
    for (int i = threadIdx.x; i < N; i += blockDim.x) tw[i] = reinterpret_cast<const f2*>(p.tw)[i];
    for (int i = threadIdx.x; i < M; i += blockDim.x) prev[i] = p.prev_re[(size_t) c * M + i];
    if (threadIdx.x == 0) turn[0] = 0;
    __syncthreads();

    const double nyquist = p.nyquist;
    const double rnyq = 1.0 / nyquist;
    const double frpb = nyquist / (double) M;          // ref SpectralCharacteristics.h:64,105
    const float  scale = 1.0f / (float) N;             // JUCE inverse scale

#ifdef FX_STAMPS
    unsigned long long stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last) :: "memory");
#endif
    for (int t = wave; t < T; t += nwaves) {
        int lane = opaque(lane0);
        FX_STAMP(11);
        // uniform per-frame results go to LDS as soon as they exist instead of occupying ~28 VGPRs
        // in every lane for the whole frame
        FramePart* fpl = parts + wave;
        if (lane == 0) { fpl->inh = 0.0; fpl->her_score = 0.0; fpl->sum_normed = 1.0; fpl->flags = 0; fpl->pad_ = 0; fpl->spare_ = 0.0; }

FX_MARK("load");
        // ---------------- a1: window assembly (ref RealTimeAudioAnalysis.h:205-219) ----------------
        {
            const size_t esz = p.sample_format == FX_SAMPLE_F16 ? 2 : 4;
            const unsigned char* in = static_cast<const unsigned char*>(p.in);
            float* tail_dst = (t == T - 1) ? p.tail_out + (size_t) c * HALF : nullptr;
            const bool f16 = p.sample_format == FX_SAMPLE_F16;
            const void* src_a; const void* src_b; float gain_a, gain_b; bool f16_a = f16;
            if (p.hop_mode) {
                gain_a = gain_b = p.gain;
                src_b = in + ((size_t) c * T + t) * HALF * esz;
                if (t == 0) { src_a = p.tail_in + (size_t) c * HALF; f16_a = false; gain_a = 1.0f; }   // tail is fp32, already gained
                else        src_a = in + ((size_t) c * T + (t - 1)) * HALF * esz;
            } else {
                gain_a = gain_b = 1.0f;
                src_a = in + ((size_t) c * T + t) * N * esz;
                src_b = static_cast<const unsigned char*>(src_a) + HALF * esz;
            }
            if constexpr (N >= 512) {
                if (f16_a && f16)       load_window<N, true,  true >(src_a, src_b, gain_a, gain_b, rbuf, tail_dst, lane);
                else if (f16)           load_window<N, false, true >(src_a, src_b, gain_a, gain_b, rbuf, tail_dst, lane);
                else                    load_window<N, false, false>(src_a, src_b, gain_a, gain_b, rbuf, tail_dst, lane);
            } else {
                load_half<HALF>(src_a, f16_a ? FX_SAMPLE_F16 : FX_SAMPLE_F32, gain_a, gain_a != 1.0f, rbuf, 0, nullptr, lane);
                load_half<HALF>(src_b, p.sample_format, gain_b, gain_b != 1.0f, rbuf, HALF, tail_dst, lane);
            }
            wave_fence();
        }

FX_MARK("rms");
        FX_STAMP(0);
        // ---------------- a2: RMS on the un-windowed frame (ref RealTimeAnalyser.h:207-208) ---------
        // the frame, in registers, in the order the first FFT pass consumes it; the LDS buffer is
        // free again after this read
        float xr[P];
#pragma unroll
        for (int g = 0; g < G::GA; g++)
#pragma unroll
            for (int j = 0; j < G::RA; j++) xr[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];
        wave_fence();
        float log_rms;
        {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < P; i++) s += (double) (xr[i] * xr[i]);
            s = wave_sum(s);
            const float rms = (float) sqrt(s / (double) N);
            // log10 of a float, correctly rounded: this value gates bins (`mag > 0.01*logRMS`), so it must
            // equal the CPU oracle's to the last bit (see oracle/fx_oracle.c)
#ifdef FX_EXP_SKIP_RMSLOG
            log_rms = __log10f(rms * 9.0f + 1.0f);
#else
            log_rms = (float) log10((double) (rms * 9.0f + 1.0f));
#endif
            if (lane == 0) fpl->log_rms = log_rms;
        }

        if constexpr (SPEC) {
        float spec_aux = 0.0f;
FX_MARK("spec_fft");
        FX_STAMP(1);
        // ---------------- spectral analyser (ref RealTimeAnalyser.h:212-224) -----------------------
        lane = opaque(lane);
        {
            float xw[P];                                                       // a3 Bartlett window
            // sample index of input j of item g is nlow + ITEMS_A*r(j) with nlow < ITEMS_A, so it lies in
            // the rising half iff r(j) < RA/2 and the gain is base + r*2/RA there, (1-base) - (r-RA/2)*2/RA
            // in the falling half -- exact dyadic arithmetic, identical to bartlett_gain<N>(index)
#pragma unroll
            for (int g = 0; g < G::GA; g++) {
                const float base = (float) rev4<G::IDIG>(lane + 64 * g) * (2.0f / N);
                const float nbase = 1.0f - base;
#pragma unroll
                for (int j = 0; j < G::RA; j++) {
                    const int r = (G::RA == 4) ? j : (G::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3));
                    const float gain = r < G::RA / 2 ? base + (float) r * (2.0f / G::RA)
                                                     : nbase - (float) (r - G::RA / 2) * (2.0f / G::RA);
                    xw[g * G::RA + j] = xr[g * G::RA + j] * gain;
                }
            }
            spec_aux = fft_from_regs<N, false, OUT_RE_LOW_MAXABS>(xw, cbuf, tw, p.first_tw, lane);   // a4
        }
FX_MARK("spec_sums");
        FX_STAMP(2);
        {
            // lane owns bins [U*lane, U*lane + U)
            float re[U];
            // ref SpectralCharacteristics.h:153: getMagnitude over the first M floats of the interleaved
            // buffer = max |re|, |im| over bins [0, M/2)
            float maxabs = spec_aux;
            lds_load_block<U>(reinterpret_cast<const float*>(cbuf) + U * lane, re);
            const double eps = 0.01 * (double) log_rms;                        // :108
            double mag_sum = 0.0, lhr = 0.0, wsum = 0.0, flat_sum = 0.0;
            float max_re = 0.0f;       // max |re|: (double) re^2 is exact and monotone in |re|, so max mag = max_re^2
            int cnt = 0;               // wave-uniform: bins that pass the flatness gate, counted from the compare masks
            // bins m <= M/5 (:86-87, inclusive) are the lanes below LQ entirely and the first LR + 1 bins of lane LQ:
            // the lane's share of `lhr` is its running magnitude sum at that point
            constexpr int LQ = (M / 5) / U, LR = (M / 5) % U;
#pragma unroll
            for (int j = 0; j < U; j++) {                                      // fillIntermediateValues :62-97
                const int m = U * lane + j;
                const double v = (double) re[j];
                const double mag = v * v;
                const double fc = (double) m * frpb + (frpb / 2.0);
                mag_sum += mag;
                if (j == LR) lhr = mag_sum;
                const bool gate = mag > eps;
                cnt += __builtin_popcountll(__ballot(gate));
                if (gate) flat_sum += mag;
                wsum += fc * mag;
                max_re = fmaxf(max_re, fabsf(re[j]));
            }
            lhr = lane < LQ ? mag_sum : (lane == LQ ? lhr : 0.0);
            mag_sum = wave_sum(mag_sum);
            lhr = wave_sum(lhr);
            wsum = wave_sum(wsum);
            flat_sum = wave_sum(flat_sum);
            max_re = wave_maxf(max_re);
            const double max_mag = (double) max_re * (double) max_re;
            maxabs = wave_maxf(maxabs);
            const bool accepted = mag_sum > 0.05;                              // :121-123

FX_MARK("flux");
            // ---- flux against the previous accepted frame; hand-off between waves ----
            double flux = 0.0;
            {
#ifndef FX_EXP_NOWAIT
                while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t)
                    __builtin_amdgcn_s_sleep(1);
#endif
                float pvf[U];
                lds_load_block<U>(prev + U * lane, pvf);
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const double pv = (double) pvf[j];
                    const double v = (double) re[j];
                    const double diff = v * v - pv * pv;                       // :76
                    if (diff > 0.0) flux += diff;                              // :77-79
                }
                if (accepted) lds_store_block<U>(prev + U * lane, re);         // :138 (only on the accepted path)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(turn, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            flux = wave_sum(flux);

            lane = opaque(lane);
FX_MARK("flatprod");
            // ---- flatness product: serial-order semantics of `magnitudeProduct *= binMagnitude` ----
            // (ref :92) including IEEE overflow (sticky inf) and gradual underflow (sticky 0):
            // exponent-extended prefix products locate the first prefix that leaves the normal
            // range; an overflow decides at once, an underflow is finished serially in IEEE double.
            double prod;
#ifdef FX_EXP_SKIP_FLATPROD
            prod = 1.0;
            if (false)
#endif
            {
                FlatProd loc = {0.5, 1};                                       // 1.0
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const double v = (double) re[j];
                    const double mag = v * v;
                    if (mag > eps) loc = fp_mul(loc, mag);
                }
                // inclusive / exclusive scan of lane totals in lane (= bin) order; identity = 1.0 = (0.5, 1)
                FlatProd inc = loc;
#define FX_FP_STEP(CTRL, ROW_MASK) { FlatProd nb; nb.mant = dpp_d<CTRL, ROW_MASK>(0.5, inc.mant); nb.exp = dpp_i<CTRL, ROW_MASK>(1, inc.exp); inc = fp_mul2(nb, inc); }
                FX_FP_STEP(DPP_ROW_SHR1, 0xF)
                FX_FP_STEP(DPP_ROW_SHR2, 0xF)
                FX_FP_STEP(DPP_ROW_SHR4, 0xF)
                FX_FP_STEP(DPP_ROW_SHR8, 0xF)
                FX_FP_STEP(DPP_BCAST15, 0xA)
                FX_FP_STEP(DPP_BCAST31, 0xC)
#undef FX_FP_STEP
                FlatProd exc;
                exc.mant = shift_up1(inc.mant, 0.5);
                exc.exp = shift_up1(inc.exp, 1);
                // replay the lane's chain from its true starting value, looking for the first prefix
                // outside the normal range:  value = mant*2^exp with mant in [0.5,1)
                //   overflow  : value >= 2^1024  <=> exp >= 1025
                //   subnormal : value <  2^-1022 <=> exp <= -1022
                int first_bad = 0x7fffffff;       // bin index of the first abnormal prefix
                int bad_kind = 0;                 // 1 overflow, 2 subnormal
                FlatProd run = exc;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const double v = (double) re[j];
                    const double mag = v * v;
                    if (mag > eps) {
                        run = fp_mul(run, mag);
                        if (first_bad == 0x7fffffff && run.mant != 0.0) {
                            if (run.exp >= 1025) { first_bad = U * lane + j; bad_kind = 1; }
                            else if (run.exp <= -1022) { first_bad = U * lane + j; bad_kind = 2; }
                        }
                    }
                }
                const int fb = wave_min_i(first_bad);
                const FlatProd total = {bcast63(inc.mant), __builtin_amdgcn_readlane(inc.exp, 63)};
                if (fb == 0x7fffffff) {
                    prod = ldexp(total.mant, total.exp);
                } else {
                    // which lane owns bin fb, and what happened there
                    const int owner = fb / U;
                    const int kind = lane_get(bad_kind, owner);
                    if (kind == 1) {
                        prod = __builtin_huge_val();                           // inf * positive finite stays inf
                    } else {
                        // value just before bin fb (normal), then IEEE double from fb onwards: the
                        // owner of bin fb continues through its own bins, hands the product to the next
                        // lane, and so on until it is exactly 0 (0 * finite stays 0) or the bins end
                        FlatProd before = exc;
#pragma unroll
                        for (int j = 0; j < U; j++) {
                            const double v = (double) re[j];
                            const double mag = v * v;
                            if (mag > eps && (U * lane + j) < fb) before = fp_mul(before, mag);
                        }
                        double pr = ldexp(lane_get(before.mant, owner), lane_get(before.exp, owner));
                        double tailf[U];           // this lane's factors from bin fb on (1.0 = not a factor; x * 1.0 is exact)
#pragma unroll
                        for (int j = 0; j < U; j++) {
                            const double v = (double) re[j];
                            const double mag = v * v;
                            tailf[j] = (mag > eps && (U * lane + j) >= fb) ? mag : 1.0;
                        }
                        for (int l = owner; l < 64; l++) {
                            double mine = pr;
#pragma unroll
                            for (int j = 0; j < U; j++) mine *= tailf[j];
                            pr = lane_get(mine, l);
                            if (pr == 0.0) break;
                        }
                        prod = pr;
                    }
                }
            }

FX_MARK("spec_pass2");
            // second pass over the lane's bins: spread needs the centroid, the slope needs the mean
            // (ref SpectralCharacteristics.h:135-139 and :182-188); everything after these sums is
            // scalar and is finished by fx_finalise_kernel
            {
                const float centroid = (float) (wsum / mag_sum);               // :127
                const double cn = (double) centroid * rnyq;
                const double mu = mag_sum * (1.0 / (double) M);
                double var = 0.0, vsum = 0.0;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const int m = U * lane + j;
                    const double v = (double) re[j];
                    const double mag = v * v;
                    const double fc = (double) m * frpb + (frpb / 2.0);
                    const double d = fc * rnyq - cn;
                    var += (d * d) * mag;
                    const double dv = mag - mu;
                    vsum += dv * dv;
                }
                var = wave_sum(var);
                vsum = wave_sum(vsum);
                if (lane == 0) { fpl->var = var; fpl->vsum = vsum; fpl->centroid = centroid; }
            }
            double max_e = (double) maxabs;                                    // :153
            if (max_mag > max_e) max_e = max_mag;                              // :161-162
            if (lane == 0) {
                fpl->mag_sum = mag_sum; fpl->lhr = lhr; fpl->flux = flux; fpl->flat_sum = flat_sum; fpl->prod = prod;
                fpl->max_e = max_e; fpl->wsum = wsum; fpl->cnt = (float) cnt;
            }
        }
        wave_fence();
        }

        if constexpr (HARM) {
FX_MARK("harm1");
        FX_STAMP(3);
        // ---------------- harmonic analyser, part 1: raw (un-windowed) spectrum ---------------------
        // ref RealTimeAnalyser.h:161 -- done before the low-pass overwrites the frame image
        lane = opaque(lane);
        fft_from_regs<N, false, OUT_RE_LOW>(xr, cbuf, tw, p.first_tw, lane);
        float hre[U];
        float h_left2, h_left1, h_right1;          // |re| of bins U*lane-2, U*lane-1, U*lane+U
        double h_sum = 0.0, h_max;
        float h_max_re = 0.0f;
        {
            const int b0 = U * lane;
            const float* relin = reinterpret_cast<const float*>(cbuf);
            lds_load_block<U>(relin + b0, hre);
            h_left2  = b0 >= 2 ? fabsf(relin[b0 - 2]) : 0.0f;
            h_left1  = b0 >= 1 ? fabsf(relin[b0 - 1]) : 0.0f;
            h_right1 = b0 + U < M ? fabsf(relin[b0 + U]) : 0.0f;
#pragma unroll
            for (int j = 0; j < U; j++) {                                      // ref HarmonicCharacteristics.h:61-69
                const double v = (double) hre[j];
                const double mag = v * v;
                h_sum += mag;
                h_max_re = fmaxf(h_max_re, fabsf(hre[j]));
            }
            h_sum = wave_sum(h_sum);
            h_max_re = wave_maxf(h_max_re);
            h_max = (double) h_max_re * (double) h_max_re;
        }
        wave_fence();

        // ---------------- pitch: low-pass -> window -> FFT -> re^2 -> inverse FFT -> lag -------------
        double f0;
        lane = opaque(lane);
        {
FX_MARK("lpf");
        FX_STAMP(4);
            // a10 AudioFilter::filterAudio, ref RealTimeAudioAnalysis.h:106-125:
            //   y[0] = x[0];  y[n] = (a*x[n]) + (b*y[n-1]) in fp32, strictly serial.
            // Lane l owns samples [P*l, P*l+P).  It starts KW samples early from a guess, and the
            // recurrence (|b| = 0.208) forgets the guess; the value it reaches at P*l-1 must be
            // bit-identical to what lane l-1 produced there, otherwise the chunk is redone from the
            // neighbour's value until every hand-over matches (exact by induction from lane 0).
            constexpr int KW = 16;
            const float a = p.lpf_a, b = p.lpf_b;
#pragma unroll
            for (int g = 0; g < G::GA; g++)
#pragma unroll
                for (int j = 0; j < G::RA; j++) (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)] = xr[g * G::RA + j];
            wave_fence();
            float x[P];
#pragma unroll
            for (int i = 0; i < P; i += 4) {
                const f4 v = *reinterpret_cast<const f4*>(&rbuf[rpad(P * lane + i)]);
                x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
            }
            float yin = 0.0f;                             // y[P*lane - 1] used as this chunk's input
            {
                const int first = P * lane;
#pragma unroll
                for (int q = 0; q < KW / 4; q++) {
                    const int n0 = first - KW + 4 * q;    // multiple of 4: the whole group is in range or not
                    if (n0 >= 0) {
                        const f4 v = *reinterpret_cast<const f4*>(&rbuf[rpad(n0)]);
                        const float w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; e++)       // sample 0 starts the filter exactly; the first
                            yin = (n0 + e == 0 || (q == 0 && e == 0)) ? w[e] : (a * w[e]) + (b * yin);   // warm-up sample is a guess
                    }
                }
            }
            wave_fence();
            float y[P];
            float ylast;
            {
                float yy = yin;
#pragma unroll
                for (int i = 0; i < P; i++) {
                    yy = (lane == 0 && i == 0) ? x[0] : (a * x[i]) + (b * yy);
                    y[i] = yy;
                }
                ylast = yy;
            }
            for (int iter = 0; iter < 64; iter++) {
                const float pe = shift_up1(ylast, 0.0f);
                const bool bad = lane > 0 && (__float_as_uint(pe) != __float_as_uint(yin));
                if (!__any(bad)) break;
                if (bad) {
                    yin = pe;
                    float yy = yin;
#pragma unroll
                    for (int i = 0; i < P; i++) { yy = (a * x[i]) + (b * yy); y[i] = yy; }
                    ylast = yy;
                }
            }
            // window the filtered frame (ref RealTimeAnalyser.h:157) and put it back in the real image.
            // A lane's P samples lie in one half of the window; the gains w0 + i*wstep are exact dyadic
            // numbers (so the fma rounds nothing) and equal bartlett_gain<N>(P*lane + i).
            lane = opaque(lane);
            const float w0 = bartlett_gain<N>(P * lane);
            const float wstep = lane < 32 ? (2.0f / N) : -(2.0f / N);
#pragma unroll
            for (int i = 0; i < P; i += 4) {
                f4 v;
                v.x = y[i]     * __builtin_fmaf(wstep, (float) i, w0);
                v.y = y[i + 1] * __builtin_fmaf(wstep, (float) (i + 1), w0);
                v.z = y[i + 2] * __builtin_fmaf(wstep, (float) (i + 2), w0);
                v.w = y[i + 3] * __builtin_fmaf(wstep, (float) (i + 3), w0);
                *reinterpret_cast<f4*>(&rbuf[rpad(P * lane + i)]) = v;
            }
            wave_fence();

FX_MARK("pitch_fft");
        FX_STAMP(5);
            lane = opaque(lane);
            float xf[P];
#pragma unroll
            for (int g = 0; g < G::GA; g++)
#pragma unroll
                for (int j = 0; j < G::RA; j++) xf[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];
            wave_fence();
            fft_from_regs<N, false, OUT_POWER>(xf, cbuf, tw, p.first_tw, lane);            // ref RealTimeAnalyser.h:160
FX_MARK("power");
        FX_STAMP(6);
            // a11 getComplexConjugateMultiplication, ref PitchAnalyser.h:83-108: re*re, imag := 0,
            // picked up directly in the order the inverse transform's first pass wants it
            lane = opaque(lane);
#pragma unroll
            for (int g = 0; g < G::GA; g++)
#pragma unroll
                for (int j = 0; j < G::RA; j++) xf[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];   // already squared
            wave_fence();
FX_MARK("ifft");
            fft_from_regs<N, true, OUT_LAG>(xf, cbuf, tw, p.first_tw, lane, scale);        // a12 inverse, ref :110-121
FX_MARK("vcalc");
        FX_STAMP(7);
            // v[s] = d[s]*d[s]*s, d = planar JUCE inverse output scaled by 1/N (ref :122-123).
            // Only s in [1, N] is ever read by the lag search; v[N] comes from imag[0].
            lane = opaque(lane);
            float* vbuf = rbuf;                                                // [N+1] plain layout
            float* sums = rbuf + N + 4;                                        // [N+1]; both fit in the buffer
            wave_fence();
FX_MARK("scan");
            // a13 running fp32 sum (ref PitchAnalyser.h:138-150) -- serial by definition, so one lane
            // adds, 64 samples at a time; after each block all lanes form cnd = v/sum (ref :146-154) for
            // that block and advance a14's search (ref :161-190), which usually ends long before N:
            //   first  = first s >= 2 with cnd[s] < 0.01
            //   stop   = first s' >= first with !(cnd[s'+1] < cnd[s'])   (or N-1)
            //   lag    = cnd[stop] <= cnd[stop+1] ? stop : stop+1        (ref :192-203)
            // otherwise the global minimum over [2, N), first occurrence (ref :171-175).
            float lag = -1.0f;
            {
                float run = 0.0f;                 // lane 0: the running sum
                float carry = 0.0f;               // cnd of the last sample of the previous block
                int first = 0x7fffffff;
                bool done = false;
                float best = 100.0f; int best_i = 0x7fffffff;
#ifdef FX_EXP_SKIP_SCAN
                for (int blk = 0; blk < (int) (scale * 0.5f) && !done; blk++) {
#else
                for (int blk = 0; blk < P && !done; blk++) {
#endif
                    if (lane == 0) {
#pragma unroll
                        for (int g = 0; g < 64; g += 4) {
                            const f4 v = *reinterpret_cast<const f4*>(&vbuf[64 * blk + g]);
                            f4 o;
                            if (g != 0 || blk != 0) run += v.x;               // the sum starts at sample 1
                            o.x = run;
                            run += v.y; o.y = run;
                            run += v.z; o.z = run;
                            run += v.w; o.w = run;
                            *reinterpret_cast<f4*>(&sums[64 * blk + g]) = o;
                        }
                    }
                    wave_fence();
                    const int s_ = 64 * blk + lane;
                    const float sm = sums[s_];
                    const float v = vbuf[s_];
                    const float c_ = (sm != 0.0f) ? v / sm : 0.0f;
                    const float p_ = shift_up1(c_, carry);                     // cnd of the previous sample
                    carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_), 63));
                    if (s_ >= 2 && c_ < best) { best = c_; best_i = s_; }
                    if (first == 0x7fffffff) {
                        const unsigned long long hit = __ballot(s_ >= 2 && c_ < 0.01f);
                        if (hit) first = 64 * blk + (int) __builtin_ctzll(hit);
                    }
                    if (first != 0x7fffffff) {
                        // sample s-1 ends the walk if it is past `first` and cnd does not keep falling
                        const unsigned long long st = __ballot(s_ - 1 >= first && !(c_ < p_));
                        if (st) {
                            const int src = (int) __builtin_ctzll(st);
                            const float pc = lane_get(p_, src), cc = lane_get(c_, src);
                            const int sstar = 64 * blk + src;
                            lag = (pc <= cc) ? (float) (sstar - 1) : (float) sstar;
                            done = true;
                        }
                    }
                }
                if (!done) {
                    if (first != 0x7fffffff) {
                        // the walk ran to N-1 (ref :178: sample + 1 < numSamples); compare with cnd[N]
                        float cn = 0.0f;
                        if (lane == 0) { run += vbuf[N]; cn = (run != 0.0f) ? vbuf[N] / run : 0.0f; }
                        cn = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(cn)));
                        lag = (carry <= cn) ? (float) (N - 1) : (float) N;
                    } else {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) {
                            const float ov = __shfl_xor(best, o, 64);
                            const int oi = __shfl_xor(best_i, o, 64);
                            if (ov < best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
                        }
                        lag = best_i == 0x7fffffff ? -1.0f : (float) best_i;
                    }
                }
            }
            f0 = (nyquist * 2.0) / (double) lag;                               // ref PitchAnalyser.h:57
            if (lane == 0) fpl->lag = lag;
        }
        wave_fence();

FX_MARK("harm2");
        FX_STAMP(8);
        // ---------------- harmonic analyser, part 2 (ref HarmonicCharacteristics.h:71-105) ----------
        lane = opaque(lane);
#ifdef FX_EXP_SKIP_HARM2
        if (h_sum < -1.0) {
#else
        if (!(h_sum < 0.005)) {                                                // :88-89
#endif
            float* normed  = reinterpret_cast<float*>(cbuf);                   // [M] floats
            int*   peaks   = reinterpret_cast<int*>(cbuf) + M;                 // [<= M] peak bins
            float* peak_re = reinterpret_cast<float*>(cbuf) + 2 * M;           // [<= M] re of those bins
            double mean_mag = h_sum / (double) M;                              // :86
            // binIsPeak's `mag > mean` (:132) is an exact tie when the spectrum is flat (an impulse at
            // sample 0 or N/2 of the window): then the last bit of the reference's serial sum (:61-66)
            // decides for every bin at once.  If any bin sits within rounding distance of the mean,
            // redo the sum in the reference's order, handing the running value from lane to lane.
            {
                // |re| within ~1e-6 of sqrt(mean) brackets every bin whose magnitude is within 1e-12 of the mean
                // (a wider band only means the exact recomputation below runs a little more often)
                const float root_mean = (float) sqrt(mean_mag);
                const float band = root_mean * 1e-6f;
                bool near = false;
#pragma unroll
                for (int j = 0; j < U; j++) near |= fabsf(fabsf(hre[j]) - root_mean) <= band;
                if (__any(near)) {
                    double run = 0.0;
                    for (int l = 0; l < 64; l++) {
                        double mine = run;
#pragma unroll
                        for (int j = 0; j < U; j++) { const double v = (double) hre[j]; mine += v * v; }
                        run = lane_get(mine, l);
                    }
                    h_sum = run;
                    mean_mag = h_sum / (double) M;
                }
            }
            unsigned peak_mask = 0;
            float nrm[U];
            const double r_hmax = 1.0 / h_max;
            const double sum_normed = h_sum * r_hmax;                          // :77 sum of mag / max over all bins
#pragma unroll
            for (int j = 0; j < U; j++) {
                const double v = (double) hre[j];
                const double mag = v * v;
                const double nm = mag * r_hmax;                                // :75 (mag / max, via one reciprocal)
                nrm[j] = (float) nm;
                // binIsPeak :127-145: above the mean and none of bins -2,-1,+1 larger (windows are
                // clipped at the ends, :136-138: no +1 neighbour for the last two bins).
                // mag = (double) re^2 is exact, so comparing |re| compares magnitudes exactly.
                const int m = U * lane + j;
                const float me = fabsf(hre[j]);
                const float l2 = j >= 2 ? fabsf(hre[j >= 2 ? j - 2 : 0]) : (j == 1 ? h_left1 : h_left2);
                const float l1 = j >= 1 ? fabsf(hre[j >= 1 ? j - 1 : 0]) : h_left1;
                const float r1 = j + 1 < U ? fabsf(hre[j + 1 < U ? j + 1 : 0]) : h_right1;
                bool pk = mag > mean_mag;
                if (m >= 2 && l2 > me) pk = false;
                if (m >= 1 && l1 > me) pk = false;
                if (m < M - 2 && r1 > me) pk = false;
                if (pk) peak_mask |= 1u << j;
            }
            lds_store_block<U>(normed + U * lane, nrm);
            // compact the peak list
            const int npk_lane = __popc(peak_mask);
            const int pre = wave_scan_incl_i(npk_lane);
            const int total_peaks = __builtin_amdgcn_readlane(pre, 63);
            int woff = pre - npk_lane;
#pragma unroll
            for (int j = 0; j < U; j++)
                if (peak_mask & (1u << j)) { peaks[woff] = U * lane + j; peak_re[woff] = hre[j]; woff++; }
            wave_fence();

            const double fr = nyquist / (double) M;                            // :93
            // calculateHarmonicEnergyCharacteristics :147-198 with numLower = 15, numHarmonics = 3:
            // 18 probes, one lane each; lane 18 divides f0 itself, which is getBinForFrequency(f0) (:246-249)
            double probe = 0.0;
            int f0_bin;
            {
                // f0 / pow(2, lane+1) is an exact scaling; f0 * h for the harmonics
                const double freq = lane < 15 ? ldexp(f0, -(lane + 1)) : (lane < 18 ? f0 * (double) (lane - 14) : f0);
                int bin = (int) floor(freq / fr);
                f0_bin = __builtin_amdgcn_readlane(bin, 18);
                if (lane < 15) { if (bin == f0_bin) bin = -1; }                // :163-164
                else if (lane < 18) { if (bin >= M) bin = -1; }                // :174-175 (monotone, so break == skip)
                else bin = -1;
                if (bin >= 0 && bin < M) {
                    // getMaxBinInNeighbourhood :200-210 : [max(0,c-2), min(c+2, M)), start value normed[c]
                    const int s0 = bin - 2 >= 0 ? bin - 2 : 0;
                    const int e0 = bin + 2 < M ? bin + 2 : M;
                    float mx = normed[bin];
                    for (int q = s0; q < e0; q++) { const float v = normed[q]; if (v > mx) mx = v; }
                    probe = (double) mx;
                }
            }
            const double score = wave_sum(probe);                              // / sum_normed, clamped: fx_finalise_kernel

            // calculateInharmonicity :212-244
            double inh = 0.0;
            if (f0 > 0.0) {                                                    // :98
                for (int i = lane; i < total_peaks; i += 64) {
                    const int bin = peaks[i];
                    if (bin == f0_bin) continue;                               // :220-221
                    double fs = (double) bin * fr;
                    if (fs == 0.0) fs = fr * 0.5;                              // :225-226
                    const double fe = (double) (bin + 1) * fr;
                    // getFrequencyRatio :251-259: higher / lower (1.0 when equal: x / x is exactly 1)
                    const double rs  = (fs > f0 ? fs : f0) / (fs > f0 ? f0 : fs);
                    const double re_ = (fe > f0 ? fe : f0) / (fe > f0 ? f0 : fe);
                    if (floor(rs) != floor(re_)) continue;                     // :232-233
                    const double r = rs < re_ ? rs : re_;
                    const double v = (double) peak_re[i];
                    inh += (r - floor(r)) * ((v * v) / h_sum);                 // :236-239
                }
            }
            inh = wave_sum(inh);
            if (lane == 0) { fpl->inh = inh; fpl->her_score = score; fpl->sum_normed = sum_normed; fpl->flags = 1; }
        }
        wave_fence();
        }

FX_MARK("store");
        FX_STAMP(9);
        wave_fence();
        static_assert(sizeof(FramePart) % 16 == 0, "16-byte stores");
#ifndef FX_EXPERIMENT_NOSTORE
        if (lane < (int) (sizeof(FramePart) / 16))
            reinterpret_cast<uint4*>(p.part + ((size_t) c * T + t))[lane] = reinterpret_cast<const uint4*>(fpl)[lane];
#endif
        wave_fence();
        FX_STAMP(10);
    }

#ifdef FX_STAMPS
    if (lane0 == 0 && p.debug)
        for (int i = 0; i < 16; i++) atomicAdd(p.debug + i, stamp_acc[i]);
#endif
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) p.prev_re[(size_t) c * M + i] = prev[i];
}

// ---------------------------------------------------------------------------------------------
// fx_finalise_kernel: thread = frame.  The scalar tail of calculateSpectralCharacteristicsFrom-
// Intermediates (ref SpectralCharacteristics.h:116-143), calculateNormalisedSpectralSlope (:189-199),
// the harmonic logs (ref HarmonicCharacteristics.h:101-105) and the slot mapping of
// RealTimeAnalyser.h:165-172,219-224.  Output: raw[C][T][12] with the onset slot still 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
fx_finalise_kernel(const EpilogueParams p)
{
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long) p.C * p.T) return;
    const FramePart f = p.part[idx];
    const int M = p.window / 2;
    const double nyquist = p.nyquist;
    float out[FX_NUM_FEATURES];
#pragma unroll
    for (int i = 0; i < FX_NUM_FEATURES; i++) out[i] = 0.0f;
    out[FX_RMS] = f.log_rms;
    const double eps = 0.01 * (double) f.log_rms;                              // :108

    const bool spec = p.analysers & 1, harm = p.analysers & 2;
    if (spec && f.mag_sum > 0.05) {                                            // :121-123
        const float centroid = f.centroid;
        const double dcnt = (double) f.cnt;
        const double inv_n = 1.0 / (dcnt > 0.0 ? dcnt : 1.0);                  // :129-130
        const float flatness = f.flat_sum > eps ? (float) (pow(f.prod, inv_n) / (inv_n * f.flat_sum)) : 0.0f;   // :57-60
        out[FX_FLATNESS] = (float) log10((double) flatness * 9.0 + 1.0);       // :132
        const float cc = centroid / (float) (nyquist / 2.0);                   // :133
        out[FX_CENTROID] = (float) log10((double) (cc * 9.0f + 1.0f));         // :134
        const double cn = (double) centroid / nyquist;
        const float max_spread = (float) (cn * (1.0 - cn));                    // :140
        out[FX_SPREAD] = (float) ((f.var / f.mag_sum) / (double) max_spread);  // :141
        out[FX_LER] = (float) (f.lhr / f.mag_sum);                             // :125
        const float max_flux = (float) (M * (M + 1)) / 2.0f;                   // :111
        out[FX_FLUX] = (float) (f.flux / (double) max_flux);
    }
    if (spec && f.max_e > 0.0001) {                                            // :165-167
        // normedEnergy = mag / max (ref :172): the sums over bins were taken before the division
        const double rmax = 1.0 / f.max_e;
        const double se = f.mag_sum * rmax;
        const double frpb = nyquist / (double) M;
        const double s1 = (f.wsum - (frpb / 2.0) * f.mag_sum) / frpb;          // sum m * mag
        const double ps = s1 * rmax;                                           // :175
        const double mean_e = se / (double) M;                                 // :177
        const double ev = f.vsum * rmax * rmax / (double) M;                   // :187,190
        const double bin_std = sqrt(p.bin_var), e_std = sqrt(ev);              // :191-192
        const double r = (ps - ((double) M * mean_e * 0.5)) / (double) ((float) M - 1.0f) * e_std * bin_std;   // :195
        out[FX_SLOPE] = (float) (r * (bin_std / e_std));                       // :198
    }
    if (harm) {
        const double f0 = (nyquist * 2.0) / (double) f.lag;                    // ref PitchAnalyser.h:57
        out[FX_F0] = (float) (f0 / 5000.0);                                    // ref RealTimeAnalyser.h:165-166
    }
    if (harm && (f.flags & 1)) {
        double her = f.her_score / f.sum_normed;                               // ref HarmonicCharacteristics.h:186-188
        if (her > 1.0) her = 1.0;
        if (her < 0.0) her = 0.0;
        her = (double) (float) her;                                            // struct of floats, :197
        const float log_her = (float) log10(her * 9.0 + 1.0);                  // :101
        out[FX_HER] = log_her;
        out[FX_OER] = log_her;                                                 // ref RealTimeAnalyser.h:171 writes HER into the OER slot
        out[FX_INHARM] = (float) log10(f.inh * 9.0 + 1.0);                     // :102
    }
    f4* dst = reinterpret_cast<f4*>(p.raw + idx * FX_NUM_FEATURES);
    dst[0] = f4{out[0], out[1], out[2], out[3]};
    dst[1] = f4{out[4], out[5], out[6], out[7]};
    dst[2] = f4{out[8], out[9], out[10], out[11]};
}

// ---------------------------------------------------------------------------------------------
// smoothing (ValueHistory, ref RealTimeAudioAnalysis.h:40-96; AudioFeatures, ref
// RealTimeAnalyser.h:70-88) and onset detection (ref SpectralCharacteristics.h:249-306,
// RealTimeAnalyser.h:236-242).
//
// A ValueHistory of length L after its k-th insert holds the last min(k, L) inserted values,
// oldest first, padded on the left with the zeros it was created with; getTotal() adds them left to
// right in fp32.  Every smoothed value and every onset decision of frame t is therefore a pure
// function of the raw values of frames t-HLEN+1 .. t, and all (channel, frame) pairs are evaluated
// in parallel: thread = (channel, frame).  Frames before this call come from hist_in.
// ---------------------------------------------------------------------------------------------
struct RawView {
    const float* raw; const float* hist; int T; long long frames_before;
    // raw value of slot s at frame index tau relative to this call (tau may be negative);
    // frames before the stream began read as "not recorded"
    __device__ __forceinline__ bool valid(int tau) const { return frames_before + (long long) tau >= 0 && tau > -HLEN - 1; }
    __device__ __forceinline__ float get(int tau, int s) const
    {
        return tau >= 0 ? raw[(size_t) tau * FX_NUM_FEATURES + s] : hist[(size_t) (HLEN + tau) * FX_NUM_FEATURES + s];
    }
};

// smoothed RMS as AudioFeatures::getValue(enRMS) returns it when `pushes_after` of frame tau's
// two RMS inserts have happened (shared AudioFeatures: two inserts per hop; isolated: one)
__device__ __forceinline__ float rms_value(const RawView& v, int tau, int order_mode, int pushes_of_tau)
{
    float total = 0.0f;
    long long recorded;
    if (order_mode == FX_ORDER_ISOLATED) {
        // spectral analyser's own AudioFeatures: one insert per frame, window = frames tau-9 .. tau
#pragma unroll
        for (int i = 0; i < 10; i++) { const int f = tau - 9 + i; total += v.valid(f) ? v.get(f, FX_RMS) : 0.0f; }
        recorded = v.frames_before + tau + 1;
    } else {
        // inserts are numbered 2*g (first analyser of frame g) and 2*g+1; the newest insert present
        // is 2*tau + pushes_of_tau - 1 and the history holds the 10 newest
        const long long newest = 2 * (v.frames_before + tau) + pushes_of_tau - 1;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const long long q = newest - 9 + i;                     // global insert index
            const long long g = q >> 1;                             // its frame (floor for q >= 0)
            const int f = (int) (g - v.frames_before);
            total += (q >= 0 && v.valid(f)) ? v.get(f, FX_RMS) : 0.0f;
        }
        recorded = newest + 1;
    }
    if (recorded > 10) recorded = 10;
    return total / (float) recorded;
}

__global__ void __launch_bounds__(256)
fx_epilogue_kernel(const EpilogueParams p)
{
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long) p.C * p.T) return;
    const int c = (int) (idx / p.T), t = (int) (idx % p.T);
    RawView v;
    v.raw = p.raw + (size_t) c * p.T * FX_NUM_FEATURES;
    v.hist = p.hist_in + (size_t) c * HLEN * FX_NUM_FEATURES;
    v.T = p.T;
    v.frames_before = p.frames_before;

    float sm[FX_NUM_FEATURES];
    float rw[FX_NUM_FEATURES];
#pragma unroll
    for (int s = 0; s < FX_NUM_FEATURES; s++) rw[s] = v.get(t, s);

    const bool spec = p.analysers & 1, harm = p.analysers & 2;
    // with a single analyser the RMS slot gets one insert per hop, like an isolated AudioFeatures
    const int order_mode = (spec && harm) ? p.order_mode : FX_ORDER_ISOLATED;
    const float never = __int_as_float(0x7fc00000);                  // getValue() of a slot nobody wrote: 0.0f / 0
    // 10-deep slots (ref RealTimeAnalyser.h:73)
    long long rec10 = p.frames_before + t + 1; if (rec10 > 10) rec10 = 10;
#pragma unroll
    for (int s = 0; s < FX_NUM_FEATURES; s++) {
        if (s == FX_ONSET || s == FX_FLUX || s == FX_RMS) continue;
        const bool harm_slot = s == FX_F0 || s == FX_HER || s == FX_OER || s == FX_INHARM;
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; i++) { const int f = t - 9 + i; total += v.valid(f) ? v.get(f, s) : 0.0f; }
        sm[s] = (harm_slot ? harm : spec) ? total / (float) rec10 : never;
    }
    sm[FX_FLUX] = spec ? (0.0f + rw[FX_FLUX]) / 1.0f : never;        // history length 1
    // RMS after every analyser of this hop has inserted (what the OSC timer samples)
    sm[FX_RMS] = rms_value(v, t, order_mode, 2);

    // OnsetDetector::detectOnset, ref SpectralCharacteristics.h:249-306.  The detector's histories
    // hold (getValue(enFlux), getValue(enRMS)) as seen by detectOnset() of each frame
    // (ref RealTimeAnalyser.h:236-242): flux of that frame, and the RMS mean at that moment.
    const int L = p.onset_window;
    const int rms_pushes_at_detect = (order_mode == FX_ORDER_HARMONIC_THEN_SPECTRAL) ? 2 : 1;
    const long long g = p.frames_before + t;
    long long recorded = g - p.onset_reset_frame + 1;
    if (recorded > L) recorded = L;
    bool onset = false;
    if (spec && recorded >= L && L > 0) {                           // :253-258 both histories full
        int cand = L - 1;                                           // :263-266
        const bool use_amp = p.onset_type == FX_ONSET_AMPLITUDE || p.onset_type == FX_ONSET_COMBINATION;
        const bool use_flux = p.onset_type == FX_ONSET_SPECTRAL || p.onset_type == FX_ONSET_COMBINATION;
        if (use_flux) cand = L / 2;
        const float cand_amp = rms_value(v, t - L + 1 + cand, order_mode, rms_pushes_at_detect);
        const float cand_sf = (0.0f + v.get(t - L + 1 + cand, FX_FLUX)) / 1.0f;
        bool ok = !(cand_amp < 0.01f);                              // :271-274
        float tot_amp = 0.0f, tot_flux = 0.0f;
#pragma unroll 1
        for (int i = 0; i < L; i++) {                               // :260-261 totals, :276-289 neighbours
            const int f = t - L + 1 + i;
            const float amp_i = rms_value(v, f, order_mode, rms_pushes_at_detect);
            const float flx_i = (0.0f + v.get(f, FX_FLUX)) / 1.0f;
            tot_amp += amp_i;
            tot_flux += flx_i;
            if (i != cand) {
                if (amp_i >= cand_amp && use_amp) ok = false;
                if (flx_i >= cand_sf && use_flux) ok = false;
            }
        }
        const float mean_flux = tot_flux / (float) recorded;
        const float mean_amp = tot_amp / (float) recorded;
        const bool on_sf = cand_sf > mean_flux * p.onset_multiplier;    // :291-292
        const bool on_amp = cand_amp > mean_amp * p.onset_multiplier;
        bool res = false;
        if (p.onset_type == FX_ONSET_AMPLITUDE) res = on_amp;
        else if (p.onset_type == FX_ONSET_SPECTRAL) res = on_sf;
        else if (p.onset_type == FX_ONSET_COMBINATION) res = on_amp && on_sf;
        onset = ok && res;
    }
    rw[FX_ONSET] = onset ? 1.0f : 0.0f;
    sm[FX_ONSET] = spec ? (0.0f + rw[FX_ONSET]) / 1.0f : never;      // history length 1

    const size_t o = ((size_t) c * p.T + t) * FX_NUM_FEATURES;
    if (p.out_raw) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.out_raw[o + s] = rw[s];
    }
    if (p.out_smoothed) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.out_smoothed[o + s] = sm[s];
    }
    if (t == p.T - 1) {
#pragma unroll
        for (int s = 0; s < FX_NUM_FEATURES; s++) p.latest[(size_t) c * FX_NUM_FEATURES + s] = sm[s];
    }
}

// carry the newest HLEN frames of raw values over to the next call
__global__ void __launch_bounds__(256)
fx_history_kernel(const EpilogueParams p)
{
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long) p.C * HLEN * FX_NUM_FEATURES;
    if (idx >= total) return;
    const int s = (int) (idx % FX_NUM_FEATURES);
    const int h = (int) ((idx / FX_NUM_FEATURES) % HLEN);
    const int c = (int) (idx / ((long long) FX_NUM_FEATURES * HLEN));
    const int tau = p.T - HLEN + h;                                  // frame relative to this call
    float val;
    if (tau >= 0) val = p.raw[((size_t) c * p.T + tau) * FX_NUM_FEATURES + s];
    else          val = p.hist_in[((size_t) c * HLEN + (HLEN + tau)) * FX_NUM_FEATURES + s];
    p.hist_out[idx] = val;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
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

template <int N> static size_t lds_bytes_t(int waves)
{
    typedef Geo<N> G;
    return sizeof(f2) * N + sizeof(float) * G::M + 16 + (size_t) waves * (sizeof(FramePart) + sizeof(f2) * G::CBUF);
}

size_t frame_kernel_lds_bytes(int n, int waves)
{
    switch (n) {
        case 256:  return lds_bytes_t<256>(waves);
        case 512:  return lds_bytes_t<512>(waves);
        case 1024: return lds_bytes_t<1024>(waves);
        case 2048: return lds_bytes_t<2048>(waves);
        case 4096: return lds_bytes_t<4096>(waves);
        default:   return 0;
    }
}

template <int N> static hipError_t prepare_t()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, true, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_frame_kernel<N, false, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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

template <int N> static hipError_t launch_t(const FrameParams& p, int analysers, int waves, hipStream_t stream)
{
    const size_t lds = lds_bytes_t<N>(waves);
    const dim3 grid((unsigned) p.C), block((unsigned) waves * 64);
    if (analysers == 3)      hipLaunchKernelGGL((fx_frame_kernel<N, true, true>), grid, block, lds, stream, p);
    else if (analysers == 1) hipLaunchKernelGGL((fx_frame_kernel<N, true, false>), grid, block, lds, stream, p);
    else if (analysers == 2) hipLaunchKernelGGL((fx_frame_kernel<N, false, true>), grid, block, lds, stream, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_frame_kernel(int n, const FrameParams& p, int analysers, int waves, hipStream_t stream)
{
    if (p.C <= 0 || p.T <= 0) return hipSuccess;
    switch (n) {
        case 256:  return launch_t<256>(p, analysers, waves, stream);
        case 512:  return launch_t<512>(p, analysers, waves, stream);
        case 1024: return launch_t<1024>(p, analysers, waves, stream);
        case 2048: return launch_t<2048>(p, analysers, waves, stream);
        case 4096: return launch_t<4096>(p, analysers, waves, stream);
        default:   return hipErrorInvalidValue;
    }
}

hipError_t launch_epilogue_kernels(const EpilogueParams& p, hipStream_t stream)
{
    if (p.C <= 0 || p.T <= 0) return hipSuccess;
    const long long n1 = (long long) p.C * p.T;
    hipLaunchKernelGGL(fx_finalise_kernel, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, stream, p);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    hipLaunchKernelGGL(fx_epilogue_kernel, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long n2 = (long long) p.C * HLEN * FX_NUM_FEATURES;
    hipLaunchKernelGGL(fx_history_kernel, dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

} // namespace fxk
