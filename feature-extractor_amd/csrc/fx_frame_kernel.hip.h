// Re-materialisation points of lane-derived values (see opaque()).  (Letting the compiler hoist at some of them took 5 % of the
// instructions out at 1024 points and not a microsecond: round 3, profiles/r03_variants.txt (e).)
// A kernel with registers to spare (HOIST: the spectral analyser alone at 1024 points, 77 of the 128 VGPRs four wavefronts per SIMD allow) is
// transparent at the top of the frame loop (point 15) and at the start of the spectral section (point 0), so that the frame's addresses and the
// sixteen window gains of a lane are formed once per wavefront instead of once per frame: that kernel is bound by VALU issue (round 4:
// SQ_ACTIVE_INST_VALU x 4 waves = 1.01), so instructions are time -- 1.031 -> 0.981 ms per 524 288 frames (+5 %), 117 VGPRs, no scratch.
// (All points transparent: 128 VGPRs + 16 B of scratch, 1.014 ms; points 15 alone 1.010-1.025; 15 + 1: 1.017.  profiles/r04_variants.txt.)
#define FX_OPQ(n, x) ((HOIST && ((n) == 15 || (n) == 0)) ? (x) : opaque<N>(x))
// Costing builds only (-DFX_EXP_STOP_AT=k, tools/section_costs.sh): the frame's work ends at stop point k, the values it has
// formed so far kept alive; the differences of the instruction counters between consecutive k are the sections' dynamic
// costs.  Results are garbage and nothing waits for a frame that stopped early (the flux turn is taken at stop 8, which every
// frame of a build reaches or none does).
#ifdef FX_EXP_STOP_AT
#define FX_STOP(k, ...) if (FX_EXP_STOP_AT == (k)) { __VA_ARGS__; }      // (no do-while: the arguments `continue` the frame loop)
#else
#define FX_STOP(k, ...)
#endif
#define FX_KEEP(v) asm volatile("" :: "v"(v))
// fx_frame_kernel.hip.h -- frame load and fx_frame_kernel: every reduction over samples, bins and lags of a frame
// Included by fx_kernels.hip inside namespace fxk (one translation unit: every kernel sees the same
// inlined helpers); not a stand-alone header.

// ---------------------------------------------------------------------------------------------
// frame load: global -> real LDS image, 16 B per lane, coalesced
// ---------------------------------------------------------------------------------------------
// Sample formats on the wire (FX_SAMPLE_*, include/fx.h): fp32, fp16, or signed PCM of 16 bits (v / 2^15) or 24 bits packed in three
// bytes, little endian (v / 2^23) -- exactly what the WAV reader makes of such files (include/fx_wav.hpp, JUCE's int -> float scaling:
// the sample left-justified in an int32, times 2^-31), so integer audio crosses PCIe at two / three bytes a sample and loses
// nothing.  The carried-over window tail is always fp32.
__device__ __forceinline__ constexpr int sample_bytes(int fmt) { return fmt == FX_SAMPLE_F32 ? 4 : (fmt == FX_SAMPLE_S24 ? 3 : 2); }
// a 24-bit sample left-justified in 32 bits -> float: 24 significant bits, so the conversion and the power-of-two scaling are exact
__device__ __forceinline__ float from_left_justified(unsigned w) { return (float) (int) w * (1.0f / 2147483648.0f); }
// LAST_USE: the frame's last read of its window (the split sizes fetch it three times): a non-temporal load, so that the L2 lets go of
// a window nobody will ask for again before it lets go of one that is between its reads.  The windows in flight on an XCD are about
// the size of its L2 at 4096 points; measured there (1024 channels x 64 whole frames): counter traffic 2.24 -> 1.53 x the algorithmic
// bytes, kernel 1.557 -> 1.548 ms; on overlapping hops 1.32 -> 1.21 x; 2048 points unchanged (profiles/r04_4096.txt (8)).
template <int FMT, bool LAST_USE = false> __device__ __forceinline__ float widen_one(const void* at)
{
    if (LAST_USE && FMT == FX_SAMPLE_F32) return __builtin_nontemporal_load(static_cast<const float*>(at));
    if (LAST_USE && FMT == FX_SAMPLE_S16) return (float) (int) __builtin_nontemporal_load(static_cast<const short*>(at)) * (1.0f / 32768.0f);
    if (LAST_USE && FMT == FX_SAMPLE_F16) return __half2float(__ushort_as_half(__builtin_nontemporal_load(static_cast<const unsigned short*>(at))));
    if (FMT == FX_SAMPLE_F16) return __half2float(*static_cast<const __half*>(at));
    if (FMT == FX_SAMPLE_S16) return (float) (int) *static_cast<const short*>(at) * (1.0f / 32768.0f);
    if (FMT == FX_SAMPLE_S24) {
        // (three byte loads: a packed sample starts at any byte, and a wider load at the last sample of a buffer would read past its end)
        const unsigned char* b = static_cast<const unsigned char*>(at);
        return from_left_justified(((unsigned) b[0] << 8) | ((unsigned) b[1] << 16) | ((unsigned) b[2] << 24));
    }
    return *static_cast<const float*>(at);
}
// four consecutive samples: r = the 16 bytes of four floats, (r.x, r.y) = the 8 bytes of four 16-bit samples, or (r.x, r.y, r.z) = the
// 12 bytes of four packed 24-bit samples
template <int FMT> __device__ __forceinline__ f4 widen_four(uint4 r)
{
    if (FMT == FX_SAMPLE_S24)
        return f4{from_left_justified(r.x << 8), from_left_justified((r.y << 16) | ((r.x >> 16) & 0xff00u)),
                  from_left_justified((r.z << 24) | ((r.y >> 8) & 0xffff00u)), from_left_justified(r.z & 0xffffff00u)};
    if (FMT == FX_SAMPLE_F16) {
        const float2 a = __half22float2(*reinterpret_cast<const __half2*>(&r.x));
        const float2 b = __half22float2(*reinterpret_cast<const __half2*>(&r.y));
        return f4{a.x, a.y, b.x, b.y};
    }
    if (FMT == FX_SAMPLE_S16) {
        const f4 v = f4{(float) ((int) (r.x << 16) >> 16), (float) ((int) r.x >> 16), (float) ((int) (r.y << 16) >> 16), (float) ((int) r.y >> 16)};
        return v * (1.0f / 32768.0f);                                          // exact: a power of two
    }
    return f4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}
template <int FMT> __device__ __forceinline__ uint4 fetch_four(const void* src, int i)       // samples i .. i + 3 of src
{
    if (FMT == FX_SAMPLE_F32) return *reinterpret_cast<const uint4*>(static_cast<const float*>(src) + i);
    if (FMT == FX_SAMPLE_S24) {             // i is a multiple of 4: twelve bytes from a 4-byte boundary
        const uint3 v = *reinterpret_cast<const uint3*>(static_cast<const unsigned char*>(src) + 3 * i);
        return uint4{v.x, v.y, v.z, 0u};
    }
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(src) + i);
    return uint4{v.x, v.y, 0u, 0u};
}
// run `CALL` with the constants FA / FB set to the formats of a window's two halves: fb is the call's sample format, fa is
// either the same or fp32 (the tail)
#define FX_FORMATS(fa, fb, CALL) do {                                                                                              \
        if ((fb) == FX_SAMPLE_F16)      { if ((fa) == FX_SAMPLE_F16) { constexpr int FA = FX_SAMPLE_F16, FB = FX_SAMPLE_F16; CALL; }   \
                                          else                       { constexpr int FA = FX_SAMPLE_F32, FB = FX_SAMPLE_F16; CALL; } } \
        else if ((fb) == FX_SAMPLE_S16) { if ((fa) == FX_SAMPLE_S16) { constexpr int FA = FX_SAMPLE_S16, FB = FX_SAMPLE_S16; CALL; }   \
                                          else                       { constexpr int FA = FX_SAMPLE_F32, FB = FX_SAMPLE_S16; CALL; } } \
        else if ((fb) == FX_SAMPLE_S24) { if ((fa) == FX_SAMPLE_S24) { constexpr int FA = FX_SAMPLE_S24, FB = FX_SAMPLE_S24; CALL; }   \
                                          else                       { constexpr int FA = FX_SAMPLE_F32, FB = FX_SAMPLE_S24; CALL; } } \
        else                            { constexpr int FA = FX_SAMPLE_F32, FB = FX_SAMPLE_F32; CALL; }                                \
    } while (0)

template <int N, int HALF>
__device__ __forceinline__ void load_half(const void* src, int sample_format, float gain, bool apply_gain,
                                          float* rbuf, int dst_off, float* tail_out, int lane)
{
    // HALF is a multiple of 128 samples; 4 samples per lane per step
    for (int i = lane * 4; i < HALF; i += 256) {
        f4 v;
        if (sample_format == FX_SAMPLE_F16)      v = widen_four<FX_SAMPLE_F16>(fetch_four<FX_SAMPLE_F16>(src, i));
        else if (sample_format == FX_SAMPLE_S16) v = widen_four<FX_SAMPLE_S16>(fetch_four<FX_SAMPLE_S16>(src, i));
        else if (sample_format == FX_SAMPLE_S24) v = widen_four<FX_SAMPLE_S24>(fetch_four<FX_SAMPLE_S24>(src, i));
        else                                     v = widen_four<FX_SAMPLE_F32>(fetch_four<FX_SAMPLE_F32>(src, i));
        if (apply_gain) v *= gain;                       // ref AudioDataCollector.h:88
        *reinterpret_cast<f4*>(&rbuf[rimg<N>(dst_off + i)]) = v;
        if (tail_out) *reinterpret_cast<f4*>(tail_out + i) = v;
    }
}

// Both halves of a window with every global load issued before the first one is consumed (one memory
// round trip per frame instead of one per 1 KB piece).  FMT_A / FMT_B: sample format of the source of
// the first / second half (the carried-over tail is always fp32).  N >= 512.
// BLOCKS: the second half is the first N/2 samples of the stream `bs` (a channel's pending samples followed by its new block,
// fx_blocks.hip.h) instead of a hop at src_b.
template <int N, int FMT_A, int FMT_B, bool BLOCKS = false>
__device__ __forceinline__ double load_window(const void* src_a, const void* src_b, float gain_a, float gain_b,
                                              float* rbuf, float* tail_out, int lane, const BlockStream bs = BlockStream{}, const bool a_stream = false,
                                              const BlockStream bs_a = BlockStream{})
{
    double ssq = 0.0;           // this lane's share of getRMSLevel's sum (ref RealTimeAnalyser.h:207): float squares, double sum
    constexpr int HALF = N / 2, QH = HALF / 256;
    uint4 ra[QH], rb[QH];
#pragma unroll
    for (int q = 0; q < QH; q++) {
        // (BLOCKS, frames after a call's first: the first half is the hop before, out of the same stream -- bs_a, wave-uniform)
        if (BLOCKS && a_stream) ra[q] = stream_four<sample_bytes(FMT_A)>(bs_a, 256 * q + 4 * lane);
        else                           ra[q] = fetch_four<FMT_A>(src_a, 256 * q + 4 * lane);
    }
#pragma unroll
    for (int q = 0; q < QH; q++) {
        if constexpr (BLOCKS) rb[q] = stream_four<sample_bytes(FMT_B)>(bs, 256 * q + 4 * lane);
        else                  rb[q] = fetch_four<FMT_B>(src_b, 256 * q + 4 * lane);
    }
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        const f4 v = widen_four<FMT_A>(ra[q]) * gain_a;        // ref AudioDataCollector.h:88 (x * 1.0f is exact)
        *reinterpret_cast<f4*>(&rbuf[rimg<N>(i)]) = v;
        if (Geo<N>::SPLIT) ssq += (double) (v.x * v.x) + (double) (v.y * v.y) + (double) (v.z * v.z) + (double) (v.w * v.w);
    }
#pragma unroll
    for (int q = 0; q < QH; q++) {
        const int i = 256 * q + 4 * lane;
        const f4 v = widen_four<FMT_B>(rb[q]) * gain_b;
        *reinterpret_cast<f4*>(&rbuf[rimg<N>(HALF + i)]) = v;
        if (tail_out) *reinterpret_cast<f4*>(tail_out + i) = v;
        if (Geo<N>::SPLIT) ssq += (double) (v.x * v.x) + (double) (v.y * v.y) + (double) (v.z * v.z) + (double) (v.w * v.w);
    }
    return ssq;
}

// The same window straight from global memory into registers in the order the first FFT pass consumes it (split
// sizes: the raw frame is not kept in registers across the four transforms; it is fetched again -- from L2 -- when
// the spectral and the harmonic analyser need it).  For a fixed (g, j) the 64 lanes read 64 consecutive samples.
template <int N, int FMT_A, int FMT_B, bool LAST_USE = false, bool BLOCKS = false>
__device__ __forceinline__ void load_window_first_pass_order(const void* src_a, const void* src_b, float gain_a, float gain_b,
                                                             int lane, float (&x)[Geo<N>::P], const BlockStream bs = BlockStream{}, const bool a_stream = false,
                                                             const BlockStream bs_a = BlockStream{})
{
    typedef Geo<N> G;
    // sample index = rev4(lane + 64*g) + ITEMS_A*r(j), and rev4(lane + 64*g) = rev4(lane) + g (lane's three base-4 digits
    // go to the top of the item index, g's digit to the bottom): ONE per-lane offset, everything else is a compile-time
    // constant, so each load is a scalar base + that offset + an immediate
    static_assert(G::IDIG == 4 && G::GA == 4, "split sizes: 256 first-pass items, 4 per lane");
    // (byte offsets kept in 32 bits: scalar base + 32-bit lane offset + immediate is one addressing mode, a 64-bit
    // element index is a vector add per load)
    const unsigned low = (unsigned) rev4<G::IDIG>(lane);
    const unsigned off_a = low * (unsigned) sample_bytes(FMT_A), off_b = low * (unsigned) sample_bytes(FMT_B);
#pragma unroll
    for (int g = 0; g < G::GA; g++) {
#pragma unroll
        for (int j = 0; j < G::RA; j++) {
            const int r = (G::RA == 4) ? j : (G::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3));
            const bool second = r >= G::RA / 2;                                // low + g < ITEMS_A <= N/2
            const int k = g + G::ITEMS_A * (second ? r - G::RA / 2 : r);       // compile-time part of the sample index
            if constexpr (BLOCKS) {
                if (second) {
                    // (the sample's place in the stream decides which of its two pieces holds it: a compare and two selects per load)
                    x[g * G::RA + j] = widen_one<FMT_B, LAST_USE>(stream_sample<sample_bytes(FMT_B)>(bs, (int) low + k));
                    continue;
                }
                if (a_stream) {
                    x[g * G::RA + j] = widen_one<FMT_A, LAST_USE>(stream_sample<sample_bytes(FMT_A)>(bs_a, (int) low + k));
                    continue;
                }
            }
            const char* at = static_cast<const char*>(second ? src_b : src_a) + (second ? off_b : off_a) + k * sample_bytes(second ? FMT_B : FMT_A);
            x[g * G::RA + j] = second ? widen_one<FMT_B, LAST_USE>(at) : widen_one<FMT_A, LAST_USE>(at);
        }
    }
    if (gain_a != 1.0f || gain_b != 1.0f) {                                   // ref AudioDataCollector.h:88 (wave-uniform)
#pragma unroll
        for (int g = 0; g < G::GA; g++)
#pragma unroll
            for (int j = 0; j < G::RA; j++) {
                const int r = (G::RA == 4) ? j : (G::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3));
                x[g * G::RA + j] *= (r >= G::RA / 2) ? gain_b : gain_a;
            }
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

// a13 / a14 (ref PitchAnalyser.h:129-217), one 64-sample block at a time, for ONE wavefront whose lane l holds v[64*blk + l]:
// the running fp32 sum (:138-150) -- serial by definition -- as a chain across the lanes, then cnd = v/sum (:146-154) and
// a14's search (:161-190) advanced over the block:
//   first  = first s >= 2 with cnd[s] < 0.01
//   stop   = first s' >= first with !(cnd[s'+1] < cnd[s'])   (or N-1)
//   lag    = cnd[stop] <= cnd[stop+1] ? stop : stop+1        (ref :192-203)
//   otherwise the global minimum over [2, N), first occurrence (ref :171-175).
// Shared by the one-wavefront-per-frame kernel (FrameWave::lag_search) and the pair kernel (PairWave).
template <int N> struct LagSearch {
    float lag, run, carry, best;
    int first, best_i;
    bool done;
    __device__ __forceinline__ void begin()
    {
        lag = -1.0f;
        run = 0.0f;                 // the running sum after the previous block (wave-uniform)
        carry = 0.0f;               // cnd of the last sample of the previous block
        first = 0x7fffffff;
        done = false;
        best = 100.0f; best_i = 0x7fffffff;
    }
    // (Round 3 tried the running sum as a parallel prefix -- six DPP additions instead of the chain of 63 -- with every
    // comparison taken only outside a guard band of N * 6.5e-8 and an exact redo otherwise: bit-exact, 30 VALU instructions
    // fewer per frame at 1024 points and no faster; at 2048 / 4096 points the wider band sent enough searches to the redo
    // that instructions went UP 3-4 %.  The chain's latency is covered by the other wavefronts; its 63 issue slots are
    // 2 % of a frame.  Not kept.)
    __device__ __forceinline__ void block(int lane, int blk, float v)
    {
        // The 64 dependent adds of the block run as a chain across the lanes: x[l] = x[l-1] + v[l] with a
        // wave_shr:1 DPP operand, 63 times.  After pass k lanes 0..k hold their final prefix sums (a lane whose
        // left neighbour is final recomputes the same value), lane 0 is never written (no source lane), so every
        // lane ends with the running sum of its own sample -- the same additions in the same order as the
        // reference's loop, with no LDS traffic at all.
        const int s_ = 64 * blk + lane;
        const float addend = (s_ == 0) ? 0.0f : v;                     // the sum starts at sample 1
        float sm = lane == 0 ? run + addend : addend;
#pragma unroll
        for (int k = 1; k < 64; k++)                                   // (s_nop: a DPP read needs 2 wait states after the write)
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(sm) : "v"(addend));
        run = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 63));
        const float c_ = (sm != 0.0f) ? v / sm : 0.0f;
        const float p_ = shift_up1(c_, carry);                         // cnd of the previous sample
        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_), 63));
        if (s_ >= 2 && c_ < best) { best = c_; best_i = s_; }
        if (first == 0x7fffffff) {
            // (lane-index conditions as scalar masks: a ballot of anything but a plain compare costs a select and a second compare)
            const unsigned long long hit = wave_ballot(c_ < 0.01f) & (blk == 0 ? ~3ull : ~0ull);           // s_ >= 2
            if (hit) first = 64 * blk + (int) __builtin_ctzll(hit);
        }
        if (first != 0x7fffffff) {
            // sample s-1 ends the walk if it is past `first` and cnd does not keep falling
            const int k0 = first + 1 - 64 * blk;                                                           // s_ - 1 >= first <=> lane >= k0
            const unsigned long long st = wave_ballot(!(c_ < p_)) & (k0 <= 0 ? ~0ull : (k0 >= 64 ? 0ull : ~0ull << k0));
            if (st) {
                const int src = (int) __builtin_ctzll(st);
                const float pc = lane_get(p_, src), cc = lane_get(c_, src);
                const int sstar = 64 * blk + src;
                lag = (pc <= cc) ? (float) (sstar - 1) : (float) sstar;
                done = true;
            }
        }
    }
    // v_end = v[N] (from imag[0] of the inverse transform), valid in lane 0
    __device__ __forceinline__ float finish(int lane, float v_end)
    {
        if (!done) {
            if (first != 0x7fffffff) {
                // the walk ran to N-1 (ref :178: sample + 1 < numSamples); compare with cnd[N]
                float cn = 0.0f;
                if (lane == 0) { run += v_end; cn = (run != 0.0f) ? v_end / run : 0.0f; }
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
        return lag;
    }
};

// waves per SIMD the register allocator must leave room for (LDS bounds residency as well)
template <int N> struct Occ {
    // Residency is set by the LDS; the waves of a CU should spread evenly over its four SIMDs (a workgroup's waves go
    // round the SIMDs, so 6 waves are 2+2+1+1 and two such workgroups in the same rotation leave two SIMDs with 4 waves
    // and two with 2 -- measured no faster than 8 waves per CU).  Hence workgroups of CH channels x K waves with
    // CH*K a multiple of 4, sharing one twiddle table:
    //   <= 512 points : 1 x 8, three workgroups per CU    -> 6 waves per SIMD, <= 80 VGPRs (the 4 KB images leave the LDS room;
    //                   +6 % at 512 points over 4 waves per SIMD, round 3: the kernels are bound by latency, not by issue)
    //   1024 points   : 1 x 8, two workgroups per CU      -> 4 waves per SIMD, <= 128 VGPRs (all the LDS holds)
    //   2048 points   : 3 x 4 = 12 waves, one workgroup   -> 3 per SIMD, <= 168 VGPRs
    //   4096 points   : 1 x 8 (the 160 KB to the byte: 24 KB of twiddles -- CompactTw -- and eight 17 KB buffers, the flux state
    //                   in global memory; until round 4 seven wavefronts beside the whole 32 KB table and the state's LDS image)
    //                   -> 2 per SIMD, <= 256 VGPRs (the split transform keeps a lane's 64 second-pass results in registers)
    static constexpr int WAVES_PER_SIMD = N <= 512 ? 8 : (N <= 1024 ? 4 : 2);
    static constexpr int MAX_THREADS = 512;
};

// The frame kernel's LDS, shared with the host's size computation (fx_kernels.hip, lds_bytes_t).
template <int N> struct FrameLds {
    static constexpr bool WIDE = N == 4096;
    static constexpr int TW_ENTRIES = WIDE ? CompactTw<N>::ENTRIES : N;
    static constexpr int WIDE_TURN_ENTRY = WIDE ? CompactTw<N>::OFF_GAP : 0;      // [channels <= 8][2] ints in the image's 128-byte gap
    __host__ __device__ static constexpr int prev_floats(bool direct) { return (direct || WIDE) ? 0 : Geo<N>::BIMG + (Geo<N>::BQ ? 0 : 4); }
    __host__ __device__ static constexpr size_t bytes(int ch, int k, bool direct)
    {
        return sizeof(f2) * TW_ENTRIES + (size_t) ch * sizeof(float) * prev_floats(direct) + (size_t) ch * k * Geo<N>::BUF_BYTES;
    }
};
static_assert(FrameLds<4096>::bytes(1, 8, false) <= 160 * 1024 && FrameLds<4096>::bytes(8, 1, true) <= 160 * 1024, "eight 4096-point wavefronts per CU");

// One wavefront's view of the frame it is analysing: where its buffers are and the constants every section
// uses.  The member functions are the sections of the reference's two run() loops in the order the kernel
// calls them; all of them are inlined into fx_frame_kernel.
// DIRECT: a call of ONE frame per channel reads and replaces the channel's flux state where it lives, in global memory: every
// element is read once and written once, there is no next frame in this launch to hand it to, and without the LDS copy a
// workgroup is eight channels with nothing but a transform buffer each -- 16 wavefronts per CU at 1024 points instead of 12.
// HOIST: two of the re-materialisation points are transparent (FX_OPQ).
// WIDE: the frame kernel's 4096-point layout -- the flux state stays in global memory for calls of any length (handed from frame to
// frame through the `turn` counter, as the LDS image is elsewhere) and `tw` is the compact twiddle image (CompactTw): what it takes
// to give a CU an eighth wavefront at this size.
// BLOCKS (one-frame calls, hop mode): p.in holds every channel's new BLOCK (rows of p.blk_in_row_bytes) and the hop analysed is the first
// N/2 samples of [the channel's pending samples | its block] -- fx_push_samples without the re-blocking pass (FrameParams::block_mode).
template <int N, bool DIRECT = false, bool HOIST = false, bool WIDE = false, bool BLOCKS = false> struct FrameWave {
    typedef Geo<N> G;
    static constexpr int M = G::M, P = G::P, U = G::U, HALF = N / 2;

    const FrameParams& p;
    const f2* tw;       // [N] pass-ordered twiddles (workgroup LDS); WIDE: the compact image
    __device__ __forceinline__ TwGlobal tw_global() const { return TwGlobal{reinterpret_cast<const f2*>(p.tw), p.tw_quarter_turn != 0, f2{p.tw_at_quarter[0], p.tw_at_quarter[1]}}; }
    const TwRegs<N>* twr;   // the lane's second- / last-pass twiddles in registers (2048 points), else unused
    float* prev;        // [M] re of the channel's last accepted spectral frame (workgroup LDS)
    int*   turn;        // index of the frame whose turn it is to read / replace `prev`
    f2*    cbuf;        // this wave's transform buffer ...
    float* rbuf;        // ... and the same memory viewed as the real image
    FramePart* fpl;     // this frame's result record in global memory: lane 0 stores each value as it appears (single-lane LDS
                        // writes cost a full LDS instruction each, and the LDS pipe is the scarcer resource; the vector-memory
                        // path is nearly idle)
    double nyquist, rnyq, frpb;   // frpb: ref SpectralCharacteristics.h:64,105
    float  scale;       // JUCE inverse-transform scale 1/N
    int    c, T, t;     // channel, frames in this call, this frame

    // |re| of the raw spectrum around and inside the lane's bins, kept from the harmonic FFT to the harmonic tail
    struct HarmonicSpectrum { float hre[U]; float left2, left1, right1; double sum, max; };

    // where the two halves of this frame's window come from (a1, ref RealTimeAudioAnalysis.h:205-219)
    struct Sources { const void* a; const void* b; float gain_a, gain_b; int fmt_a, fmt_b; BlockStream bs, bs_a; bool a_stream; };   // BLOCKS: bs / bs_a = the stream from this frame's hop / the hop before on
    __device__ __forceinline__ Sources sources() const
    {
        const size_t esz = (size_t) sample_bytes(p.sample_format);
        const unsigned char* in = static_cast<const unsigned char*>(p.in);
        Sources s;
        s.fmt_a = s.fmt_b = p.sample_format;
        if constexpr (BLOCKS) {
            // frame t's second half is hop blk_hop0 + t of the stream [pending | block], its first half the hop before -- the carried-over tail
            // for the call's first frame.  The stream is shifted to the hop with wave-uniform arithmetic (stream_from).
            const BlockStream base{p.blk_carry_in + (size_t) c * (size_t) p.blk_carry_row_bytes, in + (size_t) c * (size_t) p.blk_in_row_bytes,
                                   p.blk_carry_bytes, p.blk_in_row_bytes};
            const long long hop_bytes = (long long) HALF * (long long) esz;
            // (the frame index is wave-uniform by construction; saying so keeps the shifted pointers in scalar registers)
            const int tt = DIRECT ? 0 : __builtin_amdgcn_readfirstlane(t);
            s.bs = stream_from(base, (long long) (p.blk_hop0 + tt) * hop_bytes);
            s.b = nullptr; s.gain_b = p.gain;
            s.a_stream = !DIRECT && tt > 0;                                      // (the one-frame forms only ever see a call's first frame)
            if (DIRECT || tt == 0) { s.a = p.tail_in + (size_t) c * HALF; s.fmt_a = FX_SAMPLE_F32; s.gain_a = 1.0f; s.bs_a = s.bs; }
            else        { s.a = nullptr; s.gain_a = p.gain; s.bs_a = stream_from(base, (long long) (p.blk_hop0 + tt - 1) * hop_bytes); }
            return s;
        }
        s.a_stream = false;
        if (p.hop_mode) {
            s.gain_a = s.gain_b = p.gain;
            const size_t row = (size_t) c * (size_t) (p.in_hop_stride ? p.in_hop_stride : T) + (size_t) p.in_hop0;    // (one-frame launches over several hops)
            s.b = in + (row + t) * HALF * esz;
            if (t == 0) { s.a = p.tail_in + (size_t) c * HALF; s.fmt_a = FX_SAMPLE_F32; s.gain_a = 1.0f; }   // tail is fp32, already gained
            else        s.a = in + (row + (t - 1)) * HALF * esz;
        } else {
            s.gain_a = s.gain_b = 1.0f;
            s.a = in + ((size_t) c * T + t) * N * esz;
            s.b = static_cast<const unsigned char*>(s.a) + HALF * esz;
        }
        return s;
    }

    // split sizes: the raw window again, from global memory (L2), in first-pass order
    template <bool LAST_USE = false>
    __device__ __forceinline__ void load_raw(int lane, float (&x)[P]) const
    {
        asm volatile("" ::: "memory");        // a fetch of its own each time: the point is not to keep x live in between
        const Sources s = sources();
        FX_FORMATS(s.fmt_a, s.fmt_b, (load_window_first_pass_order<N, FA, FB, LAST_USE, BLOCKS>(s.a, s.b, s.gain_a, s.gain_b, lane, x, s.bs, s.a_stream, s.bs_a)));
    }

    // returns the lane's share of the frame's sum of squares (split sizes only; otherwise sum_squares() computes it)
    __device__ __forceinline__ double load_frame(int lane) const
    {
FX_MARK("load");
        double ssq = 0.0;
        // ---------------- a1: window assembly (ref RealTimeAudioAnalysis.h:205-219) ----------------
        {
            float* tail_dst = (t == T - 1) ? p.tail_out + (size_t) c * HALF : nullptr;
            const Sources sr = sources();
            const void* src_a = sr.a; const void* src_b = sr.b;
            const float gain_a = sr.gain_a, gain_b = sr.gain_b;
            if constexpr (N >= 512) {
                FX_FORMATS(sr.fmt_a, sr.fmt_b, (ssq = load_window<N, FA, FB, BLOCKS>(src_a, src_b, gain_a, gain_b, rbuf, tail_dst, lane, sr.bs, sr.a_stream, sr.bs_a)));
            } else {
                load_half<N, HALF>(src_a, sr.fmt_a, gain_a, gain_a != 1.0f, rbuf, 0, nullptr, lane);
                load_half<N, HALF>(src_b, sr.fmt_b, gain_b, gain_b != 1.0f, rbuf, HALF, tail_dst, lane);
            }
            wave_fence();
        }
        return ssq;
    }

    // fills xr with the raw frame in first-pass order and returns the sum of its squares: the numerator of
    // getRMSLevel (a2, ref RealTimeAnalyser.h:207-208).  logRMS itself -- (float) log10(rms * 9 + 1) with
    // rms = (float) sqrt(sum / N) -- is a ~130-instruction fp64 computation of one number per wave; it is left
    // to fx_finalise_kernel (one thread per frame), and the spectral section brackets it instead (gate_threshold).
    __device__ __forceinline__ double sum_squares(int lane, float (&xr)[P]) const
    {
FX_MARK("rms");
        // the frame, in registers, in the order the first FFT pass consumes it; the LDS buffer is
        // free again after this read
#pragma unroll
        for (int g = 0; g < G::GA; g++)
#pragma unroll
            for (int j = 0; j < G::RA; j++) xr[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];
        wave_fence();
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) s += (double) (xr[i] * xr[i]);
        s = wave_sum(s);
        if (lane == 0) fpl->sum_sq = s;
        return s;
    }

    // log10 of a float, correctly rounded: this value gates bins (`mag > 0.01*logRMS`), so it must
    // equal the CPU oracle's to the last bit (see oracle/fx_oracle.c)
    static __device__ __forceinline__ float exact_log_rms(double sum_sq)
    {
        const float rms = (float) sqrt(sum_sq / (double) N);
        return (float) log10((double) (rms * 9.0f + 1.0f));
    }

    // The flatness gate is `mag > eps`, eps = 0.01 * logRMS (ref SpectralCharacteristics.h:89-94,108), with mag =
    // (double) re^2 exact -- so it is a comparison of |re| with a float threshold: the largest float t with t*t <= eps.
    // Only the comparisons matter here, so logRMS is bracketed from single-precision hardware sqrt / log2 with a margin
    // far above their error (relative 1e-4, absolute 4e-6 on logRMS, against ~1e-6): if no bin's |re| falls inside the
    // bracket -- the usual case -- its upper end gates every bin exactly as the true threshold would; otherwise (rare,
    // wave-uniform) the exact value is computed.  Returns tg: bin j passes the gate iff |re[j]| > tg.
    __device__ __forceinline__ float gate_threshold(double sum_sq, const float (&re)[U]) const
    {
#ifdef FX_EXP_WIDE_BAND
        const float rel = 0.5f, abs_ = 0.5f;        // test builds: the bracket catches almost every frame
#else
        const float rel = 1e-4f, abs_ = 4e-6f;
#endif
        const float rms_a = __builtin_amdgcn_sqrtf((float) (sum_sq * (1.0 / (double) N)));
        const float log_a = __builtin_amdgcn_logf(rms_a * 9.0f + 1.0f) * 0.30103f;      // v_log_f32 is log2
        const float eps_lo = 0.01f * (log_a * (1.0f - rel) - abs_);
        const float eps_hi = 0.01f * (log_a * (1.0f + rel) + abs_);
        const float t_lo = __builtin_amdgcn_sqrtf(fmaxf(eps_lo, 0.0f)) * (1.0f - 1e-6f);
        const float t_hi = __builtin_amdgcn_sqrtf(eps_hi) * (1.0f + 1e-6f);
        bool inside = false;
#pragma unroll
        for (int j = 0; j < U; j++) inside |= (fabsf(re[j]) > t_lo) && !(fabsf(re[j]) > t_hi);
        if (!wave_any(inside)) return t_hi;              // (a NaN anywhere above lands here too: NaN thresholds gate nothing, as `mag > NaN`)
        // exact: eps as the reference forms it, then the largest float whose square does not exceed it
        const double eps = 0.01 * (double) exact_log_rms(sum_sq);          // :108
        float t = (float) sqrt(eps);
        const float up = __uint_as_float(__float_as_uint(t) + 1u);
        if ((double) t * (double) t > eps) t = __uint_as_float(__float_as_uint(t) - 1u);      // t > 0 here: 0*0 > eps is false
        else if ((double) up * (double) up <= eps) t = up;
        return t;
    }

    // The flatness product with the serial-order semantics of `magnitudeProduct *= binMagnitude` (ref
    // SpectralCharacteristics.h:92) over the bins whose magnitude exceeds eps, including IEEE overflow (sticky inf)
    // and gradual underflow (precision loss, sticky 0).  Exponent-extended products (mantissa in [0.5,1) + int
    // exponent) are formed per lane and scanned across lanes; each lane also keeps the range of exponents its own
    // prefixes pass through.  If no prefix of the serial product can have left the normal range, the product is the
    // scan's total.  Otherwise the product is continued in plain IEEE double from the start of the first lane where
    // that may happen, lane to lane in bin order, until it is exactly 0 or inf (both absorbing) or the bins end.
    // part 1, per lane: the exponent-extended product of the lane's gated magnitudes in bin order and the range of exponents its
    // prefixes pass through.  Called right behind the spectral sums' loop over the same bins, so that the conversions, the
    // squares and the gate compares are shared.
    __device__ __forceinline__ void flatness_local(const float (&re)[U], float tg, FlatProd& loc, int& emin, int& emax) const
    {
        loc = FlatProd{0.5, 1};                                            // 1.0
        emin = 1; emax = 1;                                                // exponents of the lane's own prefixes (1 = the empty one)
#pragma unroll
        for (int j = 0; j < U; j++) {
            const double v = (double) re[j];
            const double mag = v * v;
            if (fabsf(re[j]) > tg) {
                loc = fp_mul(loc, mag);
                emin = loc.exp < emin ? loc.exp : emin;
                emax = loc.exp > emax ? loc.exp : emax;
            }
        }
    }
    // part 2: across the lanes
    __device__ __forceinline__ double flatness_product(int lane, const float (&re)[U], float tg, FlatProd loc, int emin, int emax) const
    {
FX_MARK("flatprod");
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
        // A prefix inside this lane is (exc.mant * m) * 2^(exc.exp + e) with m in [0.5,1) and e in [emin, emax]: the
        // mantissa product lies in [0.25,1), so its normalised exponent is exc.exp + e - 1 or exc.exp + e.
        //   overflow  : value >= 2^1024  <=> exponent >= 1025
        //   subnormal : value <  2^-1022 <=> exponent <= -1022
        const unsigned long long risky_lanes = wave_ballot(exc.exp + emax >= 1025) | wave_ballot(exc.exp + emin - 1 <= -1022);
        if (risky_lanes == 0)
            return ldexp(bcast63(inc.mant), __builtin_amdgcn_readlane(inc.exp, 63));
        // every prefix before the first risky lane is normal, so the serial IEEE product equals the scan there (to
        // rounding); from that lane on the factors are multiplied one by one in double, as the reference does
        const int owner = (int) __builtin_ctzll(risky_lanes);
        double pr = ldexp(lane_get(exc.mant, owner), lane_get(exc.exp, owner));
        double factor[U];                  // 1.0 = not a factor (x * 1.0 is exact)
#pragma unroll
        for (int j = 0; j < U; j++) {
            const double v = (double) re[j];
            const double mag = v * v;
            factor[j] = fabsf(re[j]) > tg ? mag : 1.0;
        }
        for (int l = owner; l < 64; l++) {
            double mine = pr;
#pragma unroll
            for (int j = 0; j < U; j++) mine *= factor[j];
            pr = lane_get(mine, l);
            if (pr == 0.0 || pr == __builtin_huge_val()) break;           // 0 * finite = 0, inf * positive = inf
        }
        return pr;
    }

    __device__ __forceinline__ void spectral(int lane, const float (&xr)[P], double sum_sq) const
    {
        float spec_aux = 0.0f;
FX_MARK("spec_fft");
        // ---------------- spectral analyser (ref RealTimeAnalyser.h:212-224) -----------------------
        lane = FX_OPQ(0, lane);
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
            spec_aux = fft_from_regs<N, false, OUT_RE_LOW_MAXABS, WIDE>(xw, cbuf, tw, p.first_tw, lane, 0.0f, nullptr, twr, tw_global());   // a4
        }
        FX_STOP(6, FX_KEEP(spec_aux); return);
FX_MARK("spec_sums");
        {
            // lane owns bins [U*lane, U*lane + U)
            float re[U];
            // ref SpectralCharacteristics.h:153: getMagnitude over the first M floats of the interleaved
            // buffer = max |re|, |im| over bins [0, M/2)
            float maxabs = spec_aux;
            lds_load_block<U>(reinterpret_cast<const float*>(cbuf) + bimg<N>(U * lane), re);
            const float tg = gate_threshold(sum_sq, re);
            // fillIntermediateValues :62-97 over the lane's bins m = U*lane + j.  The sums over bins that the features
            // need are moments of the magnitudes: B0 = sum mag (magnitudeSum), B1 = sum m*mag, B2 = sum m^2*mag, because
            // fc[m] = (m + 1/2) * frpb (:70): weightedMagnitudeSum = frpb * (B1 + B0/2) (:95) and the spread's
            // sum ((fc - centroid)/nyq)^2 * mag (:135-139) = (B2 + B1 + B0/4)/M^2 - 2*cn*(B1 + B0/2)/M + cn^2*B0.  Inside a
            // lane they come from running suffix sums (T_j = sum_{i>=j} mag_i, V_j = sum_{i>=j} T_i, W = sum_{j>=1} V_j:
            // sum j*mag_j = V_1, sum j^2*mag_j = 2W - V_1): three adds per bin, no multiplications.  All terms are
            // positive, so nothing cancels inside the sums; the order differs from the serial loops by ~1e-16.
            double Ts = 0.0, Vs = 0.0, Ws = 0.0, t_after = 0.0;
            double flat_sum = 0.0;     // flatnessMagnitudeSum (:91)
            float max_re = 0.0f;       // max |re|: (double) re^2 is exact and monotone in |re|, so max mag = max_re^2
            int cnt = 0;               // wave-uniform: bins that pass the flatness gate, counted from the compare masks
            // bins m <= M/5 (:86-87, inclusive) are the lanes below LQ entirely and the first LR + 1 bins of lane LQ
            constexpr int LQ = (M / 5) / U, LR = (M / 5) % U;
#pragma unroll
            for (int j = U - 1; j >= 0; j--) {
                const double v = (double) re[j];
                const double mag = v * v;
                Ts += mag;
                if (j >= 1) { Vs += Ts; Ws += Vs; }
                if (j == LR + 1) t_after = Ts;                                 // sum of the lane's bins after bin LR
                const bool gate = fabsf(re[j]) > tg;                           // :89 `binMagnitude > epsilon`
                cnt += __builtin_popcountll(wave_ballot(gate));
                if (gate) flat_sum += mag;
                max_re = fmaxf(max_re, fabsf(re[j]));
            }
            FlatProd floc; int femin, femax;
            // (beside the loop above the two share conversions and squares; at the split sizes, with 16 / 32 bins a lane and the
            // registers full, that sharing spills, and the product is formed where it is needed)
            if constexpr (!G::SPLIT) flatness_local(re, tg, floc, femin, femax);
            const double ul = (double) (U * lane);
            double mag_sum = Ts;                                               // B0
            double b1 = ul * Ts + Vs;
            double b2 = (ul * ul) * Ts + ((ul + ul) * Vs + ((Ws + Ws) - Vs));
            double lhr = lane < LQ ? Ts : (lane == LQ ? Ts - t_after : 0.0);
            wave_sum4(lane, mag_sum, b1, b2, lhr);
            // (flat_sum, like flux and the second pass's sums, is only recorded: they share one reduction further down)
            max_re = wave_maxf(max_re);
            const double max_mag = (double) max_re * (double) max_re;
            maxabs = wave_maxf(maxabs);
            const bool accepted = mag_sum > 0.05;                              // :121-123

            FX_STOP(7, FX_KEEP(mag_sum); FX_KEEP(b1); FX_KEEP(b2); FX_KEEP(lhr); FX_KEEP(max_mag); FX_KEEP(maxabs); FX_KEEP(flat_sum); FX_KEEP(cnt);
                       FX_KEEP(floc.mant); FX_KEEP(floc.exp); FX_KEEP(femin); FX_KEEP(femax); return);
FX_MARK("flux");
            // ---- flux against the previous accepted frame; hand-off between waves ----
            double flux = 0.0;
            {
                float pvf[U];
                if constexpr (DIRECT) {
                    float* state = p.prev_re + (size_t) c * M + U * lane;              // the lane's U bins, coalesced (16-byte pieces from 512 points on)
                    lds_load_block<U>(state, pvf);                                      // (plain wide loads / stores: global memory here)
                    if (accepted) lds_store_block<U>(state, re);                        // :138 (only on the accepted path)
                } else if constexpr (WIDE) {
                    // the same in global memory, in turn: the wavefronts of a workgroup share their CU's vector cache, so workgroup-scope
                    // acquire / release order the state's loads and stores between them (the release waits for the stores)
                    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t)
                        __builtin_amdgcn_s_sleep(1);
                    float* state = p.prev_re + (size_t) c * M + U * lane;
                    lds_load_block<U>(state, pvf);
                    if (accepted) lds_store_block<U>(state, re);                        // :138
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(turn, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    // (the hand-over costs 2 % of the kernel, and not because of the length of this section: round 2, profiles/NOTEBOOK.md (ix))
                    while (__hip_atomic_load(turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != t)
                        __builtin_amdgcn_s_sleep(1);
                    // the turn is held for two LDS reads and two writes only: the next frame's wave is usually waiting for it
                    lds_load_block<U>(prev + bimg<N>(U * lane), pvf);
                    if (accepted) lds_store_block<U>(prev + bimg<N>(U * lane), re);     // :138 (only on the accepted path)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(turn, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const double pv = (double) pvf[j];
                    const double v = (double) re[j];
                    const double diff = v * v - pv * pv;                       // :76
                    flux += fmax(diff, 0.0);                                   // :77-79 (a NaN difference adds nothing, as `if (diff > 0)`)
                }
            }
            FX_STOP(8, FX_KEEP(mag_sum); FX_KEEP(b1); FX_KEEP(b2); FX_KEEP(lhr); FX_KEEP(max_mag); FX_KEEP(maxabs); FX_KEEP(flat_sum); FX_KEEP(cnt);
                       FX_KEEP(floc.mant); FX_KEEP(floc.exp); FX_KEEP(femin); FX_KEEP(femax); FX_KEEP(flux); return);
            lane = FX_OPQ(1, lane);
            if constexpr (G::SPLIT) flatness_local(re, tg, floc, femin, femax);
            const double prod = flatness_product(lane, re, tg, floc, femin, femax);
            FX_STOP(9, FX_KEEP(mag_sum); FX_KEEP(b1); FX_KEEP(b2); FX_KEEP(lhr); FX_KEEP(max_mag); FX_KEEP(maxabs); FX_KEEP(flat_sum); FX_KEEP(cnt);
                       FX_KEEP(prod); FX_KEEP(flux); return);

FX_MARK("spec_pass2");
            // The slope needs sum (mag - mean)^2 (:182-188): a second pass over the lane's bins.  The spread (:135-141) and the
            // centroid (:127) are formed by fx_finalise_kernel from the moments B0 (mag_sum), B1, B2 -- one thread per frame
            // instead of a wave-wide fp64 division here -- unless the moment form, which loses (centroid / bandwidth)^2 in
            // relative precision (~1e-11 for any windowed signal: the Bartlett main lobe is several bins wide), cannot be
            // trusted: the weighted variance below 1e-9 of cn^2 * B0, or not finite.  In moments that test is
            // S2 * B0 - W^2 > 1e-9 * W^2 with S2 = B2 + B1 + B0/4 and W = B1 + B0/2 (no division); where it fails the sum is
            // taken here as the reference writes it.
            {
                const double mu = mag_sum * (1.0 / (double) M);
                double vsum = 0.0;
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const double v = (double) re[j];
                    const double dv = __builtin_fma(v, v, -mu);                // = v * v - mu (v * v is exact in fp64)
                    vsum = __builtin_fma(dv, dv, vsum);                        // (one rounding fewer than the reference's; ~1e-16)
                }
                const double wm = b1 + 0.5 * mag_sum, s2b0 = (b2 + b1 + 0.25 * mag_sum) * mag_sum, ww = wm * wm;
                const bool refine = accepted && (!(s2b0 - ww > 1e-9 * ww) || !(s2b0 < __builtin_huge_val()));
                double direct = 0.0;
                if (refine) {
                    const float centroid = (float) ((frpb * wm) / mag_sum);    // :95, :127
                    const double cn = (double) centroid * rnyq;
#pragma unroll
                    for (int j = 0; j < U; j++) {
                        const double v = (double) re[j];
                        const double d = ((double) (U * lane + j) * frpb + (frpb / 2.0)) * rnyq - cn;
                        direct += (d * d) * (v * v);
                    }
                }
                wave_sum4(lane, flux, vsum, flat_sum, direct);                 // one reduction for everything that is only recorded
                if (lane == 0) { fpl->var = direct; fpl->refined = refine ? 1 : 0; fpl->vsum = vsum; }
            }
            double max_e = (double) maxabs;                                    // :153
            if (max_mag > max_e) max_e = max_mag;                              // :161-162
            if (lane == 0) {
                fpl->mag_sum = mag_sum; fpl->lhr = lhr; fpl->flux = flux; fpl->flat_sum = flat_sum; fpl->prod = prod;
                fpl->max_e = max_e; fpl->b1 = b1; fpl->b2 = b2; fpl->cnt = (float) cnt;
            }
        }
        wave_fence();
    }

    __device__ __forceinline__ void harmonic_spectrum(int lane, const float (&xr)[P], HarmonicSpectrum& hs) const
    {
        float (&hre)[U] = hs.hre;
        float &h_left2 = hs.left2, &h_left1 = hs.left1, &h_right1 = hs.right1;          // |re| of bins U*lane-2, U*lane-1, U*lane+U
        double &h_sum = hs.sum, &h_max = hs.max;
        h_sum = 0.0;
        float h_max_re = 0.0f;
FX_MARK("harm1");
        // ---------------- harmonic analyser, part 1: raw (un-windowed) spectrum ---------------------
        // ref RealTimeAnalyser.h:161
        lane = FX_OPQ(2, lane);
        fft_from_regs<N, false, OUT_RE_LOW, WIDE>(xr, cbuf, tw, p.first_tw, lane, 0.0f, nullptr, twr, tw_global());
        {
            const int b0 = U * lane;
            const float* relin = reinterpret_cast<const float*>(cbuf);
            lds_load_block<U>(relin + bimg<N>(b0), hre);
            h_left2  = b0 >= 2 ? fabsf(relin[bimg<N>(b0 - 2)]) : 0.0f;
            h_left1  = b0 >= 1 ? fabsf(relin[bimg<N>(b0 - 1)]) : 0.0f;
            h_right1 = b0 + U < M ? fabsf(relin[bimg<N>(b0 + U)]) : 0.0f;
#pragma unroll
            for (int j = 0; j < U; j++) {                                      // ref HarmonicCharacteristics.h:61-69
                const double v = (double) hre[j];
                h_sum += v * v;
                h_max_re = fmaxf(h_max_re, fabsf(hre[j]));
            }
            h_sum = wave_sum(h_sum);
            h_max_re = wave_maxf(h_max_re);
            h_max = (double) h_max_re * (double) h_max_re;
        }
        wave_fence();
    }

    // out[i] = a * in[i], formed two at a time and hidden from the vectoriser, which otherwise pairs a*x with b*y of the
    // recurrence below (a packed multiply, a move and a horizontal add per sample instead of a multiply and an add)
    template <int K> static __device__ __forceinline__ void scaled_pairs(float a, const float* in, float* out)
    {
#pragma unroll
        for (int i = 0; i < K; i += 2) {
            const f2 t = f2{in[i], in[i + 1]} * a;
            out[i] = t.x; out[i + 1] = t.y;
            asm volatile("" : "+v"(out[i]), "+v"(out[i + 1]));
        }
    }
    // returns f0 = sampleRate / lag (ref PitchAnalyser.h:57) and records the lag
    // a10 + ref RealTimeAnalyser.h:157: the one-pole low-pass of the raw frame, windowed, left in the real image
    __device__ __forceinline__ void lowpass_window(int lane) const
    {
FX_MARK("lpf");
        // a10 AudioFilter::filterAudio, ref RealTimeAudioAnalysis.h:106-125:
        //   y[0] = x[0];  y[n] = (a*x[n]) + (b*y[n-1]) in fp32, strictly serial.
        // Lane l owns samples [P*l, P*l+P).  It starts KW samples early from a guess, and the
        // recurrence (|b| = 0.208) forgets the guess; the value it reaches at P*l-1 must be
        // bit-identical to what lane l-1 produced there, otherwise the chunk is redone from the
        // neighbour's value until every hand-over matches (exact by induction from lane 0).
                constexpr int KW = 16;
        const float a = p.lpf_a, b = p.lpf_b;
        // (the pitch path runs first, while the real image of the raw frame that load_frame left in the buffer is
        // still intact)
        float x[P];
#pragma unroll
        for (int i = 0; i < P; i += 4) {
            const f4 v = *reinterpret_cast<const f4*>(&rbuf[rimg<N>(P * lane + i)]);
            x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
        }
        float yin = 0.0f;                             // y[P*lane - 1] used as this chunk's input
        {
            const int first = P * lane;
#pragma unroll
            for (int q = 0; q < KW / 4; q++) {
                const int n0 = first - KW + 4 * q;    // multiple of 4: the whole group is in range or not
                if (n0 >= 0) {
                    const f4 v = *reinterpret_cast<const f4*>(&rbuf[rimg<N>(n0)]);
                    const float w[4] = {v.x, v.y, v.z, v.w};
                    float aw[4];
                    scaled_pairs<4>(a, w, aw);
#pragma unroll
                    for (int e = 0; e < 4; e++)       // sample 0 starts the filter exactly; the first
                        yin = (n0 + e == 0 || (q == 0 && e == 0)) ? w[e] : aw[e] + (b * yin);        // warm-up sample is a guess
                }
            }
        }
        wave_fence();
        float y[P];
        float ylast;
        // a * x[n], two products per instruction; the chain is then one multiply and one add a sample.  All P of them where
        // the registers have room (they serve the redo below as well), four at a time at the split sizes.
        constexpr int AXN = G::SPLIT ? 4 : P;
        float ax[AXN];
        {
            float yy = yin;
#pragma unroll
            for (int i = 0; i < P; i += AXN) {
                scaled_pairs<AXN>(a, &x[i], ax);
#pragma unroll
                for (int e = 0; e < AXN; e++) {
                    yy = (lane == 0 && i + e == 0) ? x[0] : ax[e] + (b * yy);
                    y[i + e] = yy;
                }
            }
            ylast = yy;
        }
        for (int iter = 0; iter < 64; iter++) {
            const float pe = shift_up1(ylast, 0.0f);
            const unsigned long long bad_lanes = wave_ballot(__float_as_uint(pe) != __float_as_uint(yin)) & ~1ull;     // lanes > 0
            if (!bad_lanes) break;
            if ((bad_lanes >> lane) & 1) {
                yin = pe;
                float yy = yin;
#pragma unroll
                for (int i = 0; i < P; i++) { yy = (AXN == P ? ax[i % AXN] : a * x[i]) + (b * yy); y[i] = yy; }
                ylast = yy;
            }
        }
        // window the filtered frame (ref RealTimeAnalyser.h:157) and put it back in the real image.
        // A lane's P samples lie in one half of the window; the gains w0 + i*wstep are exact dyadic
        // numbers (so the fma rounds nothing) and equal bartlett_gain<N>(P*lane + i).
        lane = FX_OPQ(3, lane);
        const float w0 = bartlett_gain<N>(P * lane);
        const float wstep = lane < 32 ? (2.0f / N) : -(2.0f / N);
#pragma unroll
        for (int i = 0; i < P; i += 4) {
            f4 v;
            v.x = y[i]     * __builtin_fmaf(wstep, (float) i, w0);
            v.y = y[i + 1] * __builtin_fmaf(wstep, (float) (i + 1), w0);
            v.z = y[i + 2] * __builtin_fmaf(wstep, (float) (i + 2), w0);
            v.w = y[i + 3] * __builtin_fmaf(wstep, (float) (i + 3), w0);
            *reinterpret_cast<f4*>(&rbuf[rimg<N>(P * lane + i)]) = v;
        }
        wave_fence();
    }

    // a13 / a14 (ref PitchAnalyser.h:129-217) on v[s] = (d[s]/N)^2 * s, which the inverse transform left in
    // registers: lane l holds v[l + 64*m] in vreg[m], i.e. its own sample of every 64-sample block; v_end = v[N]
    // (from imag[0]) is valid in lane 0.
    //   a13: running fp32 sum (:138-150) -- serial by definition -- taken 64 samples at a time; after each block all
    //        lanes form cnd = v/sum (:146-154) and advance a14's search (:161-190), which usually ends in the first
    //        block or two:
    //   first  = first s >= 2 with cnd[s] < 0.01
    //   stop   = first s' >= first with !(cnd[s'+1] < cnd[s'])   (or N-1)
    //   lag    = cnd[stop] <= cnd[stop+1] ? stop : stop+1        (ref :192-203)
    //   otherwise the global minimum over [2, N), first occurrence (ref :171-175).
    // LAZY (1024 points): vreg[0], vreg[1] are filled; blocks 2 and 3 and then all the others are taken from `lz` only if
    // the search gets there (LazyLag).
    template <bool LAZY>
    __device__ __forceinline__ float lag_search(int lane, float (&vreg)[P], float v_end, LazyLag<LAZY ? N : 1024>* lz) const
    {
FX_MARK("scan");
        lane = FX_OPQ(4, lane);
        LagSearch<N> ls;
        ls.begin();
        ls.block(lane, 0, vreg[0]);
        if (!ls.done && P > 1) ls.block(lane, 1, vreg[1]);
        if (!ls.done && P > 2) {
            // rare: the search goes past the second block; the remaining blocks pick their samples from the buffer
            float* vbuf = rbuf;                                            // [N] plain layout
            if constexpr (LAZY) {
                vbuf[64 * 2 + lane] = lz->head(2);
                vbuf[64 * 3 + lane] = lz->head(3);
                wave_fence();
                for (int blk = 2; blk < P && !ls.done; blk++) {
                    if (blk == 4) {                                        // past sample 255: the rest of the transform
                        v_end = lz->rest(vreg);
#pragma unroll
                        for (int m = 4; m < P; m++) vbuf[64 * m + lane] = vreg[m];
                        wave_fence();
                    }
                    ls.block(lane, blk, vbuf[64 * blk + lane]);
                }
            } else {
#pragma unroll
                for (int m = 2; m < P; m++) vbuf[64 * m + lane] = vreg[m];
                wave_fence();
                for (int blk = 2; blk < P && !ls.done; blk++) ls.block(lane, blk, vbuf[64 * blk + lane]);
            }
        }
        return ls.finish(lane, v_end);
    }

    // pitch: low-pass -> window -> FFT -> re^2 -> inverse FFT -> lag; returns f0 = sampleRate / lag (ref PitchAnalyser.h:57)
    __device__ __forceinline__ double pitch(int lane) const
    {
        lowpass_window(FX_OPQ(5, lane));
        FX_STOP(2, return 1.0);
FX_MARK("pitch_fft");
        lane = FX_OPQ(6, lane);
        float xf[P];
#pragma unroll
        for (int g = 0; g < G::GA; g++)
#pragma unroll
            for (int j = 0; j < G::RA; j++) xf[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];
        wave_fence();
        // a11 getComplexConjugateMultiplication, ref PitchAnalyser.h:83-108: re*re, imag := 0, delivered in the order the
        // inverse transform's first pass wants it
        float xp[P];
        fft_from_regs<N, false, OUT_POWER, WIDE>(xf, cbuf, tw, p.first_tw, lane, 0.0f, xp, twr, tw_global());  // ref RealTimeAnalyser.h:160
        FX_STOP(3, for (int j = 0; j < P; j++) FX_KEEP(xp[j]); return 1.0);
FX_MARK("power");
        lane = FX_OPQ(7, lane);
        if constexpr (G::GA == 1) {
#pragma unroll
            for (int j = 0; j < P; j++) xf[j] = xp[j];
        } else {
#pragma unroll
            for (int g = 0; g < G::GA; g++)
#pragma unroll
                for (int j = 0; j < G::RA; j++) xf[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];   // already squared
            wave_fence();
        }
FX_MARK("ifft");
        float vreg[P];
        float lag;
        if constexpr (N == 1024) {
            LazyLag<N> lz;                                                 // a12 inverse, ref :110-121, its last pass on demand
            lz.load(xf, cbuf, tw, p.first_tw, lane, scale);
            vreg[0] = lz.head(0);
            vreg[1] = lz.head(1);
            FX_STOP(4, FX_KEEP(vreg[0]); FX_KEEP(vreg[1]); return 1.0);
            lag = lag_search<true>(lane, vreg, 0.0f, &lz);
        } else
        {
            const float v_end = fft_from_regs<N, true, OUT_LAG, WIDE>(xf, cbuf, tw, p.first_tw, lane, scale, vreg, twr, tw_global());   // a12 inverse, ref :110-121
            lag = lag_search<false>(lane, vreg, v_end, nullptr);
        }
        if (lane == 0) fpl->lag = lag;
        wave_fence();
        return (nyquist * 2.0) / (double) lag;
    }

    __device__ __forceinline__ void harmonic_tail(int lane, HarmonicSpectrum& hs, double f0) const
    {
        float (&hre)[U] = hs.hre;
        const float h_left2 = hs.left2, h_left1 = hs.left1, h_right1 = hs.right1;
        double &h_sum = hs.sum;
        const double h_max = hs.max;
FX_MARK("harm2");
        // ---------------- harmonic analyser, part 2 (ref HarmonicCharacteristics.h:71-105) ----------
        lane = FX_OPQ(8, lane);
        if (!(h_sum < 0.005)) {                                                // :88-89
            float* normed = reinterpret_cast<float*>(cbuf);                    // bins image: re of every bin (normalised on demand)
            unsigned short* peaks = reinterpret_cast<unsigned short*>(normed + G::BIMG);   // [<= M] peak bins (< 2048: 16 bits)
            double mean_mag = h_sum / (double) M;                              // :86
            // binIsPeak's `mag > mean` (:132) is an exact tie when the spectrum is flat (an impulse at
            // sample 0 or N/2 of the window): then the last bit of the reference's serial sum (:61-66)
            // decides for every bin at once.  If any bin sits within rounding distance of the mean,
            // redo the sum in the reference's order, handing the running value from lane to lane.
            {
                // |re| within ~3e-6 of sqrt(mean) brackets every bin whose magnitude is within 1e-12 of the mean, with
                // room for the hardware square root's own error (a wider band only means the exact recomputation
                // below runs a little more often)
                const float root_mean = __builtin_amdgcn_sqrtf((float) mean_mag);
                const float band = root_mean * 3e-6f;
                unsigned long long near = 0;                                   // (lane masks: a bool would be packed into bytes)
#pragma unroll
                for (int j = 0; j < U; j++) near |= wave_ballot(fabsf(fabsf(hre[j]) - root_mean) <= band);
                if (near) {
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
            bool pk[U];                                                        // lane masks (SGPR pairs), not a per-lane bit field
            const double r_hmax = 1.0 / h_max;
            const double sum_normed = h_sum * r_hmax;                          // :77 sum of mag / max over all bins
            int npk_lane = 0;
            // binIsPeak :127-145: above the mean and none of bins -2,-1,+1 larger (windows are clipped at the ends, :136-138: no
            // +1 neighbour for the last two bins).  mag = (double) re^2 is exact, so comparing |re| compares magnitudes exactly.
#pragma unroll
            for (int j = 0; j < U; j++) {
                const double v = (double) hre[j];
                const float me = fabsf(hre[j]);
                const float l2 = j >= 2 ? fabsf(hre[j >= 2 ? j - 2 : 0]) : (j == 1 ? h_left1 : h_left2);
                const float l1 = j >= 1 ? fabsf(hre[j >= 1 ? j - 1 : 0]) : h_left1;
                const float r1 = j + 1 < U ? fabsf(hre[j + 1 < U ? j + 1 : 0]) : h_right1;
                // Neighbours that do not exist (below bin 0, above bin M-1) were loaded as 0 and are never greater;
                // the one clipped neighbour that does exist is bin M-1 seen from bin M-2 (lane 63, j = U-2).
                bool is_peak = v * v > mean_mag && !(l2 > me) && !(l1 > me);
                if (j == U - 2) is_peak = is_peak && (!(r1 > me) || lane == 63);
                else            is_peak = is_peak && !(r1 > me);
                pk[j] = is_peak;
                npk_lane += is_peak ? 1 : 0;
            }
            // the probes below need the normalised magnitudes (float)(mag / max) (:75) of a few bins only; the
            // normalisation is monotone, so the largest of a neighbourhood is the normalised largest |re|
            // (the bins image is still as the raw frame's transform left it: harmonic_spectrum only read it)
            // compact the peak list
            const int pre = wave_scan_incl_i(npk_lane);
            const int total_peaks = __builtin_amdgcn_readlane(pre, 63);
            {
                unsigned short* wp = peaks + (pre - npk_lane);
#pragma unroll
                for (int j = 0; j < U; j++)
                    if (pk[j]) { *wp = (unsigned short) (U * lane + j); wp++; }
            }
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
                    // getMaxBinInNeighbourhood :200-210 : [max(0,c-2), min(c+2, M)) = bins c-2, c-1, c, c+1 where they
                    // exist, start value normed[c], replaced only by a strictly greater one (`if (v > mx) mx = v`)
                    float mx = fabsf(normed[bimg<N>(bin)]);
                    const float a2 = bin >= 2 ? fabsf(normed[bimg<N>(bin >= 2 ? bin - 2 : 0)]) : mx;
                    const float a1 = bin >= 1 ? fabsf(normed[bimg<N>(bin >= 1 ? bin - 1 : 0)]) : mx;
                    const float b1 = bin + 1 < M ? fabsf(normed[bimg<N>(bin + 1 < M ? bin + 1 : 0)]) : mx;
                    // in the reference's order: c-2, c-1, (c itself changes nothing), c+1
                    float run = mx;
                    if (a2 > run) run = a2;
                    if (a1 > run) run = a1;
                    if (b1 > run) run = b1;
                    mx = run;
                    const double pm = (double) mx;
                    probe = (double) (float) ((pm * pm) * r_hmax);                 // (mag / max, via one reciprocal), as a float (:75)
                }
            }

            // calculateInharmonicity :212-244
            double inh = 0.0;
            const double r_hsum = 1.0 / h_sum;
            if (f0 > 0.0) {                                                    // :98
                for (int i = lane; i < total_peaks; i += 64) {
                    const int bin = (int) peaks[i];
                    if (bin == f0_bin) continue;                               // :220-221
                    double fs = (double) bin * fr;
                    if (fs == 0.0) fs = fr * 0.5;                              // :225-226
                    const double fe = (double) (bin + 1) * fr;
                    // getFrequencyRatio :251-259: higher / lower (1.0 when equal: x / x is exactly 1)
                    const double rs  = (fs > f0 ? fs : f0) / (fs > f0 ? f0 : fs);
                    const double re_ = (fe > f0 ? fe : f0) / (fe > f0 ? f0 : fe);
                    if (floor(rs) != floor(re_)) continue;                     // :232-233
                    const double r = rs < re_ ? rs : re_;
                    const double v = (double) normed[bimg<N>(bin)];
                    inh += (r - floor(r)) * ((v * v) * r_hsum);                // :236-239 (binMagnitude / magnitudeSum)
                }
            }
            double score = probe;                                              // / sum_normed, clamped: fx_finalise_kernel
            wave_sum2(lane, score, inh);
            if (lane == 0) { fpl->inh = inh; fpl->her_score = score; fpl->sum_normed = sum_normed; fpl->flags = 1; }
        }
        wave_fence();
    }
};

// SPEC / HARM: which of the reference's two analysers run (RealTimeSpectralAnalyser,
// RealTimeHarmonicAnalyser -- both by default, as AnalyserTrackController constructs them)
// DIRECT: calls of one frame per channel (FrameWave): p.T == 1, p.waves_per_ch == 1, a workgroup is p.ch_per_wg channels, no flux
// state in LDS.
// (the kernel's body as a device function: fx_frame_tail_kernel -- fx_tail_kernels.hip.h -- runs it too and finishes the hop itself)
template <int N, bool SPEC, bool HARM, bool DIRECT, bool BLOCKS = false>
__device__ __forceinline__ void frame_kernel_body(const FrameParams& p_arg)
{
    static_assert(!BLOCKS || N >= 512, "the block-fed forms exist from 512 points on (load_window)");
    FrameParams p = p_arg;
    if (p.dyn) { p.gain = p.dyn->gain; p.nyquist = p.dyn->nyquist; }         // captured step (hipGraph): per-call scalars
    typedef Geo<N> G;
    constexpr int M = G::M, P = G::P;

    // One workgroup = CH channels x K waves (K frames of a channel in flight), sharing one twiddle table.  LDS:
    //   twiddles [N] | CH x { bins image of the channel's flux state (+ its hand-over counter) } | one buffer per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    //   WIDE (4096 points): compact twiddle image, the hand-over counters (8 bytes per channel) in its gap | one buffer per wave
    constexpr bool WIDE = FrameLds<N>::WIDE;
    constexpr int PREV_FLOATS = FrameLds<N>::prev_floats(DIRECT);
    constexpr size_t WAVE_BYTES = G::BUF_BYTES;
    const int CH = p.ch_per_wg, K = p.waves_per_ch;
    f2*    tw    = reinterpret_cast<f2*>(smem);                             // [N], or the compact image
    f2*    tw_lds = tw;
    float* prev0 = reinterpret_cast<float*>(tw_lds + FrameLds<N>::TW_ENTRIES);
    unsigned char* per_wave = reinterpret_cast<unsigned char*>(prev0 + (size_t) CH * PREV_FLOATS);

    // (the wavefront's index is wave-uniform by construction; saying so keeps everything derived from it -- channel, frame index,
    // buffer and record addresses, the frame loop itself -- in scalar registers: 126 -> 106 VGPRs at 1024 points, 256 + 60 B of
    // scratch -> 245 at 4096, 80 + 16 B -> 64 at 512)
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int lane0 = threadIdx.x & 63;
    const int chl = wave / K, slot = wave % K;          // channel within the workgroup, frame slot within the channel
    const int T = p.T;
    // which channels, which frames: the whole call for channel group blockIdx.x, or -- a call cut in time (FrameParams::
    // num_chunks) -- the unit a ticket names, chunk-major: every earlier chunk of these channels has a lower ticket
    int group = (int) blockIdx.x, chunk = 0;
    if (p.num_chunks > 1) {
        unsigned* ticket_s = reinterpret_cast<unsigned*>(per_wave);           // (wave 0's buffer is not in use yet)
        if (threadIdx.x == 0) *ticket_s = __hip_atomic_fetch_add(p.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *ticket_s, groups = gridDim.x / (unsigned) p.num_chunks;
        __syncthreads();                                                       // (before anyone writes that buffer)
        chunk = (int) (ticket / groups);
        group = (int) (ticket % groups);
    }
    const int c = group * CH + chl;
    const bool live = c < p.C;                          // the last workgroup may hold fewer channels
    const int t_begin = p.num_chunks > 1 ? p_arg.chunk_begin[chunk] : 0;
    const int t_end = p.num_chunks > 1 ? p_arg.chunk_begin[chunk + 1] : T;

    float* prev = prev0 + (size_t) chl * PREV_FLOATS;   // bins image: re of the channel's last accepted frame
    // the hand-over counter lives in the first padding gap of the bins image when there is one, else behind it; 4096 points, which
    // has no state image (WIDE), keeps its counters in the gap of the compact twiddle image
    int*   turn = WIDE ? reinterpret_cast<int*>(tw_lds + FrameLds<N>::WIDE_TURN_ENTRY) + 2 * chl : reinterpret_cast<int*>(G::BQ ? prev + G::U : prev + G::BIMG);
    f2*    cbuf = reinterpret_cast<f2*>(per_wave + WAVE_BYTES * wave);
    float* rbuf = reinterpret_cast<float*>(cbuf);      // the same memory viewed as the real image

    // workgroup prologue: twiddle table + the channels' flux state into LDS
    if constexpr (WIDE) {
        // (the image as the host laid it out -- build_twiddle_image; its gap holds the hand-over counters, set below)
        typedef CompactTw<N> CT;
        const f2* image = reinterpret_cast<const f2*>(p.tw_image);
        for (int i = threadIdx.x; i < CT::ENTRIES; i += blockDim.x)
            if (i < CT::OFF_GAP || i >= CT::OFF_GAP + CT::Q1_GAP) tw_lds[i] = image[i];
    } else {
        for (int i = threadIdx.x; i < N; i += blockDim.x) tw_lds[i] = reinterpret_cast<const f2*>(p.tw)[i];
    }
    if (chunk > 0) {
        // the flux state comes from the chunk before, written by another workgroup (another CU): its count, then an
        // agent-scope acquire, then the barrier (MI355X guide: one relaxed poll -> one acquire -> vmcnt(0) -> barrier ->
        // plain loads).  The predecessor holds a lower ticket, so it is running or done; it was dispatched a whole round
        // of channels earlier and is normally long finished.  The spin is bounded all the same (FrameParams::spin_limit polls), and a
        // unit that runs into the bound reports it (FrameParams::err) instead of carrying on silently.
        if (live && slot == 0 && lane0 == 0) {
            unsigned spins = 0;
            bool there;
            while (!(there = __hip_atomic_load(p.queue + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned) chunk) && ++spins < p.spin_limit)
                __builtin_amdgcn_s_sleep(8);
            // gave up: the state loaded below is stale.  Say so where the host looks at its next synchronisation.
            if (!there) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (live && !DIRECT) {
        if constexpr (!WIDE)
            for (int i = lane0 + 64 * slot; i < M; i += 64 * K) prev[bimg<N>(i)] = p.prev_re[(size_t) c * M + i];
        if (lane0 == 0 && slot == 0) { turn[0] = t_begin; turn[1] = t_begin; }
    }
    __syncthreads();

    TwRegs<N> twr;
    if constexpr (TwRegs<N>::USE) twr.load(tw, lane0);

    const double nyquist = p.nyquist;
    const double rnyq = 1.0 / nyquist;
    const double frpb = nyquist / (double) M;          // ref SpectralCharacteristics.h:64,105
    const float  scale = 1.0f / (float) N;             // JUCE inverse scale

    // Which frame a wavefront takes next.  Round robin (frame slot, slot + K, ...) at most sizes.  At 4096 points the wavefronts CLAIM
    // frames, in order, from a counter beside the hand-over counter (`turn[1]`; the frame before any claimed frame has been claimed by
    // a wavefront that is running, so the flux hand-over cannot deadlock): a frame's time depends on its data (the lag search), a
    // wavefront gets through eight frames or fewer per call at the bench shape, and with two wavefronts per SIMD nothing else evens a
    // slow one out.  Which wavefront analyses a frame changes no arithmetic.  Measured (profiles/r04_4096.txt), 1024 channels x 64 frames:
    // seven wavefronts per CU (2 + 2 + 2 + 1 on the SIMDs) 1.785 -> 1.652 ms, eight 1.718 -> 1.593 ms; at 2048 / 1024 points, with 8 / 16
    // wavefronts per CU and more frames each, -0.3 % / +0.5 %: not used there.
    constexpr bool CLAIM = (N == 4096) && !DIRECT;
    auto next_frame = [&](int t_now) -> int {
        if constexpr (!CLAIM) return t_now + K;
        int v = 0;
        if (lane0 == 0) v = __hip_atomic_fetch_add(turn + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(v);
    };
    for (int t = live ? (CLAIM ? next_frame(0) : t_begin + slot) : t_end; t < t_end; t = next_frame(t)) {
        constexpr bool HOIST = SPEC && !HARM && N == 1024;       // (see FX_OPQ)
        const int lane = FX_OPQ(15, lane0);
        // uniform per-frame results go to LDS as soon as they exist instead of occupying ~28 VGPRs
        // in every lane for the whole frame
        FramePart* fpl = p.part + ((size_t) c * T + t);
        if (lane == 0) fpl->flags = 0;            // the harmonic tail sets it; the other fields are read only where written
        const FrameWave<N, DIRECT, HOIST, WIDE, BLOCKS> w{p, tw, &twr, prev, turn, cbuf, rbuf, fpl, nyquist, rnyq, frpb, scale, c, T, t};

        const double ssq_lane = w.load_frame(lane);
        float xr[P];
        double sum_sq;
        if constexpr (G::SPLIT) {
            // split sizes do not hold the raw frame in registers across the transforms (a lane's second-pass results
            // alone are 64 or 128 registers): the sum of squares comes from the load, the frame is fetched again below
            sum_sq = wave_sum(ssq_lane);
            if (lane == 0) fpl->sum_sq = sum_sq;
        } else {
            sum_sq = w.sum_squares(lane, xr);
        }
        // The harmonic analyser's pitch estimate comes first: its low-pass reads the raw frame's real image, which the
        // transforms below overwrite.  (The order of the two analysers only matters for the smoothed RMS, which
        // fx_epilogue_kernel derives from the order flag.)
        double f0 = 0.0;
        FX_STOP(1, FX_KEEP(sum_sq); FX_KEEP(xr[0]); continue);
        if constexpr (HARM) f0 = w.pitch(lane);
        FX_STOP(2, continue); FX_STOP(3, continue); FX_STOP(4, continue);
        FX_STOP(5, FX_KEEP(f0); continue);
        if constexpr (SPEC) {
            if constexpr (G::SPLIT) w.load_raw(lane, xr);
            w.spectral(lane, xr, sum_sq);
        }
        FX_STOP(6, continue); FX_STOP(7, continue); FX_STOP(8, continue); FX_STOP(9, continue); FX_STOP(10, continue);
        if constexpr (HARM) {
            typename FrameWave<N, DIRECT, HOIST, WIDE, BLOCKS>::HarmonicSpectrum hs;
            if constexpr (G::SPLIT) w.template load_raw<true>(lane, xr);
            w.harmonic_spectrum(lane, xr, hs);
            FX_STOP(11, FX_KEEP(hs.sum); FX_KEEP(hs.max); FX_KEEP(hs.left2); FX_KEEP(hs.left1); FX_KEEP(hs.right1); for (int j = 0; j < G::U; j++) FX_KEEP(hs.hre[j]); continue);
            w.harmonic_tail(lane, hs, f0);
        }
        // BLOCKS: what the block leaves over becomes the channel's pending samples (the other of the context's two carry buffers).  Measured
        // at 8192 channels x 1024 points (tools/ab_blocks.sh): here, right behind the window's loads, or with its loads issued in front of
        // the window's and its stores behind them (8 - 32 more registers across the load stage: scratch at 4096 points) -- 66.5 us per call
        // all three.  What did matter was the row's last piece (stream_piece16): byte by byte it cost 1.7 us per launch.
        if constexpr (BLOCKS)
            if (p.blk_keep_rest && t == T - 1)                                  // (the wavefront that analyses the call's last frame)
                stream_keep_rest(w.sources().bs, (long long) (N / 2) * sample_bytes(p.sample_format), p.blk_carry_out + (size_t) c * (size_t) p.blk_carry_row_bytes, lane, 64);
    }

    if constexpr (DIRECT) return;                       // (the flux state was replaced in place)
    __syncthreads();
    if constexpr (!WIDE)                                // (WIDE: in place as well)
        if (live)
            for (int i = lane0 + 64 * slot; i < M; i += 64 * K) p.prev_re[(size_t) c * M + i] = prev[bimg<N>(i)];
    if (chunk + 1 < p.num_chunks) {
        // hand the flux state to the next chunk's workgroup: every storing wave's stores done, the barrier, one lane's
        // agent-scope release, then the count (MI355X guide, producer form)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (live && slot == 0 && lane0 == 0 && !(p.debug_flags & 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.queue + 1 + c, (unsigned) (chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int N, bool SPEC, bool HARM, bool DIRECT = false, bool BLOCKS = false>
__global__ void __launch_bounds__(Occ<N>::MAX_THREADS, Occ<N>::WAVES_PER_SIMD)
fx_frame_kernel(const FrameParams p_arg)
{
    frame_kernel_body<N, SPEC, HARM, DIRECT, BLOCKS>(p_arg);
}
