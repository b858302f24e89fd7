// fx_fft.hip.h -- padded LDS images and the bit-exact JUCE-order FFT of one wavefront
// Included by fx_kernels.hip inside namespace fxk (one translation unit: every kernel sees the same
// inlined helpers); not a stand-alone header.

// (The un-split transforms (N <= 1024) let the compiler hoist their few lane-derived addresses out of the frame loop: +1 %, 126
// VGPRs.  Everywhere else lane-derived values are re-materialised per frame -- opaque(): hoisting them all spills 85 registers at
// 1024 points.)

// ---------------------------------------------------------------------------------------------
// LDS images
//   complex image: position p at p + (p >> 4)            (one float2 of padding per 16)
//   real image   : sample  n at n + 4 * (n >> 4)         (16 B of padding per 16 floats, keeps
//                                                          16-byte alignment of 4-sample groups)
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr int cpad(int p) { return p + (p >> 4); }

template <int N> struct Geo {
    static constexpr int M      = N / 2;            // numMagnitudes (ref SpectralCharacteristics.h:104)
    static constexpr int P      = N / 64;           // samples per lane
    static constexpr int U      = M / 64;           // bins per lane
    // Windows >= 2048 exchange the complex image between FFT passes in two halves (every lane owns an even number of
    // items in the consuming pass, so the first half of its items is read before the producers write the second
    // half): the per-wave buffer is half a complex image instead of a whole one, which is what bounds how many
    // wavefronts a CU holds at these sizes.
    static constexpr bool SPLIT = N >= 2048;
    // real image: 4 floats of padding per RQ samples; RQ = the lane's own run of samples (>= 16), so that the 16-byte
    // accesses of lanes P samples apart fall on different banks (lane stride = 4 * odd dwords)
    static constexpr int RQ     = P > 16 ? P : 16;
    static constexpr int RIMG   = N + 4 * (N / RQ);                       // floats
    // bins image (re of bins < M, one run of U bins per lane): the same idea, 4 floats per U bins (U >= 8)
    static constexpr int BQ     = U >= 8 ? U : 0;
    static constexpr int BIMG   = M + (BQ ? 4 * (M / U) : 0);             // floats
    // ONE LDS buffer per wavefront, reused as: real image of the frame, complex image of each transform (the whole
    // image, or one half at a time), v array of the lag scan, harmonic scratch (bins image + 16-bit peak list).
    static constexpr int QSLOTS = N / 2 + (N == 2048 ? 8 * (N / 128) : 0);                  // second exchange (qpad)
    static constexpr int CSLOTS = SPLIT ? (cpad(N / 2 - 1) + 1 > QSLOTS ? cpad(N / 2 - 1) + 1 : QSLOTS) : cpad(N) + 2;  // float2 elements
    static constexpr int CBUF_B = 8 * CSLOTS;
    static constexpr int RBUF_B = 4 * (SPLIT ? RIMG - 4 : RIMG + 4);      // the last sample sits at RIMG - 5
    static constexpr int HBUF_B = 4 * BIMG + 2 * M + 16;
    static constexpr int BUF_B0 = CBUF_B > RBUF_B ? (CBUF_B > HBUF_B ? CBUF_B : HBUF_B) : (RBUF_B > HBUF_B ? RBUF_B : HBUF_B);
    static constexpr int BUF_BYTES = (BUF_B0 > 4 * N ? BUF_B0 : 4 * N) + 15 & ~15;     // 16-byte multiple, >= N floats (lag-scan spill)
    static constexpr int LOG2N  = (N == 256) ? 8 : (N == 512) ? 9 : (N == 1024) ? 10 : (N == 2048) ? 11 : 12;
    // first FFT pass: R inputs per item, G items per lane, R*G == P
    static constexpr int RA     = (N == 256) ? 4 : ((N == 512 || N == 2048) ? 8 : 16);
    static constexpr int LOG2RA = (RA == 4) ? 2 : (RA == 8) ? 3 : 4;
    static constexpr int ITEMS_A = N / RA;
    static constexpr int GA     = ITEMS_A / 64;
    static constexpr int IDIG   = (LOG2N - LOG2RA) / 2;   // base-4 digits of an item index
};

// position of sample n in the real image / of bin b in the bins image
template <int N> __host__ __device__ constexpr int rimg(int n) { return n + 4 * (int) ((unsigned) n / (unsigned) Geo<N>::RQ); }
template <int N> __host__ __device__ constexpr int bimg(int b) { return Geo<N>::BQ ? b + 4 * (int) ((unsigned) b / (unsigned) (Geo<N>::BQ ? Geo<N>::BQ : 1)) : b; }
// bin (or sample) lane + c with c a compile-time multiple of 64: one per-lane base plus a constant (64 is a multiple of
// every padding quantum), so the stores of a pass are one address register and immediate offsets
template <int N> __host__ __device__ constexpr int bimg_step(int c) { return Geo<N>::BQ ? c + 4 * (c / (Geo<N>::BQ ? Geo<N>::BQ : 1)) : c; }
template <int N> __host__ __device__ constexpr int rimg_step(int c) { return c + 4 * (c / Geo<N>::RQ); }
static_assert(64 % Geo<1024>::RQ == 0 && 64 % Geo<2048>::RQ == 0 && 64 % Geo<4096>::RQ == 0 && 64 % Geo<4096>::U == 0, "lane + 64*c splits");

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
#ifdef FX_EXP_FMA_TWIDDLES
    // EXPERIMENT BUILD ONLY (round 5, profiles/r05_fma_experiment.txt): the second product fused into the sum -- two packed instructions, one
    // rounding fewer per component, and spectra that are no longer the reference's.  What the FMA lever is worth on this chip, measured;
    // the library is never built with it (profiles/r05_fma_experiment.txt).
    if (INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(t), "v"(p));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(t), "v"(p));
    }
    (void) q;
    return r;
#else
    if (INV) {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(q) : "v"(a), "v"(t));
    } else {
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(a), "v"(t));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(q) : "v"(a), "v"(t));
    }
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(p), "v"(q));
    return r;
#endif
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
// The last stage of a transform whose consumer reads real parts only (re of bins < N/2 for the two analysers, re^2 for
// the pitch path, the squared real part for the lag search): x of outputs 0 and 1 (or of all four when ALL4), and y of
// output 0 when D0Y (getMagnitude's raw floats, imag[0] of the inverse transform).  Every value that is formed has the
// operands, the order and the roundings of twmul + bfly4_core; what no consumer reads is not computed: s1 needs its real
// part only (three plain instructions instead of three packed ones), s3 / s4 one half each (one packed add for both).
template <bool INV, bool ALL4, bool D0Y>
__device__ __forceinline__ void bfly4_re(f2& d0, f2& d1, f2& d2, f2& d3, f2 w1, f2 w2, f2 w3)
{
    const f2 s0 = twmul<INV>(d1, w1);
    const f2 s2 = twmul<INV>(d3, w3);
    float s1x, s1y = 0.0f;
    if (D0Y) { const f2 s1 = twmul<INV>(d2, w2); s1x = s1.x; s1y = s1.y; }
    else     s1x = INV ? (d2.x * w2.x) + (d2.y * w2.y) : (d2.x * w2.x) - (d2.y * w2.y);
    f2 t;                                           // (s3.x, s4.y) = (s0.x + s2.x, s0.y - s2.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(t) : "v"(s0), "v"(s2));
    const float ax = d0.x + s1x, s5x = d0.x - s1x;  // a.x, s5.x
    const float y0 = D0Y ? (d0.y + s1y) + (s0.y + s2.y) : 0.0f;
    d0 = f2{ax + t.x, y0};
    d1 = f2{INV ? s5x - t.y : s5x + t.y, 0.0f};
    if (ALL4) {
        d2 = f2{ax - t.x, 0.0f};
        d3 = f2{INV ? s5x + t.y : s5x - t.y, 0.0f};
    }
}
// Real part of output 0 alone: (d0.x + s1.x) + (s0.x + s2.x), each s a twiddle product's real part (one packed
// multiply + one add / subtract each) -- what the head of the lag search reads of an inverse transform.
template <bool INV>
__device__ __forceinline__ float bfly4_x0(f2 d0, f2 d1, f2 d2, f2 d3, f2 w1, f2 w2, f2 w3)
{
    const float s0x = INV ? (d1.x * w1.x) + (d1.y * w1.y) : (d1.x * w1.x) - (d1.y * w1.y);
    const float s1x = INV ? (d2.x * w2.x) + (d2.y * w2.y) : (d2.x * w2.x) - (d2.y * w2.y);
    const float s2x = INV ? (d3.x * w3.x) + (d3.y * w3.y) : (d3.x * w3.x) - (d3.y * w3.y);
    return (d0.x + s1x) + (s0x + s2x);
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

// COMPACT twiddle image (the 4096-point frame kernel): 24 KB instead of the pass-ordered table's 32, which together with the flux
// state left in global memory is what an EIGHTH wavefront's transform buffer needs (160 KB = 24 KB + 8 x 17 KB).  The last pass's
// fifteen rows of L2 = 256 entries (row i of item k: i < 3 the first stage tw[4*k*(i+1)], i = 3 + 3*jin + (q-1) the second stage
// tw[(k + 256*jin) * q]) shrink to eleven because the q = 1 rows of the four jin are one contiguous run -- tw[0 .. 1023], the first
// quarter of the reference's table -- and the q = 2 rows are the same run's even entries (2*k + 512*jin) for jin < 2 and a QUARTER
// TURN of those for jin = 2, 3: tw[j + N/4] = (y, -x) of tw[j] = (x, y) -- IF the float table has that symmetry.  The host checks the
// entries concerned (FrameParams::tw_quarter_turn); should it ever not hold, those two rows are read from the whole table in global
// memory instead (slower, same values).  The run is stored even entries first, odd entries behind them and 16 entries (32 banks)
// further on: the q = 2 rows then read consecutive entries, and of the q = 1 rows' 32 lanes per LDS pass the even ones fall on one
// half of the banks and the odd ones on the other.  The 16-entry gap holds the workgroup's hand-over counters (FrameLds).
//   [0, 240)      the second pass's 15 * L1 entries, as in the pass-ordered table
//   [240, 1008)   S : rows 0, 1, 2
//   [1008, 2048)  Q1: tw[j] at q1_pos(j) = j/2 + (j odd ? 528 : 0), j < 1024; [1008 + 512, 1008 + 528) free
//   [2048, 3072)  R3: the q = 3 rows, tw[3*k + 768*jin]
// (One entry never has the symmetry: tw[N/4] = ((float) cos(pi/2 in double), -1) = (6.1e-17, -1), not the quarter turn (-0, -1) of
// tw[0] = (1, -0).  It is item 0's twiddle 10 alone and travels as a kernel argument: TwGlobal::at_quarter.)
struct TwGlobal { const f2* table; bool quarter_turn; f2 at_quarter; };       // the pass-ordered table in global memory; tw[N/4]
template <int N> struct CompactTw {
    typedef Plan<N> PL;
    static constexpr int L2 = PL::L2;
    static constexpr int Q1_GAP = 16;
    static constexpr int OFF_S = 15 * PL::L1, OFF_Q1 = OFF_S + 3 * L2, OFF_GAP = OFF_Q1 + 2 * L2, OFF_R3 = OFF_Q1 + 4 * L2 + Q1_GAP, ENTRIES = OFF_R3 + 4 * L2;
    __host__ __device__ static constexpr int q1_pos(int j) { return (j >> 1) + ((j & 1) ? 2 * L2 + Q1_GAP : 0); }
    // where entry e of the image comes from in the pass-ordered table (build_twiddle_image lays the image out on the host; the kernels copy it as it is)
    __host__ __device__ static constexpr int source(int e)
    {
        if (e < OFF_S)  return PL::OFF1 + e;
        if (e < OFF_Q1) return PL::OFF2 + (e - OFF_S);                                               // rows 0..2 are contiguous
        if (e < OFF_R3) {
            const int pos = e - OFF_Q1;
            if (pos >= 2 * L2 && pos < 2 * L2 + Q1_GAP) return -1;                                   // the gap: not a twiddle
            const int j = pos < 2 * L2 ? 2 * pos : 2 * (pos - 2 * L2 - Q1_GAP) + 1;                  // tw[j]: row 3 + 3*(j / 256) of the table
            return PL::OFF2 + (3 + 3 * (j / L2)) * L2 + j % L2;
        }
        return PL::OFF2 + (5 + 3 * ((e - OFF_R3) / L2)) * L2 + (e - OFF_R3) % L2;
    }
};
static_assert(CompactTw<4096>::ENTRIES == 3072 && CompactTw<4096>::OFF_S == 240 && CompactTw<4096>::q1_pos(1) == 528 && CompactTw<4096>::q1_pos(1022) == 511, "compact twiddle image");

// offset of element i of an item inside the padded complex image, relative to cpad(base):
// cpad(base + L0*i) - cpad(base) is a compile-time constant because base = blk*(R*L0) + k, k < L0
__host__ __device__ constexpr int item_off(int L0, int i) { return L0 * i + (L0 >= 16 ? (L0 / 16) * i : ((L0 * i) >> 4)); }

// The first exchange of a REAL-input transform need not carry the whole item.  A 16-element first-pass item is a 16-point
// transform of real data: e[3], e[7], e[11], e[15] are formed as the bitwise conjugates of e[13], e[9], e[5], e[1] (first_pass_item),
// e[12] is the bitwise conjugate of e[4] (bfly4_real), and the consumer -- the next pass's item k = lane % 16 reads element k of
// sixteen first-pass items -- can take the twin and flip a sign instead.  Eleven stores per item instead of sixteen: LDS stores are
// what a wavefront's exchange waits for (round 3: five stores more per transform cost the 1024-point kernel 4 %).
// Layout: [item][11] float2, no padding (rows of 88 B: the 16 lanes of a ds_write_b64 group and the 32 of a ds_read_b64 group fall
// on distinct banks; lanes that want the same element of the same item read one address).
template <int N> struct RealExchange {
    // 1024 (un-split, one item per lane) and 4096 (split, two per half): 16-element items, next pass at stride 16.
    // 2048 (split) has 8-element items (a radix-2 and a radix-4 stage) of which only e[6] = conj(e[2]) is exact: 7 stores of 8 are
    // the same four store instructions and the sign flips cost 1 % (measured, profiles/r03_variants.txt): not used there.
    static constexpr bool USE = N == 1024 || N == 4096;
    static constexpr int RA = Geo<N>::RA;
    static constexpr int SLOTS = RA == 16 ? 11 : 7;
    // float2 of padding behind every 16 items: 7-slot rows need it to keep the 32 lanes of a read group on distinct banks
    static constexpr int GROUP_PAD = RA == 16 ? 0 : 8;
    // element k of an item -> the slot that holds it or its twin (4 bits each), and whether it is the twin (conjugate)
    static constexpr unsigned long long SLOT_OF = RA == 16 ? 0x1a93487675439210ull     // stored, in slot order: 0 1 2 4 5 6 8 9 10 13 14
                                                           : 0x62543210ull;            // 0 1 2 3 4 5 7
    static constexpr unsigned TWIN = RA == 16 ? (1u << 3) | (1u << 7) | (1u << 11) | (1u << 12) | (1u << 15) : (1u << 6);
    __host__ __device__ static constexpr int stored(int q)
    {
        constexpr int s16[11] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 13, 14};
        constexpr int s8[7] = {0, 1, 2, 3, 4, 5, 7};
        return RA == 16 ? s16[q < 11 ? q : 0] : s8[q < 7 ? q : 0];
    }
    // position (float2) of slot 0 of first-pass item `item` (an index within the image being exchanged)
    __host__ __device__ static constexpr int row(int item) { return item * SLOTS + GROUP_PAD * (item / 16); }
};

// A later pass: every item of R elements (stride L0) is loaded from the complex image, its 1 or 2
// radix-4 stages run in registers, and it is stored back to the same positions.
// FROM_REAL: this is the pass behind the first exchange of a transform whose first pass used RealExchange.
template <int N, int R, int L0, int TWOFF, bool INV, bool FROM_REAL = false>
__device__ __forceinline__ void fft_pass(f2* cbuf, const f2* tw, int lane)
{
    constexpr int ITEMS = N / R;
    for (int it = lane; it < ITEMS; it += 64) {
        f2 e[R];
        const int k = it % L0;
        const int base = (it / L0) * (R * L0) + k;
        f2* img = cbuf + cpad(base);
        const f2* t1 = tw + TWOFF + k;
        if constexpr (FROM_REAL) {
            // element k of the first-pass items (it / 16) * 16 + i: from its slot, or from its twin's with the sign of im flipped
            static_assert(R == 16 && L0 == 16 && ITEMS == 64, "the pass behind a 16-element real first pass");
            const int slot = (int) ((RealExchange<N>::SLOT_OF >> (4 * k)) & 15ull);
            const unsigned flip = ((RealExchange<N>::TWIN >> k) & 1u) << 31;
            const f2* src = cbuf + RealExchange<N>::row((it / L0) * 16) + slot;
#pragma unroll
            for (int i = 0; i < R; i++) {
                const f2 v = src[i * RealExchange<N>::SLOTS];
                e[i] = f2{v.x, __uint_as_float(__float_as_uint(v.y) ^ flip)};
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; i++) e[i] = img[item_off(L0, i)];
        }
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
// ITEMS_A is a multiple of RQ, so rimg(low + ITEMS_A*r) = rimg(low) + (ITEMS_A + 4*ITEMS_A/RQ)*r and the
// RA accesses of an item are one address register plus immediate offsets.
template <int N> __device__ __forceinline__ int first_pass_rbase(int lane, int g) { return rimg<N>(rev4<Geo<N>::IDIG>(lane + 64 * g)); }
template <int N> __host__ __device__ constexpr int first_pass_rstep(int j)
{
    return (Geo<N>::ITEMS_A + 4 * (Geo<N>::ITEMS_A / Geo<N>::RQ))
         * ((Geo<N>::RA == 4) ? j : (Geo<N>::RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3)));
}
static_assert(Geo<256>::ITEMS_A % Geo<256>::RQ == 0 && Geo<512>::ITEMS_A % Geo<512>::RQ == 0 && Geo<1024>::ITEMS_A % Geo<1024>::RQ == 0
              && Geo<2048>::ITEMS_A % Geo<2048>::RQ == 0 && Geo<4096>::ITEMS_A % Geo<4096>::RQ == 0, "the real image splits only at multiples of RQ");

// One first-pass item: R REAL inputs (imag = 0, as in performRealOnlyForwardTransform and in PitchAnalyser's re*re
// spectrum), already in first_pass_index order.  Stages at length 1 have unit twiddles and real operands; the stage
// after them sees real operands in half of its butterflies.
template <int N, bool INV>
__device__ __forceinline__ void first_pass_item(const float* x, const f2 (&ta)[9], f2 (&e)[Geo<N>::RA])
{
    constexpr int R = Geo<N>::RA;
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
        // The item is a 16-point transform of real data, and the jin = 3 butterfly is the mirror image of the
        // jin = 1 butterfly: its operands are the conjugates (bfly4_real: d3 = conj(d1)) and its twiddles W^3, W^6,
        // W^9 are -i*conj(W^1), -conj(W^2), i*conj(W^3) -- bit for bit, which the host verifies on the float table
        // before any launch (first_pass_twiddles_hermitian).  Every product and sum then comes out as the exact
        // conjugate (negation and operand order do not change a rounding): out'[0..3] = conj(out[3], out[2],
        // out[1], out[0]).  (The jin = 2 butterfly has no such shortcut: the table's cos(pi/2) is 6e-17, not 0.)
        e[3] = f2{e[13].x, -e[13].y}; e[7] = f2{e[9].x, -e[9].y}; e[11] = f2{e[5].x, -e[5].y}; e[15] = f2{e[1].x, -e[1].y};
    }
}

// First pass of an un-split transform: the lane's P real inputs -> the complex image.
template <int N, bool INV>
__device__ __forceinline__ void fft_first_pass(const float (&xin)[Geo<N>::P], f2* cbuf, const float (&ftw)[18], int lane)
{
    typedef Geo<N> G;
    constexpr int R = G::RA;
    f2 ta[9];                      // wave-uniform: kernel arguments, not LDS
#pragma unroll
    for (int i = 0; i < 9; i++) ta[i] = f2{ftw[2 * i], ftw[2 * i + 1]};
#pragma unroll
    for (int g = 0; g < G::GA; g++) {
        f2 e[R];
        first_pass_item<N, INV>(&xin[g * R], ta, e);
        if constexpr (RealExchange<N>::USE) {
            static_assert(R == 16 && G::GA == 1, "one 16-element item per lane");
            f2* img = cbuf + RealExchange<N>::row(lane);
#pragma unroll
            for (int q = 0; q < RealExchange<N>::SLOTS; q++) img[q] = e[RealExchange<N>::stored(q)];
        } else {
            f2* img = cbuf + cpad((lane + 64 * g) * R);
#pragma unroll
            for (int i = 0; i < R; i++) img[i] = e[i];         // R <= 16 contiguous positions: no pad inside
        }
    }
    wave_fence();
}

// The two radix-4 stages of a 16-element item (stride L0 between elements), in registers.
template <int L0, bool INV>
__device__ __forceinline__ void item16_stages(f2 (&e)[16], const f2* t1)
{
    {
        const f2 w1 = t1[0], w2 = t1[L0], w3 = t1[2 * L0];
#pragma unroll
        for (int g = 0; g < 4; g++)
            bfly4_core<INV>(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3],
                            twmul<INV>(e[4 * g + 1], w1), twmul<INV>(e[4 * g + 2], w2), twmul<INV>(e[4 * g + 3], w3));
    }
    const f2* t2 = t1 + 3 * L0;
#pragma unroll
    for (int jin = 0; jin < 4; jin++) {
        const f2 w1 = t2[(jin * 3 + 0) * L0], w2 = t2[(jin * 3 + 1) * L0], w3 = t2[(jin * 3 + 2) * L0];
        bfly4_core<INV>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12],
                        twmul<INV>(e[jin + 4], w1), twmul<INV>(e[jin + 8], w2), twmul<INV>(e[jin + 12], w3));
    }
}

// The same with the item's 15 twiddles already in registers (w[0..2]: first stage; w[3 + 3*jin + q]: second stage).
template <bool INV>
__device__ __forceinline__ void item16_stages_r(f2 (&e)[16], const f2 (&w)[15])
{
#pragma unroll
    for (int g = 0; g < 4; g++)
        bfly4_core<INV>(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3],
                        twmul<INV>(e[4 * g + 1], w[0]), twmul<INV>(e[4 * g + 2], w[1]), twmul<INV>(e[4 * g + 3], w[2]));
#pragma unroll
    for (int jin = 0; jin < 4; jin++)
        bfly4_core<INV>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12],
                        twmul<INV>(e[jin + 4], w[3 + 3 * jin]), twmul<INV>(e[jin + 8], w[4 + 3 * jin]), twmul<INV>(e[jin + 12], w[5 + 3 * jin]));
}

// What the last pass leaves in the wave's LDS buffer.
enum { OUT_RE_LOW = 1,        // float re[M] (plain layout): all the harmonic analyser reads (ref HarmonicCharacteristics.h:63)
       OUT_RE_LOW_MAXABS = 2, // the same + max(|re|,|im|) over bins [0, M/2) returned per lane (ref SpectralCharacteristics.h:153)
       OUT_POWER = 3,         // re*re of all N bins (ref PitchAnalyser.h:97-103) as the NEXT transform's first-pass inputs: in registers
                              // (regs_out, one lane permutation) when a lane owns one first-pass item, else in the real image (rpad layout)
       OUT_LAG = 4 };         // v[s] = (re_s/N)^2 * s of the lane's own samples s = lane + 64*m in registers (regs_out[m]); returns v[N], from imag[0], in lane 0: ref :119-123

// The two stages of a LAST-pass 16-element item: the second stage forms only what the consumer of the spectrum reads
// (bfly4_re; OUT as in fft_from_regs; `first_item`: the item whose element 0 is bin 0).  w(i): the item's 15 twiddles.
template <bool INV, int OUT, typename W>
__device__ __forceinline__ void item16_last(f2 (&e)[16], W w, bool first_item)
{
    constexpr bool ALL4 = !(OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS);          // those read bins < N/2: elements i < 8
#pragma unroll
    for (int g = 0; g < 4; g++)
        bfly4_core<INV>(e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3],
                        twmul<INV>(e[4 * g + 1], w(0)), twmul<INV>(e[4 * g + 2], w(1)), twmul<INV>(e[4 * g + 3], w(2)));
#pragma unroll
    for (int jin = 0; jin < 4; jin++) {
        const f2 w1 = w(3 + 3 * jin), w2 = w(4 + 3 * jin), w3 = w(5 + 3 * jin);
        if (OUT == OUT_RE_LOW_MAXABS)                       // max(|re|, |im|) over elements i < 4: y of every output 0
            bfly4_re<INV, ALL4, true>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12], w1, w2, w3);
        else if (OUT == OUT_LAG && jin == 0 && first_item)  // imag[0] of the inverse transform
            bfly4_re<INV, ALL4, true>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12], w1, w2, w3);
        else
            bfly4_re<INV, ALL4, false>(e[jin], e[jin + 4], e[jin + 8], e[jin + 12], w1, w2, w3);
    }
}

// Twiddles a lane needs for the second and the last pass of a split transform, held in registers for the wave's whole
// life (2048 points: the kernel runs at 2 waves per SIMD there, which leaves the registers, and VALU issue and the LDS
// pipe are co-limiting, so the 180 twiddle reads per frame -- a fifth of its LDS instructions -- are worth removing).
// The second pass's twiddles depend on it % L1 only, which is the same for all of a lane's items.
template <int N> struct TwRegs {
    static constexpr bool USE = N == 2048;
    static constexpr bool USE_C = USE;
    static constexpr int GB = USE_C ? (N / 16) / 64 : 1;
    f2 b[15];
    f2 c[GB][15];
    __device__ __forceinline__ void load(const f2* tw, int lane)
    {
        typedef Plan<N> PL;
        const f2* t1 = tw + PL::OFF1 + lane % PL::L1;
#pragma unroll
        for (int i = 0; i < 15; i++) b[i] = t1[i * PL::L1];
#pragma unroll
        for (int g = 0; g < (USE_C ? GB : 0); g++) {
            const f2* t2 = tw + PL::OFF2 + lane + 64 * g;
#pragma unroll
            for (int i = 0; i < 15; i++) c[g][i] = t2[i * PL::L2];
        }
    }
};

// Split transforms (N >= 2048): position of a complex element inside the HALF image of the second exchange (second
// pass -> last pass).  q = k' + (L2/2)*i with k' < L2/2 the last-pass item within its half and i its element: rows of
// L2/2 consecutive k'.  The last pass reads a row with 64 consecutive lanes (conflict-free for any row offset); the
// second pass writes runs of L1 consecutive k' that are L2/2 apart, which need the rows 8 slots apart in bank
// position when L1 = 8 (N = 2048) and nothing when L1 = 16.
template <int N> __host__ __device__ constexpr int qpad(int q) { return N == 2048 ? q + 8 * (q >> 6) : q; }
static_assert(qpad<2048>(1023) < Geo<2048>::CSLOTS && qpad<4096>(2047) < Geo<4096>::CSLOTS, "half image fits the wave buffer");

// Last pass -- radix 4 at length N/4 (N <= 1024) or radix 16 at length N/16 -- with all of a lane's items in
// registers (128 VGPRs of them at N = 4096, which runs one wave per SIMD anyway), fused with the consumer of the spectrum, so the full complex image is never written back and
// re-read: the spectral / harmonic analysers only read re of bins < N/2, the pitch analyser only re*re, the
// lag search only the squared, lag-weighted real part.
// Last pass, compute half: e[g][*] holds the items of this lane (item k = lane + 64*g).  Radix 4 at length N/4
// (N <= 1024) or radix 16 at length N/16, fused with the consumer of the spectrum, so the full complex image is never
// written back and re-read: the spectral / harmonic analysers only read re of bins < N/2, the pitch analyser only
// re*re, the lag search only the squared, lag-weighted real part.  The wave's buffer must be free (every lane has read
// its inputs) when this is called.
template <int N, bool INV, int OUT>
__device__ __forceinline__ float fft_last_pass_consume(f2 (&e)[(N / Plan<N>::R2) / 64][Plan<N>::R2], f2* cbuf, const f2* tw, int lane, float scale, float* regs_out)
{
    typedef Plan<N> PL;
    constexpr int R = PL::R2, L0 = PL::L2, GI = (N / R) / 64, TWOFF = PL::OFF2;
    float* fbuf = reinterpret_cast<float*>(cbuf);
    float aux = 0.0f;
#pragma unroll
    for (int g = 0; g < GI; g++) {
        const int k = lane + 64 * g;
        const f2* t1 = tw + TWOFF + k;
        if constexpr (R == 16) {
            item16_stages<L0, INV>(e[g], t1);
        } else {
            const f2 w1 = t1[0], w2 = t1[L0], w3 = t1[2 * L0];
            constexpr bool ALL4 = !(OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS);
            if (OUT == OUT_RE_LOW_MAXABS || (OUT == OUT_LAG && g == 0)) bfly4_re<INV, ALL4, true>(e[g][0], e[g][1], e[g][2], e[g][3], w1, w2, w3);
            else bfly4_re<INV, ALL4, false>(e[g][0], e[g][1], e[g][2], e[g][3], w1, w2, w3);
        }
        // e[g][i] is bin k + L0*i
#pragma unroll
        for (int i = 0; i < R; i++) {
            const int bin = k + L0 * i;
            if (OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS) {
                if (i < R / 2) fbuf[bimg<N>(lane) + bimg_step<N>(64 * g + L0 * i)] = e[g][i].x;     // bins >= N/2 are never read
                if (OUT == OUT_RE_LOW_MAXABS && i < R / 4) aux = fmaxf(aux, fmaxf(fabsf(e[g][i].x), fabsf(e[g][i].y)));
            } else if (OUT == OUT_POWER) {
                if constexpr (Geo<N>::GA == 1) {
                    // Lane l holds the bins congruent to l mod 64: bin = l + 64*m, m = g + (L0/64)*i.  The next
                    // transform's first pass wants, in lane l', the samples rev4(l') + 64*r(j): the whole set of lane
                    // rev4(l'), in the order m = r(j).  One ds_bpermute per register instead of a store + gather.
                    constexpr int RA = Geo<N>::RA;
                    const int m = g + (L0 / 64) * i;
                    // j with r(j) == m  (r is its own inverse up to the digit swap used by first_pass_index)
                    const int j = (RA == 4) ? m : (RA == 8) ? (2 * (m & 3) + (m >> 2)) : (4 * (m & 3) + (m >> 2));
                    regs_out[j] = __int_as_float(__builtin_amdgcn_ds_bpermute(
                        rev4<Geo<N>::IDIG>(lane) << 2, __float_as_int(e[g][i].x * e[g][i].x)));
                } else {
                    fbuf[rimg<N>(lane) + rimg_step<N>(64 * g + L0 * i)] = e[g][i].x * e[g][i].x;
                }
            } else {
                const float d = e[g][i].x * scale;
                regs_out[g + (L0 / 64) * i] = d * d * (float) bin;                // bin = lane + 64 * (g + (L0/64) * i)
            }
        }
        if (OUT == OUT_LAG && g == 0) { const float d = e[0][0].y * scale; aux = d * d * (float) N; }
    }
    wave_fence();
    return aux;
}

// Last pass of an un-split transform: all of a lane's items are read from the complex image first.
template <int N, bool INV, int OUT>
__device__ __forceinline__ float fft_last_pass_fused(f2* cbuf, const f2* tw, int lane, float scale, float* regs_out)
{
    typedef Plan<N> PL;
    constexpr int R = PL::R2, L0 = PL::L2, GI = (N / R) / 64;
    f2 e[GI][R];
#pragma unroll
    for (int g = 0; g < GI; g++) {
        const f2* img = cbuf + cpad(lane + 64 * g);
#pragma unroll
        for (int i = 0; i < R; i++) e[g][i] = img[item_off(L0, i)];
    }
    wave_fence();                 // the wave has read the whole complex image; the buffer may be rewritten
    return fft_last_pass_consume<N, INV, OUT>(e, cbuf, tw, lane, scale, regs_out);
}

// What the consumer keeps of a finished 16-element last-pass item (item k: element i is bin k + (N/16)*i).
template <int N, int OUT>
__device__ __forceinline__ void last_item_reduce(const f2 (&e)[16], int k, float scale, float (&res)[16], float& aux)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS) {
            if (i < 8) res[i] = e[i].x;                                        // bins >= N/2 are never read
            if (OUT == OUT_RE_LOW_MAXABS && i < 4) aux = fmaxf(aux, fmaxf(fabsf(e[i].x), fabsf(e[i].y)));
        } else if (OUT == OUT_POWER) {
            res[i] = e[i].x * e[i].x;
        } else {
            const float d = e[i].x * scale;
            res[i] = d * d * (float) (k + (N / 16) * i);
        }
    }
}

// The inverse transform of the pitch path at 1024 points with its LAST pass left to the lag search, which reads
// v[s] = (re_s / N)^2 * s from s = 0 upwards and is usually decided within the first 64 or 128 lags: the four last-pass
// butterflies of a lane are held as operands (e[g], butterfly k = lane + 64*g, output i = sample k + 256*i), and
//   head(g) : v[lane + 64*g], g = 0..3 -- output 0 of butterfly g alone (bfly4_x0)
//   rest()  : every other sample, and v[N] from imag[0] (lane 0), when the search goes past sample 255.
// Each value has the operands and roundings of the whole pass (fft_last_pass_consume, OUT_LAG).
template <int N> struct LazyLag {
    static_assert(N == 1024, "four last-pass butterflies per lane, one 64-sample block per butterfly's output 0");
    typedef Plan<N> PL;
    static constexpr int L0 = PL::L2;
    f2 e[4][4];
    const f2* t1;       // tw + OFF2 + lane
    float scale;
    int lane;
    __device__ __forceinline__ void load(const float (&xin)[Geo<N>::P], f2* cbuf, const f2* tw, const float (&ftw)[18], int lane_, float scale_)
    {
        fft_first_pass<N, true>(xin, cbuf, ftw, lane_);
        fft_pass<N, PL::R1, PL::L1, PL::OFF1, true, RealExchange<N>::USE>(cbuf, tw, lane_);
        lane = lane_;
        scale = scale_;
        t1 = tw + PL::OFF2 + lane;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f2* img = cbuf + cpad(lane + 64 * g);
#pragma unroll
            for (int i = 0; i < 4; i++) e[g][i] = img[item_off(L0, i)];
        }
        wave_fence();                 // the wave has read the whole complex image; the buffer may be rewritten
    }
    __device__ __forceinline__ float head(int g) const          // g: compile-time after unrolling
    {
        const f2* t = t1 + 64 * g;
        const float d = bfly4_x0<true>(e[g][0], e[g][1], e[g][2], e[g][3], t[0], t[L0], t[2 * L0]) * scale;
        return d * d * (float) (lane + 64 * g);
    }
    __device__ __forceinline__ float rest(float (&vreg)[Geo<N>::P])
    {
        float v_end = 0.0f;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f2* t = t1 + 64 * g;
            if (g == 0) bfly4_re<true, true, true>(e[g][0], e[g][1], e[g][2], e[g][3], t[0], t[L0], t[2 * L0]);
            else        bfly4_re<true, true, false>(e[g][0], e[g][1], e[g][2], e[g][3], t[0], t[L0], t[2 * L0]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float d = e[g][i].x * scale;
                vreg[g + 4 * i] = d * d * (float) (lane + 64 * g + L0 * i);
            }
            if (g == 0) { const float d = e[0][0].y * scale; v_end = d * d * (float) N; }
        }
        return v_end;
    }
};

// Split transform (N >= 2048): three passes with both exchanges done one half at a time through a half-size image.
//   first pass  : RA real inputs per item, ITEMS_A/64 = 4 items per lane  (items 0,1 = half 0; items 2,3 = half 1)
//   second pass : 16-element items at stride L1, GB = N/1024 items per lane: item lane + 64*j reads positions
//                 (it/L1)*16*L1 + it%L1 + L1*i, which lie in the first-pass half [it >= ITEMS_B/2]
//   last pass   : 16-element items at stride L2 = N/16, GB items per lane: item k = lane + 64*g reads positions
//                 k + L2*i, produced as element i' of second-pass items with (k'' + L1*i') mod L2 = k, i.e. by
//                 elements i' < 8 for the first half of the g's and i' >= 8 for the second half.
// CTW: `tw` is the compact image (CompactTw<N>), `tg` the whole table in global memory.
template <int N, bool INV, int OUT, bool CTW = false>
__device__ __forceinline__ float fft_split(const float (&xin)[Geo<N>::P], f2* cbuf, const f2* tw, const float (&ftw)[18],
                                           int lane, float scale, float* regs_out, const TwRegs<N>* twr, TwGlobal tg = TwGlobal{nullptr, false, f2{0.0f, 0.0f}})
{
    typedef Geo<N> G;
    typedef Plan<N> PL;
    constexpr int RA = G::RA, GA = G::GA, L1 = PL::L1, L2 = PL::L2, GB = (N / 16) / 64, HB = GB / 2;
    static_assert(PL::R1 == 16 && PL::R2 == 16 && GA == 4 && GB >= 2 && GB % 2 == 0, "split plan: three passes, even item counts");
    lane = opaque<N>(lane);
    f2 ta[9];
#pragma unroll
    for (int i = 0; i < 9; i++) ta[i] = f2{ftw[2 * i], ftw[2 * i + 1]};
    f2 eb[GB][16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // first-pass items of this half -> half image (item lane + 64*g' at local position (lane + 64*g')*RA)
#pragma unroll
        for (int gl = 0; gl < GA / 2; gl++) {
            f2 e[RA];
            first_pass_item<N, INV>(&xin[(h * (GA / 2) + gl) * RA], ta, e);
            if constexpr (RealExchange<N>::USE) {
                static_assert(RA == L1, "the next pass's stride is the item's length: its lanes read ONE element index each");
                f2* img = cbuf + RealExchange<N>::row(lane + 64 * gl);
#pragma unroll
                for (int q = 0; q < RealExchange<N>::SLOTS; q++) img[q] = e[RealExchange<N>::stored(q)];
            } else {
                f2* img = cbuf + cpad((lane + 64 * gl) * RA);
#pragma unroll
                for (int i = 0; i < RA; i++) img[i] = e[i];
            }
        }
        wave_fence();
        // second-pass items of this half: local item lane + 64*jl
#pragma unroll
        for (int jl = 0; jl < HB; jl++) {
            const int it = lane + 64 * jl;
            if constexpr (RealExchange<N>::USE) {
                // element k = it % L1 of the first-pass items (it / L1) * 16 + i: its slot, or its twin's with the sign of im flipped
                const int k = it % L1;
                const int slot = (int) ((RealExchange<N>::SLOT_OF >> (4 * k)) & 15ull);
                const unsigned flip = ((RealExchange<N>::TWIN >> k) & 1u) << 31;
                const f2* src = cbuf + RealExchange<N>::row((it / L1) * 16) + slot;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const f2 v = src[i * RealExchange<N>::SLOTS];
                    eb[h * HB + jl][i] = f2{v.x, __uint_as_float(__float_as_uint(v.y) ^ flip)};
                }
            } else {
                const f2* img = cbuf + cpad((it / L1) * (16 * L1) + it % L1);
#pragma unroll
                for (int i = 0; i < 16; i++) eb[h * HB + jl][i] = img[item_off(L1, i)];
            }
        }
        wave_fence();             // every lane has its inputs: the next half may overwrite the image
#pragma unroll
        for (int jl = 0; jl < HB; jl++) {
            const int it = lane + 64 * jl;
            if constexpr (TwRegs<N>::USE) item16_stages_r<INV>(eb[h * HB + jl], twr->b);
            else item16_stages<L1, INV>(eb[h * HB + jl], tw + PL::OFF1 + it % L1);
        }
    }
    // second exchange.  Element i of second-pass item it = lane + 64*j sits at position (it/L1)*L2 + it%L1 + L1*i
    // (16*L1 == L2): row it/L1 of the last pass's input (its element index), column k = it%L1 + L1*i (its item), and
    // the half of the last-pass items that column belongs to is k >= L2/2, i.e. i >= 8.  The last-pass items of a half
    // are finished as soon as they are read (and reduced to what the consumer keeps: <= 16 floats per item), so that the
    // other half's second-pass results are the only other large live set.
    static_assert(16 * L1 == L2, "a second-pass item spans exactly one row of the last pass");
    float res[GB][16];
    float aux = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int j = 0; j < GB; j++) {
            const int it = lane + 64 * j;
            f2* img = cbuf + qpad<N>((L2 / 2) * (it / L1)) + it % L1;          // row start + column of element 8*h
#pragma unroll
            for (int i = 0; i < 8; i++) img[L1 * i] = eb[j][8 * h + i];
        }
        wave_fence();
        f2 ec[HB][16];
#pragma unroll
        for (int gl = 0; gl < HB; gl++) {
            const f2* img = cbuf + (lane + 64 * gl);
#pragma unroll
            for (int i = 0; i < 16; i++) ec[gl][i] = img[qpad<N>((L2 / 2) * i)];
        }
        wave_fence();
#pragma unroll
        for (int gl = 0; gl < HB; gl++) {
            const int g = h * HB + gl, k = lane + 64 * g;
            if constexpr (TwRegs<N>::USE_C) { const f2 (&wr)[15] = twr->c[g]; item16_last<INV, OUT>(ec[gl], [&](int i) { return wr[i]; }, g == 0); }
            else if constexpr (CTW) {
                static_assert(!TwRegs<N>::USE, "compact image: twiddles from LDS");
                typedef CompactTw<N> CT;
                // (two per-lane addresses for all of a lane's items: everything else is a compile-time offset -- item k = lane + 64*g)
                const f2* at_lane = tw + lane;
                const f2* at_half = tw + CT::q1_pos(lane);
                const f2* ts = at_lane + (CT::OFF_S + 64 * g);
                const f2* q1 = at_half + (CT::OFF_Q1 + 32 * g);           // q1_pos(lane + 64*g); tw[k + 256*jin] sits 128 entries further per jin
                const f2* q2 = at_lane + (CT::OFF_Q1 + 64 * g);           // tw[2*k + 512*jin]: entry k + 256*jin of the even half
                const f2* r3 = at_lane + (CT::OFF_R3 + 64 * g);
                item16_last<INV, OUT>(ec[gl], [&](int i) -> f2 {
                    if (i < 3) return ts[i * L2];
                    const int jin = (i - 3) / 3, q = (i - 3) % 3 + 1;
                    if (q == 1) return q1[jin * (L2 / 2)];
                    if (q == 3) return r3[jin * L2];
                    if (jin < 2) return q2[jin * L2];
                    if (!tg.quarter_turn) return tg.table[PL::OFF2 + k + i * L2];
                    const f2 t = q2[(jin - 2) * L2];
                    if (jin == 2 && g == 0 && lane == 0) return tg.at_quarter;         // k == 0: tw[N/4] itself
                    return f2{t.y, -t.x};
                }, g == 0);
            }
            else { const f2* t2 = tw + PL::OFF2 + k; item16_last<INV, OUT>(ec[gl], [&](int i) { return t2[i * L2]; }, g == 0); }
            last_item_reduce<N, OUT>(ec[gl], k, scale, res[g], aux);
            if (OUT == OUT_LAG && g == 0) { const float d = ec[0][0].y * scale; aux = d * d * (float) N; }
        }
    }
    // the buffer is free: hand the results to the consumer
    float* fbuf = reinterpret_cast<float*>(cbuf);
#pragma unroll
    for (int g = 0; g < GB; g++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS) { if (i < 8) fbuf[bimg<N>(lane) + bimg_step<N>(64 * g + L2 * i)] = res[g][i]; }
            else if (OUT == OUT_POWER) fbuf[rimg<N>(lane) + rimg_step<N>(64 * g + L2 * i)] = res[g][i];
            else regs_out[g + (L2 / 64) * i] = res[g][i];                      // bin = lane + 64 * (g + (L2/64) * i)
        }
    }
    wave_fence();
    return aux;
}

// Whole transform of one wavefront: P real inputs per lane (first-pass order) -> OUT (see above).
template <int N, bool INV, int OUT, bool CTW = false>
__device__ __forceinline__ float fft_from_regs(const float (&xin)[Geo<N>::P], f2* cbuf, const f2* tw, const float (&ftw)[18],
                                               int lane, float scale = 0.0f, float* regs_out = nullptr, const TwRegs<N>* twr = nullptr,
                                               TwGlobal tg = TwGlobal{nullptr, false, f2{0.0f, 0.0f}})
{
    typedef Plan<N> PL;
    static_assert(!CTW || Geo<N>::SPLIT, "the compact twiddle image belongs to a split transform");
    if constexpr (Geo<N>::SPLIT) {
        return fft_split<N, INV, OUT, CTW>(xin, cbuf, tw, ftw, lane, scale, regs_out, twr, tg);
    } else {
        fft_first_pass<N, INV>(xin, cbuf, ftw, lane);
        fft_pass<N, PL::R1, PL::L1, PL::OFF1, INV, RealExchange<N>::USE>(cbuf, tw, lane);
        return fft_last_pass_fused<N, INV, OUT>(cbuf, tw, lane, scale, regs_out);
    }
}
