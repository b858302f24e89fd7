// fx_pair_kernel.hip.h -- fx_pair_kernel<N>: windows of 2048 and 4096 points with ONE FRAME ACROSS TWO WAVEFRONTS.
// Included by fx_kernels.hip inside namespace fxk; not a stand-alone header.
//
// Why: with a whole frame in one wavefront (fx_frame_kernel) these sizes hold 8 (2048-pt) and 7 (4096-pt) wavefronts per
// CU -- the transform buffer and a lane's 64-128 second-pass results bound them -- and run latency-bound (VALU 76 % / 53 %
// busy, round 2).  Here the 128 lanes of a wavefront PAIR share one frame and one 17 KB buffer: half the registers per
// lane, twice the wavefronts per CU for the same LDS, every section of the reference's two run() loops data-parallel
// over the 128 lanes.  Same butterfly DAG, same operands, same roundings as the one-wavefront kernel (the FFT helpers are
// shared), so spectra stay bit-identical and every discrete decision agrees; the fp64 sums over bins are added in a
// different order (~1e-16).
//
// The split follows the transform's own structure (kiss-style decimation in time, three passes of radix 16 / 16 / 16, or
// 8 / 16 / 16 at 2048 points): up to the second pass the transform is 16 independent sub-transforms, so wave w owns
// sub-transforms [8w, 8w + 8) and exchanges first-pass -> second-pass data through a PRIVATE region of the buffer (no
// cross-wave synchronisation); only the second-pass -> last-pass exchange is all-to-all between the two waves.
//
// Cross-wave synchronisation is a two-flag handshake in LDS (pair_sync): gfx950 has one hardware barrier per workgroup and
// a workgroup here is several pairs sharing a twiddle table, so s_barrier cannot serve a pair.  Small results travel
// through a per-pair mailbox (two alternating halves, so that a slot is never rewritten before the partner has read it).

template <int N> struct PGeo {
    typedef Geo<N> G;
    typedef Plan<N> PL;
    static_assert(N == 2048 || N == 4096, "pair kernel: split window sizes");
    static constexpr int M = N / 2;
    static constexpr int P2 = N / 128;            // samples per lane
    static constexpr int U2 = M / 128;            // bins per lane
    static constexpr int R = N / 2048;            // second-pass / last-pass items per lane (= private rounds = exchange rounds)
    static constexpr int RA = G::RA;              // first-pass radix: 8 (2048), 16 (4096)
    static constexpr int FA = 1024 / RA / 64;     // first-pass items per lane per round: 2 (2048), 1 (4096)
    static constexpr int L1 = PL::L1, L2 = PL::L2;
    static constexpr int REGION = 1088;           // float2 slots of a wave's private region (1024 positions, cpad layout: 1087 used)
    // second exchange: 16 rows (element index of the last pass) x 128 columns (last-pass item within the round).  Readers
    // take a row with consecutive lanes (conflict-free at any pitch); writers cover, per 16 lanes, one row (L1 = 16) or two
    // rows (L1 = 8), which then need a pitch of 8 mod 16 slots.
    static constexpr int XPITCH = (N == 2048) ? 136 : 128;
    static constexpr int CSLOTS = 2 * REGION;     // 2176 float2 = 17408 B
    static_assert(8 * XPITCH <= REGION, "a wave's eight rows of the second exchange fit its region");
    // real image of the frame: 4 floats of padding per RQ samples.  2048: a lane's own run (16).  4096: 64 (two lanes' runs)
    // -- the 17408 bytes hold no more, and with a 32-sample run per lane 8l + (l >> 1) is still distinct mod 16 within
    // every 16-lane group of a 16-byte access.
    static constexpr int RQ = (N == 4096) ? 64 : 16;
    static constexpr int RIMG = N + 4 * (N / RQ);
    static constexpr int BQ = U2;                 // bins image: 4 floats of padding per lane's run of U2 bins
    static constexpr int BIMG = M + 4 * (M / BQ);
    static constexpr int BUF_BYTES = 8 * CSLOTS;
    static_assert(4 * RIMG <= BUF_BYTES && 4 * N <= BUF_BYTES && 4 * BIMG + 2 * M <= BUF_BYTES, "images fit the pair's buffer");
    static constexpr int MBOX_DOUBLES = 32;       // two halves of 16: [half][wave][8]
    static constexpr int PAIR_EXTRA = 8 * MBOX_DOUBLES + 16;       // mailbox + the two flags (+ padding to 16 bytes)
    static constexpr int PREV_FLOATS = M + 4;     // flux state of a channel, plain layout, + its hand-over counter
};
template <int N> __host__ __device__ constexpr int prim(int n) { return n + 4 * (int) ((unsigned) n / (unsigned) PGeo<N>::RQ); }
template <int N> __host__ __device__ constexpr int prim_step(int c) { return c + 4 * (c / PGeo<N>::RQ); }
template <int N> __host__ __device__ constexpr int pbim(int b) { return b + 4 * (int) ((unsigned) b / (unsigned) PGeo<N>::BQ); }
template <int N> __host__ __device__ constexpr int pbim_step(int c) { return c + 4 * (c / PGeo<N>::BQ); }

// waves per SIMD the register allocator must leave room for: 2048 points 16 waves per CU (8 frames), 4096 points 12 (6 frames)
template <int N> struct POcc {
    static constexpr int MAX_PAIRS = (N == 2048) ? 8 : 6;
    static constexpr int MAX_THREADS = 128 * MAX_PAIRS;
    static constexpr int WAVES_PER_SIMD = (N == 2048) ? 4 : 3;
};

// One wavefront's view of the frame its pair is analysing.
template <int N> struct PairWave {
    typedef PGeo<N> PG;
    typedef Plan<N> PL;
    static constexpr int M = PG::M, P2 = PG::P2, U2 = PG::U2, R = PG::R, RA = PG::RA, FA = PG::FA, L1 = PG::L1, L2 = PG::L2, HALF = N / 2;

    const FrameParams& p;
    const f2* tw;           // [N] pass-ordered twiddles (workgroup LDS)
    float* prev;            // [M] re of the channel's last accepted spectral frame (workgroup LDS, plain layout)
    int*   turn2;           // 2 * (index of the frame whose turn it is) + waves of that frame that are done with `prev`
    f2*    cbuf;            // the pair's buffer ...
    float* rbuf;            // ... viewed as floats (real image, bins image, lag array)
    double* mbox;           // the pair's mailbox: [2 halves][2 waves][8] doubles
    unsigned* flags;        // the pair's two arrival counters
    FramePart* fpl;
    double nyquist, rnyq, frpb;
    float  scale;
    int    c, T, t;
    int    w;               // which wave of the pair (wave-uniform)
    mutable unsigned gen;   // arrivals so far (wave-uniform; the same in both waves at every pair_sync)
    mutable unsigned xch;   // mailbox exchanges so far

    // ---- pair synchronisation ------------------------------------------------------------------------------------
    // Every LDS access this wave has issued is complete, then its arrival is published; returns when the partner's is
    // there.  LDS operations of one wavefront execute in order and the LDS is one memory for the CU, so what the partner
    // wrote before its arrival is visible to everything issued after the poll that saw it.
    // Split in two so that work which touches no shared LDS can sit between them: arrive() publishes this wave's arrival
    // (no wait for its own LDS traffic: the LDS executes a wavefront's instructions in issue order, so the flag lands after
    // every write and after every read issued before it has taken its data), wait() returns when the partner's is there.
    // Every arrive() is matched by one wait() before the next arrive().
    __device__ __forceinline__ void arrive(int lane) const
    {
        gen++;
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_store(flags + w, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void wait() const
    {
        // (a poll without s_sleep: the partner is at most one LDS round trip away; with a sleep, measured no different -- r03_pair.txt)
        while ((int) (__hip_atomic_load(flags + (w ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - gen) < 0) { }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ void pair_sync(int lane) const { arrive(lane); wait(); }
    // mailbox slot i of wave `who` in the half the current exchange uses; next_exchange() after the pair_sync that ends it
    __device__ __forceinline__ double* slot(int who, int i) const { return mbox + 16 * (int) (xch & 1u) + 8 * who + i; }
    __device__ __forceinline__ void next_exchange() const { xch++; }

    struct Sources { const void* a; const void* b; float gain_a, gain_b; int fmt_a, fmt_b; };
    __device__ __forceinline__ Sources sources() const
    {
        const size_t esz = (size_t) sample_bytes(p.sample_format);
        const unsigned char* in = static_cast<const unsigned char*>(p.in);
        Sources s;
        s.fmt_a = s.fmt_b = p.sample_format;
        if (p.hop_mode) {
            s.gain_a = s.gain_b = p.gain;
            s.b = in + ((size_t) c * T + t) * HALF * esz;
            if (t == 0) { s.a = p.tail_in + (size_t) c * HALF; s.fmt_a = FX_SAMPLE_F32; s.gain_a = 1.0f; }   // tail is fp32, already gained
            else        s.a = in + ((size_t) c * T + (t - 1)) * HALF * esz;
        } else {
            s.gain_a = s.gain_b = 1.0f;
            s.a = in + ((size_t) c * T + t) * N * esz;
            s.b = static_cast<const unsigned char*>(s.a) + HALF * esz;
        }
        return s;
    }

    // a1 (ref RealTimeAudioAnalysis.h:205-219): wave w brings half w of the window into the real image, 16 bytes per lane,
    // every load issued before the first is consumed; returns the wave's share of getRMSLevel's sum (float squares, double sum)
    template <int FMT>
    __device__ __forceinline__ double load_half_window_t(int lane, const void* src, float gain, float* tail_dst) const
    {
        constexpr int QH = HALF / 256;
        uint4 r[QH];
#pragma unroll
        for (int q = 0; q < QH; q++) r[q] = fetch_four<FMT>(src, 256 * q + 4 * lane);
        double ssq = 0.0;
        float* img = rbuf + prim<N>(HALF * w + 4 * lane);                  // 256 is a multiple of the padding quantum: immediates from here
#pragma unroll
        for (int q = 0; q < QH; q++) {
            const f4 v = widen_four<FMT>(r[q]) * gain;                     // ref AudioDataCollector.h:88 (x * 1.0f is exact)
            *reinterpret_cast<f4*>(img + prim_step<N>(256 * q)) = v;
            if (tail_dst) *reinterpret_cast<f4*>(tail_dst + 256 * q + 4 * lane) = v;
            ssq += (double) (v.x * v.x) + (double) (v.y * v.y) + (double) (v.z * v.z) + (double) (v.w * v.w);
        }
        return wave_sum(ssq);
    }
    __device__ __forceinline__ double load_half_window(int lane) const
    {
FX_MARK("p_load");
        const Sources sr = sources();
        const void* src = w ? sr.b : sr.a;
        const int fmt = w ? sr.fmt_b : sr.fmt_a;
        const float gain = w ? sr.gain_b : sr.gain_a;
        float* tail_dst = (w == 1 && t == T - 1) ? p.tail_out + (size_t) c * HALF : nullptr;
        if (fmt == FX_SAMPLE_F16) return load_half_window_t<FX_SAMPLE_F16>(lane, src, gain, tail_dst);
        if (fmt == FX_SAMPLE_S16) return load_half_window_t<FX_SAMPLE_S16>(lane, src, gain, tail_dst);
        if (fmt == FX_SAMPLE_S24) return load_half_window_t<FX_SAMPLE_S24>(lane, src, gain, tail_dst);
        return load_half_window_t<FX_SAMPLE_F32>(lane, src, gain, tail_dst);
    }

    // ---- first-pass operands -------------------------------------------------------------------------------------
    // The lane's first-pass items are it = 64 * d3 + lane with d3 = 2w + (round or item), and the sample that feeds input j
    // of item it is rev4(it) + 256 * r(j) = 4 * rev3(lane) + d3 + 256 * r(j): one per-lane part, the rest compile-time /
    // wave-uniform.
    static __device__ __forceinline__ int rj(int j) { return (RA == 8) ? ((j >> 1) + 4 * (j & 1)) : ((j >> 2) + 4 * (j & 3)); }
    __device__ __forceinline__ int low_of(int lane, int d) const { return 4 * rev4<3>(lane) + 2 * w + d; }

    // from the real image (the pitch path's two transforms read what the low-pass / the power spectrum left there)
    __device__ __forceinline__ void inputs_from_image(int lane, float (&x)[P2]) const
    {
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const float* at = rbuf + prim<N>(low_of(lane, d));
#pragma unroll
            for (int j = 0; j < RA; j++) x[d * RA + j] = at[prim_step<N>(256 * rj(j))];
        }
    }
    // the raw window from global memory (L2-hot after load_half_window), optionally through the Bartlett window (a3).  Every
    // load is a scalar base (the source of its half, advanced by this wave's 2w samples) + ONE per-lane 32-bit offset + an
    // immediate (as load_window_first_pass_order does it: a 64-bit element index would be a vector add per load).
    template <bool WINDOWED, int FMT_A, int FMT_B>
    __device__ __forceinline__ void inputs_from_global_t(int lane, float (&x)[P2], const Sources& s) const
    {
        const unsigned low4 = 4u * (unsigned) rev4<3>(lane);
        const unsigned off_a = low4 * (unsigned) sample_bytes(FMT_A), off_b = low4 * (unsigned) sample_bytes(FMT_B);
        const char* base_a = static_cast<const char*>(s.a) + 2 * w * sample_bytes(FMT_A);
        const char* base_b = static_cast<const char*>(s.b) + 2 * w * sample_bytes(FMT_B);
#pragma unroll
        for (int d = 0; d < 2; d++) {
#pragma unroll
            for (int j = 0; j < RA; j++) {
                const int r = rj(j);
                const bool second = r >= RA / 2;                               // low < 256 <= N/2
                const int k = d + 256 * (second ? r - RA / 2 : r);            // compile-time part of the sample index
                const char* at = (second ? base_b : base_a) + (second ? off_b : off_a) + k * sample_bytes(second ? FMT_B : FMT_A);
                x[d * RA + j] = second ? widen_one<FMT_B>(at) : widen_one<FMT_A>(at);
            }
        }
        if (s.gain_a != 1.0f || s.gain_b != 1.0f) {                           // ref AudioDataCollector.h:88 (wave-uniform)
#pragma unroll
            for (int d = 0; d < 2; d++)
#pragma unroll
                for (int j = 0; j < RA; j++) x[d * RA + j] *= (rj(j) >= RA / 2) ? s.gain_b : s.gain_a;
        }
        if (WINDOWED) {
            // exact dyadic arithmetic, identical to bartlett_gain<N>(index): rising half base + r * 2/RA, falling half
            // (1 - base) - (r - RA/2) * 2/RA, with base = (sample index mod 256) * 2/N
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const float base = (float) (low4 + (unsigned) (2 * w + d)) * (2.0f / N);
                const float nbase = 1.0f - base;
#pragma unroll
                for (int j = 0; j < RA; j++) {
                    const int r = rj(j);
                    const float gain = r < RA / 2 ? base + (float) r * (2.0f / RA) : nbase - (float) (r - RA / 2) * (2.0f / RA);
                    x[d * RA + j] *= gain;
                }
            }
        }
    }
    template <bool WINDOWED>
    __device__ __forceinline__ void inputs_from_global(int lane, float (&x)[P2]) const
    {
        asm volatile("" ::: "memory");
        const Sources s = sources();
        FX_FORMATS(s.fmt_a, s.fmt_b, (inputs_from_global_t<WINDOWED, FA, FB>(lane, x, s)));
    }

    // ---- the transform ---------------------------------------------------------------------------------------------
    // x: the lane's P2 real inputs (first-pass order, item d at x[d * RA ..]).  The pair's buffer must be free when this is
    // called (a pair_sync after whatever read it last).  OUT as in fft_from_regs; results:
    //   OUT_RE_LOW / _MAXABS : re of bins < N/2 in the bins image (after the closing pair_sync); returns max(|re|,|im|) over the
    //                          lane's share of bins [0, M/2) for _MAXABS
    //   OUT_POWER            : re^2 of every bin in the real image (the next transform's inputs)
    //   OUT_LAG              : v[s] = (re_s / N)^2 * s for every sample in the buffer, plain layout; returns v[N] in lane 0 of wave 0
    // `entry_pending`: the caller has issued an arrive() after its last read of the buffer and left the wait() to us -- it is
    // taken after the first-pass arithmetic, right before the first write.
    // `early_b` / `before_last`: the one-hop kernel's pitch pair runs its first transform before the workgroup's twiddle table is in
    // LDS -- the second pass's 15 twiddles come from registers it filled from global memory at kernel entry, and `before_last()`
    // (the wait for the table) is called where the last pass is about to read its own.
    struct NoWait { __device__ __forceinline__ void operator()() const {} };
    template <bool INV, int OUT, typename Hook = NoWait>
    __device__ __forceinline__ float transform(int lane, const float (&x)[P2], bool entry_pending = false,
                                               const f2 (*early_b)[15] = nullptr, Hook before_last = Hook{}) const
    {
FX_MARK("p_fft_ab");
        f2 ta[9];
#pragma unroll
        for (int i = 0; i < 9; i++) ta[i] = f2{p.first_tw[2 * i], p.first_tw[2 * i + 1]};
        f2* priv = cbuf + PG::REGION * w;
        f2 eb[R][16];
        // First two passes: R rounds through the wave's private region, 64 second-pass items per round.  The first-pass
        // arithmetic of round q + 1 is placed behind the reads of round q, whose latency it covers; the second-pass
        // arithmetic follows when all rounds are read.
        {
            f2 e[FA][RA];
#pragma unroll
            for (int f = 0; f < FA; f++) first_pass_item<N, INV>(&x[f * RA], ta, e[f]);
            if (entry_pending) wait();
#pragma unroll
            for (int q = 0; q < R; q++) {
#pragma unroll
                for (int f = 0; f < FA; f++) {
                    if constexpr (RealExchange<N>::USE) {
                        // (fx_fft.hip.h: only the elements of a real-input item that are not bitwise conjugates of stored ones)
                        f2* img = priv + RealExchange<N>::row(lane + 64 * f);
#pragma unroll
                        for (int sl = 0; sl < RealExchange<N>::SLOTS; sl++) img[sl] = e[f][RealExchange<N>::stored(sl)];
                    } else {
                        f2* img = priv + cpad((lane + 64 * f) * RA);
#pragma unroll
                        for (int i = 0; i < RA; i++) img[i] = e[f][i];
                    }
                }
                wave_fence();
                if constexpr (RealExchange<N>::USE) {
                    static_assert(!RealExchange<N>::USE || RA == L1, "the next pass's lanes read one element index each");
                    const int k = lane % L1;
                    const int slot = (int) ((RealExchange<N>::SLOT_OF >> (4 * k)) & 15ull);
                    const unsigned flip = ((RealExchange<N>::TWIN >> k) & 1u) << 31;
                    const f2* src = priv + RealExchange<N>::row((lane / L1) * 16) + slot;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const f2 v = src[i * RealExchange<N>::SLOTS];
                        eb[q][i] = f2{v.x, __uint_as_float(__float_as_uint(v.y) ^ flip)};
                    }
                } else {
                    const f2* img = priv + cpad((lane / L1) * (16 * L1) + lane % L1);
#pragma unroll
                    for (int i = 0; i < 16; i++) eb[q][i] = img[item_off(L1, i)];
                }
                if (q + 1 < R) {
#pragma unroll
                    for (int f = 0; f < FA; f++) first_pass_item<N, INV>(&x[((q + 1) * FA + f) * RA], ta, e[f]);
                }
                wave_fence();
            }
        }
        if (early_b) {
#pragma unroll
            for (int q = 0; q < R; q++) item16_stages_r<INV>(eb[q], *early_b);
        } else {
#pragma unroll
            for (int q = 0; q < R; q++) item16_stages<L1, INV>(eb[q], tw + PL::OFF1 + lane % L1);
        }
        before_last();
FX_MARK("p_fft_c");
        // Second exchange, all-to-all between the two waves: round h carries the elements that feed last-pass items
        // k = 128 h + (0..127).  Element i of second-pass item it sits in row it / L1, column it % L1 + L1 * i of the last
        // pass's input.  Rows 0-7 come from wave 0's items and rows 8-15 from wave 1's, and each wave keeps its rows in its
        // own region: a wave writes only memory that is its own until the partner reads it, so the first write needs no
        // synchronisation, and a round's last-pass arithmetic sits between "I have read" and "may I write again".
        float res[R][16];
        float aux = 0.0f;
#pragma unroll
        for (int h = 0; h < R; h++) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int itl = 64 * q + lane;                                 // second-pass item within this wave: its row is 8w + itl / L1
                f2* img = priv + PG::XPITCH * (itl / L1) + itl % L1;
#pragma unroll
                for (int i = 0; i < 16 / R; i++) img[L1 * i] = eb[q][(16 / R) * h + i];
            }
            pair_sync(lane);                   // both waves' rows of this round are in place
            f2 ec[16];
            {
                const f2* img = cbuf + 64 * w + lane;
#pragma unroll
                for (int i = 0; i < 16; i++) ec[i] = img[i < 8 ? PG::XPITCH * i : PG::REGION + PG::XPITCH * (i - 8)];
            }
            arrive(lane);                      // "I have read": the arithmetic below covers the partner's way here
            const int k = 128 * h + 64 * w + lane;
            const f2* t2 = tw + PL::OFF2 + k;
            item16_last<INV, OUT>(ec, [&](int i) { return t2[i * L2]; }, h == 0);
            last_item_reduce<N, OUT>(ec, k, scale, res[h], aux);
            if (OUT == OUT_LAG && h == 0) { const float d = ec[0].y * scale; aux = d * d * (float) N; }      // (meaningful in lane 0 of wave 0)
            wait();                            // the partner has read too: the rows may be rewritten
        }
FX_MARK("p_fft_out");
        // hand the results to the consumer
        float* fbuf = rbuf;
#pragma unroll
        for (int h = 0; h < R; h++) {
            const int k = 128 * h + 64 * w + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OUT == OUT_RE_LOW || OUT == OUT_RE_LOW_MAXABS) { if (i < 8) fbuf[pbim<N>(k) + pbim_step<N>(L2 * i)] = res[h][i]; }
                else if (OUT == OUT_POWER) fbuf[prim<N>(k) + prim_step<N>(L2 * i)] = res[h][i];
                else fbuf[k + L2 * i] = res[h][i];
            }
        }
        pair_sync(lane);
        return aux;
    }

    // ---- pitch (ref PitchAnalyser.h:24-217, RealTimeAnalyser.h:152-166) -------------------------------------------------
    // `before_transforms`: called by both waves once the filtered window is written, before the first twiddle is read (the one-hop
    // kernel joins its workgroup barrier there: the other pairs load the twiddle table while this pair loads and filters)
    // `early_b` (optional): the second pass's twiddles of this lane in registers; the first transform then starts without the
    // table and `before_transforms` is called before its last pass instead.
    template <typename Hook>
    __device__ __forceinline__ float pitch(int lane, Hook before_transforms, const f2 (*early_b)[15] = nullptr) const
    {
        const int gl = 64 * w + lane;
        // a10 AudioFilter::filterAudio (ref RealTimeAudioAnalysis.h:106-125): y[0] = x[0]; y[n] = (a*x[n]) + (b*y[n-1]) in fp32,
        // strictly serial.  Lane gl owns samples [P2*gl, P2*gl + P2), starts 16 samples early from a guess, and the value it
        // reaches at P2*gl - 1 must be bit-identical to what its left neighbour produced there; lanes that disagree redo
        // their chunk from the neighbour's value (exact by induction from lane 0 of wave 0).  Wave 1's first lane takes its
        // neighbour's value -- wave 0's last -- from the mailbox.
FX_MARK("p_lpf");
        constexpr int KW = 16;
        const float a = p.lpf_a, b = p.lpf_b;
        float x[P2];
#pragma unroll
        for (int i = 0; i < P2; i += 4) {
            const f4 v = *reinterpret_cast<const f4*>(&rbuf[prim<N>(P2 * gl + i)]);
            x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
        }
        float yin = 0.0f;
        {
            const int first = P2 * gl;
#pragma unroll
            for (int q = 0; q < KW / 4; q++) {
                const int n0 = first - KW + 4 * q;
                if (n0 >= 0) {
                    const f4 v = *reinterpret_cast<const f4*>(&rbuf[prim<N>(n0)]);
                    const float wv[4] = {v.x, v.y, v.z, v.w};
                    float aw[4];
                    FrameWave<N>::template scaled_pairs<4>(a, wv, aw);        // (see FrameWave::lowpass_window)
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        yin = (n0 + e == 0 || (q == 0 && e == 0)) ? wv[e] : aw[e] + (b * yin);
                }
            }
        }
        float y[P2];
        float ylast;
        {
            float yy = yin;
#pragma unroll
            for (int i = 0; i < P2; i += 4) {
                float ax[4];
                FrameWave<N>::template scaled_pairs<4>(a, &x[i], ax);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    yy = (gl == 0 && i + e == 0) ? x[0] : ax[e] + (b * yy);
                    y[i + e] = yy;
                }
            }
            ylast = yy;
        }
        auto converge = [&](float from_left) {
            for (int iter = 0; iter < 130; iter++) {
                const float pe = shift_up1(ylast, from_left);
                const unsigned long long bad_lanes = wave_ballot(__float_as_uint(pe) != __float_as_uint(yin)) & (w == 0 ? ~1ull : ~0ull);   // gl > 0
                if (!bad_lanes) break;
                if ((bad_lanes >> lane) & 1) {
                    yin = pe;
                    float yy = yin;
#pragma unroll
                    for (int i = 0; i < P2; i++) { yy = (a * x[i]) + (b * yy); y[i] = yy; }
                    ylast = yy;
                }
            }
        };
        if (w == 0) {
            converge(0.0f);
            if (lane == 63) *reinterpret_cast<float*>(slot(0, 0)) = ylast;
        }
        pair_sync(lane);              // every read of the raw image is done (both waves); wave 0's last value is in the mailbox
        if (w == 1) converge(*reinterpret_cast<const float*>(slot(0, 0)));
        next_exchange();
        // window the filtered frame (ref RealTimeAnalyser.h:157) and put it back in the real image: a lane's samples lie in
        // one half of the window, the gains w0 + i * wstep are exact dyadic numbers and equal bartlett_gain<N>(P2*gl + i)
        {
            const float w0 = bartlett_gain<N>(P2 * gl);
            const float wstep = w == 0 ? (2.0f / N) : -(2.0f / N);
#pragma unroll
            for (int i = 0; i < P2; i += 4) {
                f4 v;
                v.x = y[i]     * __builtin_fmaf(wstep, (float) i, w0);
                v.y = y[i + 1] * __builtin_fmaf(wstep, (float) (i + 1), w0);
                v.z = y[i + 2] * __builtin_fmaf(wstep, (float) (i + 2), w0);
                v.w = y[i + 3] * __builtin_fmaf(wstep, (float) (i + 3), w0);
                *reinterpret_cast<f4*>(&rbuf[prim<N>(P2 * gl + i)]) = v;
            }
        }
        pair_sync(lane);
        if (!early_b) before_transforms();
FX_MARK("p_pitch_in");
        float xf[P2];
        inputs_from_image(lane, xf);
        arrive(lane);                 // the image has been read (the transform waits for the partner's before it writes)
        if (early_b) transform<false, OUT_POWER>(lane, xf, true, early_b, before_transforms);
        else         transform<false, OUT_POWER>(lane, xf, true);              // ref RealTimeAnalyser.h:160; a11: re * re, imag := 0
        inputs_from_image(lane, xf);                                           // already squared
        arrive(lane);
        const float v_end = transform<true, OUT_LAG>(lane, xf, true);          // a12 inverse, ref PitchAnalyser.h:110-121
        // a13 / a14 on wave 0: the running sum is serial, 64 samples at a time, and the search usually ends in the first block or two
FX_MARK("p_lag");
        float lag = 0.0f;
        if (w == 0) {
            LagSearch<N> ls;
            ls.begin();
            for (int blk = 0; blk < N / 64 && !ls.done; blk++) ls.block(lane, blk, rbuf[64 * blk + lane]);
            lag = ls.finish(lane, v_end);
            if (lane == 0) { *reinterpret_cast<float*>(slot(0, 0)) = lag; fpl->lag = lag; }
        }
        pair_sync(lane);              // (also: wave 0 is done with the lag array)
        lag = *reinterpret_cast<const float*>(slot(0, 0));
        next_exchange();
        return lag;
    }

    __device__ __forceinline__ float pitch(int lane) const { return pitch(lane, [] {}); }

    // ---- the flatness gate (see FrameWave::gate_threshold): each wave brackets for its own bins ---------------------------
    __device__ __forceinline__ float gate_threshold(double sum_sq, const float (&re)[U2]) const
    {
        const float rel = 1e-4f, abs_ = 4e-6f;
        const float rms_a = __builtin_amdgcn_sqrtf((float) (sum_sq * (1.0 / (double) N)));
        const float log_a = __builtin_amdgcn_logf(rms_a * 9.0f + 1.0f) * 0.30103f;
        const float eps_lo = 0.01f * (log_a * (1.0f - rel) - abs_);
        const float eps_hi = 0.01f * (log_a * (1.0f + rel) + abs_);
        const float t_lo = __builtin_amdgcn_sqrtf(fmaxf(eps_lo, 0.0f)) * (1.0f - 1e-6f);
        const float t_hi = __builtin_amdgcn_sqrtf(eps_hi) * (1.0f + 1e-6f);
        bool inside = false;
#pragma unroll
        for (int j = 0; j < U2; j++) inside |= (fabsf(re[j]) > t_lo) && !(fabsf(re[j]) > t_hi);
        if (!wave_any(inside)) return t_hi;          // gates this wave's bins exactly as the true threshold would
        const double eps = 0.01 * (double) FrameWave<N>::exact_log_rms(sum_sq);
        float tt = (float) sqrt(eps);
        const float up = __uint_as_float(__float_as_uint(tt) + 1u);
        if ((double) tt * (double) tt > eps) tt = __uint_as_float(__float_as_uint(tt) - 1u);
        else if ((double) up * (double) up <= eps) tt = up;
        return tt;
    }

    // ---- spectral analyser (ref RealTimeAnalyser.h:212-224, SpectralCharacteristics.h:62-200) ------------------------------
    __device__ __forceinline__ void spectral(int lane, double sum_sq) const
    {
        const int gl = 64 * w + lane;
        float spec_aux;
        {
FX_MARK("p_spec_in");
            float xw[P2];
            inputs_from_global<true>(lane, xw);                                // a3 Bartlett window
            spec_aux = transform<false, OUT_RE_LOW_MAXABS>(lane, xw);          // a4
        }
FX_MARK("p_spec_sums");
        float re[U2];
        lds_load_block<U2>(rbuf + pbim<N>(U2 * gl), re);
        const float tg = gate_threshold(sum_sq, re);
        // fillIntermediateValues :62-97 over the lane's bins m = U2*gl + j, as moments of the magnitudes (see FrameWave::spectral)
        double Ts = 0.0, Vs = 0.0, Ws = 0.0, t_after = 0.0, flat_sum = 0.0;
        float max_re = 0.0f;
        int cnt = 0;
        constexpr int LQ = (M / 5) / U2, LR = (M / 5) % U2;
#pragma unroll
        for (int j = U2 - 1; j >= 0; j--) {
            const double v = (double) re[j];
            const double mag = v * v;
            Ts += mag;
            if (j >= 1) { Vs += Ts; Ws += Vs; }
            if (j == LR + 1) t_after = Ts;
            const bool gate = fabsf(re[j]) > tg;                               // :89
            cnt += __builtin_popcountll(wave_ballot(gate));
            if (gate) flat_sum += mag;
            max_re = fmaxf(max_re, fabsf(re[j]));
        }
        const double ul = (double) (U2 * gl);
        double mag_sum = Ts;
        double b1 = ul * Ts + Vs;
        double b2 = (ul * ul) * Ts + ((ul + ul) * Vs + ((Ws + Ws) - Vs));
        double lhr = gl < LQ ? Ts : (gl == LQ ? Ts - t_after : 0.0);
        wave_sum4(lane, mag_sum, b1, b2, lhr);
        flat_sum = wave_sum(flat_sum);
        max_re = wave_maxf(max_re);
        float maxabs = wave_maxf(spec_aux);
        // exchange 1: the wave totals; both waves add them in the same order (wave 0 + wave 1) and hold the same sums
        if (lane == 0) {
            double* s = slot(w, 0);
            s[0] = mag_sum; s[1] = b1; s[2] = b2; s[3] = lhr; s[4] = flat_sum; s[5] = (double) max_re; s[6] = (double) maxabs; s[7] = (double) cnt;
        }
        pair_sync(lane);
        {
            const double* s0 = slot(0, 0); const double* s1 = slot(1, 0);
            mag_sum = s0[0] + s1[0]; b1 = s0[1] + s1[1]; b2 = s0[2] + s1[2]; lhr = s0[3] + s1[3]; flat_sum = s0[4] + s1[4];
            max_re = fmaxf((float) s0[5], (float) s1[5]); maxabs = fmaxf((float) s0[6], (float) s1[6]); cnt = (int) s0[7] + (int) s1[7];
        }
        next_exchange();
        const double wsum = frpb * (b1 + 0.5 * mag_sum);                       // :95
        const double max_mag = (double) max_re * (double) max_re;
        const bool accepted = mag_sum > 0.05;                                  // :121-123

        // ---- flux against the previous accepted frame; the channel's frames take turns, both waves of a frame within its turn ----
FX_MARK("p_flux");
        double flux = 0.0;
        {
            while (__hip_atomic_load(turn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2 * t) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            float pvf[U2];
            lds_load_block<U2>(prev + U2 * gl, pvf);
            if (accepted) lds_store_block<U2>(prev + U2 * gl, re);             // :138 (only on the accepted path)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(turn2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (LDS operations of a wave execute in order)
#pragma unroll
            for (int j = 0; j < U2; j++) {
                const double pv = (double) pvf[j];
                const double v = (double) re[j];
                flux += fmax(v * v - pv * pv, 0.0);                            // :76-79
            }
        }
        flux = wave_sum(flux);

        // ---- flatness product, serial-order semantics (see FrameWave::flatness_product): lane totals scanned per wave ----
FX_MARK("p_flat");
        FlatProd loc = {0.5, 1};
        int emin = 1, emax = 1;
#pragma unroll
        for (int j = 0; j < U2; j++) {
            const double v = (double) re[j];
            if (fabsf(re[j]) > tg) {
                loc = fp_mul(loc, v * v);
                emin = loc.exp < emin ? loc.exp : emin;
                emax = loc.exp > emax ? loc.exp : emax;
            }
        }
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
        // the factors of this lane, for the serial continuation (1.0 = not a factor)
        auto serial_from = [&](double pr, int first_lane) {
            for (int l = first_lane; l < 64; l++) {
                double mine = pr;
#pragma unroll
                for (int j = 0; j < U2; j++) { const double v = (double) re[j]; mine *= fabsf(re[j]) > tg ? v * v : 1.0; }
                pr = lane_get(mine, l);
                if (pr == 0.0 || pr == __builtin_huge_val()) break;
            }
            return pr;
        };

        // second pass over the lane's bins: sum (mag - mean)^2 for the slope (:182-188) and, if the moment form of the
        // spread is not trustworthy (see FrameWave::spectral), the reference's own sum
FX_MARK("p_pass2");
        const float centroid = (float) (wsum / mag_sum);                       // :127
        const double cn = (double) centroid * rnyq;
        const double rm = 1.0 / (double) M;
        double var = ((b2 + b1 + 0.25 * mag_sum) * rm - (cn + cn) * (b1 + 0.5 * mag_sum)) * rm + (cn * cn) * mag_sum;
        const double mu = mag_sum * (1.0 / (double) M);
        double vsum = 0.0, direct = 0.0;
#pragma unroll
        for (int j = 0; j < U2; j++) {
            const double v = (double) re[j];
            const double dv = v * v - mu;
            vsum += dv * dv;
        }
        const bool refine = !(var > 1e-9 * (cn * cn) * mag_sum) || !(var < __builtin_huge_val());
        if (refine) {
#pragma unroll
            for (int j = 0; j < U2; j++) {
                const double v = (double) re[j];
                const double d = ((double) (U2 * gl + j) * frpb + (frpb / 2.0)) * rnyq - cn;
                direct += (d * d) * (v * v);
            }
        }
        wave_sum2(lane, direct, vsum);

FX_MARK("p_spec_x2");
        // exchange 2: wave 0 -> wave 1, which finishes flux, product, vsum (and the direct spread) and records them
        if (w == 0) {
            const unsigned long long risky_lanes = wave_ballot(exc.exp + emax >= 1025) | wave_ballot(exc.exp + emin - 1 <= -1022);
            double pr_end = 0.0;
            if (risky_lanes) {
                const int owner = (int) __builtin_ctzll(risky_lanes);
                pr_end = serial_from(ldexp(lane_get(exc.mant, owner), lane_get(exc.exp, owner)), owner);
            }
            if (lane == 0) {
                double* s = slot(0, 0);
                s[0] = bcast63(inc.mant); s[1] = (double) __builtin_amdgcn_readlane(inc.exp, 63); s[2] = risky_lanes ? 1.0 : 0.0; s[3] = pr_end;
                s[4] = flux; s[5] = vsum; s[6] = direct;
                const double max_e = max_mag > (double) maxabs ? max_mag : (double) maxabs;        // :153, :161-162
                fpl->mag_sum = mag_sum; fpl->lhr = lhr; fpl->flat_sum = flat_sum; fpl->max_e = max_e; fpl->b1 = b1; fpl->b2 = b2;
                fpl->cnt = (float) cnt;
            }
        }
        pair_sync(lane);
        if (w == 1) {
            const double* s = slot(0, 0);
            const FlatProd tot0 = {s[0], (int) s[1]};
            const bool risky0 = s[2] != 0.0;
            const double pr0_end = s[3];
            double prod;
            if (risky0) {
                // the serial product is already in plain IEEE double when it reaches this wave: carry on, unless it is
                // absorbed (0 or inf) already
                prod = (pr0_end == 0.0 || pr0_end == __builtin_huge_val()) ? pr0_end : serial_from(pr0_end, 0);
            } else {
                const FlatProd ex = fp_mul2(tot0, exc);                        // prefix before this lane, wave 0's bins included
                const unsigned long long risky_lanes = wave_ballot(ex.exp + emax >= 1025) | wave_ballot(ex.exp + emin - 1 <= -1022);
                if (risky_lanes) {
                    const int owner = (int) __builtin_ctzll(risky_lanes);
                    prod = serial_from(ldexp(lane_get(ex.mant, owner), lane_get(ex.exp, owner)), owner);
                } else {
                    const FlatProd all = fp_mul2(tot0, FlatProd{bcast63(inc.mant), __builtin_amdgcn_readlane(inc.exp, 63)});
                    prod = ldexp(all.mant, all.exp);
                }
            }
            if (lane == 0) {
                fpl->flux = s[4] + flux; fpl->prod = prod; fpl->vsum = s[5] + vsum;
                fpl->var = refine ? s[6] + direct : var;        // (this kernel forms the spread's sum itself, from its own centroid)
                fpl->refined = 1;
            }
        }
        next_exchange();
    }

    // ---- harmonic analyser (ref RealTimeAnalyser.h:161,169-172; HarmonicCharacteristics.h:46-244) ---------------------------
    // What the tail needs of the raw spectrum: |re| of the lane's own bins and of the three bins around them, the sums
    struct HarmonicSpectrum { float hre[U2]; float left2, left1, right1; double sum; float max_re; };

    // part 1: the raw (un-windowed) frame's spectrum and its sums (:161, HarmonicCharacteristics.h:61-69) -- needs no pitch yet
    __device__ __forceinline__ void harmonic_spectrum(int lane, HarmonicSpectrum& hs) const
    {
        const int gl = 64 * w + lane;
        {
FX_MARK("p_harm_in");
            float xr[P2];
            inputs_from_global<false>(lane, xr);
            transform<false, OUT_RE_LOW>(lane, xr);                            // the raw (un-windowed) frame's spectrum, :161
        }
FX_MARK("p_harm_sums");
        const float* relin = rbuf;                                             // bins image: re of every bin < M
        const int b0 = U2 * gl;
        lds_load_block<U2>(relin + pbim<N>(b0), hs.hre);
        hs.left2  = b0 >= 2 ? fabsf(relin[pbim<N>(b0 - 2)]) : 0.0f;
        hs.left1  = b0 >= 1 ? fabsf(relin[pbim<N>(b0 - 1)]) : 0.0f;
        hs.right1 = b0 + U2 < M ? fabsf(relin[pbim<N>(b0 + U2 < M ? b0 + U2 : 0)]) : 0.0f;
        double h_sum = 0.0;
        float h_max_re = 0.0f;
#pragma unroll
        for (int j = 0; j < U2; j++) {                                         // ref HarmonicCharacteristics.h:61-69
            const double v = (double) hs.hre[j];
            h_sum += v * v;
            h_max_re = fmaxf(h_max_re, fabsf(hs.hre[j]));
        }
        h_sum = wave_sum(h_sum);
        h_max_re = wave_maxf(h_max_re);
        if (lane == 0) { double* s = slot(w, 0); s[0] = h_sum; s[1] = (double) h_max_re; }
        pair_sync(lane);
        {
            const double* s0 = slot(0, 0); const double* s1 = slot(1, 0);
            hs.sum = s0[0] + s1[0];
            hs.max_re = fmaxf((float) s0[1], (float) s1[1]);
        }
        next_exchange();
    }

    // part 2: peaks, harmonic energy ratio, inharmonicity (HarmonicCharacteristics.h:71-105), once the pitch is known
    __device__ __forceinline__ void harmonic_tail(int lane, HarmonicSpectrum& hs, double f0) const
    {
        const int gl = 64 * w + lane;
        const float* relin = rbuf;
        float (&hre)[U2] = hs.hre;
        const float h_left2 = hs.left2, h_left1 = hs.left1, h_right1 = hs.right1;
        double h_sum = hs.sum;
        const double h_max = (double) hs.max_re * (double) hs.max_re;
        if (h_sum < 0.005) return;                                             // :88-89 (the same decision in both waves)

        double mean_mag = h_sum / (double) M;                                  // :86
        // binIsPeak's `mag > mean` (:132) is an exact tie for a flat spectrum: then the last bit of the reference's SERIAL
        // sum decides for every bin at once.  If any bin sits within rounding distance of the mean, redo the sum in the
        // reference's order, lane to lane, wave 0 then wave 1 (rare; see FrameWave::harmonic_tail).
        {
            const float root_mean = __builtin_amdgcn_sqrtf((float) mean_mag);
            const float band = root_mean * 3e-6f;
            unsigned long long near = 0;                                       // (lane masks: a bool would be packed into bytes)
#pragma unroll
            for (int j = 0; j < U2; j++) near |= wave_ballot(fabsf(fabsf(hre[j]) - root_mean) <= band);
            const bool near_wave = near != 0;
            if (lane == 0) *slot(w, 0) = near_wave ? 1.0 : 0.0;
            pair_sync(lane);
            const bool near_any = *slot(0, 0) != 0.0 || *slot(1, 0) != 0.0;
            next_exchange();
            if (near_any) {
                auto serial_sum = [&](double run) {
                    for (int l = 0; l < 64; l++) {
                        double mine = run;
#pragma unroll
                        for (int j = 0; j < U2; j++) { const double v = (double) hre[j]; mine += v * v; }
                        run = lane_get(mine, l);
                    }
                    return run;
                };
                if (w == 0) { const double r0 = serial_sum(0.0); if (lane == 0) *slot(0, 0) = r0; }
                pair_sync(lane);
                if (w == 1) { const double r1 = serial_sum(*slot(0, 0)); if (lane == 0) *slot(1, 0) = r1; }
                pair_sync(lane);
                h_sum = *slot(1, 0);
                next_exchange();
                mean_mag = h_sum / (double) M;
            }
        }
        // peaks (binIsPeak :127-145): above the mean and none of bins -2, -1, +1 larger; each wave lists the peaks among its own
        // bins (at most M/2 of them), behind the bins image
FX_MARK("p_harm_tail");
        unsigned short* peaks = reinterpret_cast<unsigned short*>(rbuf + PG::BIMG) + (M / 2) * w;
        bool pk[U2];
        int npk_lane = 0;
#pragma unroll
        for (int j = 0; j < U2; j++) {
            const double v = (double) hre[j];
            const double mag = v * v;
            const float me = fabsf(hre[j]);
            const float l2 = j >= 2 ? fabsf(hre[j >= 2 ? j - 2 : 0]) : (j == 1 ? h_left1 : h_left2);
            const float l1 = j >= 1 ? fabsf(hre[j >= 1 ? j - 1 : 0]) : h_left1;
            const float r1 = j + 1 < U2 ? fabsf(hre[j + 1 < U2 ? j + 1 : 0]) : h_right1;
            bool is_peak = (mag > mean_mag) && !(l2 > me) && !(l1 > me);
            // (neighbours that do not exist were loaded as 0; the one clipped neighbour that does exist is bin M-1 seen from
            // bin M-2, :136-138)
            if (j == U2 - 2) is_peak = is_peak && (!(r1 > me) || gl == 127);
            else             is_peak = is_peak && !(r1 > me);
            pk[j] = is_peak;
            npk_lane += is_peak ? 1 : 0;
        }
        const int pre = wave_scan_incl_i(npk_lane);
        const int wave_peaks = __builtin_amdgcn_readlane(pre, 63);
        {
            unsigned short* wp = peaks + (pre - npk_lane);
#pragma unroll
            for (int j = 0; j < U2; j++)
                if (pk[j]) { *wp = (unsigned short) (U2 * gl + j); wp++; }
        }
        wave_fence();

        const double r_hmax = 1.0 / h_max;
        const double fr = nyquist / (double) M;                                // :93
        // calculateHarmonicEnergyCharacteristics :147-198 (numLower = 15, numHarmonics = 3): 18 probes, one lane each, on
        // wave 0; every wave needs the bin of f0 itself (lane 18)
        double probe = 0.0;
        int f0_bin;
        {
            const double freq = lane < 15 ? ldexp(f0, -(lane + 1)) : (lane < 18 ? f0 * (double) (lane - 14) : f0);
            int bin = (int) floor(freq / fr);
            f0_bin = __builtin_amdgcn_readlane(bin, 18);
            if (lane < 15) { if (bin == f0_bin) bin = -1; }                    // :163-164
            else if (lane < 18) { if (bin >= M) bin = -1; }                    // :174-175
            else bin = -1;
            if (w == 0 && bin >= 0 && bin < M) {
                // getMaxBinInNeighbourhood :200-210 in the reference's order: c-2, c-1, (c), c+1
                float mx = fabsf(relin[pbim<N>(bin)]);
                const float a2 = bin >= 2 ? fabsf(relin[pbim<N>(bin >= 2 ? bin - 2 : 0)]) : mx;
                const float a1 = bin >= 1 ? fabsf(relin[pbim<N>(bin >= 1 ? bin - 1 : 0)]) : mx;
                const float b1 = bin + 1 < M ? fabsf(relin[pbim<N>(bin + 1 < M ? bin + 1 : 0)]) : mx;
                float run = mx;
                if (a2 > run) run = a2;
                if (a1 > run) run = a1;
                if (b1 > run) run = b1;
                const double pm = (double) run;
                probe = (double) (float) ((pm * pm) * r_hmax);                 // (float) (mag / max), :75
            }
        }
        // calculateInharmonicity :212-244 over this wave's own peaks
        double inh = 0.0;
        const double r_hsum = 1.0 / h_sum;
        if (f0 > 0.0) {                                                        // :98
            for (int i = lane; i < wave_peaks; i += 64) {
                const int bin = (int) peaks[i];
                if (bin == f0_bin) continue;                                   // :220-221
                double fs = (double) bin * fr;
                if (fs == 0.0) fs = fr * 0.5;                                  // :225-226
                const double fe = (double) (bin + 1) * fr;
                const double rs  = (fs > f0 ? fs : f0) / (fs > f0 ? f0 : fs);  // getFrequencyRatio :251-259
                const double re_ = (fe > f0 ? fe : f0) / (fe > f0 ? f0 : fe);
                if (floor(rs) != floor(re_)) continue;                         // :232-233
                const double r = rs < re_ ? rs : re_;
                const double v = (double) relin[pbim<N>(bin)];
                inh += (r - floor(r)) * ((v * v) * r_hsum);                    // :236-239
            }
        }
        double score = probe;
        wave_sum2(lane, score, inh);
        if (w == 1 && lane == 0) *slot(1, 0) = inh;
        pair_sync(lane);              // (also: both waves are done with the bins image and the peak lists)
        if (w == 0 && lane == 0) {
            fpl->inh = inh + *slot(1, 0); fpl->her_score = score; fpl->sum_normed = h_sum * r_hmax; fpl->flags = 1;
        }
        next_exchange();
    }
};

template <int N>
__global__ void __launch_bounds__(POcc<N>::MAX_THREADS, POcc<N>::WAVES_PER_SIMD)
fx_pair_kernel(const FrameParams p_arg)
{
    FrameParams p = p_arg;
    if (p.dyn) { p.gain = p.dyn->gain; p.nyquist = p.dyn->nyquist; }
    typedef PGeo<N> PG;
    constexpr int M = PG::M;

    // One workgroup = CH channels x K pairs (K frames of a channel in flight), sharing one twiddle table.  LDS:
    //   twiddles [N] | CH x { flux state of the channel + its hand-over counter } | per pair { buffer, mailbox, flags }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int CH = p.ch_per_wg, K = p.waves_per_ch;
    f2*    tw = reinterpret_cast<f2*>(smem);
    float* prev0 = reinterpret_cast<float*>(tw + N);
    unsigned char* per_pair = reinterpret_cast<unsigned char*>(prev0 + (size_t) CH * PG::PREV_FLOATS);
    constexpr size_t PAIR_BYTES = PG::BUF_BYTES + PG::PAIR_EXTRA;

    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));     // wave-uniform: scalar registers for all that follows from it
    const int lane = threadIdx.x & 63;
    const int pair = wave >> 1, w = wave & 1;
    const int chl = pair / K, slot = pair % K;
    const int T = p.T;
    int group = (int) blockIdx.x, chunk = 0;
    if (p.num_chunks > 1) {
        unsigned* ticket_s = reinterpret_cast<unsigned*>(per_pair);            // (pair 0's buffer is not in use yet)
        if (threadIdx.x == 0) *ticket_s = __hip_atomic_fetch_add(p.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned ticket = *ticket_s, groups = gridDim.x / (unsigned) p.num_chunks;
        __syncthreads();
        chunk = (int) (ticket / groups);
        group = (int) (ticket % groups);
    }
    const int c = group * CH + chl;
    const bool live = c < p.C;
    const int t_begin = p.num_chunks > 1 ? p_arg.chunk_begin[chunk] : 0;
    const int t_end = p.num_chunks > 1 ? p_arg.chunk_begin[chunk + 1] : T;

    float* prev = prev0 + (size_t) chl * PG::PREV_FLOATS;
    int*   turn2 = reinterpret_cast<int*>(prev + M);
    unsigned char* mine = per_pair + PAIR_BYTES * pair;
    f2*    cbuf = reinterpret_cast<f2*>(mine);
    double* mbox = reinterpret_cast<double*>(mine + PG::BUF_BYTES);
    unsigned* flags = reinterpret_cast<unsigned*>(mine + PG::BUF_BYTES + 8 * PG::MBOX_DOUBLES);

    for (int i = threadIdx.x; i < N; i += blockDim.x) tw[i] = reinterpret_cast<const f2*>(p.tw)[i];
    if (chunk > 0) {
        // the flux state comes from the chunk before (another workgroup): see fx_frame_kernel
        if (live && slot == 0 && w == 0 && lane == 0) {
            unsigned spins = 0;
            bool there;
            while (!(there = __hip_atomic_load(p.queue + 1 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned) chunk) && ++spins < p.spin_limit)
                __builtin_amdgcn_s_sleep(8);
            if (!there) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (live) {
        for (int i = lane + 64 * (2 * slot + w); i < M; i += 128 * K) prev[i] = p.prev_re[(size_t) c * M + i];
        if (lane == 0 && slot == 0 && w == 0) turn2[0] = 2 * t_begin;
    }
    if (lane == 0) flags[w] = 0u;
    __syncthreads();

    const double nyquist = p.nyquist;
    PairWave<N> pw{p, tw, prev, turn2, cbuf, reinterpret_cast<float*>(cbuf), mbox, flags, nullptr,
                   nyquist, 1.0 / nyquist, nyquist / (double) M, 1.0f / (float) N, c, T, 0, w, 0u, 0u};
    for (int t = live ? t_begin + slot : t_end; t < t_end; t += K) {
        const int ln = opaque<N>(lane);
        pw.t = t;
        pw.fpl = p.part + ((size_t) c * T + t);
        if (w == 0 && ln == 0) pw.fpl->flags = 0;
        // a1 + a2: the window into the real image, the sum of its squares
        const double ssq_wave = pw.load_half_window(ln);
        if (ln == 0) *pw.slot(w, 0) = ssq_wave;
        pw.pair_sync(ln);
        const double sum_sq = *pw.slot(0, 0) + *pw.slot(1, 0);
        pw.next_exchange();
        if (w == 0 && ln == 0) pw.fpl->sum_sq = sum_sq;
        // the harmonic analyser's pitch estimate first: its low-pass reads the raw frame's image
        const float lag = pw.pitch(ln);
        const double f0 = (nyquist * 2.0) / (double) lag;                      // ref PitchAnalyser.h:57
        pw.spectral(opaque<N>(ln), sum_sq);
        { typename PairWave<N>::HarmonicSpectrum hs; pw.harmonic_spectrum(opaque<N>(ln), hs); pw.harmonic_tail(opaque<N>(ln), hs, f0); }
    }

    __syncthreads();
    if (live)
        for (int i = lane + 64 * (2 * slot + w); i < M; i += 128 * K) p.prev_re[(size_t) c * M + i] = prev[i];
    if (chunk + 1 < p.num_chunks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (live && slot == 0 && w == 0 && lane == 0 && !(p.debug_flags & 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.queue + 1 + c, (unsigned) (chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
