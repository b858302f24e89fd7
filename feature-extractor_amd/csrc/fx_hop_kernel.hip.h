// fx_hop_kernel.hip.h -- fx_hop_kernel: ONE hop per channel per call, end to end in one launch (the streaming ring at
// one hop per call, BASELINE configs[4]).  Included by fx_kernels.hip inside namespace fxk; not a stand-alone header.
//
// The batch path (fx_frame_kernel + tail kernels) is built for throughput: a frame lives in one wavefront and a step is
// two to four launches.  For a single hop that is all latency: ~9.4 k dependent-issue instructions on one SIMD at 4096
// points, a kernel boundary, a graph replay and an event wait.  Here a workgroup is one channel and THREE wavefronts, one
// per independent part of the reference's two run() loops (ref RealTimeAnalyser.h:141-177, :201-234):
//   wave 0  pitch estimate: low-pass, window, transform, re^2, inverse transform, lag search      (a10-a14)
//   wave 1  spectral analyser: RMS, Bartlett window, transform, spectral sums, flux state         (a2-a7)
//   wave 2  harmonic analyser: transform of the raw frame, its sums; then -- once wave 0 has the
//           lag -- peaks, harmonic energy ratio, inharmonicity                                     (a15-a18)
// each with its own transform buffer, running the very sections of FrameWave the batch kernel runs (same bits).  The
// frame's record stays in LDS; wave 0 then finishes the hop the way fx_tail_fused_kernel does (scalar tail, smoothing,
// onset, history) and the last workgroup to arrive stores the call's sequence number where the host is polling -- in
// pinned host memory, next to the results, which the kernel also writes there directly.  No second launch, no graph,
// no event: the critical path is the pitch estimate plus the harmonic tail.

// The flux state is read and replaced where it lives, in global memory, by the one wavefront that uses it (FrameWave's DIRECT
// form, as in one-frame calls of the frame kernel), and 4096 points keeps the frame kernel's 24 KB twiddle image (WIDE / CompactTw)
// instead of the 32 KB table: a 4096-point workgroup is 80 KB, so a CU holds TWO -- 512 channels per round of the chip instead of 256.
template <int N> struct HopGeo {
    typedef Geo<N> G;
    static constexpr bool   WIDE = FrameLds<N>::WIDE;
    static constexpr size_t OFF_WAVES = sizeof(f2) * FrameLds<N>::TW_ENTRIES;
    static constexpr size_t OFF_PART = OFF_WAVES + 3 * (size_t) G::BUF_BYTES;
    static constexpr size_t OFF_HIST = OFF_PART + sizeof(FramePart);
    static constexpr size_t OFF_RAW = OFF_HIST + sizeof(float) * HLEN * FX_NUM_FEATURES;
    static constexpr size_t OFF_ONSET = OFF_RAW + sizeof(float) * 16;       // detect_onset_wave's scratch
    static constexpr size_t BYTES = OFF_ONSET + sizeof(float) * 64;
    static_assert(OFF_WAVES % 16 == 0 && OFF_PART % 16 == 0, "16-byte aligned sections");
};

template <int N, bool BLOCKS = false>
// (4096 points: two workgroups per CU, so <= 256 VGPRs -- the harmonic wave, which is not the one the hop waits for, then keeps
// six register pairs of its transform in scratch, 52 bytes per lane; measured 34.1 -> 32.6 us per hop all the same)
__global__ void __launch_bounds__(192, HopGeo<N>::WIDE ? 2 : 1)
fx_hop_kernel(const FrameParams p_arg, const EpilogueParams ep_arg, const HopSignal sig)
{
    // BLOCKS: p.in is the channels' new block and the hop is the head of [pending samples | block] (FrameParams::block_mode)
    FrameParams p = p_arg;
    typedef Geo<N> G;
    typedef HopGeo<N> HG;
    constexpr int M = G::M, P = G::P;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2*    tw   = reinterpret_cast<f2*>(smem);
    FramePart* part = reinterpret_cast<FramePart*>(smem + HG::OFF_PART);
    float* s_hist = reinterpret_cast<float*>(smem + HG::OFF_HIST);
    float* s_raw  = reinterpret_cast<float*>(smem + HG::OFF_RAW);
    float* s_onset = reinterpret_cast<float*>(smem + HG::OFF_ONSET);

    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));     // wave-uniform: scalar registers for all that follows from it
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x;
    f2*    cbuf = reinterpret_cast<f2*>(smem + HG::OFF_WAVES + (size_t) G::BUF_BYTES * wave);
    float* rbuf = reinterpret_cast<float*>(cbuf);

    // prologue: twiddles and the history of raw values -- 16 bytes per lane per load, all of them issued before the first is
    // used (this is one trip to memory, not twenty)
    {
        const uint4* src = reinterpret_cast<const uint4*>(HG::WIDE ? p.tw_image : p.tw);
        uint4* dst = reinterpret_cast<uint4*>(tw);
#pragma unroll 4
        for (int i = threadIdx.x; i < FrameLds<N>::TW_ENTRIES / 2; i += 192) dst[i] = src[i];
        const uint4* hs4 = reinterpret_cast<const uint4*>(ep_arg.hist + (size_t) c * HLEN * FX_NUM_FEATURES);     // the channel's ring, row for row
        for (int i = threadIdx.x; i < HLEN * FX_NUM_FEATURES / 4; i += 192) reinterpret_cast<uint4*>(s_hist)[i] = hs4[i];
    }
    if (threadIdx.x == 0) part->flags = 0;
    if (!BLOCKS && sig.stage) {
        // the hop itself: out of the pinned host slot into device memory, 16 bytes per lane, once
        const size_t hop_bytes = (size_t) (N / 2) * (size_t) sample_bytes(p.sample_format);
        const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(p.in) + (size_t) c * hop_bytes);
        uint4* dst = reinterpret_cast<uint4*>(static_cast<unsigned char*>(sig.stage) + (size_t) c * hop_bytes);
        for (int i = threadIdx.x; i < (int) (hop_bytes / 16); i += blockDim.x) dst[i] = src[i];
        p.in = sig.stage;
    }
    __syncthreads();                                            // (waits for the stores above as well)

    TwRegs<N> twr;
    if constexpr (TwRegs<N>::USE) twr.load(tw, lane);
    const double nyquist = p.nyquist;
    typedef FrameWave<N, true, false, HG::WIDE, BLOCKS> Wave;
    const Wave w{p, tw, &twr, nullptr, nullptr, cbuf, rbuf, part, nyquist, 1.0 / nyquist, nyquist / (double) M,
                 1.0f / (float) N, c, 1, 0};

    // every wave assembles the window in its own buffer (a1): three reads of the same 2-8 KB, two of them from L2
    const double ssq_lane = w.load_frame(lane);
    typename Wave::HarmonicSpectrum hs;
    if (wave == 0) {
        (void) w.pitch(lane);                                   // leaves the lag in the record
    } else if (wave == 1) {
        float xr[P];
        double sum_sq;
        if constexpr (G::SPLIT) {
            sum_sq = wave_sum(ssq_lane);
            if (lane == 0) part->sum_sq = sum_sq;
            w.load_raw(lane, xr);
        } else {
            sum_sq = w.sum_squares(lane, xr);
        }
        w.spectral(lane, xr, sum_sq);
        // this analyser's slots of the raw vector (ref RealTimeAnalyser.h:209-226), while the pitch estimate is still running.
        // (Round 3 also finished their smoothing and the onset here, leaving four slots behind the last barrier: measured 1-3 us
        // SLOWER per hop at every size -- the early stores to the pinned host slot are in the way of the later ones -- so everything
        // is finished at the end: profiles/r03_variants.txt.)
        EpilogueParams e1 = ep_arg; e1.analysers = 1;
        float out[FX_NUM_FEATURES];
        finalise_wave(e1, *part, lane, out);                    // (every lane: the logarithms side by side)
        if (lane == 0) {
            s_raw[FX_ONSET] = 0.0f; s_raw[FX_RMS] = out[FX_RMS]; s_raw[FX_CENTROID] = out[FX_CENTROID]; s_raw[FX_SPREAD] = out[FX_SPREAD];
            s_raw[FX_FLATNESS] = out[FX_FLATNESS]; s_raw[FX_LER] = out[FX_LER]; s_raw[FX_FLUX] = out[FX_FLUX]; s_raw[FX_SLOPE] = out[FX_SLOPE];
        }
        wave_fence();
        if constexpr (BLOCKS) {
            // what the block leaves over becomes the channel's pending samples: this wavefront is done and the hop still waits for the pitch
            // estimate (in the prologue the copy's trip to memory stood in front of all three: 20.7 us per launch against 18.8)
            const long long hop_bytes = (long long) (N / 2) * sample_bytes(p.sample_format);
            const BlockStream bs = stream_from(BlockStream{p.blk_carry_in + (size_t) c * (size_t) p.blk_carry_row_bytes,
                                                           static_cast<const unsigned char*>(p.in) + (size_t) c * (size_t) p.blk_in_row_bytes, p.blk_carry_bytes, p.blk_in_row_bytes},
                                               (long long) p.blk_hop0 * hop_bytes);
            if (p.blk_keep_rest) stream_keep_rest(bs, hop_bytes, p.blk_carry_out + (size_t) c * (size_t) p.blk_carry_row_bytes, lane, 64);
        }
    } else {
        float xr[P];
        if constexpr (G::SPLIT) {
            w.load_raw(lane, xr);
        } else {
#pragma unroll
            for (int g = 0; g < G::GA; g++)
#pragma unroll
                for (int j = 0; j < G::RA; j++) xr[g * G::RA + j] = (rbuf + first_pass_rbase<N>(lane, g))[first_pass_rstep<N>(j)];
            wave_fence();
        }
        w.harmonic_spectrum(lane, xr, hs);
    }
    __syncthreads();                                            // the lag is known
    if (wave == 2) {
        const double f0 = (nyquist * 2.0) / (double) part->lag; // ref PitchAnalyser.h:57
        w.harmonic_tail(lane, hs, f0);
        EpilogueParams e2 = ep_arg; e2.analysers = 2;
        float out[FX_NUM_FEATURES];
        finalise_wave(e2, *part, lane, out);
        if (lane == 0) {                                        // the harmonic analyser's slots (ref RealTimeAnalyser.h:150-172)
            s_raw[FX_F0] = out[FX_F0]; s_raw[FX_HER] = out[FX_HER]; s_raw[FX_OER] = out[FX_OER]; s_raw[FX_INHARM] = out[FX_INHARM];
        }
    }
    __syncthreads();                                            // the record is complete
    if (wave != 0) return;

    // the rest of the hop -- smoothing, onset, history -- as fx_tail_fused_kernel does it for T = 1, from LDS, one slot per lane
    EpilogueParams ep = ep_arg;
    ep.raw = s_raw - (size_t) c * FX_NUM_FEATURES;              // epilogue_hop / history_value index by channel
    ep.hist = s_hist - (size_t) c * HLEN * FX_NUM_FEATURES;     // (the LDS copy of the ring)
    epilogue_hop(ep, c, lane, s_onset);
    ep.hist = ep_arg.hist;                                      // this hop's row of the ring, and nothing else
    if (lane < FX_NUM_FEATURES) history_value(ep, c, 0, lane);

    // completion: this workgroup's results (pinned host memory) are visible system-wide before it counts itself in
    if (lane == 0 && sig.host_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned before = __hip_atomic_fetch_add(sig.arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1 == gridDim.x) {
            __hip_atomic_store(sig.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sig.host_flag, sig.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same for windows of 2048 / 4096 points with every analyser on a PAIR of wavefronts (fx_pair_kernel.hip.h): a workgroup
// is one channel and six wavefronts -- pitch pair, spectral pair, harmonic pair -- running the sections of PairWave, i.e.
// bit for bit what fx_pair_kernel computes.  A frame across two wavefronts halves the arithmetic on the critical path (the
// pitch estimate: two transforms, the low-pass, the lag search), which is what a one-hop call waits for.
// ---------------------------------------------------------------------------------------------------------------------
template <int N> struct HopPairGeo {
    typedef PGeo<N> PG;
    static constexpr size_t PAIR_BYTES = PG::BUF_BYTES + PG::PAIR_EXTRA;
    static constexpr size_t OFF_PREV = sizeof(f2) * N;
    static constexpr size_t OFF_PAIRS = OFF_PREV + sizeof(float) * PG::PREV_FLOATS;
    static constexpr size_t OFF_PART = OFF_PAIRS + 3 * PAIR_BYTES;
    static constexpr size_t OFF_HIST = OFF_PART + sizeof(FramePart);
    static constexpr size_t OFF_RAW = OFF_HIST + sizeof(float) * HLEN * FX_NUM_FEATURES;
    static constexpr size_t OFF_ONSET = OFF_RAW + sizeof(float) * 16;
    static constexpr size_t OFF_READY = OFF_ONSET + sizeof(float) * 64;          // loader wavefronts that have finished the prologue's tables
    static constexpr size_t BYTES = OFF_READY + 16;
    static_assert(OFF_PAIRS % 16 == 0 && OFF_PART % 16 == 0 && PAIR_BYTES % 16 == 0, "16-byte aligned sections");
};

template <int N>
__global__ void __launch_bounds__(384, 1)
fx_hop_pair_kernel(const FrameParams p_arg, const EpilogueParams ep_arg, const HopSignal sig)
{
    FrameParams p = p_arg;
    typedef PGeo<N> PG;
    typedef HopPairGeo<N> HG;
    constexpr int M = PG::M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2*    tw   = reinterpret_cast<f2*>(smem);
    float* prev = reinterpret_cast<float*>(smem + HG::OFF_PREV);
    int*   turn2 = reinterpret_cast<int*>(prev + M);
    FramePart* part = reinterpret_cast<FramePart*>(smem + HG::OFF_PART);
    float* s_hist = reinterpret_cast<float*>(smem + HG::OFF_HIST);
    float* s_raw  = reinterpret_cast<float*>(smem + HG::OFF_RAW);
    float* s_onset = reinterpret_cast<float*>(smem + HG::OFF_ONSET);
    unsigned* s_ready = reinterpret_cast<unsigned*>(smem + HG::OFF_READY);

    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));     // wave-uniform: scalar registers for all that follows from it
    const int lane = threadIdx.x & 63;
    const int pair = wave >> 1, w = wave & 1;
    const int c = blockIdx.x;
    unsigned char* mine = smem + HG::OFF_PAIRS + HG::PAIR_BYTES * pair;
    f2* cbuf = reinterpret_cast<f2*>(mine);
    double* mbox = reinterpret_cast<double*>(mine + PG::BUF_BYTES);
    unsigned* flags = reinterpret_cast<unsigned*>(mine + PG::BUF_BYTES + 8 * PG::MBOX_DOUBLES);

    // Prologue, split by who needs what first.  The pitch pair -- the hop's critical path -- copies the hop out of the pinned
    // slot (16 bytes per lane, once: the slot is un-cached memory across PCIe) and starts on it at once; meanwhile the other two
    // pairs bring in the twiddle table, the flux state and the history of raw values.  Nobody waits for anybody he does not
    // need: each of the four loading wavefronts counts itself in when its loads are in LDS and goes on when all four are; the
    // pitch pair looks at that count only after its low-pass, right before its first transform reads a twiddle.
    auto tables_loaded = [&] {
        while (__hip_atomic_load(s_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u) { }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    if (threadIdx.x == 0) { turn2[0] = 0; part->flags = 0; s_ready[0] = 0u; s_ready[1] = 0u; }
    if (lane == 0) flags[w] = 0u;
    __syncthreads();                                            // (every pair's arrival counters are zero before any pair synchronises on them)
    f2 early_b[15];
    if (pair == 0) {
        {
            typedef Plan<N> PL;
            const f2* g = reinterpret_cast<const f2*>(p.tw) + PL::OFF1 + lane % PL::L1;
#pragma unroll
            for (int i = 0; i < 15; i++) early_b[i] = g[i * PL::L1];
        }
        if (sig.stage) {
            const size_t hop_bytes = (size_t) (N / 2) * (size_t) sample_bytes(p.sample_format);
            const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(p.in) + (size_t) c * hop_bytes);
            uint4* dst = reinterpret_cast<uint4*>(static_cast<unsigned char*>(sig.stage) + (size_t) c * hop_bytes);
            for (int i = threadIdx.x; i < (int) (hop_bytes / 16); i += 128) dst[i] = src[i];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wavefront's part of the copy is in memory ...
            if (lane == 0) __hip_atomic_fetch_add(s_ready + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ... before it says so
        }
    } else {
        const int tid = threadIdx.x - 128;
        const uint4* src = reinterpret_cast<const uint4*>(p.tw);
        uint4* dst = reinterpret_cast<uint4*>(tw);
#pragma unroll 4
        for (int i = tid; i < N / 2; i += 256) dst[i] = src[i];
        const f4* ps = reinterpret_cast<const f4*>(p.prev_re + (size_t) c * M);
        for (int i = tid; i < M / 4; i += 256) reinterpret_cast<f4*>(prev)[i] = ps[i];
        const uint4* hs4 = reinterpret_cast<const uint4*>(ep_arg.hist + (size_t) c * HLEN * FX_NUM_FEATURES);     // the channel's ring, row for row
        for (int i = tid; i < HLEN * FX_NUM_FEATURES / 4; i += 256) reinterpret_cast<uint4*>(s_hist)[i] = hs4[i];
    }
    if (sig.stage) p.in = sig.stage;
    if (pair != 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wavefront's share of the tables is in LDS
        if (lane == 0) __hip_atomic_fetch_add(s_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tables_loaded();
        if (sig.stage) {                                        // the hop itself: staged by the pitch pair's two wavefronts
            while (__hip_atomic_load(s_ready + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 2u) { }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }

    const double nyquist = p.nyquist;
    PairWave<N> pw{p, tw, prev, turn2, cbuf, reinterpret_cast<float*>(cbuf), mbox, flags, part,
                   nyquist, 1.0 / nyquist, nyquist / (double) M, 1.0f / (float) N, c, 1, 0, w, 0u, 0u};
    typename PairWave<N>::HarmonicSpectrum hs;
    if (pair == 0) {
        // a1 + the pitch estimate (a10-a14): the window into this pair's real image, low-pass, two transforms, lag search
        pw.pair_sync(lane);                                     // both halves of the hop are staged (and this pair's flags are initialised)
        (void) pw.load_half_window(lane);
        pw.pair_sync(lane);
        // the tables are awaited where the first transform's LAST pass is about to read them: its second pass runs on the 15
        // twiddles this lane fetched from global memory at kernel entry (`early_b`, the same values as the LDS copy)
        (void) pw.pitch(lane, tables_loaded, &early_b);         // leaves the lag in the record
    } else if (pair == 1) {
        // a2 + the spectral analyser (a3-a7): the sum of squares exactly as fx_pair_kernel takes it (each wave its half of the
        // window, wave 0 + wave 1), then the windowed transform and its sums
        const double ssq_wave = pw.load_half_window(lane);
        if (lane == 0) *pw.slot(w, 0) = ssq_wave;
        pw.pair_sync(lane);
        const double sum_sq = *pw.slot(0, 0) + *pw.slot(1, 0);
        pw.next_exchange();
        if (w == 0 && lane == 0) part->sum_sq = sum_sq;
        // (the image itself is not used here; each wave wrote its half into its own region before the exchange above)
        pw.spectral(lane, sum_sq);
        pw.pair_sync(lane);                                     // both waves' shares of the record are written, the flux state is final
        for (int i = 64 * w + lane; i < M; i += 128) p.prev_re[(size_t) c * M + i] = prev[i];
        if (w == 0) {
            // this analyser's slots (ref RealTimeAnalyser.h:209-226), while the pitch estimate is still running
            EpilogueParams e1 = ep_arg; e1.analysers = 1;
            float out[FX_NUM_FEATURES];
            finalise_wave(e1, *part, lane, out);                // (every lane: the logarithms side by side)
            if (lane == 0) {
                s_raw[FX_ONSET] = 0.0f; s_raw[FX_RMS] = out[FX_RMS]; s_raw[FX_CENTROID] = out[FX_CENTROID]; s_raw[FX_SPREAD] = out[FX_SPREAD];
                s_raw[FX_FLATNESS] = out[FX_FLATNESS]; s_raw[FX_LER] = out[FX_LER]; s_raw[FX_FLUX] = out[FX_FLUX]; s_raw[FX_SLOPE] = out[FX_SLOPE];
            }
            wave_fence();
        }
    } else {
        pw.harmonic_spectrum(lane, hs);                         // a15, up to where the pitch is needed
    }
    __syncthreads();                                            // the lag is known
    if (pair == 2) {
        const double f0 = (nyquist * 2.0) / (double) part->lag; // ref PitchAnalyser.h:57
        pw.harmonic_tail(lane, hs, f0);                         // a16-a18
        if (w == 0) {                                           // the harmonic analyser's slots (ref RealTimeAnalyser.h:150-172)
            EpilogueParams e2 = ep_arg; e2.analysers = 2;
            float out[FX_NUM_FEATURES];
            finalise_wave(e2, *part, lane, out);
            if (lane == 0) { s_raw[FX_F0] = out[FX_F0]; s_raw[FX_HER] = out[FX_HER]; s_raw[FX_OER] = out[FX_OER]; s_raw[FX_INHARM] = out[FX_INHARM]; }
        }
    }
    __syncthreads();                                            // the record is complete
    if (wave != 0) return;

    EpilogueParams ep = ep_arg;
    ep.raw = s_raw - (size_t) c * FX_NUM_FEATURES;
    ep.hist = s_hist - (size_t) c * HLEN * FX_NUM_FEATURES;     // (the LDS copy of the ring)
    epilogue_hop(ep, c, lane, s_onset);
    ep.hist = ep_arg.hist;                                      // this hop's row of the ring, and nothing else
    if (lane < FX_NUM_FEATURES) history_value(ep, c, 0, lane);
    if (lane == 0 && sig.host_flag) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned before = __hip_atomic_fetch_add(sig.arrivals, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1 == gridDim.x) {
            __hip_atomic_store(sig.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sig.host_flag, sig.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int N> hipError_t hop_pair_prepare_t()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_hop_pair_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int N> hipError_t hop_pair_launch_t(const FrameParams& p, const EpilogueParams& ep, const HopSignal& sig, hipStream_t stream)
{
    hipLaunchKernelGGL((fx_hop_pair_kernel<N>), dim3((unsigned) p.C), dim3(384), HopPairGeo<N>::BYTES, stream, p, ep, sig);
    return hipGetLastError();
}

template <int N> hipError_t hop_prepare_t()
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_hop_kernel<N, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    return e != hipSuccess ? e : hipFuncSetAttribute(reinterpret_cast<const void*>(&fx_hop_kernel<N, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int N> hipError_t hop_launch_t(const FrameParams& p, const EpilogueParams& ep, const HopSignal& sig, hipStream_t stream)
{
    if (p.block_mode) {
        if (sig.stage || !block_feed_valid(N, p)) return hipErrorInvalidValue;
        hipLaunchKernelGGL((fx_hop_kernel<N, true>), dim3((unsigned) p.C), dim3(192), HopGeo<N>::BYTES, stream, p, ep, sig);
    } else {
        hipLaunchKernelGGL((fx_hop_kernel<N, false>), dim3((unsigned) p.C), dim3(192), HopGeo<N>::BYTES, stream, p, ep, sig);
    }
    return hipGetLastError();
}
