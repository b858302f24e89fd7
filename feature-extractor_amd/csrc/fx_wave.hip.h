// fx_wave.hip.h -- wavefront-level primitives: fences, DPP reductions and scans, lane exchange
// Included by fx_kernels.hip inside namespace fxk (one translation unit: every kernel sees the same
// inlined helpers); not a stand-alone header.
#define FX_MARK(name) asm volatile("; FXMARK " name)


typedef float  __attribute__((ext_vector_type(2))) f2;
typedef float  __attribute__((ext_vector_type(4))) f4;

// lane mask of a condition, straight from the compare (HIP's __ballot / __any take an int: a select to 0 / 1 and a second compare)
__device__ __forceinline__ unsigned long long wave_ballot(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool wave_any(bool b) { return __builtin_amdgcn_ballot_w64(b) != 0; }

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
// (a lane index: saying so lets the compiler drop the sign handling of its divisions and the trip tests of per-lane item loops)
template <int N = 0> __device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    __builtin_assume((unsigned) v < 64u);
    return v;
}

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
// Four sums at once.  The first two exchange steps fold the four registers into one -- after them lane l holds the
// partial of sum (l & 3) over its quad -- so the remaining steps move one register instead of four: row_ror:4 / :8
// add the quads of a row position-wise, v_permlane16_swap / v_permlane32_swap (gfx950) add the rows.  45 VALU
// instructions against 4 x 20.
enum { DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128 };
__device__ __forceinline__ double swap_add_rows16(double v)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap_add_halves32(double v)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum4(int lane, double& a, double& b, double& c, double& d)
{
    const bool odd = lane & 1, upper = lane & 2;
    const double ab = (odd ? b : a) + dppz_d<DPP_XOR1>(odd ? a : b);      // even lanes: a over the pair, odd lanes: b
    const double cd = (odd ? d : c) + dppz_d<DPP_XOR1>(odd ? c : d);
    double q = (upper ? cd : ab) + dppz_d<DPP_XOR2>(upper ? ab : cd);     // lane & 3 = 0, 1, 2, 3: a, b, c, d over the quad
    q += dppz_d<DPP_ROW_ROR4>(q);
    q += dppz_d<DPP_ROW_ROR8>(q);                                         // ... over the row
    q = swap_add_rows16(q);
    q = swap_add_halves32(q);                                             // ... over the wave, in every lane
    a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 0), __builtin_amdgcn_readlane(__double2loint(q), 0));
    b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 1), __builtin_amdgcn_readlane(__double2loint(q), 1));
    c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 2), __builtin_amdgcn_readlane(__double2loint(q), 2));
    d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 3), __builtin_amdgcn_readlane(__double2loint(q), 3));
}
// Two sums at once, the same way (one folding step).
__device__ __forceinline__ void wave_sum2(int lane, double& a, double& b)
{
    const bool odd = lane & 1;
    double q = (odd ? b : a) + dppz_d<DPP_XOR1>(odd ? a : b);             // even lanes: a over the pair, odd lanes: b
    q += dppz_d<DPP_XOR2>(q);
    q += dppz_d<DPP_ROW_ROR4>(q);
    q += dppz_d<DPP_ROW_ROR8>(q);
    q = swap_add_rows16(q);
    q = swap_add_halves32(q);
    a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 0), __builtin_amdgcn_readlane(__double2loint(q), 0));
    b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(q), 1), __builtin_amdgcn_readlane(__double2loint(q), 1));
}
// (Moving the exchange steps to ds_swizzle -- the LDS crossbar instead of VALU DPP moves -- was measured 3 % slower:
// the LDS pipe is the kernel's second limiter.  Moving whole sums to the idle matrix pipe -- v_mfma_f64_16x16x4_f64 with
// B = ones adds lanes l, l+16, l+32, l+48; three adds and a second product give every lane the total: 2 MFMAs + 3 adds
// instead of 18 DPP / add instructions -- gave the same bits and was 2 % (single sums) to 6 % (all sums) SLOWER at 1024
// points: each MFMA needs 18 wait states before its result can be read and holds the SIMD's issue port meanwhile.)
// fp32 sum of one value per lane (tree order; for quantities that only need ~1e-6)
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += dppz_f<DPP_XOR1>(v);
    v += dppz_f<DPP_XOR2>(v);
    v += dppz_f<DPP_HALF_MIRROR>(v);
    v += dppz_f<DPP_MIRROR>(v);
    v += dppz_f<DPP_BCAST15>(v);
    v += dppz_f<DPP_BCAST31>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// maximum of values that are >= 0 (or NaN, which never wins -- as in `if (x > max) max = x`).
// One v_max_f32 with a DPP operand per step, written out: from fmaxf() the compiler emits the DPP move, a v_max x, x to quiet
// a signalling NaN that cannot be there, and the maximum -- 19 instructions per reduction instead of 7.  (s_nop 1: a DPP read
// needs two wait states behind the VALU write of its source; rows without a source lane read 0.)
#define FX_MAX_DPP(v, ctrl) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v))
__device__ __forceinline__ float wave_maxf(float v)
{
    FX_MAX_DPP(v, "quad_perm:[1,0,3,2]");
    FX_MAX_DPP(v, "quad_perm:[2,3,0,1]");
    FX_MAX_DPP(v, "row_half_mirror");
    FX_MAX_DPP(v, "row_mirror");
    FX_MAX_DPP(v, "row_bcast:15");
    FX_MAX_DPP(v, "row_bcast:31");
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
