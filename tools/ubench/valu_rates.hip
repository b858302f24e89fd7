// valu_rates.hip -- issue-rate microbenchmark for gfx950: ns per wave-instruction per SIMD of the opcodes the feature
// kernels use, at 4 and 2 waves per SIMD (256 workgroups x w per CU-quarter, 256 threads each; every kernel is 16 asm
// blocks of 8 independent copies of one instruction -- on lane-private registers 0..7 (fp32 / b32) or 8..15 (fp64 /
// packed) -- so the measured time is issue cost, not latency).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates && tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

// one instruction applied to operand i (F: registers %0..%7, D: %8..%15); %16 = c, %17 = c2 (floats), %18 = dc (double)
#define S(x) #x
#define F8(T) T(0) "\n" T(1) "\n" T(2) "\n" T(3) "\n" T(4) "\n" T(5) "\n" T(6) "\n" T(7)
#define D8(T) T(8) "\n" T(9) "\n" T(10) "\n" T(11) "\n" T(12) "\n" T(13) "\n" T(14) "\n" T(15)
#define OPS : "+v"(a[0]),"+v"(a[1]),"+v"(a[2]),"+v"(a[3]),"+v"(a[4]),"+v"(a[5]),"+v"(a[6]),"+v"(a[7]), \
              "+v"(d[0]),"+v"(d[1]),"+v"(d[2]),"+v"(d[3]),"+v"(d[4]),"+v"(d[5]),"+v"(d[6]),"+v"(d[7])   \
            : "v"(c),"v"(c2),"v"(dc) : "vcc","s20","s21"
#define BLOCK(BODY) asm volatile(BODY OPS);
#define X16(B) B B B B B B B B B B B B B B B B
#define KERNEL(NAME, BODY)                                                                                          \
__global__ void __launch_bounds__(256) NAME(float* out, int iters, float seed) {                                    \
    float a[8]; double d[8];                                                                                        \
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x + i; d[i] = a[i] * 1.5; }                               \
    float c = seed * 0.5f + 1.0f, c2 = seed + 3.0f; double dc = c;                                                  \
    for (int i = 0; i < iters; i++) { X16(BLOCK(BODY)) }                                                            \
    float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float) d[i];                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                 \
}

#define T_MUL(i)     "v_mul_f32 %" S(i) ", %" S(i) ", %16"
#define T_ADD(i)     "v_add_f32 %" S(i) ", %" S(i) ", %16"
#define T_FMA(i)     "v_fma_f32 %" S(i) ", %" S(i) ", %16, %17"
#define T_MAX(i)     "v_max_f32 %" S(i) ", %" S(i) ", %16"
#define T_ADDU(i)    "v_add_u32 %" S(i) ", %" S(i) ", %16"
#define T_SHL(i)     "v_lshlrev_b32 %" S(i) ", 1, %" S(i)
#define T_AND(i)     "v_and_b32 %" S(i) ", %" S(i) ", %16"
#define T_MOV(i)     "v_mov_b32 %" S(i) ", %16"
#define T_CND(i)     "v_cndmask_b32 %" S(i) ", %" S(i) ", %16, vcc"
#define T_CND64(i)   "v_cndmask_b32_e64 %" S(i) ", %" S(i) ", %16, s[20:21]"
#define T_CMP(i)     "v_cmp_gt_f32 vcc, %" S(i) ", %16"
#define T_CMP64(i)   "v_cmp_gt_f32_e64 s[20:21], %" S(i) ", %16"
#define T_CMPCND(i)  "v_cmp_gt_f32 vcc, %" S(i) ", %16\n v_cndmask_b32 %" S(i) ", %" S(i) ", %16, vcc"
#define T_ADDDPP(i)  "v_add_f32_dpp %" S(i) ", %" S(i) ", %" S(i) " row_shr:1 row_mask:0xf bank_mask:0xf"
#define T_MOVDPP(i)  "v_mov_b32_dpp %" S(i) ", %" S(i) " row_shr:1 row_mask:0xf bank_mask:0xf"
#define T_RDLANE(i)  "v_readlane_b32 s20, %" S(i) ", 3"
#define T_RDFIRST(i) "v_readfirstlane_b32 s20, %" S(i)
#define T_MULLO(i)   "v_mul_lo_u32 %" S(i) ", %" S(i) ", %16"
#define T_MAD24(i)   "v_mad_u32_u24 %" S(i) ", %" S(i) ", %16, %17"
#define T_BFE(i)     "v_bfe_u32 %" S(i) ", %" S(i) ", 3, 5"
#define T_RCP(i)     "v_rcp_f32 %" S(i) ", %" S(i)
#define T_LOG(i)     "v_log_f32 %" S(i) ", %" S(i)
#define T_RCP64(i)   "v_rcp_f64 %" S(i) ", %" S(i)
#define T_SQRT64(i)  "v_sqrt_f64 %" S(i) ", %" S(i)
#define T_LDEXP64(i) "v_ldexp_f64 %" S(i) ", %" S(i) ", 1"
#define T_FREXP64(i) "v_frexp_mant_f64 %" S(i) ", %" S(i)
#define T_ADD64(i)   "v_add_f64 %" S(i) ", %" S(i) ", %18"
#define T_FMA64(i)   "v_fma_f64 %" S(i) ", %" S(i) ", %18, %18"
#define T_MAX64(i)   "v_max_f64 %" S(i) ", %" S(i) ", %18"
#define T_PKMUL(i)   "v_pk_mul_f32 %" S(i) ", %" S(i) ", %18"
#define T_PKMULS(i)  "v_pk_mul_f32 %" S(i) ", %" S(i) ", %18 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]"
#define T_PKADD(i)   "v_pk_add_f32 %" S(i) ", %" S(i) ", %18"
#define T_MOV64(i)   "v_mov_b64 %" S(i) ", %18"
#define T_NOP(i)     "s_nop 0"
// one compare, then 8 selects on the same mask: through VCC / through an SGPR pair
#define CMP_8CND   "v_cmp_gt_f32 vcc, %0, %16\n" F8(T_CND)
#define CMP_8CND64 "v_cmp_gt_f32_e64 s[20:21], %0, %16\n" F8(T_CND64)
#define SMOV_8CND  "s_mov_b64 vcc, exec\n" F8(T_CND)
// the same select with VCC named as an explicit VOP3 operand; and a compare / select pair with other work in between
#define T_CNDV3(i)   "v_cndmask_b32_e64 %" S(i) ", %" S(i) ", %16, vcc"
#define SMOV_8CNDV3 "s_mov_b64 vcc, exec\n" F8(T_CNDV3)
#define T_CMP_X_CND(i) "v_cmp_gt_f32 vcc, %" S(i) ", %16\n v_add_f32 %" S(i) ", %" S(i) ", %17\n v_cndmask_b32 %" S(i) ", %" S(i) ", %16, vcc"
#define T_CMP_XX_CND(i) "v_cmp_gt_f32 vcc, %" S(i) ", %16\n v_add_f32 %" S(i) ", %" S(i) ", %17\n v_mul_f32 %" S(i) ", %" S(i) ", %17\n v_add_f32 %" S(i) ", %" S(i) ", %17\n v_cndmask_b32 %" S(i) ", %" S(i) ", %16, vcc"
// the two conversions: float register j <-> double register j + 8
#define CVT_DOWN "v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15"
#define CVT_UP   "v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %10, %2\n v_cvt_f64_f32 %11, %3\n v_cvt_f64_f32 %12, %4\n v_cvt_f64_f32 %13, %5\n v_cvt_f64_f32 %14, %6\n v_cvt_f64_f32 %15, %7"

KERNEL(k_mul, F8(T_MUL))        KERNEL(k_add, F8(T_ADD))        KERNEL(k_fma, F8(T_FMA))        KERNEL(k_max, F8(T_MAX))
KERNEL(k_addu, F8(T_ADDU))      KERNEL(k_shl, F8(T_SHL))        KERNEL(k_and, F8(T_AND))        KERNEL(k_mov, F8(T_MOV))
KERNEL(k_cnd, F8(T_CND))        KERNEL(k_cnd64, F8(T_CND64))    KERNEL(k_cmp, F8(T_CMP))        KERNEL(k_cmp64, F8(T_CMP64))
KERNEL(k_cmpcnd, F8(T_CMPCND))  KERNEL(k_adddpp, F8(T_ADDDPP))  KERNEL(k_movdpp, F8(T_MOVDPP))  KERNEL(k_rdlane, F8(T_RDLANE))
KERNEL(k_rdfirst, F8(T_RDFIRST)) KERNEL(k_mullo, F8(T_MULLO))   KERNEL(k_mad24, F8(T_MAD24))    KERNEL(k_bfe, F8(T_BFE))
KERNEL(k_cvtdown, CVT_DOWN)     KERNEL(k_cvtup, CVT_UP)         KERNEL(k_rcp, F8(T_RCP))        KERNEL(k_log, F8(T_LOG))
KERNEL(k_rcp64, D8(T_RCP64))    KERNEL(k_sqrt64, D8(T_SQRT64))  KERNEL(k_ldexp64, D8(T_LDEXP64)) KERNEL(k_frexp64, D8(T_FREXP64))
KERNEL(k_add64, D8(T_ADD64))    KERNEL(k_fma64, D8(T_FMA64))    KERNEL(k_max64, D8(T_MAX64))    KERNEL(k_pkmul, D8(T_PKMUL))
KERNEL(k_pkmuls, D8(T_PKMULS))  KERNEL(k_pkadd, D8(T_PKADD))    KERNEL(k_mov64, D8(T_MOV64))    KERNEL(k_nop, F8(T_NOP))
KERNEL(k_cmp8cnd, CMP_8CND)     KERNEL(k_cmp8cnd64, CMP_8CND64) KERNEL(k_smov8cnd, SMOV_8CND)
KERNEL(k_smov8cndv3, SMOV_8CNDV3) KERNEL(k_cmpxcnd, F8(T_CMP_X_CND)) KERNEL(k_cmpxxcnd, F8(T_CMP_XX_CND))

typedef void (*kern)(float*, int, float);
struct E { const char* name; kern k; int per; };
static E es[] = {{"v_mul_f32", k_mul, 8}, {"v_add_f32", k_add, 8}, {"v_fma_f32 (3 distinct src)", k_fma, 8}, {"v_max_f32", k_max, 8},
                 {"v_add_u32", k_addu, 8}, {"v_lshlrev_b32", k_shl, 8}, {"v_and_b32", k_and, 8}, {"v_mov_b32", k_mov, 8},
                 {"v_cndmask vcc", k_cnd, 8}, {"v_cndmask e64 s[20:21]", k_cnd64, 8}, {"v_cmp_gt_f32 vcc", k_cmp, 8},
                 {"v_cmp_gt_f32 e64 sgpr", k_cmp64, 8}, {"v_cmp+v_cndmask pair", k_cmpcnd, 16}, {"v_add_f32_dpp row_shr", k_adddpp, 8},
                 {"v_mov_b32_dpp", k_movdpp, 8}, {"v_readlane_b32", k_rdlane, 8}, {"v_readfirstlane_b32", k_rdfirst, 8},
                 {"v_mul_lo_u32", k_mullo, 8}, {"v_mad_u32_u24", k_mad24, 8}, {"v_bfe_u32", k_bfe, 8}, {"v_cvt_f32_f64", k_cvtdown, 8},
                 {"v_cvt_f64_f32", k_cvtup, 8}, {"v_rcp_f32", k_rcp, 8}, {"v_log_f32", k_log, 8}, {"v_rcp_f64", k_rcp64, 8},
                 {"v_sqrt_f64", k_sqrt64, 8}, {"v_ldexp_f64", k_ldexp64, 8}, {"v_frexp_mant_f64", k_frexp64, 8}, {"v_add_f64", k_add64, 8},
                 {"v_fma_f64", k_fma64, 8}, {"v_max_f64", k_max64, 8}, {"v_pk_mul_f32", k_pkmul, 8}, {"v_pk_mul_f32 op_sel", k_pkmuls, 8},
                 {"v_pk_add_f32", k_pkadd, 8}, {"v_mov_b64", k_mov64, 8}, {"s_nop 0", k_nop, 8},
                 {"1 v_cmp vcc + 8 v_cndmask vcc", k_cmp8cnd, 9}, {"1 v_cmp sgpr + 8 v_cndmask sgpr", k_cmp8cnd64, 9},
                 {"s_mov vcc + 8 v_cndmask vcc", k_smov8cnd, 9}, {"s_mov vcc + 8 v_cndmask_e64 ..., vcc", k_smov8cndv3, 9},
                 {"v_cmp vcc, v_add, v_cndmask vcc", k_cmpxcnd, 24}, {"v_cmp vcc, 3 VALU, v_cndmask vcc", k_cmpxxcnd, 40}};

int main()
{
    float* d; (void) hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int iters = 1000;
    for (int w : {4, 2}) for (auto& e : es) {
        hipLaunchKernelGGL(e.k, dim3(256 * w), dim3(256), 0, 0, d, 10, 1.0f); (void) hipDeviceSynchronize();
        (void) hipEventRecord(e0); hipLaunchKernelGGL(e.k, dim3(256 * w), dim3(256), 0, 0, d, iters, 1.0f); (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        const double n = (double) iters * 16 * e.per * w;
        std::printf("%-28s w/SIMD %d  %7.3f ms  %.2f ns/instr/SIMD  (%.2f cyc @2.4GHz)\n", e.name, w, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
    }
    return 0;
}
