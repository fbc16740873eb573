// Cycles a SIMD needs per 64-lane instruction, by opcode: each kernel runs a loop of 64 independent copies of one instruction
// (inline assembly: the compiler neither fuses nor reorders them) on every SIMD of the chip at 4 and 8 waves per SIMD, after
// some tens of milliseconds of load.  cycles = SIMDs x clock x time / instructions, clock taken as 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float *out, float a, float b)                               \
    {                                                                                                       \
        float x0 = threadIdx.x + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;                      \
        float y = a, z = b; double dy = a, dz = b;                                                          \
        for (int it = 0; it < N_ITER; it++) {                                                               \
            __asm__ volatile(REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM) REP8(ASM) \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),     \
                               "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)      \
                             : "v"(y), "v"(z), "v"(dy), "v"(dz), "s"(a), "s"(b) : "vcc", "s10", "s11");                       \
        }                                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }
// operands: %0..%7 float registers, %8..%15 double pairs, %16 y, %17 z, %18 dy, %19 dz
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %16, %17\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define A_ADD(i) "v_add_f32 %" #i ", %" #i ", %17\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n"
#define A_CMP_S(i) "v_cmp_lt_f32 s[10:11], %" #i ", %16\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %16\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %16\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 13\n"
#define A_XAD(i) "v_xad_u32 %" #i ", %" #i ", %16, %17\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %16, %" #i ", %17\n"
#define A_MIN(i) "v_min_f32 %" #i ", %" #i ", %16\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %16, %17\n"
#define A_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %16, %" #i "\n"
#define A_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %16, %17\n"
#define A_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %16, %17\n"
#define A_FREXP(i) "v_frexp_exp_i32_f32 %" #i ", %" #i "\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_CVTU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %16\n"
#define A_FMA64_(i, j) "v_fma_f64 %" #j ", %" #j ", %18, %19\n"
#define A_FMA64(i) A_FMA64_(i, i)
#define A_DPP(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define A_READLANE(i) "v_readlane_b32 s10, %" #i ", 3\n"
#define A_PKFMA_(i, j) "v_pk_fma_f32 %" #j ", %" #j ", %18, %19\n"
#define A_NOP(i) "s_nop 0\n"
// %20, %21: the same values in scalar registers
#define A_MUL_S(i) "v_mul_f32 %" #i ", %20, %" #i "\n"
#define A_ADD_S(i) "v_add_f32 %" #i ", %21, %" #i "\n"
#define A_FMA_S(i) "v_fma_f32 %" #i ", %" #i ", %20, %17\n"
#define A_FMA_SS(i) "v_fma_f32 %" #i ", %" #i ", %20, %20\n"
#define A_FMA_K(i) "v_fma_f32 %" #i ", %" #i ", %16, 1.0\n"
#define A_MUL_K(i) "v_mul_f32 %" #i ", 0.5, %" #i "\n"
#define A_ADD_L(i) "v_add_f32 %" #i ", 0x3f7fbe77, %" #i "\n"
#define A_FMAC(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define A_SUB(i) "v_sub_f32 %" #i ", %" #i ", %17\n"
#define A_MAX(i) "v_max_f32 %" #i ", %" #i ", %16\n"
#define A_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
#define A_CMPCND_S(i) "v_cmp_lt_f32 s[10:11], %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, s[10:11]\n"
#define A_CMPU(i) "v_cmp_lt_u32 vcc, %" #i ", %16\n"
#define A_CLASS(i) "v_cmp_class_f32 vcc, %" #i ", %16\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %16\n"
#define A_OR3(i) "v_or3_b32 %" #i ", %" #i ", %16, %17\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %16, %17\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %16\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %16, %17\n"
#define A_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %16\n"
#define A_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define A_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %" #i ", %16\n"
#define A_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %16\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %16, %17\n"
#define A_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define A_BPERM(i) "ds_bpermute_b32 %" #i ", %16, %" #i "\ns_waitcnt lgkmcnt(0)\n"
#define A_BPERM_NW(i) "ds_bpermute_b32 %" #i ", %16, %" #i "\n"
// double-pair operands are %8..%15: a second set of kernels maps i -> i+8
#define D8(X) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL_D(NAME, ASM)                                                                                 \
    __global__ void __launch_bounds__(256) NAME(float *out, float a, float b)                               \
    {                                                                                                       \
        float x0 = threadIdx.x + 1.0f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;                      \
        float y = a, z = b; double dy = a, dz = b;                                                          \
        for (int it = 0; it < N_ITER; it++) {                                                               \
            __asm__ volatile(D8(ASM) D8(ASM) D8(ASM) D8(ASM) D8(ASM) D8(ASM) D8(ASM) D8(ASM)                \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7),     \
                               "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)      \
                             : "v"(y), "v"(z), "v"(dy), "v"(dz), "s"(a), "s"(b) : "vcc");                                   \
        }                                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }
#define B_FMA64(j) "v_fma_f64 %" #j ", %" #j ", %18, %19\n"
#define B_MUL64(j) "v_mul_f64 %" #j ", %" #j ", %18\n"
#define B_ADD64(j) "v_add_f64 %" #j ", %" #j ", %19\n"
#define B_PKFMA(j) "v_pk_fma_f32 %" #j ", %" #j ", %18, %19\n"
#define B_PKMUL(j) "v_pk_mul_f32 %" #j ", %" #j ", %18\n"
#define B_CVT64(j) "v_cvt_f64_f32 %" #j ", %16\n"
#define B_RNDNE64(j) "v_rndne_f64 %" #j ", %" #j "\n"
#define B_FMA64_S(j) "v_fma_f64 %" #j ", %" #j ", %18, s[10:11]\n"
#define B_LSHLADD64(j) "v_lshl_add_u64 %" #j ", %" #j ", 2, %18\n"
KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_mov, A_MOV) KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp, A_CMP) KERNEL(k_cmp_s, A_CMP_S) KERNEL(k_addu, A_ADDU) KERNEL(k_and, A_AND) KERNEL(k_lshl, A_LSHL) KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_xad, A_XAD) KERNEL(k_bfi, A_BFI) KERNEL(k_min, A_MIN) KERNEL(k_max3, A_MAX3) KERNEL(k_divscale, A_DIVSCALE)
KERNEL(k_divfmas, A_DIVFMAS) KERNEL(k_divfixup, A_DIVFIXUP) KERNEL(k_frexp, A_FREXP) KERNEL(k_rcp, A_RCP) KERNEL(k_sqrt, A_SQRT)
KERNEL(k_cvtu, A_CVTU) KERNEL(k_mullo, A_MULLO) KERNEL(k_dpp, A_DPP) KERNEL(k_readlane, A_READLANE) KERNEL(k_nop, A_NOP)
KERNEL(k_mul_sgpr, A_MUL_S) KERNEL(k_add_sgpr, A_ADD_S) KERNEL(k_fma_sgpr, A_FMA_S) KERNEL(k_fma_2sgpr, A_FMA_SS) KERNEL(k_fma_const, A_FMA_K)
KERNEL(k_mul_const, A_MUL_K) KERNEL(k_add_literal, A_ADD_L) KERNEL(k_fmac, A_FMAC) KERNEL(k_sub, A_SUB) KERNEL(k_max, A_MAX)
KERNEL(k_cmp_cndmask_vcc, A_CMPCND) KERNEL(k_cmp_cndmask_sgpr, A_CMPCND_S) KERNEL(k_cmp_u32, A_CMPU) KERNEL(k_cmp_class, A_CLASS)
KERNEL(k_xor, A_XOR) KERNEL(k_or3, A_OR3) KERNEL(k_add3, A_ADD3) KERNEL(k_lshl_add, A_LSHLADD) KERNEL(k_and_or, A_ANDOR) KERNEL(k_subu, A_SUBU)
KERNEL(k_lshr, A_LSHR) KERNEL(k_mbcnt, A_MBCNT) KERNEL(k_ldexp, A_LDEXP) KERNEL(k_med3, A_MED3) KERNEL(k_cvt_i32, A_CVTI)
KERNEL(k_bpermute_wait, A_BPERM) KERNEL(k_bpermute, A_BPERM_NW)
KERNEL_D(k_cvt_f64_f32, B_CVT64) KERNEL_D(k_rndne_f64, B_RNDNE64) KERNEL_D(k_fma64_sgpr, B_FMA64_S) KERNEL_D(k_lshl_add_u64, B_LSHLADD64)
KERNEL_D(k_fma64, B_FMA64) KERNEL_D(k_mul64, B_MUL64) KERNEL_D(k_add64, B_ADD64) KERNEL_D(k_pkfma, B_PKFMA) KERNEL_D(k_pkmul, B_PKMUL)
typedef void (*kern_t)(float *, float, float);
static void run(const char *name, kern_t k, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s", name);
    for (int waves = 4; waves <= 8; waves *= 2) {
        const int blocks = 256 * waves;
        float ms = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            float t = 0;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 0.999f, 0.001f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1);
            if (t < ms) ms = t;
        }
        const double insts = (double)blocks * 4 * N_ITER * 64;   /* kernels named *_cndmask_* and k_bpermute_wait issue two instructions per copy: their figures are per pair */
        printf("  %d waves/SIMD: %6.1f G/s = %.2f cycles", waves, insts / ms / 1e6, 1024.0 * 2.4e9 * (ms * 1e-3) / insts);
    }
    printf("\n");
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int rep = 0; rep < 40; rep++) hipLaunchKernelGGL(k_fma, dim3(2048), dim3(256), 0, 0, d, 0.999f, 0.001f);
    hipDeviceSynchronize();
#define RUN(k) run(#k, k, d);
    RUN(k_fma) RUN(k_mul) RUN(k_add) RUN(k_min) RUN(k_max3) RUN(k_mov) RUN(k_cndmask) RUN(k_cmp) RUN(k_cmp_s) RUN(k_addu) RUN(k_and) RUN(k_lshl)
    RUN(k_alignbit) RUN(k_xad) RUN(k_bfi) RUN(k_mullo) RUN(k_cvtu) RUN(k_frexp) RUN(k_divscale) RUN(k_divfmas) RUN(k_divfixup)
    RUN(k_mul_sgpr) RUN(k_add_sgpr) RUN(k_fma_sgpr) RUN(k_fma_2sgpr) RUN(k_fma_const) RUN(k_mul_const) RUN(k_add_literal) RUN(k_fmac) RUN(k_sub) RUN(k_max)
    RUN(k_cmp_cndmask_vcc) RUN(k_cmp_cndmask_sgpr) RUN(k_cmp_u32) RUN(k_cmp_class) RUN(k_xor) RUN(k_or3) RUN(k_add3) RUN(k_lshl_add) RUN(k_and_or) RUN(k_subu)
    RUN(k_lshr) RUN(k_mbcnt) RUN(k_ldexp) RUN(k_med3) RUN(k_cvt_i32) RUN(k_bpermute_wait) RUN(k_bpermute)
    RUN(k_cvt_f64_f32) RUN(k_rndne_f64) RUN(k_fma64_sgpr) RUN(k_lshl_add_u64)
    RUN(k_rcp) RUN(k_sqrt) RUN(k_fma64) RUN(k_mul64) RUN(k_add64) RUN(k_pkfma) RUN(k_pkmul) RUN(k_dpp) RUN(k_readlane) RUN(k_nop)
    return 0;
}
