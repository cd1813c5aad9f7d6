// valu_issue_bench.hip — measured ISSUE cost (SIMD cycles per wave64 instruction) of the instruction classes the
// GPIS kernels are made of, on gfx950, at 1 / 2 / 4 resident waves per SIMD.
//
// Why: the march kernels are bound by vector-instruction issue, not by HBM or MFMA, so their
// roofline is   sum_class(count_class * cycles_class) / (SIMD cycles available).   The per-class cycle costs
// are measured here rather than assumed: MI355X_MICROARCH.md gives v_fma_f32 = 2 cycles at >= 2 waves/SIMD
// (4 for a lone wave) and 8 for transcendentals, and nothing for 64-bit integer multiplies, f64 or packed f32.
//
// Method: one kernel per class; every wave runs REPS iterations of a block of 32 independent instructions of the
// class (8 rotating destination registers, so no instruction waits for the previous one) and stamps s_memtime
// around the loop.  W blocks of 256 threads per CU put W waves on every SIMD.  Reported: the median over waves of
//     elapsed_cycles / (REPS * 32 * W)        = SIMD cycles per instruction when W waves share the SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue_bench tools/valu_issue_bench.hip
// Run:   tools/valu_issue_bench > profiles/r02_valu_issue_cycles.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define REPS 2000
#define STR2(x) #x
#define STR(x) STR2(x)

// 32 instructions: BODY(d) is one instruction writing register set d (d = 0..7), sources never written here
#define X8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define X32(I) X8(I) X8(I) X8(I) X8(I)

#define KERNEL(NAME, CLOBBERS, ...)                                                                  \
    __global__ void __launch_bounds__(256) NAME(unsigned long long *out, float fa, float fb, int ia) \
    {                                                                                                \
        __shared__ float lds[1024];                                                                  \
        lds[threadIdx.x] = fa;                                                                       \
        __syncthreads();                                                                             \
        unsigned long long t0, t1;                                                                   \
        asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n v_mov_b32 v42, %2\n v_mov_b32 v43, %0\n" \
                     "v_mov_b32 v44, %1\n v_mov_b32 v45, %0\n v_mov_b32 v46, %1\n v_mov_b32 v47, %0\n" \
                     "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n"                                         \
                     "s_mov_b32 s40, 0x3039\n s_mov_b32 s41, 0x4c957f2d\n s_mov_b32 s42, 5\n s_mov_b64 s[46:47], 0x5555\n"        \
                     :: "v"(fa), "v"(fb), "v"(ia) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "s40", "s41", "s42", "s46", "s47"); \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
        for (int r = 0; r < REPS; ++r)                                                               \
            asm volatile(__VA_ARGS__ ::: CLOBBERS);                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
        if ((threadIdx.x & 63) == 0)                                                                 \
            out[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                              \
    }

#define CLOB32 "scc", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "vcc", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "memory"

// 32-bit destinations v0..v7; 64-bit destinations v[0:1]..v[14:15]
#define I_ADD_F32(d) "v_add_f32 v" #d ", v40, v41\n"
#define I_MUL_F32(d) "v_mul_f32 v" #d ", v40, v41\n"
#define I_FMA_F32(d) "v_fma_f32 v" #d ", v40, v41, v43\n"
#define I_SUB_F32(d) "v_sub_f32 v" #d ", v40, v41\n"
#define I_MAX_F32(d) "v_max_f32 v" #d ", v40, v41\n"
#define I_CMP_F32(d) "v_cmp_lt_f32 vcc, v40, v41\n"
#define I_CNDMASK(d) "v_cndmask_b32 v" #d ", v40, v41, vcc\n"
#define I_MOV_B32(d) "v_mov_b32 v" #d ", v40\n"
#define I_ADD_U32(d) "v_add_u32 v" #d ", v42, v40\n"
#define I_ADD3_U32(d) "v_add3_u32 v" #d ", v42, v40, v41\n"
#define I_XOR_B32(d) "v_xor_b32 v" #d ", v42, v40\n"
#define I_AND_B32(d) "v_and_b32 v" #d ", v42, v40\n"
#define I_OR_B32(d) "v_or_b32 v" #d ", 1.0, v40\n"
#define I_LSHR_B32(d) "v_lshrrev_b32 v" #d ", 13, v40\n"
#define I_ALIGNBIT(d) "v_alignbit_b32 v" #d ", v40, v41, 27\n"
#define I_ALIGNBIT_V(d) "v_alignbit_b32 v" #d ", v40, v40, v42\n"
#define I_MUL_LO_U32(d) "v_mul_lo_u32 v" #d ", v42, s41\n"
#define I_MUL_HI_U32(d) "v_mul_hi_u32 v" #d ", v42, s41\n"
#define I_MUL_U32_U24(d) "v_mul_u32_u24 v" #d ", v42, v40\n"
#define I_MAD_U64_U32(d) "v_mad_u64_u32 v[" STR(PAIR##d) "], s[44:45], v42, s41, v[48:49]\n"
#define I_LSHL_ADD_U64(d) "v_lshl_add_u64 v[" STR(PAIR##d) "], v[40:41], 0, v[42:43]\n"
#define I_PK_ADD_F32(d) "v_pk_add_f32 v[" STR(PAIR##d) "], v[40:41], v[42:43]\n"
#define I_PK_MUL_F32(d) "v_pk_mul_f32 v[" STR(PAIR##d) "], v[40:41], v[42:43]\n"
#define I_PK_FMA_F32(d) "v_pk_fma_f32 v[" STR(PAIR##d) "], v[40:41], v[42:43], v[44:45]\n"
#define I_ADD_F64(d) "v_add_f64 v[" STR(PAIR##d) "], v[40:41], v[42:43]\n"
#define I_MUL_F64(d) "v_mul_f64 v[" STR(PAIR##d) "], v[40:41], v[42:43]\n"
#define I_FMA_F64(d) "v_fma_f64 v[" STR(PAIR##d) "], v[40:41], v[42:43], v[44:45]\n"
#define I_CVT_F64_F32(d) "v_cvt_f64_f32 v[" STR(PAIR##d) "], v40\n"
#define I_CVT_F32_F64(d) "v_cvt_f32_f64 v" #d ", v[40:41]\n"
#define I_CVT_F32_U32(d) "v_cvt_f32_u32 v" #d ", v42\n"
#define I_EXP_F32(d) "v_exp_f32 v" #d ", v40\n"
#define I_RCP_F32(d) "v_rcp_f32 v" #d ", v40\n"
#define I_SQRT_F32(d) "v_sqrt_f32 v" #d ", v40\n"
#define I_RSQ_F64(d) "v_rsq_f64 v[" STR(PAIR##d) "], v[40:41]\n"
#define I_READLANE(d) "v_readlane_b32 s" STR(SREG##d) ", v40, s42\n"
#define I_READFIRST(d) "v_readfirstlane_b32 s" STR(SREG##d) ", v40\n"
#define I_DS_READ_B128(d) "ds_read_b128 v[" STR(QUAD##d) "], v49\n"
#define I_DS_READ_B64(d) "ds_read_b64 v[" STR(PAIR##d) "], v49\n"
#define I_DS_READ_B32(d) "ds_read_b32 v" #d ", v49\n"
#define I_S_MUL(d) "s_mul_i32 s" STR(SREG##d) ", s40, s41\n"
#define I_S_ADD(d) "s_add_u32 s" STR(SREG##d) ", s40, s41\n"
#define I_SUB_F32X(d) "v_sub_f32 v" #d ", v40, v41\n"
#define I_MIN_F32(d) "v_min_f32 v" #d ", v40, v41\n"
#define I_FMAC_F32(d) "v_fmac_f32 v" #d ", v40, v41\n"
#define I_FLOOR_F32(d) "v_floor_f32 v" #d ", v40\n"
#define I_CVT_I32_F32(d) "v_cvt_i32_f32 v" #d ", v40\n"
#define I_AND_B32X(d) "v_and_b32 v" #d ", v42, v40\n"
#define I_LSHL_B32(d) "v_lshlrev_b32 v" #d ", 3, v40\n"
#define I_BFE_U32(d) "v_bfe_u32 v" #d ", v42, 5, 7\n"
#define I_BFI_B32(d) "v_bfi_b32 v" #d ", v42, v40, v41\n"
#define I_LSHL_OR(d) "v_lshl_or_b32 v" #d ", v42, 8, v40\n"
#define I_AND_OR(d) "v_and_or_b32 v" #d ", v42, v40, v41\n"
#define I_OR3(d) "v_or3_b32 v" #d ", v42, v40, v41\n"
#define I_LSHL_ADD_U32(d) "v_lshl_add_u32 v" #d ", v42, 2, v40\n"
#define I_PERM(d) "v_perm_b32 v" #d ", v42, v40, v41\n"
#define I_BITOP3(d) "v_bitop3_b32 v" #d ", v42, v40, v41 bitop3:0xc6\n"
#define I_MOV_B64(d) "v_mov_b64 v[" STR(PAIR##d) "], v[40:41]\n"
#define I_FMAC_F64(d) "v_fmac_f64 v[" STR(PAIR##d) "], v[40:41], v[42:43]\n"
#define I_LDEXP_F64(d) "v_ldexp_f64 v[" STR(PAIR##d) "], v[40:41], v42\n"
#define I_CMP_E64(d) "v_cmp_gt_f32 s[" STR(SPAIR##d) "], v40, v41\n"
#define I_CMP_U32(d) "v_cmp_lt_u32 vcc, v42, v40\n"
#define I_CNDMASK_E64(d) "v_cndmask_b32 v" #d ", v40, v41, s[46:47]\n"
#define I_MBCNT(d) "v_mbcnt_lo_u32_b32 v" #d ", s46, v42\n"
#define I_MUL_U64(d) "v_mul_lo_u32 v" #d ", v42, v40\n"
#define I_ADD_CO(d) "v_add_co_u32 v" #d ", vcc, v42, v40\n"
#define I_ADDC_CO(d) "v_addc_co_u32 v" #d ", vcc, v42, v40, vcc\n"
#define I_S_AND_B64(d) "s_and_b64 s[" STR(SPAIR##d) "], s[40:41], s[40:41]\n"
#define I_S_FF1(d) "s_ff1_i32_b64 s" STR(SREG##d) ", s[40:41]\n"
#define I_DS_WRITE_B32(d) "ds_write_b32 v49, v40\n"
#define SPAIR0 44:45
#define SPAIR1 46:47
#define SPAIR2 48:49
#define SPAIR3 50:51
#define SPAIR4 44:45
#define SPAIR5 46:47
#define SPAIR6 48:49
#define SPAIR7 50:51
#define PAIR0 0:1
#define PAIR1 2:3
#define PAIR2 4:5
#define PAIR3 6:7
#define PAIR4 8:9
#define PAIR5 10:11
#define PAIR6 12:13
#define PAIR7 14:15
#define QUAD0 0:3
#define QUAD1 4:7
#define QUAD2 8:11
#define QUAD3 12:15
#define QUAD4 0:3
#define QUAD5 4:7
#define QUAD6 8:11
#define QUAD7 12:15
#define SREG0 44
#define SREG1 45
#define SREG2 46
#define SREG3 47
#define SREG4 48
#define SREG5 49
#define SREG6 50
#define SREG7 51

KERNEL(k_add_f32, CLOB32, X32(I_ADD_F32))
KERNEL(k_mul_f32, CLOB32, X32(I_MUL_F32))
KERNEL(k_fma_f32, CLOB32, X32(I_FMA_F32))
KERNEL(k_max_f32, CLOB32, X32(I_MAX_F32))
KERNEL(k_cmp_f32, CLOB32, X32(I_CMP_F32))
KERNEL(k_cndmask, CLOB32, X32(I_CNDMASK))
KERNEL(k_mov_b32, CLOB32, X32(I_MOV_B32))
KERNEL(k_add_u32, CLOB32, X32(I_ADD_U32))
KERNEL(k_add3_u32, CLOB32, X32(I_ADD3_U32))
KERNEL(k_xor_b32, CLOB32, X32(I_XOR_B32))
KERNEL(k_or_b32, CLOB32, X32(I_OR_B32))
KERNEL(k_lshr_b32, CLOB32, X32(I_LSHR_B32))
KERNEL(k_alignbit, CLOB32, X32(I_ALIGNBIT))
KERNEL(k_alignbit_v, CLOB32, X32(I_ALIGNBIT_V))
KERNEL(k_mul_lo_u32, CLOB32, X32(I_MUL_LO_U32))
KERNEL(k_mul_hi_u32, CLOB32, X32(I_MUL_HI_U32))
KERNEL(k_mul_u32_u24, CLOB32, X32(I_MUL_U32_U24))
KERNEL(k_mad_u64_u32, CLOB32, X32(I_MAD_U64_U32))
KERNEL(k_lshl_add_u64, CLOB32, X32(I_LSHL_ADD_U64))
KERNEL(k_pk_add_f32, CLOB32, X32(I_PK_ADD_F32))
KERNEL(k_pk_mul_f32, CLOB32, X32(I_PK_MUL_F32))
KERNEL(k_pk_fma_f32, CLOB32, X32(I_PK_FMA_F32))
KERNEL(k_add_f64, CLOB32, X32(I_ADD_F64))
KERNEL(k_mul_f64, CLOB32, X32(I_MUL_F64))
KERNEL(k_fma_f64, CLOB32, X32(I_FMA_F64))
KERNEL(k_cvt_f64_f32, CLOB32, X32(I_CVT_F64_F32))
KERNEL(k_cvt_f32_f64, CLOB32, X32(I_CVT_F32_F64))
KERNEL(k_cvt_f32_u32, CLOB32, X32(I_CVT_F32_U32))
KERNEL(k_exp_f32, CLOB32, X32(I_EXP_F32))
KERNEL(k_rcp_f32, CLOB32, X32(I_RCP_F32))
KERNEL(k_sqrt_f32, CLOB32, X32(I_SQRT_F32))
KERNEL(k_rsq_f64, CLOB32, X32(I_RSQ_F64))
KERNEL(k_readlane, CLOB32, X32(I_READLANE))
KERNEL(k_readfirstlane, CLOB32, X32(I_READFIRST))
KERNEL(k_ds_read_b128, CLOB32, X32(I_DS_READ_B128) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ds_read_b64, CLOB32, X32(I_DS_READ_B64) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ds_read_b32, CLOB32, X32(I_DS_READ_B32) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_sub_f32, CLOB32, X32(I_SUB_F32X))
KERNEL(k_min_f32, CLOB32, X32(I_MIN_F32))
KERNEL(k_fmac_f32, CLOB32, X32(I_FMAC_F32))
KERNEL(k_floor_f32, CLOB32, X32(I_FLOOR_F32))
KERNEL(k_cvt_i32_f32, CLOB32, X32(I_CVT_I32_F32))
KERNEL(k_and_b32, CLOB32, X32(I_AND_B32X))
KERNEL(k_lshl_b32, CLOB32, X32(I_LSHL_B32))
KERNEL(k_bfe_u32, CLOB32, X32(I_BFE_U32))
KERNEL(k_bfi_b32, CLOB32, X32(I_BFI_B32))
KERNEL(k_lshl_or, CLOB32, X32(I_LSHL_OR))
KERNEL(k_and_or, CLOB32, X32(I_AND_OR))
KERNEL(k_or3, CLOB32, X32(I_OR3))
KERNEL(k_lshl_add_u32, CLOB32, X32(I_LSHL_ADD_U32))
KERNEL(k_perm, CLOB32, X32(I_PERM))
KERNEL(k_bitop3, CLOB32, X32(I_BITOP3))
KERNEL(k_mov_b64, CLOB32, X32(I_MOV_B64))
KERNEL(k_fmac_f64, CLOB32, X32(I_FMAC_F64))
KERNEL(k_ldexp_f64, CLOB32, X32(I_LDEXP_F64))
KERNEL(k_cmp_e64, CLOB32, X32(I_CMP_E64))
KERNEL(k_cmp_u32, CLOB32, X32(I_CMP_U32))
KERNEL(k_cndmask_e64, CLOB32, X32(I_CNDMASK_E64))
KERNEL(k_mbcnt, CLOB32, X32(I_MBCNT))
KERNEL(k_add_co, CLOB32, X32(I_ADD_CO))
KERNEL(k_addc_co, CLOB32, X32(I_ADDC_CO))
KERNEL(k_s_and_b64, CLOB32, X32(I_S_AND_B64))
KERNEL(k_s_ff1, CLOB32, X32(I_S_FF1))
KERNEL(k_ds_write_b32, CLOB32, X32(I_DS_WRITE_B32) "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_s_mul_i32, CLOB32, X32(I_S_MUL))
KERNEL(k_s_add_u32, CLOB32, X32(I_S_ADD))

typedef void (*kern_t)(unsigned long long *, float, float, int);
struct Entry { const char *name; kern_t fn; };

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
    // optional: index of the first entry and how many to run (the session script runs entries one by one under a
    // timeout so that one bad instruction form cannot take the table down)
    const int first = argc > 1 ? atoi(argv[1]) : 0;
    const int count = argc > 2 ? atoi(argv[2]) : 1000;
    const Entry entries[] = {
        {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_max_f32", k_max_f32},
        {"v_cmp_lt_f32", k_cmp_f32}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b32", k_mov_b32},
        {"v_add_u32", k_add_u32}, {"v_add3_u32", k_add3_u32}, {"v_xor_b32", k_xor_b32}, {"v_or_b32", k_or_b32},
        {"v_lshrrev_b32", k_lshr_b32}, {"v_alignbit_b32(imm)", k_alignbit}, {"v_alignbit_b32(vgpr)", k_alignbit_v},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64},
        {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f32_u32", k_cvt_f32_u32},
        {"v_exp_f32", k_exp_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rsq_f64", k_rsq_f64},
        {"v_readlane_b32", k_readlane}, {"v_readfirstlane_b32", k_readfirstlane},
        {"ds_read_b128(broadcast)", k_ds_read_b128}, {"ds_read_b64(broadcast)", k_ds_read_b64}, {"ds_read_b32(broadcast)", k_ds_read_b32},
        {"s_mul_i32", k_s_mul_i32}, {"s_add_u32", k_s_add_u32},
        {"v_sub_f32", k_sub_f32}, {"v_min_f32", k_min_f32}, {"v_fmac_f32", k_fmac_f32}, {"v_floor_f32", k_floor_f32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
        {"v_and_b32", k_and_b32}, {"v_lshlrev_b32", k_lshl_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_bfi_b32", k_bfi_b32}, {"v_lshl_or_b32", k_lshl_or},
        {"v_and_or_b32", k_and_or}, {"v_or3_b32", k_or3}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_perm_b32", k_perm}, {"v_bitop3_b32", k_bitop3},
        {"v_mov_b64", k_mov_b64}, {"v_fmac_f64", k_fmac_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_cmp_gt_f32(sgpr pair)", k_cmp_e64},
        {"v_cmp_lt_u32", k_cmp_u32}, {"v_cndmask_b32(sgpr pair)", k_cndmask_e64}, {"v_mbcnt_lo_u32_b32", k_mbcnt},
        {"v_add_co_u32", k_add_co}, {"v_addc_co_u32", k_addc_co}, {"s_and_b64", k_s_and_b64}, {"s_ff1_i32_b64", k_s_ff1},
        {"ds_write_b32", k_ds_write_b32},
    };
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int waves_per_simd[] = {1, 2, 4, 8};
    unsigned long long *d_out;
    const size_t max_waves = (size_t)cus * 4 * 8;
    CK(hipMalloc(&d_out, max_waves * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(max_waves);
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    float wall_ms[4] = {0, 0, 0, 0};
    const int ne = (int)(sizeof entries / sizeof entries[0]);
    if (first == 0 && count >= ne)
        printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"reps\": %d, \"block\": 32, \"unit\": \"SIMD cycles per wave64 instruction (median over waves of elapsed s_memtime / (instructions per wave * waves per SIMD))\",\n \"cycles\": {\n",
               prop.gcnArchName, cus, prop.clockRate, REPS);
    for (int e = first; e < ne && e < first + count; ++e) {
        printf("  \"%s\": {", entries[e].name);
        fflush(stdout);
        for (int wi = 0; wi < 4; ++wi) {
            const int W = waves_per_simd[wi];
            const int blocks = cus * W;
            hipLaunchKernelGGL(entries[e].fn, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f, 0.75f, 12345);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(ev0, 0));
            hipLaunchKernelGGL(entries[e].fn, dim3(blocks), dim3(256), 0, 0, d_out, 1.25f, 0.75f, 12345);
            CK(hipEventRecord(ev1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, ev0, ev1));
            wall_ms[wi] = ms;
            CK(hipMemcpy(h.data(), d_out, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v(h.begin(), h.begin() + (size_t)blocks * 4);
            std::sort(v.begin(), v.end());
            const double med = (double)v[v.size() / 2];
            printf("\"%d\": %.3f, ", W, med / ((double)REPS * 32.0 * W));
        }
        // cross-check that does not rely on all W waves being co-resident: whole-kernel wall time (launch overhead
        // included, ~10 us) as SIMD cycles at the device's peak clock per instruction and SIMD
        printf("\"wall_ms\": [%.4f, %.4f, %.4f, %.4f], \"wall_cycles_at_peak_clock\": [", wall_ms[0], wall_ms[1], wall_ms[2], wall_ms[3]);
        for (int wi = 0; wi < 4; ++wi)
            printf("%.3f%s", wall_ms[wi] * 1e-3 * (double)prop.clockRate * 1e3 / ((double)REPS * 32.0 * waves_per_simd[wi]), wi < 3 ? ", " : "]");
        printf("}%s\n", e + 1 < ne ? "," : "");
        fflush(stdout);
    }
    if (first == 0 && count >= ne)
        printf(" }\n}\n");
    CK(hipFree(d_out));
    return 0;
}
