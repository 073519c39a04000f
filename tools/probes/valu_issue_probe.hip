// What does one SIMD of gfx950 sustain per cycle for the vector instructions the render kernel is made of?
// The render kernel's roofline (bench.py) prices its instruction stream in SIMD issue cycles; the guide
// (MI355X_MICROARCH.md) gives 4 cycles per wave64 VALU instruction for ONE wave alone and 2 as the SIMD-32 rate, and the kernel
// runs 4 waves per SIMD -- so the price has to be measured at that occupancy.  This probe runs streams of independent
//   v_fma_f32 | v_pk_fma_f32 | v_exp_f32 | v_add_u32 | v_cndmask_b32 | v_mov_b32 dpp | v_cvt_pk_f16_f32 | ds_read_b128
//   v_mfma_f32_16x16x32_f16 alone and with K = 1..6 v_fma_f32 fillers per MFMA, v_mfma_f32_16x16x4_f32 alone,
//   and the render kernel's own mix (2 transcendentals per 8 plain per 1 f16 MFMA)
// at 1, 2 and 4 waves per SIMD (one 256/512/1024-lane workgroup per CU, 100 KB of LDS keeps a second one away), every wave
// timing its own loop with s_memtime (tick = shader cycle).  Reported per (stream, waves/SIMD):
//   wave_cyc_per_inst  = a wave's cycles / its instructions           (what the wave sees)
//   simd_cyc_per_inst  = slowest wave of the SIMD / waves on that SIMD (what the SIMD sustains: the price the roofline needs)
// plus the SIMD each wave landed on (HW_ID) so that the waves-per-SIMD assumption is checked, not assumed.
// build + run: tools/probes/valu_issue_probe.sh  ->  gpurun_out/valu_issue_probe.json  (committed as profiles/r03_valu_issue_probe.json)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>
#include <map>

#define CLOB_V "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", \
               "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", \
               "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57"

// one instruction per destination register v10..v41 (32 independent chains; each depends only on its own previous value)
#define X32(OP) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15) OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) OP(24) OP(25) \
                OP(26) OP(27) OP(28) OP(29) OP(30) OP(31) OP(32) OP(33) OP(34) OP(35) OP(36) OP(37) OP(38) OP(39) OP(40) OP(41)
#define I_FMA(n)   "v_fma_f32 v" #n ", v" #n ", v42, v43\n\t"
#define I_EXP(n)   "v_exp_f32 v" #n ", v" #n "\n\t"
#define I_ZERO(n)  "v_mov_b32 v" #n ", 0\n\t"
#define I_ADDU(n)  "v_add_u32 v" #n ", v" #n ", v42\n\t"
#define I_CND(n)   "v_cndmask_b32 v" #n ", v" #n ", v42, vcc\n\t"
#define I_DPP(n)   "v_mov_b32_dpp v" #n ", v" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_CVT(n)   "v_cvt_pk_f16_f32 v" #n ", v" #n ", v42\n\t"
#define I_MAX(n)   "v_max_f32 v" #n ", v" #n ", v42\n\t"
#define I_MUL24(n) "v_mul_u32_u24 v" #n ", v" #n ", v42\n\t"
#define I_MULF(n)  "v_mul_f32 v" #n ", v" #n ", v42\n\t"
#define I_ADDF(n)  "v_add_f32 v" #n ", v" #n ", v42\n\t"
#define I_SUBF(n)  "v_sub_f32 v" #n ", v" #n ", v42\n\t"
#define I_FMAC(n)  "v_fmac_f32 v" #n ", v42, v43\n\t"
#define I_MOV(n)   "v_mov_b32 v" #n ", v42\n\t"
#define I_MAXI(n)  "v_max_i32 v" #n ", v" #n ", v42\n\t"
#define I_MINI(n)  "v_min_i32 v" #n ", v" #n ", v42\n\t"
#define I_MAXU(n)  "v_max_u32 v" #n ", v" #n ", v42\n\t"
#define I_MAX3(n)  "v_max3_f32 v" #n ", v" #n ", v42, v43\n\t"
#define I_MED3(n)  "v_med3_f32 v" #n ", v" #n ", v42, v43\n\t"
#define I_AND(n)   "v_and_b32 v" #n ", v" #n ", v42\n\t"
#define I_LSHL(n)  "v_lshlrev_b32 v" #n ", 2, v" #n "\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 v" #n ", v" #n ", 2, v42\n\t"
#define I_MAD24(n) "v_mad_u32_u24 v" #n ", v" #n ", v42, v43\n\t"
#define I_FLOOR(n) "v_floor_f32 v" #n ", v" #n "\n\t"
#define I_CVTI(n)  "v_cvt_i32_f32 v" #n ", v" #n "\n\t"
#define I_CMP(n)   "v_cmp_lt_f32 vcc, v" #n ", v42\n\t"
#define I_CMP64(n) "v_cmp_lt_f32_e64 s[30:31], v" #n ", v42\n\t"
#define I_CND64(n) "v_cndmask_b32_e64 v" #n ", v" #n ", v42, s[30:31]\n\t"
#define I_ADDC(n)  "v_addc_co_u32 v" #n ", vcc, v" #n ", v42, vcc\n\t"
#define I_CMPADDC(n) "v_cmp_lt_f32 vcc, v42, v43\n\tv_addc_co_u32 v" #n ", vcc, 0, v" #n ", vcc\n\t"
#define I_CMPCND(n) "v_cmp_lt_f32 vcc, v" #n ", v43\n\tv_cndmask_b32 v" #n ", v" #n ", v42, vcc\n\t"
#define I_LOG(n)   "v_log_f32 v" #n ", v" #n "\n\t"
#define I_RCP(n)   "v_rcp_f32 v" #n ", v" #n "\n\t"
#define I_FMAMIX(n) "v_fma_mixlo_f16 v" #n ", v" #n ", -1.0, v42 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
#define I_DPPADD(n) "v_add_f32_dpp v" #n ", v" #n ", v" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_READLANE(n) "v_readlane_b32 s30, v" #n ", 3\n\t"
#define I_READFIRST(n) "v_readfirstlane_b32 s30, v" #n "\n\t"
#define I_MULLO(n) "v_mul_lo_u32 v" #n ", v" #n ", v42\n\t"
#define I_DOT2(n)  "v_dot2_f32_f16 v" #n ", v42, v43, v" #n "\n\t"
#define I_CVTF16(n) "v_cvt_f32_f16 v" #n ", v" #n "\n\t"
#define I_FMAMIX32(n) "v_fma_mix_f32 v" #n ", v" #n ", v42, v43 op_sel_hi:[1,0,0]\n\t"
#define I_PKFMA16(n) "v_pk_fma_f16 v" #n ", v" #n ", v42, v43\n\t"
#define I_SUBU(n)  "v_sub_u32 v" #n ", v" #n ", v42\n\t"
#define I_OR(n)    "v_or_b32 v" #n ", v" #n ", v42\n\t"
#define I_XOR(n)   "v_xor_b32 v" #n ", v" #n ", v42\n\t"
#define I_ADD3(n)  "v_add3_u32 v" #n ", v" #n ", v42, v43\n\t"
#define I_LSHLOR(n) "v_lshl_or_b32 v" #n ", v" #n ", 2, v42\n\t"
#define I_BFE(n)   "v_bfe_u32 v" #n ", v" #n ", 3, 5\n\t"
#define I_CVTFI(n) "v_cvt_f32_i32 v" #n ", v" #n "\n\t"
#define I_CVTUF(n) "v_cvt_u32_f32 v" #n ", v" #n "\n\t"
#define I_LDEXP(n) "v_ldexp_f32 v" #n ", v" #n ", v42\n\t"
#define I_FRACT(n) "v_fract_f32 v" #n ", v" #n "\n\t"
#define I_CMPI(n)  "v_cmp_lt_i32 vcc, v" #n ", v42\n\t"
#define I_PERMB(n) "v_perm_b32 v" #n ", v" #n ", v42, v43\n\t"
#define I_ALIGN(n) "v_alignbit_b32 v" #n ", v" #n ", v42, 16\n\t"
#define I_MADI24(n) "v_mad_i32_i24 v" #n ", v" #n ", v42, v43\n\t"
#define I_ASHR(n)  "v_ashrrev_i32 v" #n ", 3, v" #n "\n\t"
#define I_LSHR(n)  "v_lshrrev_b32 v" #n ", 3, v" #n "\n\t"
#define I_ADDNEG(n) "v_add_f32_e64 v" #n ", -|v" #n "|, v42\n\t"
#define I_MIN3(n)  "v_min3_f32 v" #n ", v" #n ", v42, v43\n\t"
#define I_FMAK(n)  "v_fma_f32 v" #n ", v" #n ", 0.5, v43\n\t"
#define I_MADAK(n) "v_fmaak_f32 v" #n ", v" #n ", v42, 0x3f9d70a4\n\t"
#define I_SUBREV(n) "v_subrev_f32 v" #n ", v42, v" #n "\n\t"
#define I_EXPMOD(n) "v_exp_f32_e64 v" #n ", -|v" #n "|\n\t"
#define I_WRITELANE(n) "v_writelane_b32 v" #n ", s20, 3\n\t"
#define I_PERM16(a, b) "v_permlane16_swap_b32 v" #a ", v" #b "\n\t"
#define I_PKMUL(a, b) "v_pk_mul_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[42:43]\n\t"
// packed: 16 register pairs
#define X16P(OP) OP(10, 11) OP(12, 13) OP(14, 15) OP(16, 17) OP(18, 19) OP(20, 21) OP(22, 23) OP(24, 25) \
                 OP(26, 27) OP(28, 29) OP(30, 31) OP(32, 33) OP(34, 35) OP(36, 37) OP(38, 39) OP(40, 41)
#define I_PKFMA(a, b) "v_pk_fma_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[42:43], v[44:45]\n\t"
#define I_PKADD(a, b) "v_pk_add_f32 v[" #a ":" #b "], v[" #a ":" #b "], v[42:43]\n\t"
// matrix: 8 independent accumulators of 4 registers, A = v[42:45], B = v[46:49]
#define X8M(OP) OP(10, 13) OP(14, 17) OP(18, 21) OP(22, 25) OP(26, 29) OP(30, 33) OP(34, 37) OP(38, 41)
#define I_MFMA16(a, b) "v_mfma_f32_16x16x32_f16 v[" #a ":" #b "], v[42:45], v[46:49], v[" #a ":" #b "]\n\t"
#define I_MFMA32(a, b) "v_mfma_f32_16x16x4_f32 v[" #a ":" #b "], v42, v46, v[" #a ":" #b "]\n\t"
#define I_MFMA16K16(a, b) "v_mfma_f32_16x16x16_f16 v[" #a ":" #b "], v[42:43], v[46:47], v[" #a ":" #b "]\n\t"
#define I_MULHI(r) "v_mul_hi_u32 v" #r ", v" #r ", v42\n\t"
// fillers beside the MFMAs write v50..v57 (independent of the accumulators)
#define F1 "v_fma_f32 v50, v50, v42, v43\n\t"
#define F2 F1 "v_fma_f32 v51, v51, v42, v43\n\t"
#define F3 F2 "v_fma_f32 v52, v52, v42, v43\n\t"
#define F4 F3 "v_fma_f32 v53, v53, v42, v43\n\t"
#define F6 F4 "v_fma_f32 v54, v54, v42, v43\n\t" "v_fma_f32 v55, v55, v42, v43\n\t"
#define F8 F6 "v_fma_f32 v56, v56, v42, v43\n\t" "v_fma_f32 v57, v57, v42, v43\n\t"
#define E1 "v_exp_f32 v50, v50\n\t"
#define E2 E1 "v_exp_f32 v51, v51\n\t"

#define TIMED_LOOP(BODY)                                                                                         \
    asm volatile("s_mov_b32 s20, %2\n\t"                                                                         \
                 "s_barrier\n\t"                                                                                 \
                 "s_memtime %0\n\t"                                                                              \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                      \
                 "1:\n\t" BODY                                                                                   \
                 "s_sub_u32 s20, s20, 1\n\t"                                                                     \
                 "s_cmp_lg_u32 s20, 0\n\t"                                                                       \
                 "s_cbranch_scc1 1b\n\t"                                                                         \
                 "s_nop 7\n\t"                                                                                   \
                 "s_memtime %1\n\t"                                                                              \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                      \
                 : "=s"(t0), "=s"(t1) : "s"(iters) : "s20", "s30", "s31", "scc", "vcc", "memory", "v58", CLOB_V)

struct Stream { const char* name; int insts_per_iter; const char* what; };
static const Stream kStreams[] = {
    {"v_fma_f32", 32, "32 independent v_fma_f32"},
    {"v_pk_fma_f32", 16, "16 independent v_pk_fma_f32 (2 FMAs per lane each)"},
    {"v_exp_f32", 32, "32 independent v_exp_f32"},
    {"v_add_u32", 32, "32 independent v_add_u32"},
    {"v_cndmask_b32", 32, "32 independent v_cndmask_b32"},
    {"v_mov_b32_dpp", 32, "32 independent v_mov_b32 row_shr:1"},
    {"v_cvt_pk_f16_f32", 32, "32 independent v_cvt_pk_f16_f32"},
    {"v_max_f32", 32, "32 independent v_max_f32"},
    {"v_mul_u32_u24", 32, "32 independent v_mul_u32_u24"},
    {"v_pk_add_f32", 16, "16 independent v_pk_add_f32"},
    {"mfma_f16_16x16x32", 8, "8 independent v_mfma_f32_16x16x32_f16"},
    {"mfma_f16+1fma", 16, "per MFMA: 1 v_fma_f32 filler"},
    {"mfma_f16+2fma", 24, "per MFMA: 2 v_fma_f32 fillers"},
    {"mfma_f16+3fma", 32, "per MFMA: 3 v_fma_f32 fillers"},
    {"mfma_f16+4fma", 40, "per MFMA: 4 v_fma_f32 fillers"},
    {"mfma_f16+6fma", 56, "per MFMA: 6 v_fma_f32 fillers"},
    {"mfma_f16+8fma", 72, "per MFMA: 8 v_fma_f32 fillers"},
    {"mfma_f32_16x16x4", 8, "8 independent v_mfma_f32_16x16x4_f32"},
    {"render_mix", 88, "per f16 MFMA: 8 v_fma_f32 + 2 v_exp_f32 (the render kernel's executed ratio: ~136 plain + 21 transcendental per 6 MFMA ... scaled)"},
    {"fma_exp_4to1", 40, "32 v_fma_f32 + 8 v_exp_f32 interleaved 4:1"},
    {"ds_read_b128", 16, "16 ds_read_b128, lane i at 16 i bytes, one wait per iteration"},
    {"v_mul_f32", 32, ""}, {"v_add_f32", 32, ""}, {"v_sub_f32", 32, ""}, {"v_fmac_f32", 32, ""}, {"v_mov_b32", 32, ""},
    {"v_max_i32", 32, ""}, {"v_min_i32", 32, ""}, {"v_max_u32", 32, ""}, {"v_max3_f32", 32, ""}, {"v_med3_f32", 32, ""},
    {"v_and_b32", 32, ""}, {"v_lshlrev_b32", 32, ""}, {"v_lshl_add_u32", 32, ""}, {"v_mad_u32_u24", 32, ""}, {"v_floor_f32", 32, ""},
    {"v_cvt_i32_f32", 32, ""}, {"v_cmp_lt_f32_vcc", 32, "32 v_cmp_lt_f32 writing vcc"}, {"v_cmp_lt_f32_sgpr", 32, "32 v_cmp_lt_f32_e64 writing s[30:31]"},
    {"v_cndmask_b32_sgpr", 32, "32 v_cndmask_b32_e64 selecting on s[30:31] (written once before the loop)"},
    {"v_addc_co_u32", 32, "32 v_addc_co_u32 chained through vcc"},
    {"cmp+addc", 64, "the counting-scan idiom: v_cmp_lt_f32 vcc + v_addc_co_u32 (count as 2 instructions)"},
    {"cmp+cndmask", 64, "v_cmp_lt_f32 vcc + v_cndmask_b32 on it (count as 2 instructions)"},
    {"v_log_f32", 32, ""}, {"v_rcp_f32", 32, ""}, {"v_fma_mixlo_f16", 32, ""}, {"v_add_f32_dpp", 32, "v_add_f32 row_shr:1 (the scan step)"},
    {"v_readlane_b32", 32, ""}, {"v_readfirstlane_b32", 32, ""}, {"v_mul_lo_u32", 32, ""},
    {"v_permlane16_swap", 16, ""}, {"v_pk_mul_f32", 16, ""},
    {"v_dot2_f32_f16", 32, ""}, {"v_cvt_f32_f16", 32, ""}, {"v_fma_mix_f32", 32, ""}, {"v_pk_fma_f16", 32, ""}, {"v_sub_u32", 32, ""},
    {"v_or_b32", 32, ""}, {"v_xor_b32", 32, ""}, {"v_add3_u32", 32, ""}, {"v_lshl_or_b32", 32, ""}, {"v_bfe_u32", 32, ""},
    {"v_cvt_f32_i32", 32, ""}, {"v_cvt_u32_f32", 32, ""}, {"v_ldexp_f32", 32, ""}, {"v_fract_f32", 32, ""}, {"v_cmp_lt_i32", 32, ""},
    {"v_perm_b32", 32, ""}, {"v_alignbit_b32", 32, ""}, {"v_mad_i32_i24", 32, ""}, {"v_ashrrev_i32", 32, ""}, {"v_lshrrev_b32", 32, ""},
    {"v_add_f32_negabs", 32, "v_add_f32_e64 with neg+abs source modifiers"}, {"v_min3_f32", 32, ""}, {"v_fma_f32_const", 32, "v_fma_f32 with an inline constant"},
    {"v_fmaak_f32", 32, "v_fmaak_f32 (32-bit literal)"}, {"v_subrev_f32", 32, ""}, {"v_exp_f32_negabs", 32, "v_exp_f32_e64 with neg+abs"}, {"v_writelane_b32", 32, ""},
    {"ds_read_b128_bcast", 16, "16 ds_read_b128, all lanes the same address (the key scans)"},
    {"ds_read_b32", 16, "16 ds_read_b32, lane i at 4 i bytes"},
    {"ds_write_b128", 16, "16 ds_write_b128, lane i at 16 i bytes"},
    {"mfma_f16_16x16x16", 8, "8 independent v_mfma_f32_16x16x16_f16 (the K = 16 form: 4 halves per lane and operand)"},
    {"v_mul_hi_u32", 32, "32 independent v_mul_hi_u32 (Philox rounds)"},
};
constexpr int kNumStreams = sizeof(kStreams) / sizeof(kStreams[0]);

#define DS16(OP, A) OP " v[10:13], " A "\n\t" OP " v[14:17], " A " offset:1024\n\t" OP " v[18:21], " A " offset:2048\n\t" OP " v[22:25], " A " offset:3072\n\t" \
                    OP " v[26:29], " A " offset:4096\n\t" OP " v[30:33], " A " offset:5120\n\t" OP " v[34:37], " A " offset:6144\n\t" OP " v[38:41], " A " offset:7168\n\t" \
                    OP " v[10:13], " A " offset:8192\n\t" OP " v[14:17], " A " offset:9216\n\t" OP " v[18:21], " A " offset:10240\n\t" OP " v[22:25], " A " offset:11264\n\t" \
                    OP " v[26:29], " A " offset:12288\n\t" OP " v[30:33], " A " offset:13312\n\t" OP " v[34:37], " A " offset:14336\n\t" OP " v[38:41], " A " offset:15360\n\t" \
                    "s_waitcnt lgkmcnt(0)\n\t"
#define PER_MFMA(FILL) I_MFMA16(10, 13) FILL I_MFMA16(14, 17) FILL I_MFMA16(18, 21) FILL I_MFMA16(22, 25) FILL \
                       I_MFMA16(26, 29) FILL I_MFMA16(30, 33) FILL I_MFMA16(34, 37) FILL I_MFMA16(38, 41) FILL

__global__ __launch_bounds__(1024) void probe(int stream, int iters, unsigned long long* cyc, unsigned* hwid) {
    extern __shared__ __align__(16) float lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    // benign operand values: everything stays finite (fma multiplies by 0, exp of 0 ...)
    asm volatile("v_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\t"
                 "v_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
                 "v_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t"
                 "v_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\t"
                 X32(I_ZERO) ::: CLOB_V);
    switch (stream) {
        case 0: TIMED_LOOP(X32(I_FMA)); break;
        case 1: TIMED_LOOP(X16P(I_PKFMA)); break;
        case 2: TIMED_LOOP(X32(I_EXP)); break;
        case 3: TIMED_LOOP(X32(I_ADDU)); break;
        case 4: TIMED_LOOP(X32(I_CND)); break;
        case 5: TIMED_LOOP(X32(I_DPP)); break;
        case 6: TIMED_LOOP(X32(I_CVT)); break;
        case 7: TIMED_LOOP(X32(I_MAX)); break;
        case 8: TIMED_LOOP(X32(I_MUL24)); break;
        case 9: TIMED_LOOP(X16P(I_PKADD)); break;
        case 10: TIMED_LOOP(X8M(I_MFMA16)); break;
        case 11: TIMED_LOOP(PER_MFMA(F1)); break;
        case 12: TIMED_LOOP(PER_MFMA(F2)); break;
        case 13: TIMED_LOOP(PER_MFMA(F3)); break;
        case 14: TIMED_LOOP(PER_MFMA(F4)); break;
        case 15: TIMED_LOOP(PER_MFMA(F6)); break;
        case 16: TIMED_LOOP(PER_MFMA(F8)); break;
        case 17: TIMED_LOOP(X8M(I_MFMA32)); break;
        case 18: TIMED_LOOP(PER_MFMA(F8 E2)); break;
        case 19: TIMED_LOOP(I_FMA(10) I_FMA(11) I_FMA(12) I_FMA(13) I_EXP(50) I_FMA(14) I_FMA(15) I_FMA(16) I_FMA(17) I_EXP(51)
                            I_FMA(18) I_FMA(19) I_FMA(20) I_FMA(21) I_EXP(52) I_FMA(22) I_FMA(23) I_FMA(24) I_FMA(25) I_EXP(53)
                            I_FMA(26) I_FMA(27) I_FMA(28) I_FMA(29) I_EXP(54) I_FMA(30) I_FMA(31) I_FMA(32) I_FMA(33) I_EXP(55)
                            I_FMA(34) I_FMA(35) I_FMA(36) I_FMA(37) I_EXP(56) I_FMA(38) I_FMA(39) I_FMA(40) I_FMA(41) I_EXP(57)); break;
        case 20: {
            const unsigned addr = (threadIdx.x & 63) * 16;
            asm volatile("v_mov_b32 v58, %0" :: "v"(addr) : "v58");
            TIMED_LOOP(DS16("ds_read_b128", "v58"));
        } break;
        case 21: TIMED_LOOP(X32(I_MULF)); break;
        case 22: TIMED_LOOP(X32(I_ADDF)); break;
        case 23: TIMED_LOOP(X32(I_SUBF)); break;
        case 24: TIMED_LOOP(X32(I_FMAC)); break;
        case 25: TIMED_LOOP(X32(I_MOV)); break;
        case 26: TIMED_LOOP(X32(I_MAXI)); break;
        case 27: TIMED_LOOP(X32(I_MINI)); break;
        case 28: TIMED_LOOP(X32(I_MAXU)); break;
        case 29: TIMED_LOOP(X32(I_MAX3)); break;
        case 30: TIMED_LOOP(X32(I_MED3)); break;
        case 31: TIMED_LOOP(X32(I_AND)); break;
        case 32: TIMED_LOOP(X32(I_LSHL)); break;
        case 33: TIMED_LOOP(X32(I_LSHLADD)); break;
        case 34: TIMED_LOOP(X32(I_MAD24)); break;
        case 35: TIMED_LOOP(X32(I_FLOOR)); break;
        case 36: TIMED_LOOP(X32(I_CVTI)); break;
        case 37: TIMED_LOOP(X32(I_CMP)); break;
        case 38: TIMED_LOOP(X32(I_CMP64)); break;
        case 39: asm volatile("s_mov_b64 s[30:31], 0x5555" ::: "s30", "s31"); TIMED_LOOP(X32(I_CND64)); break;
        case 40: TIMED_LOOP(X32(I_ADDC)); break;
        case 41: TIMED_LOOP(X32(I_CMPADDC)); break;
        case 42: TIMED_LOOP(X32(I_CMPCND)); break;
        case 43: TIMED_LOOP(X32(I_LOG)); break;
        case 44: TIMED_LOOP(X32(I_RCP)); break;
        case 45: TIMED_LOOP(X32(I_FMAMIX)); break;
        case 46: TIMED_LOOP(X32(I_DPPADD)); break;
        case 47: TIMED_LOOP(X32(I_READLANE)); break;
        case 48: TIMED_LOOP(X32(I_READFIRST)); break;
        case 49: TIMED_LOOP(X32(I_MULLO)); break;
        case 50: TIMED_LOOP(X16P(I_PERM16)); break;
        case 51: TIMED_LOOP(X16P(I_PKMUL)); break;
        case 52: TIMED_LOOP(X32(I_DOT2)); break;
        case 53: TIMED_LOOP(X32(I_CVTF16)); break;
        case 54: TIMED_LOOP(X32(I_FMAMIX32)); break;
        case 55: TIMED_LOOP(X32(I_PKFMA16)); break;
        case 56: TIMED_LOOP(X32(I_SUBU)); break;
        case 57: TIMED_LOOP(X32(I_OR)); break;
        case 58: TIMED_LOOP(X32(I_XOR)); break;
        case 59: TIMED_LOOP(X32(I_ADD3)); break;
        case 60: TIMED_LOOP(X32(I_LSHLOR)); break;
        case 61: TIMED_LOOP(X32(I_BFE)); break;
        case 62: TIMED_LOOP(X32(I_CVTFI)); break;
        case 63: TIMED_LOOP(X32(I_CVTUF)); break;
        case 64: TIMED_LOOP(X32(I_LDEXP)); break;
        case 65: TIMED_LOOP(X32(I_FRACT)); break;
        case 66: TIMED_LOOP(X32(I_CMPI)); break;
        case 67: TIMED_LOOP(X32(I_PERMB)); break;
        case 68: TIMED_LOOP(X32(I_ALIGN)); break;
        case 69: TIMED_LOOP(X32(I_MADI24)); break;
        case 70: TIMED_LOOP(X32(I_ASHR)); break;
        case 71: TIMED_LOOP(X32(I_LSHR)); break;
        case 72: TIMED_LOOP(X32(I_ADDNEG)); break;
        case 73: TIMED_LOOP(X32(I_MIN3)); break;
        case 74: TIMED_LOOP(X32(I_FMAK)); break;
        case 75: TIMED_LOOP(X32(I_MADAK)); break;
        case 76: TIMED_LOOP(X32(I_SUBREV)); break;
        case 77: TIMED_LOOP(X32(I_EXPMOD)); break;
        case 78: TIMED_LOOP(X32(I_WRITELANE)); break;
        case 79: {
            asm volatile("v_mov_b32 v58, 64" ::: "v58");
            TIMED_LOOP(DS16("ds_read_b128", "v58"));
        } break;
        case 80: {
            const unsigned addr = (threadIdx.x & 63) * 4;
            asm volatile("v_mov_b32 v58, %0" :: "v"(addr) : "v58");
            TIMED_LOOP("ds_read_b32 v10, v58\n\tds_read_b32 v11, v58 offset:256\n\tds_read_b32 v12, v58 offset:512\n\tds_read_b32 v13, v58 offset:768\n\t"
                       "ds_read_b32 v14, v58 offset:1024\n\tds_read_b32 v15, v58 offset:1280\n\tds_read_b32 v16, v58 offset:1536\n\tds_read_b32 v17, v58 offset:1792\n\t"
                       "ds_read_b32 v18, v58 offset:2048\n\tds_read_b32 v19, v58 offset:2304\n\tds_read_b32 v20, v58 offset:2560\n\tds_read_b32 v21, v58 offset:2816\n\t"
                       "ds_read_b32 v22, v58 offset:3072\n\tds_read_b32 v23, v58 offset:3328\n\tds_read_b32 v24, v58 offset:3584\n\tds_read_b32 v25, v58 offset:3840\n\t"
                       "s_waitcnt lgkmcnt(0)\n\t");
        } break;
        case 82: TIMED_LOOP(X8M(I_MFMA16K16)); break;
        case 83: TIMED_LOOP(X32(I_MULHI)); break;
        default: {
            const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;        // every wave its own 1 KB: no write races that matter
            asm volatile("v_mov_b32 v58, %0" :: "v"(addr) : "v58");
            TIMED_LOOP("ds_write_b128 v58, v[10:13]\n\tds_write_b128 v58, v[14:17]\n\tds_write_b128 v58, v[18:21]\n\tds_write_b128 v58, v[22:25]\n\t"
                       "ds_write_b128 v58, v[26:29]\n\tds_write_b128 v58, v[30:33]\n\tds_write_b128 v58, v[34:37]\n\tds_write_b128 v58, v[38:41]\n\t"
                       "ds_write_b128 v58, v[10:13]\n\tds_write_b128 v58, v[14:17]\n\tds_write_b128 v58, v[18:21]\n\tds_write_b128 v58, v[22:25]\n\t"
                       "ds_write_b128 v58, v[26:29]\n\tds_write_b128 v58, v[30:33]\n\tds_write_b128 v58, v[34:37]\n\tds_write_b128 v58, v[38:41]\n\t"
                       "s_waitcnt lgkmcnt(0)\n\t");
        } break;
    }
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[w] = t1 - t0;
        hwid[w] = id;
    }
}


// ---- texel-gather rate: the render kernel's lookup pattern.  8 adjacent lanes read one 128-byte line (16 bytes each), so a wave
// instruction touches 8 lines; `scatter` = the 8 lines of an instruction are unrelated lines of a footprint of `lines` lines
// (16 KB: the CU's vector L1; 1 MB / 4 MB: the XCD's L2), `contiguous` = 8 consecutive lines (1 KB, a plain coalesced load).
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void gather_probe(const char* table, unsigned lines_mask, int scatter, int iters, float* out, unsigned long long* cyc, unsigned* hwid) {
    extern __shared__ __align__(16) float lds[];
    const unsigned lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned line = (wave * 2654435761u) >> 7;
    const unsigned step = scatter ? (grp * 2 + 1) * 40503u : 8u, first = scatter ? grp * 977u : grp;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; it++) {
        v4f v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            line += step;
            const unsigned off = ((line + first) & lines_mask) * 128u + sub * 16u;
            v[k] = *reinterpret_cast<const v4f*>(table + off);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k];
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]) : "memory");
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if (lane == 0) { cyc[wave] = t1 - t0; hwid[wave] = id; }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "valu_issue_probe.json";
    const int iters = 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_cyc; unsigned* d_id;
    hipMalloc(&d_cyc, n_cu * 16 * sizeof(unsigned long long));
    hipMalloc(&d_id, n_cu * 16 * sizeof(unsigned));
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    FILE* fo = fopen(out_path, "w");
    fprintf(fo, "{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz_reported\": %d, \"iters\": %d,\n \"how\": \"one workgroup of 256 x W lanes per CU "
                "(100 KB LDS keeps a second one off the CU), every wave times its own loop of independent instructions with s_memtime; "
                "simd_cyc_per_inst = (slowest wave of a SIMD, in cycles per instruction) / W, averaged over the SIMDs: the waves of a SIMD start together and the arbiter favours the oldest, so the SIMD has issued all W streams when its last wave ends\",\n \"streams\": [\n", prop.gcnArchName, n_cu, prop.clockRate / 1000, iters);
    bool first = true;
    for (int s = 0; s < kNumStreams; s++) {
        for (int W : {1, 2, 4}) {
            const int waves = n_cu * 4 * W;
            hipMemset(d_cyc, 0, waves * sizeof(unsigned long long));
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(probe, dim3(n_cu), dim3(256 * W), 100 * 1024, 0, s, 200, d_cyc, d_id);       // warm-up (clocks, icache)
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(n_cu), dim3(256 * W), 100 * 1024, 0, s, iters, d_cyc, d_id);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> cyc(waves); std::vector<unsigned> id(waves);
            hipMemcpy(cyc.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            hipMemcpy(id.data(), d_id, waves * sizeof(unsigned), hipMemcpyDeviceToHost);
            std::vector<double> per(waves);
            const double n_inst = double(iters) * kStreams[s].insts_per_iter;
            for (int w = 0; w < waves; w++) per[w] = double(cyc[w]) / n_inst;
            std::sort(per.begin(), per.end());
            // waves per (block, SIMD): HW_ID bits [5:4] = SIMD
            int wps_min = 1 << 30, wps_max = 0;
            for (int b = 0; b < n_cu; b++) {
                int cnt[4] = {0, 0, 0, 0};
                for (int w = 0; w < 4 * W; w++) cnt[(id[b * 4 * W + w] >> 4) & 3]++;
                for (int q = 0; q < 4; q++) { wps_min = std::min(wps_min, cnt[q]); wps_max = std::max(wps_max, cnt[q]); }
            }
            const double med = per[waves / 2];
            // per SIMD: its W waves start together (barrier) and the arbiter favours the oldest, so they END at different times;
            // the SIMD has issued W x n_inst instructions when its LAST wave ends -> SIMD rate = mean over SIMDs of (slowest wave) / W
            double simd_sum = 0; int simd_n = 0;
            for (int b = 0; b < n_cu; b++) {
                double worst[4] = {0, 0, 0, 0};
                for (int w = 0; w < 4 * W; w++) { const int q = (id[b * 4 * W + w] >> 4) & 3; worst[q] = std::max(worst[q], double(cyc[b * 4 * W + w]) / n_inst); }
                for (int q = 0; q < 4; q++) if (worst[q] > 0) { simd_sum += worst[q]; simd_n++; }
            }
            const double simd_rate = simd_sum / simd_n / W;
            // wall-clock cross-check: instructions per SIMD / elapsed -> cycles at the reported clock
            const double wall_cyc_per_inst = (ms * 1e-3 * prop.clockRate * 1e3) / (n_inst * W);
            fprintf(fo, "%s  {\"stream\": \"%s\", \"what\": \"%s\", \"waves_per_simd\": %d, \"waves_per_simd_by_hw_id\": [%d, %d], "
                        "\"wave_cyc_per_inst\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f}, \"simd_cyc_per_inst\": %.3f, "
                        "\"kernel_ms\": %.4f, \"simd_cyc_per_inst_by_wall_clock_at_reported_mhz\": %.3f}",
                    first ? "" : ",\n", kStreams[s].name, kStreams[s].what, W, wps_min, wps_max, med, per.front(), per.back(), simd_rate, ms, wall_cyc_per_inst);
            first = false;
            printf("%-22s W=%d  wave %.3f cyc/inst  simd %.3f cyc/inst  (hw_id waves/SIMD %d..%d, %.3f ms)\n", kStreams[s].name, W, med, simd_rate, wps_min, wps_max, ms);
            fflush(stdout);
        }
    }
    fprintf(fo, "\n ]");
    // ---- gather rates
    {
        const size_t table_bytes = size_t(64) << 20;
        char* table; hipMalloc(&table, table_bytes); hipMemset(table, 0, table_bytes);
        float* d_out; hipMalloc(&d_out, 16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(gather_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        const int g_iters = 2000;
        struct { const char* name; unsigned lines; int scatter; } modes[] = {
            {"contiguous_1KB_in_16KB", 128, 0}, {"scatter8_in_16KB", 128, 1}, {"scatter8_in_256KB", 2048, 1}, {"scatter8_in_2MB", 16384, 1}, {"scatter8_in_32MB", 262144, 1}};
        fprintf(fo, ",\n \"gather\": [\n");
        bool gfirst = true;
        for (auto& m : modes) for (int W : {1, 2, 4}) {
            const int waves = n_cu * 4 * W;
            hipLaunchKernelGGL(gather_probe, dim3(n_cu), dim3(256 * W), 100 * 1024, 0, table, m.lines - 1, m.scatter, 50, d_out, d_cyc, d_id);
            hipLaunchKernelGGL(gather_probe, dim3(n_cu), dim3(256 * W), 100 * 1024, 0, table, m.lines - 1, m.scatter, g_iters, d_out, d_cyc, d_id);
            if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "gather launch failed\n"); return 1; }
            std::vector<unsigned long long> cyc(waves);
            hipMemcpy(cyc.data(), d_cyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            double worst_sum = 0;
            for (int b = 0; b < n_cu; b++) { unsigned long long wmax = 0; for (int w = 0; w < 4 * W; w++) wmax = std::max(wmax, cyc[b * 4 * W + w]); worst_sum += double(wmax); }
            const double cu_cycles = worst_sum / n_cu, loads_per_cu = double(g_iters) * 8 * 4 * W;
            fprintf(fo, "%s  {\"pattern\": \"%s\", \"waves_per_simd\": %d, \"cu_cycles_per_wave_load\": %.2f, \"bytes_per_cycle_per_cu\": %.1f}",
                    gfirst ? "" : ",\n", m.name, W, cu_cycles / loads_per_cu, 1024.0 * loads_per_cu / cu_cycles);
            gfirst = false;
            printf("gather %-24s W=%d  %.2f CU-cycles per wave load (1 KB)  = %.1f B/cycle/CU\n", m.name, W, cu_cycles / loads_per_cu, 1024.0 * loads_per_cu / cu_cycles);
        }
        fprintf(fo, "\n ]");
    }
    fprintf(fo, "}\n");
    fclose(fo);
    return 0;
}
