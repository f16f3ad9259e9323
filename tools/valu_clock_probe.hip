// VALU issue rate on gfx950 in REAL shader cycles (round 5; VERDICT r04 "weak" #3 / "next" #3).
//
// tools/valu_probe.hip timed instruction streams with HIP events and converted with an ASSUMED 2.4 GHz.  Here every wave brackets its
// stream with s_memtime (shader-clock ticks: MI355X_MICROARCH.md, "s_memtime tick = shader cycle") and s_memrealtime (constant 100 MHz),
// so the table is in cycles the SIMD really ran and the clock under that load is printed next to it — no GRBM_GUI_ACTIVE bias.
// The streams are written with explicit registers, so the operand forms (VGPR banks = register number mod 4, SGPR / inline-constant
// operands, VOP2 vs VOP3) are what the text says and not what the register allocator chose.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_clock_probe.hip -o tools/valu_clock_probe && ./tools/valu_clock_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define ITERS 4000
#define STR2(x) #x
#define STR(x) STR2(x)
// eight accumulators v8, v9, ..., v15 (banks 0 1 2 3 0 1 2 3); constants v16..v23; s30/s31 scalar constants
#define R8(a)  a(8) a(9) a(10) a(11) a(12) a(13) a(14) a(15)
#define R64(a) R8(a) R8(a) R8(a) R8(a) R8(a) R8(a) R8(a) R8(a)

// operand forms (each macro = one instruction on accumulator v<n>; constants: v16..v23 = banks 0 1 2 3 0 1 2 3 if bank = number mod 4)
#define F_FMA(n)          "v_fma_f32 v" #n ", v" #n ", v17, v18\n"         /* the reference stream: acc, then two constants in different registers */
#define F_FMA_ACC2(n)     "v_fma_f32 v" #n ", v17, v18, v" #n "\n"         /* accumulator as src2 (what v_fmac encodes) */
#define F_FMA_P01(n)      "v_fma_f32 v" #n ", v" #n ", v16, v17\n"
#define F_FMA_P02(n)      "v_fma_f32 v" #n ", v" #n ", v16, v18\n"
#define F_FMA_P03(n)      "v_fma_f32 v" #n ", v" #n ", v16, v19\n"
#define F_FMA_P04(n)      "v_fma_f32 v" #n ", v" #n ", v16, v20\n"         /* 16 and 20: same bank if bank = number mod 4 */
#define F_FMA_P08(n)      "v_fma_f32 v" #n ", v" #n ", v16, v24\n"
#define F_FMA_P15(n)      "v_fma_f32 v" #n ", v" #n ", v17, v21\n"
#define F_FMA_P05(n)      "v_fma_f32 v" #n ", v" #n ", v16, v21\n"
#define F_FMA_SAMEREG(n)  "v_fma_f32 v" #n ", v" #n ", v17, v17\n"
#define F_FMA_SQ(n)       "v_fma_f32 v" #n ", v17, v17, v" #n "\n"         /* x * x + acc */
#define F_FMAC_SQ(n)      "v_fmac_f32 v" #n ", v17, v17\n"
#define F_MUL_SQ(n)       "v_mul_f32 v" #n ", v17, v17\n"
#define F_FMA_SGPR(n)     "v_fma_f32 v" #n ", v" #n ", s30, v18\n"
#define F_FMA_INLINE(n)   "v_fma_f32 v" #n ", v" #n ", 1.0, v18\n"
#define F_FMA_INLINE2(n)  "v_fma_f32 v" #n ", v" #n ", v17, 1.0\n"
#define F_FMA_2SGPR(n)    "v_fma_f32 v" #n ", v" #n ", s30, s30\n"
#define F_FMA_NEG(n)      "v_fma_f32 v" #n ", -v" #n ", v17, v18\n"
#define F_FMAC(n)         "v_fmac_f32 v" #n ", v17, v18\n"
#define F_FMAC_SGPR(n)    "v_fmac_f32 v" #n ", s30, v18\n"
#define F_FMAC_LIT(n)     "v_fmac_f32 v" #n ", 0x3f7fbe77, v18\n"          /* VOP2 with a 32-bit literal */
#define F_MUL(n)          "v_mul_f32 v" #n ", v17, v" #n "\n"
#define F_MUL_E64(n)      "v_mul_f32_e64 v" #n ", v17, v" #n "\n"
#define F_MUL_SGPR(n)     "v_mul_f32 v" #n ", s30, v" #n "\n"
#define F_MUL_LIT(n)      "v_mul_f32 v" #n ", 0x3f7fbe77, v" #n "\n"
#define F_MUL_INLINE(n)   "v_mul_f32 v" #n ", 1.0, v" #n "\n"
#define F_ADD(n)          "v_add_f32 v" #n ", v18, v" #n "\n"
#define F_SUB(n)          "v_sub_f32 v" #n ", v" #n ", v18\n"
#define F_MOV(n)          "v_mov_b32 v" #n ", v17\n"
#define F_MOV_SGPR(n)     "v_mov_b32 v" #n ", s30\n"
#define F_EXP(n)          "v_exp_f32 v" #n ", v" #n "\n"
#define F_EXP_CLAMP(n)    "v_exp_f32_e64 v" #n ", v" #n " clamp\n"
#define F_RCP(n)          "v_rcp_f32 v" #n ", v" #n "\n"
#define F_CNDMASK(n)      "v_cndmask_b32 v" #n ", v" #n ", v17, vcc\n"
#define F_CNDMASK_S(n)    "v_cndmask_b32_e64 v" #n ", v" #n ", v17, s[20:21]\n"
#define F_CNDMASK_0(n)    "v_cndmask_b32_e64 v" #n ", 0, v17, s[20:21]\n"
#define F_CMP(n)          "v_cmp_lt_f32 vcc, v" #n ", v17\n"
#define F_CMP_S(n)        "v_cmp_ge_f32_e64 s[20:21], v" #n ", v17\n"
#define F_CMP_SGPRSRC(n)  "v_cmp_ge_f32 vcc, s30, v" #n "\n"
#define F_MIN(n)          "v_min_f32 v" #n ", v17, v" #n "\n"
#define F_MAX(n)          "v_max_f32 v" #n ", v17, v" #n "\n"
#define F_MED3(n)         "v_med3_f32 v" #n ", v" #n ", v17, v18\n"
#define F_MED3_C(n)       "v_med3_f32 v" #n ", v" #n ", 0, 1.0\n"
#define F_AND(n)          "v_and_b32 v" #n ", v17, v" #n "\n"
#define F_AND_LIT(n)      "v_and_b32 v" #n ", 0xff, v" #n "\n"
#define F_AND_INL(n)      "v_and_b32 v" #n ", 15, v" #n "\n"
#define F_ADDU(n)         "v_add_u32 v" #n ", 1, v" #n "\n"
#define F_ADDU_V(n)       "v_add_u32 v" #n ", v17, v" #n "\n"
#define F_LSHL(n)         "v_lshlrev_b32 v" #n ", 2, v" #n "\n"
#define F_MADU24(n)       "v_mad_u32_u24 v" #n ", v" #n ", v17, v18\n"
#define F_MADU24_S(n)     "v_mad_u32_u24 v" #n ", v" #n ", s31, v18\n"
#define F_LSHLADD(n)      "v_lshl_add_u32 v" #n ", v" #n ", 2, v18\n"
#define F_CVT(n)          "v_cvt_f32_u32 v" #n ", v" #n "\n"
#define F_DPP_ADD(n)      "v_add_f32_dpp v" #n ", v" #n ", v" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPP_MOV(n)      "v_mov_b32_dpp v" #n ", v17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_PKFMA(n)        "v_pk_fma_f32 v[24:25], v[24:25], v[26:27], v[28:29]\n"
#define F_PKMUL(n)        "v_pk_mul_f32 v[24:25], v[24:25], v[26:27]\n"
#define F_SWAP32(n)       "v_permlane32_swap_b32 v24, v25\n"
#define F_SWAP16(n)       "v_permlane16_swap_b32 v24, v25\n"
#define F_READLANE(n)     "v_readlane_b32 s22, v" #n ", 3\n"
#define F_READFIRST(n)    "v_readfirstlane_b32 s22, v" #n "\n"
#define F_FMA_DEP(n)      "v_fma_f32 v8, v8, v17, v18\n"
#define F_SNOP(n)         "s_nop 0\n"

// pure stream: 64 x the form; mix: 2 of every 8 instructions are the form, the other 6 the reference FMA
#define R8P(a)   a(8) a(9) a(10) a(11) a(12) a(13) a(14) a(15)
#define R8M(a)   a(8) F_FMA(9) F_FMA(10) F_FMA(11) a(12) F_FMA(13) F_FMA(14) F_FMA(15)
#define PURE(a)  R8P(a) R8P(a) R8P(a) R8P(a) R8P(a) R8P(a) R8P(a) R8P(a)
#define MIX(a)   R8M(a) R8M(a) R8M(a) R8M(a) R8M(a) R8M(a) R8M(a) R8M(a)

#define CLOBBERS "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
    "v28", "v29", "v30", "v31", "v32", "v36", "v40", "v44", "v48", "v52", "s20", "s21", "s22", "s28", "s30", "s31", "vcc", "scc", "memory"

#define PROLOGUE                                                                                                  \
    "v_mov_b32 v8, 1.0\nv_mov_b32 v9, 1.0\nv_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\nv_mov_b32 v12, 1.0\n"          \
    "v_mov_b32 v13, 1.0\nv_mov_b32 v14, 1.0\nv_mov_b32 v15, 1.0\nv_mov_b32 v16, 1.0\nv_mov_b32 v17, 1.0\n"        \
    "v_mov_b32 v18, 0\nv_mov_b32 v19, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v24, 1.0\nv_mov_b32 v25, 1.0\n" \
    "v_mov_b32 v26, 1.0\nv_mov_b32 v27, 1.0\nv_mov_b32 v28, 0\nv_mov_b32 v29, 0\nv_mov_b32 v30, 1.0\nv_mov_b32 v31, 1.0\n" \
    "v_mov_b32 v32, 1.0\nv_mov_b32 v36, 1.0\nv_mov_b32 v40, 1.0\nv_mov_b32 v44, 1.0\nv_mov_b32 v48, 1.0\nv_mov_b32 v52, 1.0\n" \
    "s_mov_b32 s30, 1.0\ns_mov_b32 s31, 0\ns_mov_b64 s[20:21], exec\ns_mov_b64 vcc, exec\ns_mov_b32 s28, " STR(ITERS) "\n"                                      \
    "s_barrier\n"                                                                                                 \
    "s_memtime %0\ns_memrealtime %1\ns_waitcnt lgkmcnt(0)\n"                                                      \
    "1:\n"
#define EPILOGUE                                                                                                  \
    "s_sub_u32 s28, s28, 1\ns_cmp_lg_u32 s28, 0\ns_cbranch_scc1 1b\n"                                             \
    "s_memtime %2\ns_memrealtime %3\ns_waitcnt lgkmcnt(0)\n"

struct WaveRec { uint64_t t0, r0, t1, r1; uint32_t hw_id, xcc_id, pad0, pad1; };

#define PROBE_KERNEL(NAME, BODY)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(WaveRec* out) {                                                   \
        uint64_t t0, r0, t1, r1;                                                                                  \
        asm volatile(PROLOGUE BODY EPILOGUE : "=&s"(t0), "=&s"(r0), "=&s"(t1), "=&s"(r1) : : CLOBBERS);           \
        if ((threadIdx.x & 63u) == 0u) {                                                                          \
            uint32_t hw, xcc;                                                                                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\ns_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc)); \
            WaveRec w = {t0, r0, t1, r1, hw, xcc, 0u, 0u};                                                        \
            out[blockIdx.x * 4u + (threadIdx.x >> 6)] = w;                                                        \
        }                                                                                                         \
    }

PROBE_KERNEL(kp_FMA, PURE(F_FMA))
PROBE_KERNEL(km_FMA, MIX(F_FMA))
PROBE_KERNEL(kp_FMA_ACC2, PURE(F_FMA_ACC2))
PROBE_KERNEL(km_FMA_ACC2, MIX(F_FMA_ACC2))
PROBE_KERNEL(kp_FMA_P01, PURE(F_FMA_P01))
PROBE_KERNEL(km_FMA_P01, MIX(F_FMA_P01))
PROBE_KERNEL(kp_FMA_P02, PURE(F_FMA_P02))
PROBE_KERNEL(km_FMA_P02, MIX(F_FMA_P02))
PROBE_KERNEL(kp_FMA_P03, PURE(F_FMA_P03))
PROBE_KERNEL(km_FMA_P03, MIX(F_FMA_P03))
PROBE_KERNEL(kp_FMA_P04, PURE(F_FMA_P04))
PROBE_KERNEL(km_FMA_P04, MIX(F_FMA_P04))
PROBE_KERNEL(kp_FMA_P08, PURE(F_FMA_P08))
PROBE_KERNEL(km_FMA_P08, MIX(F_FMA_P08))
PROBE_KERNEL(kp_FMA_P15, PURE(F_FMA_P15))
PROBE_KERNEL(km_FMA_P15, MIX(F_FMA_P15))
PROBE_KERNEL(kp_FMA_P05, PURE(F_FMA_P05))
PROBE_KERNEL(km_FMA_P05, MIX(F_FMA_P05))
PROBE_KERNEL(kp_FMA_SAMEREG, PURE(F_FMA_SAMEREG))
PROBE_KERNEL(km_FMA_SAMEREG, MIX(F_FMA_SAMEREG))
PROBE_KERNEL(kp_FMA_SQ, PURE(F_FMA_SQ))
PROBE_KERNEL(km_FMA_SQ, MIX(F_FMA_SQ))
PROBE_KERNEL(kp_FMAC_SQ, PURE(F_FMAC_SQ))
PROBE_KERNEL(km_FMAC_SQ, MIX(F_FMAC_SQ))
PROBE_KERNEL(kp_MUL_SQ, PURE(F_MUL_SQ))
PROBE_KERNEL(km_MUL_SQ, MIX(F_MUL_SQ))
PROBE_KERNEL(kp_FMA_SGPR, PURE(F_FMA_SGPR))
PROBE_KERNEL(km_FMA_SGPR, MIX(F_FMA_SGPR))
PROBE_KERNEL(kp_FMA_INLINE, PURE(F_FMA_INLINE))
PROBE_KERNEL(km_FMA_INLINE, MIX(F_FMA_INLINE))
PROBE_KERNEL(kp_FMA_INLINE2, PURE(F_FMA_INLINE2))
PROBE_KERNEL(km_FMA_INLINE2, MIX(F_FMA_INLINE2))
PROBE_KERNEL(kp_FMA_2SGPR, PURE(F_FMA_2SGPR))
PROBE_KERNEL(km_FMA_2SGPR, MIX(F_FMA_2SGPR))
PROBE_KERNEL(kp_FMA_NEG, PURE(F_FMA_NEG))
PROBE_KERNEL(km_FMA_NEG, MIX(F_FMA_NEG))
PROBE_KERNEL(kp_FMAC, PURE(F_FMAC))
PROBE_KERNEL(km_FMAC, MIX(F_FMAC))
PROBE_KERNEL(kp_FMAC_SGPR, PURE(F_FMAC_SGPR))
PROBE_KERNEL(km_FMAC_SGPR, MIX(F_FMAC_SGPR))
PROBE_KERNEL(kp_FMAC_LIT, PURE(F_FMAC_LIT))
PROBE_KERNEL(km_FMAC_LIT, MIX(F_FMAC_LIT))
PROBE_KERNEL(kp_MUL, PURE(F_MUL))
PROBE_KERNEL(km_MUL, MIX(F_MUL))
PROBE_KERNEL(kp_MUL_E64, PURE(F_MUL_E64))
PROBE_KERNEL(km_MUL_E64, MIX(F_MUL_E64))
PROBE_KERNEL(kp_MUL_SGPR, PURE(F_MUL_SGPR))
PROBE_KERNEL(km_MUL_SGPR, MIX(F_MUL_SGPR))
PROBE_KERNEL(kp_MUL_LIT, PURE(F_MUL_LIT))
PROBE_KERNEL(km_MUL_LIT, MIX(F_MUL_LIT))
PROBE_KERNEL(kp_MUL_INLINE, PURE(F_MUL_INLINE))
PROBE_KERNEL(km_MUL_INLINE, MIX(F_MUL_INLINE))
PROBE_KERNEL(kp_ADD, PURE(F_ADD))
PROBE_KERNEL(km_ADD, MIX(F_ADD))
PROBE_KERNEL(kp_SUB, PURE(F_SUB))
PROBE_KERNEL(km_SUB, MIX(F_SUB))
PROBE_KERNEL(kp_MOV, PURE(F_MOV))
PROBE_KERNEL(km_MOV, MIX(F_MOV))
PROBE_KERNEL(kp_MOV_SGPR, PURE(F_MOV_SGPR))
PROBE_KERNEL(km_MOV_SGPR, MIX(F_MOV_SGPR))
PROBE_KERNEL(kp_EXP, PURE(F_EXP))
PROBE_KERNEL(km_EXP, MIX(F_EXP))
PROBE_KERNEL(kp_EXP_CLAMP, PURE(F_EXP_CLAMP))
PROBE_KERNEL(km_EXP_CLAMP, MIX(F_EXP_CLAMP))
PROBE_KERNEL(kp_RCP, PURE(F_RCP))
PROBE_KERNEL(km_RCP, MIX(F_RCP))
PROBE_KERNEL(kp_CNDMASK, PURE(F_CNDMASK))
PROBE_KERNEL(km_CNDMASK, MIX(F_CNDMASK))
PROBE_KERNEL(kp_CNDMASK_S, PURE(F_CNDMASK_S))
PROBE_KERNEL(km_CNDMASK_S, MIX(F_CNDMASK_S))
PROBE_KERNEL(kp_CNDMASK_0, PURE(F_CNDMASK_0))
PROBE_KERNEL(km_CNDMASK_0, MIX(F_CNDMASK_0))
PROBE_KERNEL(kp_CMP, PURE(F_CMP))
PROBE_KERNEL(km_CMP, MIX(F_CMP))
PROBE_KERNEL(kp_CMP_S, PURE(F_CMP_S))
PROBE_KERNEL(km_CMP_S, MIX(F_CMP_S))
PROBE_KERNEL(kp_CMP_SGPRSRC, PURE(F_CMP_SGPRSRC))
PROBE_KERNEL(km_CMP_SGPRSRC, MIX(F_CMP_SGPRSRC))
PROBE_KERNEL(kp_MIN, PURE(F_MIN))
PROBE_KERNEL(km_MIN, MIX(F_MIN))
PROBE_KERNEL(kp_MAX, PURE(F_MAX))
PROBE_KERNEL(km_MAX, MIX(F_MAX))
PROBE_KERNEL(kp_MED3, PURE(F_MED3))
PROBE_KERNEL(km_MED3, MIX(F_MED3))
PROBE_KERNEL(kp_MED3_C, PURE(F_MED3_C))
PROBE_KERNEL(km_MED3_C, MIX(F_MED3_C))
PROBE_KERNEL(kp_AND, PURE(F_AND))
PROBE_KERNEL(km_AND, MIX(F_AND))
PROBE_KERNEL(kp_AND_LIT, PURE(F_AND_LIT))
PROBE_KERNEL(km_AND_LIT, MIX(F_AND_LIT))
PROBE_KERNEL(kp_AND_INL, PURE(F_AND_INL))
PROBE_KERNEL(km_AND_INL, MIX(F_AND_INL))
PROBE_KERNEL(kp_ADDU, PURE(F_ADDU))
PROBE_KERNEL(km_ADDU, MIX(F_ADDU))
PROBE_KERNEL(kp_ADDU_V, PURE(F_ADDU_V))
PROBE_KERNEL(km_ADDU_V, MIX(F_ADDU_V))
PROBE_KERNEL(kp_LSHL, PURE(F_LSHL))
PROBE_KERNEL(km_LSHL, MIX(F_LSHL))
PROBE_KERNEL(kp_MADU24, PURE(F_MADU24))
PROBE_KERNEL(km_MADU24, MIX(F_MADU24))
PROBE_KERNEL(kp_MADU24_S, PURE(F_MADU24_S))
PROBE_KERNEL(km_MADU24_S, MIX(F_MADU24_S))
PROBE_KERNEL(kp_LSHLADD, PURE(F_LSHLADD))
PROBE_KERNEL(km_LSHLADD, MIX(F_LSHLADD))
PROBE_KERNEL(kp_CVT, PURE(F_CVT))
PROBE_KERNEL(km_CVT, MIX(F_CVT))
PROBE_KERNEL(kp_DPP_ADD, PURE(F_DPP_ADD))
PROBE_KERNEL(km_DPP_ADD, MIX(F_DPP_ADD))
PROBE_KERNEL(kp_DPP_MOV, PURE(F_DPP_MOV))
PROBE_KERNEL(km_DPP_MOV, MIX(F_DPP_MOV))
PROBE_KERNEL(kp_PKFMA, PURE(F_PKFMA))
PROBE_KERNEL(km_PKFMA, MIX(F_PKFMA))
PROBE_KERNEL(kp_PKMUL, PURE(F_PKMUL))
PROBE_KERNEL(km_PKMUL, MIX(F_PKMUL))
PROBE_KERNEL(kp_SWAP32, PURE(F_SWAP32))
PROBE_KERNEL(km_SWAP32, MIX(F_SWAP32))
PROBE_KERNEL(kp_SWAP16, PURE(F_SWAP16))
PROBE_KERNEL(km_SWAP16, MIX(F_SWAP16))
PROBE_KERNEL(kp_READLANE, PURE(F_READLANE))
PROBE_KERNEL(km_READLANE, MIX(F_READLANE))
PROBE_KERNEL(kp_READFIRST, PURE(F_READFIRST))
PROBE_KERNEL(km_READFIRST, MIX(F_READFIRST))
PROBE_KERNEL(kp_FMA_DEP, PURE(F_FMA_DEP))
PROBE_KERNEL(km_FMA_DEP, MIX(F_FMA_DEP))
PROBE_KERNEL(kp_SNOP, PURE(F_SNOP))
PROBE_KERNEL(km_SNOP, MIX(F_SNOP))

typedef void (*kern_t)(WaveRec*);

struct Result { double cyc_per_inst, wave_cyc_per_inst, ghz; int simds, waves_per_simd_max; };

// waves_per_simd W: 256 CUs x W blocks of 4 waves (a block's waves go to the CU's 4 SIMDs); dynamic LDS limits the blocks per CU to W.
// cycles per instruction PER SIMD = (last s_memtime of the SIMD's waves - first) / (instructions its waves issued): independent of how the
// arbiter shares the SIMD among the waves (round 5's first version divided a wave's own interval by W: wrong when older waves are favoured).
static Result run(kern_t k, int W, WaveRec* d, std::vector<WaveRec>& h) {
    const int blocks = 256 * W;
    const size_t lds = W >= 8 ? 0 : (size_t)(160 * 1024 / W) - 1024;
    (void)hipMemset(d, 0, sizeof(WaveRec) * blocks * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d);   // warm-up (clock ramp)
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, d);
    (void)hipDeviceSynchronize();
    h.resize((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), d, sizeof(WaveRec) * h.size(), hipMemcpyDeviceToHost);
    struct Simd { uint64_t t0 = ~0ull, t1 = 0; int n = 0; };
    std::map<uint32_t, Simd> per_simd;
    std::vector<double> wcyc, ghz, scyc;
    for (const WaveRec& w : h) {
        const uint32_t simd = (w.hw_id >> 4) & 3u, cu = (w.hw_id >> 8) & 15u, sh = (w.hw_id >> 12) & 1u, se = (w.hw_id >> 13) & 7u;
        Simd& sd = per_simd[(w.xcc_id & 15u) << 16 | se << 12 | sh << 8 | cu << 4 | simd];
        sd.t0 = std::min(sd.t0, w.t0); sd.t1 = std::max(sd.t1, w.t1); sd.n++;
        const double dt = (double)(w.t1 - w.t0), dr = (double)(w.r1 - w.r0);
        wcyc.push_back(dt / ((double)ITERS * 64.0));
        ghz.push_back(dt / (dr * 10.0));                // ticks per 10 ns
    }
    int mx = 0;
    for (auto& kv : per_simd) {
        mx = std::max(mx, kv.second.n);
        if (kv.second.n == W) scyc.push_back((double)(kv.second.t1 - kv.second.t0) / ((double)W * ITERS * 64.0));
    }
    std::sort(wcyc.begin(), wcyc.end());
    std::sort(ghz.begin(), ghz.end());
    std::sort(scyc.begin(), scyc.end());
    Result r;
    r.simds = (int)scyc.size();
    r.waves_per_simd_max = mx;
    r.cyc_per_inst = scyc.empty() ? 0.0 : scyc[scyc.size() / 2];
    r.wave_cyc_per_inst = wcyc[wcyc.size() / 2];
    r.ghz = ghz[ghz.size() / 2];
    return r;
}

static std::string text_of(const char* form) {   // "v_fma_f32 vN, vN, v17, v18\n" -> without the newline, first instruction only
    std::string t(form);
    const size_t nl = t.find('\n');
    return nl == std::string::npos ? t : t.substr(0, nl);
}

int main(int argc, char** argv) {
    WaveRec* d;
    (void)hipMalloc(&d, sizeof(WaveRec) * 256 * 8 * 4);
    std::vector<WaveRec> h;
    struct { const char* name; kern_t pure, mix; } ks[] = {
        {F_FMA(N), kp_FMA, km_FMA},
        {F_FMA_ACC2(N), kp_FMA_ACC2, km_FMA_ACC2},
        {F_FMA_P01(N), kp_FMA_P01, km_FMA_P01},
        {F_FMA_P02(N), kp_FMA_P02, km_FMA_P02},
        {F_FMA_P03(N), kp_FMA_P03, km_FMA_P03},
        {F_FMA_P04(N), kp_FMA_P04, km_FMA_P04},
        {F_FMA_P08(N), kp_FMA_P08, km_FMA_P08},
        {F_FMA_P15(N), kp_FMA_P15, km_FMA_P15},
        {F_FMA_P05(N), kp_FMA_P05, km_FMA_P05},
        {F_FMA_SAMEREG(N), kp_FMA_SAMEREG, km_FMA_SAMEREG},
        {F_FMA_SQ(N), kp_FMA_SQ, km_FMA_SQ},
        {F_FMAC_SQ(N), kp_FMAC_SQ, km_FMAC_SQ},
        {F_MUL_SQ(N), kp_MUL_SQ, km_MUL_SQ},
        {F_FMA_SGPR(N), kp_FMA_SGPR, km_FMA_SGPR},
        {F_FMA_INLINE(N), kp_FMA_INLINE, km_FMA_INLINE},
        {F_FMA_INLINE2(N), kp_FMA_INLINE2, km_FMA_INLINE2},
        {F_FMA_2SGPR(N), kp_FMA_2SGPR, km_FMA_2SGPR},
        {F_FMA_NEG(N), kp_FMA_NEG, km_FMA_NEG},
        {F_FMAC(N), kp_FMAC, km_FMAC},
        {F_FMAC_SGPR(N), kp_FMAC_SGPR, km_FMAC_SGPR},
        {F_FMAC_LIT(N), kp_FMAC_LIT, km_FMAC_LIT},
        {F_MUL(N), kp_MUL, km_MUL},
        {F_MUL_E64(N), kp_MUL_E64, km_MUL_E64},
        {F_MUL_SGPR(N), kp_MUL_SGPR, km_MUL_SGPR},
        {F_MUL_LIT(N), kp_MUL_LIT, km_MUL_LIT},
        {F_MUL_INLINE(N), kp_MUL_INLINE, km_MUL_INLINE},
        {F_ADD(N), kp_ADD, km_ADD},
        {F_SUB(N), kp_SUB, km_SUB},
        {F_MOV(N), kp_MOV, km_MOV},
        {F_MOV_SGPR(N), kp_MOV_SGPR, km_MOV_SGPR},
        {F_EXP(N), kp_EXP, km_EXP},
        {F_EXP_CLAMP(N), kp_EXP_CLAMP, km_EXP_CLAMP},
        {F_RCP(N), kp_RCP, km_RCP},
        {F_CNDMASK(N), kp_CNDMASK, km_CNDMASK},
        {F_CNDMASK_S(N), kp_CNDMASK_S, km_CNDMASK_S},
        {F_CNDMASK_0(N), kp_CNDMASK_0, km_CNDMASK_0},
        {F_CMP(N), kp_CMP, km_CMP},
        {F_CMP_S(N), kp_CMP_S, km_CMP_S},
        {F_CMP_SGPRSRC(N), kp_CMP_SGPRSRC, km_CMP_SGPRSRC},
        {F_MIN(N), kp_MIN, km_MIN},
        {F_MAX(N), kp_MAX, km_MAX},
        {F_MED3(N), kp_MED3, km_MED3},
        {F_MED3_C(N), kp_MED3_C, km_MED3_C},
        {F_AND(N), kp_AND, km_AND},
        {F_AND_LIT(N), kp_AND_LIT, km_AND_LIT},
        {F_AND_INL(N), kp_AND_INL, km_AND_INL},
        {F_ADDU(N), kp_ADDU, km_ADDU},
        {F_ADDU_V(N), kp_ADDU_V, km_ADDU_V},
        {F_LSHL(N), kp_LSHL, km_LSHL},
        {F_MADU24(N), kp_MADU24, km_MADU24},
        {F_MADU24_S(N), kp_MADU24_S, km_MADU24_S},
        {F_LSHLADD(N), kp_LSHLADD, km_LSHLADD},
        {F_CVT(N), kp_CVT, km_CVT},
        {F_DPP_ADD(N), kp_DPP_ADD, km_DPP_ADD},
        {F_DPP_MOV(N), kp_DPP_MOV, km_DPP_MOV},
        {F_PKFMA(N), kp_PKFMA, km_PKFMA},
        {F_PKMUL(N), kp_PKMUL, km_PKMUL},
        {F_SWAP32(N), kp_SWAP32, km_SWAP32},
        {F_SWAP16(N), kp_SWAP16, km_SWAP16},
        {F_READLANE(N), kp_READLANE, km_READLANE},
        {F_READFIRST(N), kp_READFIRST, km_READFIRST},
        {F_FMA_DEP(N), kp_FMA_DEP, km_FMA_DEP},
        {F_SNOP(N), kp_SNOP, km_SNOP},
    };
    const int Ws[4] = {1, 2, 4, 8};
    printf("# valu_clock_probe (MI355X, gfx950): cycles = s_memtime ticks (shader clock); clock = s_memtime ticks per s_memrealtime tick (100 MHz)\n");
    printf("# per wave %d x 64 instructions; N = the wave's accumulator register (v8..v15 in turn); v16..v31 hold constants, s30 = 1.0\n", ITERS);
    printf("# pure: the stream is 64 x the form.  mix: 2 of every 8 instructions are the form, 6 are `v_fma_f32 vN, vN, v17, v18`; the column is the\n");
    printf("#   MARGINAL cost of one instruction of the form in that stream = (8 x cycles(mix) - 6 x cycles(reference FMA)) / 2, same W\n");
    printf("| form | pure W=1 | W=2 | W=4 | W=8 | in-mix marginal W=2 | W=4 | W=8 | clock GHz (pure, W=4) | one wave's cycles/instr at W=4 |\n|---|---|---|---|---|---|---|---|---|---|\n");
    double ref[4] = {0, 0, 0, 0};
    for (auto& e : ks) {
        Result rp[4], rm[4];
        for (int i = 0; i < 4; ++i) { rp[i] = run(e.pure, Ws[i], d, h); rm[i] = run(e.mix, Ws[i], d, h); }
        if (ref[0] == 0) for (int i = 0; i < 4; ++i) ref[i] = rp[i].cyc_per_inst;
        printf("| `%s` | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.3f | %.2f |\n", text_of(e.name).c_str(), rp[0].cyc_per_inst, rp[1].cyc_per_inst, rp[2].cyc_per_inst,
               rp[3].cyc_per_inst, (8 * rm[1].cyc_per_inst - 6 * ref[1]) / 2, (8 * rm[2].cyc_per_inst - 6 * ref[2]) / 2, (8 * rm[3].cyc_per_inst - 6 * ref[3]) / 2,
               rp[2].ghz, rp[2].wave_cyc_per_inst);
        if (rp[2].simds < 900) printf("|   (only %d SIMDs held exactly 4 waves; max %d) |\n", rp[2].simds, rp[2].waves_per_simd_max);
        fflush(stdout);
    }
    return 0;
}
