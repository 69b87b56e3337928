// enc_rows_asm.hpp -- one sample of the encoder for 1-row slices (putSymbol<true,4,6,7> over the range encoder,
// llcomp.hpp:33-89, 166-206), hand-written for gfx950.  Same bins, same order, same arithmetic as enc_residual<*, true>
// in slice_kernels.hip (which the other kernel families keep using and which documents the scheme: every lane takes
// the bit-0 outcome, the lanes with a 1 patch up under an exec mask; held byte inside `low`; eager carries).
//
// Why by hand: the kernels are bound by instruction issue, vector AND scalar.  hipcc's structurised control flow spends
// 9-11 scalar instructions and 4-5 branches per bin on exec bookkeeping (a save / skip-branch / restore per `if`, an
// accumulated exit mask per loop, copies at every merge).  Written directly, the lock-step walk needs 3-5: the lane sets
// of the unary phase are nested (exec only ever shrinks until the phase ends, one restore), a lane that codes the closing
// zero of its run simply drops out with its renormalisation pending (done once for everybody behind the phase), and the
// rare carry into a held 0xFF is a subroutine shared by all renormalisation sites.
//
// LDS contract (the kernel puts ALL its LDS into the dynamic block so that it starts at address 0, and checks it):
//   [0, 1024)  model table (tables.hpp entries; successor offsets are relative to 0)
// Register contract: the asm owns v32..v53 and s36..s54 for the length of the block (clobbers); everything that lives
// across samples is an operand.
#pragma once
#include <cstdint>

namespace llcomp_mi {

// bank words / entries / temporaries: VGPRs owned by the block.  Together with what hipcc needs around the block (v0..v31,
// s0..s35) the kernel stays at 56 VGPRs / 64 SGPRs: the register files hold eight wavefronts per SIMD with room to spare
// (a first version that owned v36..v63 / s64..s86 sat exactly on the limit of both files and ran 5 % slower than hipcc's
// code with 14 % fewer scalar instructions: fewer wavefronts were resident).
#define LL_B0 "v32"
#define LL_B1 "v33"
#define LL_B "v[32:33]"
#define LL_A "v32"     // |residual| (the bank words are dead once the entries are requested)
#define LL_EX "v33"    // exponent = floor(log2 |residual|)
#define LL_E0 "v[34:35]"
#define LL_E0L "v34"
#define LL_E0H "v35"
#define LL_E1 "v[36:37]"
#define LL_E1L "v36"
#define LL_E1H "v37"
#define LL_E2 "v[38:39]"
#define LL_E2L "v38"
#define LL_E2H "v39"
#define LL_E3 "v[40:41]"
#define LL_E3L "v40"
#define LL_E3H "v41"
#define LL_E4 "v[42:43]"
#define LL_E4L "v42"
#define LL_E4H "v43"
#define LL_E5 "v[44:45]"
#define LL_E5L "v44"
#define LL_E5H "v45"
#define LL_E6 "v[46:47]"
#define LL_E6L "v46"
#define LL_E6H "v47"
#define LL_E7 "v[48:49]"
#define LL_E7L "v48"
#define LL_E7H "v49"
#define LL_LOW "v30"    // RangeEnc::low and ::range live in ONE register pair (an operand tied to it): a renormalisation
#define LL_RANGE "v31"  // shifts both with one 64-bit shift -- `low` is below 2^25, a carry out of its held byte is
#define LL_LR "v[30:31]"  // cleared by the subroutine before, so nothing of it reaches `range` -- and masks the old held byte away
#define LL_R1 "v50"    // range * P >> 8
#define LL_BITS "v51"  // mantissa bits still to code, left-aligned, sentinel 1 behind them
#define LL_N "v51"     // ones of the unary tail (ex - 3); the tail is over before the mantissa bits are formed
#define LL_OFF "v52"   // table offset of a successor entry
#define LL_T "v53"     // short-lived temporary (never live across a renormalisation)
#define LL_CK "v34"    // carry subroutine: LDS address walked backwards (slot 0's entry is dead by its first call)
#define LL_CT "v35"
#define LL_CT2 "v50"   // (r1 is dead in every renormalisation)
// scalar registers owned by the block
#define LL_SX "s[36:37]"  // exec at entry
#define LL_SA "s[38:39]"  // lanes with a non-zero residual
#define LL_ST "s[40:41]"  // exec saved around a bit-1 patch
#define LL_SW "s[42:43]"  // exec saved around a renormalisation
#define LL_SB "s[44:45]"  // lanes that code slot 5 (ex > 0)
#define LL_SU "s[46:47]"  // lanes that entered the unary tail (ex > 2)
#define LL_SM "s[56:57]"  // lanes with a mantissa run (ex > 1)
#define LL_SC "s[40:41]"  // carry subroutine (no patch is open during a renormalisation)
#define LL_SD "s[48:49]"
#define LL_SE "s[50:51]"
#define LL_SR "s[52:53]"  // return address of the carry subroutine
#define LL_SRL "s52"
#define LL_SRH "s53"
#define LL_SI "s54"       // bin counter of the unary tail
#define LL_S1 "s52"
#define LL_S2 "s53"

// experiments (make exp XFLAGS=-DLLMI_ASM_VAR=n): bit 0 = SDWA shift amounts as inline constants instead of VGPRs,
// bit 1 = no skip branch around a renormalisation that no lane needs, bit 2 = none around a bit-1 patch that no lane needs
#ifndef LLMI_ASM_VAR
#define LLMI_ASM_VAR 0
#endif
#if LLMI_ASM_VAR & 1
#define LL_C3 "3"
#else
#define LL_C3 "%[c3]"
#endif
#if LLMI_ASM_VAR & 4
#define LL_PATCH_SKIP(N)
#else
#define LL_PATCH_SKIP(N) "s_cbranch_execz .Lpatch" N "_%=\n\t"
#endif
#define LL_PATCH_END(N) ".Lpatch" N "_%=:\n\t" "s_mov_b64 exec, " LL_ST "\n\t"
#if LLMI_ASM_VAR & 2
#define LL_RENORM_SKIP(N)
#else
#define LL_RENORM_SKIP(N) "s_cbranch_execz .Lskip" N "_%=\n\t"
#endif
// range -= (r1 = range * P(entry) >> 8): the outcome of a 0
#define LL_SPLIT(EL)                                                                                              \
    "v_mul_u32_u24_sdwa " LL_R1 ", " EL ", " LL_RANGE " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_lshrrev_b32_e32 " LL_R1 ", 8, " LL_R1 "\n\t"                                                               \
    "v_sub_u32_e32 " LL_RANGE ", " LL_RANGE ", " LL_R1 "\n\t"
// the lanes in exec coded a 1 instead
#define LL_ONE                                          \
    "v_add_u32_e32 " LL_LOW ", " LL_LOW ", " LL_RANGE "\n\t"        \
    "v_mov_b32_e32 " LL_RANGE ", " LL_R1 "\n\t"
// Renormalisation site N of the lanes in exec (llcomp.hpp:62-72; one step always suffices).  The byte held in bits 16..23 of
// `low` goes to the staging area; a carry that reached a held 0xFF (bit 24) is the rare case and a subroutine.
#define LL_RENORM(N)                                                                                               \
    "v_cmp_gt_u32_e32 vcc, %[c100], " LL_RANGE "\n\t"                                                                  \
    "s_and_saveexec_b64 " LL_SW ", vcc\n\t"                                                                        \
    LL_RENORM_SKIP(N)                                                                                              \
    "ds_write_b8_d16_hi %[wp], " LL_LOW "\n\t"                                                                         \
    "v_cmp_lt_u32_e32 vcc, %[cwrap], " LL_LOW "\n\t"                                                                   \
    "s_cbranch_vccnz .Lrare" N "_%=\n"                                                                             \
    ".Lback" N "_%=:\n\t"                                                                                          \
    "v_add_u32_e32 %[wp], 1, %[wp]\n\t"                                                                            \
    "v_lshlrev_b64 " LL_LR ", 8, " LL_LR "\n\t"                                                                     \
    "v_and_b32_e32 " LL_LOW ", 0xffffff, " LL_LOW "\n"                                                                     \
    ".Lskip" N "_%=:\n\t"                                                                                          \
    "s_mov_b64 exec, " LL_SW "\n\t"
#define LL_RARE_STUB(N)                     \
    ".Lrare" N "_%=:\n\t"                   \
    "s_getpc_b64 " LL_SR "\n\t"             \
    "s_branch .Lcarry_%=\n\t"               \
    "s_branch .Lback" N "_%=\n"
// entries of slots 1..7 (their addresses are formed in the registers that receive them)
#define LL_FETCH_REST                                                                                                         \
    "v_lshlrev_b32_sdwa " LL_E1L ", " LL_C3 ", " LL_B0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"      \
    "v_lshlrev_b32_sdwa " LL_E2L ", " LL_C3 ", " LL_B0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"      \
    "v_lshlrev_b32_sdwa " LL_E3L ", " LL_C3 ", " LL_B0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"      \
    "v_lshlrev_b32_sdwa " LL_E4L ", " LL_C3 ", " LL_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"      \
    "ds_read_b64 " LL_E1 ", " LL_E1L "\n\t"                                                                                   \
    "ds_read_b64 " LL_E2 ", " LL_E2L "\n\t"                                                                                   \
    "ds_read_b64 " LL_E3 ", " LL_E3L "\n\t"                                                                                   \
    "ds_read_b64 " LL_E4 ", " LL_E4L "\n\t"                                                                                   \
    "v_lshlrev_b32_sdwa " LL_E5L ", " LL_C3 ", " LL_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"      \
    "v_lshlrev_b32_sdwa " LL_E6L ", " LL_C3 ", " LL_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"      \
    "v_lshlrev_b32_sdwa " LL_E7L ", " LL_C3 ", " LL_B1 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"      \
    "ds_read_b64 " LL_E5 ", " LL_E5L "\n\t"                                                                                   \
    "ds_read_b64 " LL_E6 ", " LL_E6L "\n\t"                                                                                   \
    "ds_read_b64 " LL_E7 ", " LL_E7L "\n\t"

// What the sample loop keeps in registers for the block next to the coder's low / range / write pointer.
struct EncRowsExtra {
    uint32_t pend;      // lane flag: a carry has to go on into the bytes already flushed to HBM (resolved by the caller)
    uint32_t hot;       // wave-uniform: most lanes had a non-zero residual last time (entries of slots 1..7 up front)
    uint32_t any_pend;  // wave-uniform: some lane set `pend`
};

#define LL_SNAP 0
#include "enc_sample_asm.inc"
#undef LL_SNAP
#define LL_SNAP 1
#include "enc_sample_asm.inc"
#undef LL_SNAP

}  // namespace llcomp_mi
