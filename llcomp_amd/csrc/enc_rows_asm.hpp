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
// one bin of the nested unary prefix on slot K (1..3): the lanes in exec code "ex > K-1"; those with a 1 stay
#define LL_UNARY(K, KM1, EL, EH, N, SAVE)                  \
    LL_SPLIT(EL)                                           \
    "ds_write_b8 %[bank], " EL " offset:" K "\n\t"         \
    "v_cmp_lt_u32_e32 vcc, " KM1 ", " LL_EX "\n\t"         \
    "s_and_b64 exec, exec, vcc\n\t"                        \
    SAVE                                                   \
    "s_cbranch_execz .Lunary_done_%=\n\t"                  \
    LL_ONE                                                 \
    "ds_write_b8 %[bank], " EH " offset:" K "\n\t"         \
    LL_RENORM(N)
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

// Codes residual `res` in the context whose 8 state bytes sit at LDS address `bank`; low_range = RangeEnc::low | ::range << 32,
// wp as in RangeEnc (slice_kernels.hip; all LDS addresses are byte addresses), `base` = LDS address of the lane's staging area (carries walk
// back to it).
__device__ __forceinline__ void enc_rows_sample_asm(unsigned long long& low_range, uint32_t& wp, EncRowsExtra& x, uint32_t bank,
                                                    int res, uint32_t base) {
    asm volatile(
        "s_mov_b64 " LL_SX ", exec\n\t"
        "ds_read2st64_b32 " LL_B ", %[bank] offset1:1\n\t"
        "s_cmp_lg_u32 %[hot], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshlrev_b32_sdwa " LL_E0L ", " LL_C3 ", " LL_B0 " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "ds_read_b64 " LL_E0 ", " LL_E0L "\n\t"
        "s_cbranch_scc0 .Lcold_%=\n\t"
        LL_FETCH_REST
        "s_waitcnt lgkmcnt(7)\n\t"
        "s_branch .Lzero_%=\n"
        ".Lcold_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        // ---- slot 0: "the residual is zero" (a 1 for the lanes whose residual IS zero)
        ".Lzero_%=:\n\t"
        LL_SPLIT(LL_E0L)
        "ds_write_b8 %[bank], " LL_E0L "\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[res]\n\t"
        "s_andn2_b64 " LL_SA ", exec, vcc\n\t"  // the lanes with a non-zero residual
        "s_and_saveexec_b64 " LL_ST ", vcc\n\t"
        LL_PATCH_SKIP("0")
        LL_ONE
        "ds_write_b8 %[bank], " LL_E0H "\n"
        LL_PATCH_END("0")
        LL_RENORM("0")
        "s_and_b64 exec, " LL_SA ", " LL_SA "\n\t"
        "s_cbranch_execz .Ldone_%=\n\t"
        "s_mov_b64 " LL_SB ", 0\n\t"  // (slot 1 may not be reached by anybody: nobody codes slot 5 then)
        "s_mov_b64 " LL_SM ", 0\n\t"
        "s_cmp_lg_u32 %[hot], 0\n\t"
        "s_cbranch_scc1 .Lhave_%=\n\t"
        LL_FETCH_REST
        ".Lhave_%=:\n\t"
        "v_sub_u32_e32 " LL_A ", 0, %[res]\n\t"
        "v_max_i32_e32 " LL_A ", %[res], " LL_A "\n\t"
        "v_ffbh_u32_e32 " LL_EX ", " LL_A "\n\t"
        "v_sub_u32_e32 " LL_EX ", 31, " LL_EX "\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // ---- unary exponent: slots 1, 2, 3 once each, then a run on slot 4; exec only shrinks until .Lunary_done
        // (the lane sets of slots 2 and 3 are the lane sets of slot 5 and of the mantissa run: kept, not compared for again)
        LL_UNARY("1", "0", LL_E1L, LL_E1H, "1", "s_mov_b64 " LL_SB ", exec\n\t")
        LL_UNARY("2", "1", LL_E2L, LL_E2H, "2", "s_mov_b64 " LL_SM ", exec\n\t")
        LL_UNARY("3", "2", LL_E3L, LL_E3H, "3", "")
        "s_mov_b64 " LL_SU ", exec\n\t"
        "v_add_u32_e32 " LL_N ", -3, " LL_EX "\n\t"
        "s_mov_b32 " LL_SI ", 0\n\t"
        ".p2align 6\n"  // (the two run loops start on a 64-byte line of the instruction cache)
        ".Ltail_%=:\n\t"
        LL_SPLIT(LL_E4L)
        "v_cmp_lt_u32_e32 vcc, " LL_SI ", " LL_N "\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Ltail_done_%=\n\t"
        "v_lshrrev_b32_e32 " LL_OFF ", 16, " LL_E4H "\n\t"
        "ds_read_b64 " LL_E4 ", " LL_OFF "\n\t"
        LL_ONE
        LL_RENORM("4")
        "s_add_i32 " LL_SI ", " LL_SI ", 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_branch .Ltail_%=\n"
        ".Ltail_done_%=:\n\t"
        "s_mov_b64 exec, " LL_SU "\n\t"
        "ds_write_b8 %[bank], " LL_E4L " offset:256\n"
        ".Lunary_done_%=:\n\t"
        "s_mov_b64 exec, " LL_SA "\n\t"
        LL_RENORM("5")  // of the lanes whose last unary bin was the closing zero
        // ---- slot 5: the mantissa bit below the leading one, then the rest of the mantissa as a run on slot 6
        "s_and_b64 exec, " LL_SB ", " LL_SB "\n\t"  // the lanes with an exponent > 0
        "s_cbranch_execz .Lsign_%=\n\t"
        LL_SPLIT(LL_E5L)
        "ds_write_b8 %[bank], " LL_E5L " offset:257\n\t"
        "v_lshl_or_b32 " LL_BITS ", " LL_A ", 1, 1\n\t"
        "v_sub_u32_e32 " LL_T ", 31, " LL_EX "\n\t"
        "v_lshlrev_b32_e32 " LL_BITS ", " LL_T ", " LL_BITS "\n\t"
        "v_add_co_u32_e32 " LL_BITS ", vcc, " LL_BITS ", " LL_BITS "\n\t"
        "s_and_saveexec_b64 " LL_ST ", vcc\n\t"
        LL_PATCH_SKIP("5")
        LL_ONE
        "ds_write_b8 %[bank], " LL_E5H " offset:257\n"
        LL_PATCH_END("5")
        LL_RENORM("6")
        ".p2align 6\n"
        ".Lman_%=:\n\t"
        "v_cmp_ne_u32_e32 vcc, %[sent], " LL_BITS "\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lman_done_%=\n\t"
        "v_add_co_u32_e32 " LL_BITS ", vcc, " LL_BITS ", " LL_BITS "\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        LL_SPLIT(LL_E6L)
        "v_lshrrev_b32_e32 " LL_OFF ", 16, " LL_E6L "\n\t"
        "s_and_saveexec_b64 " LL_ST ", vcc\n\t"
        LL_PATCH_SKIP("6")
        LL_ONE
        "v_lshrrev_b32_e32 " LL_OFF ", 16, " LL_E6H "\n"
        LL_PATCH_END("6")
        "ds_read_b64 " LL_E6 ", " LL_OFF "\n\t"
        LL_RENORM("7")
        "s_branch .Lman_%=\n"
        // ---- out of line: the carry subroutine and its call stubs
        LL_RARE_STUB("0") LL_RARE_STUB("1") LL_RARE_STUB("2") LL_RARE_STUB("3") LL_RARE_STUB("4")
        LL_RARE_STUB("5") LL_RARE_STUB("6") LL_RARE_STUB("7") LL_RARE_STUB("8")
        // exec = the lanes that renormalise, vcc = those of them whose held 0xFF took a carry: +1 into the bytes before the
        // one just stored (llcomp.hpp:40-57 resolved eagerly), walking back through the staging area; what would go on
        // below it (bytes already in HBM) is left to the caller: `pend`.
        ".Lcarry_%=:\n\t"
        "s_and_saveexec_b64 " LL_SC ", vcc\n\t"
        "v_add_u32_e32 " LL_CK ", -1, %[wp]\n"
        ".Lcloop_%=:\n\t"
        "v_cmp_lt_u32_e32 vcc, " LL_CK ", %[base]\n\t"
        "s_and_b64 " LL_SD ", exec, vcc\n\t"
        "s_cbranch_scc0 .Lcin_%=\n\t"
        "s_mov_b64 " LL_SE ", exec\n\t"
        "s_mov_b64 exec, " LL_SD "\n\t"
        "v_mov_b32_e32 %[pend], 1\n\t"
        "s_mov_b32 %[anyp], 1\n\t"
        "s_andn2_b64 exec, " LL_SE ", " LL_SD "\n\t"
        "s_cbranch_execz .Lcdone_%=\n"
        ".Lcin_%=:\n\t"
        "ds_read_u8 " LL_CT ", " LL_CK "\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_u32_e32 " LL_CT2 ", 1, " LL_CT "\n\t"
        "ds_write_b8 " LL_CK ", " LL_CT2 "\n\t"
        "v_add_u32_e32 " LL_CK ", -1, " LL_CK "\n\t"
        "v_cmp_eq_u32_e32 vcc, 0xff, " LL_CT "\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz .Lcloop_%=\n"
        ".Lcdone_%=:\n\t"
        "s_mov_b64 exec, " LL_SC "\n\t"
        "v_and_b32_e32 " LL_LOW ", 0xffffff, " LL_LOW "\n\t"  // the carry is spent (it must not reach `range` in the 64-bit shift)
        "s_add_u32 " LL_SRL ", " LL_SRL ", 4\n\t"
        "s_addc_u32 " LL_SRH ", " LL_SRH ", 0\n\t"
        "s_setpc_b64 " LL_SR "\n"
        // ---- back in line
        ".Lman_done_%=:\n\t"
        "s_mov_b64 exec, " LL_SM "\n\t"  // the lanes with an exponent > 1: those that ran
        "v_lshrrev_b32_e32 " LL_T ", 3, " LL_OFF "\n\t"
        "ds_write_b8 %[bank], " LL_T " offset:258\n"
        // ---- slot 7: the sign
        ".Lsign_%=:\n\t"
        "s_mov_b64 exec, " LL_SA "\n\t"
        LL_SPLIT(LL_E7L)
        "ds_write_b8 %[bank], " LL_E7L " offset:259\n\t"
        "v_cmp_gt_i32_e32 vcc, 0, %[res]\n\t"
        "s_and_saveexec_b64 " LL_ST ", vcc\n\t"
        LL_PATCH_SKIP("7")
        LL_ONE
        "ds_write_b8 %[bank], " LL_E7H " offset:259\n"
        LL_PATCH_END("7")
        LL_RENORM("8")
        ".Ldone_%=:\n\t"
        "s_bcnt1_i32_b64 " LL_S1 ", " LL_SA "\n\t"
        "s_bcnt1_i32_b64 " LL_S2 ", " LL_SX "\n\t"
        "s_lshl_b32 " LL_S1 ", " LL_S1 ", 1\n\t"
        "s_cmp_ge_u32 " LL_S1 ", " LL_S2 "\n\t"
        "s_cselect_b32 %[hot], 1, 0\n\t"
        "s_mov_b64 exec, " LL_SX "\n\t"
        : "+{v[30:31]}"(low_range), [wp] "+v"(wp), [pend] "+v"(x.pend), [hot] "+s"(x.hot), [anyp] "+s"(x.any_pend)
        : [bank] "v"(bank), [res] "v"(res), [base] "v"(base), [c100] "s"(0x100u), [cwrap] "s"(0xFFFFFFu),
          [sent] "s"(0x80000000u), [c3] "v"(3u)
        : "vcc", "scc", "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",
          "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43",
          "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s56", "s57");
}

}  // namespace llcomp_mi
