// dec_rows_asm.hpp -- one sample of the decoder's FAST PATH for 1-row slices (getSymbol<true,4,6,7>, llcomp.hpp:219-247, over
// the range decoder, llcomp.hpp:98-121), hand-written for gfx950.  Same bins, same order, same arithmetic as
// dec_residual<*, false, true> in slice_kernels.hip, which documents the scheme and stays in use as the CHECKED replay (the block
// never looks at the fill level of the window: dec_sample rolls back and replays a sample that emptied it) and for the other
// kernel families.
//
// What the block does that hipcc's structurised code cannot (prices: tools/ubench/valu_rate3):
//   * the lane sets of the unary exponent are NESTED (the lanes that decode a 1 on slot k are the lanes that decode slot k + 1),
//     so exec only shrinks until the phase is over; every lane takes the outcome of a 0 IN PLACE (range = r0, low -= r0: a
//     borrow says the bin is a 0 and the lane drops out; `low` is put back and the refill done once behind the phase), where
//     the compiler keeps the difference apart and selects or copies per bin;
//   * a lane that stays has range = r1, and r1 << 8 of its refill is the product with its low byte masked away (no left shift);
//   * the mantissa run gathers its bits under a MARKER bit that the same add-with-carry pushes out as its carry when the lane's
//     last bit has come in: the carry is the loop's exit mask, no compare per bin;
//   * the "most lanes non-zero" flag is three scalar instructions on masks the block has anyway; no copies at merges; the two
//     run loops start on a 64-byte line of the instruction cache.
//
// LDS contract: the entries of the model table carry absolute LDS addresses of their successors (load_table); the row bank is
// the decoder's wide one (16-bit table addresses, [word 0..3][lane] 256 bytes apart, slot k in half k & 1 of word k / 2).
// Register contract: the window lives in v[46:47] (an operand tied to the pair: the block needs its low half by name), the block
// owns v32..v45, v48..v52 and s56..s71 for its length; with what hipcc needs around it the kernel stays at 56 VGPRs / 80 SGPRs
// (eight wavefronts per SIMD with room to spare).  gfx950 hazards the assembler does not handle inside inline asm, checked by
// hand: a vector instruction that reads VCC follows the vector instruction that wrote it by two wait states (s_nop 1).
#pragma once
#include <cstdint>

namespace llcomp_mi {

#define LD_E0L "v32"
#define LD_E0H "v33"
#define LD_E0 "v[32:33]"
#define LD_E5L "v34"
#define LD_E5H "v35"
#define LD_E5 "v[34:35]"
#define LD_E6L "v36"
#define LD_E6H "v37"
#define LD_E6 "v[36:37]"
#define LD_E7L "v38"
#define LD_E7H "v39"
#define LD_E7 "v[38:39]"
#define LD_E1L "v40"
#define LD_E1H "v41"
#define LD_E1 "v[40:41]"
#define LD_E2L "v42"
#define LD_E2H "v43"
#define LD_E2 "v[42:43]"
#define LD_E3L "v44"
#define LD_E3H "v45"
#define LD_E3 "v[44:45]"
#define LD_E4L "v48"
#define LD_E4H "v49"
#define LD_E4 "v[48:49]"
#define LD_WINL "v46"
#define LD_WIN "v[46:47]"
#define LD_PROD "v50"  // range * P
#define LD_R1 "v51"    // range * P >> 8
#define LD_DIFF "v52"  // low - r0
#define LD_OFF "v52"   // LDS address of a successor entry / of a fetched slot (never live together with the difference)
#define LD_EX "v32"    // exponent                         (slot 0 is over: its entry's registers are free)
#define LD_W "v33"     // mantissa bits gathered so far, inverted, under their leading one
#define LD_RUN "v40"   // mantissa run: marker + inverted bits   (slots 1..3 are over: their entries' registers are free)
#define LD_NX "v41"    //               half-entry the last bin chose
#define LD_T "v42"     // short-lived
#define LD_SX "s[56:57]"  // exec at entry
#define LD_SA "s[58:59]"  // lanes with a non-zero residual
#define LD_SW "s[60:61]"  // exec saved around a refill
#define LD_SP "s[62:63]"  // exec saved around a bit-1 patch
#define LD_S5 "s[64:65]"  // lanes that finish the sample (non-zero residual, valid exponent)
#define LD_SU "s[66:67]"  // lanes of the run on slot 4; later: lanes that decode slot 5 (exponent > 0)
#define LD_SD "s[68:69]"  // mantissa run: lanes whose last bit has just come in
#define LD_S1L "s[68:69]"  // lanes with an exponent > 0 (the lanes that stayed behind slot 1); shares LD_SD: read before the run
#define LD_SM "s[70:71]"   // lanes with an exponent > 1; shares LD_S1 / LD_S2: read before the end
#define LD_S1 "s70"
#define LD_S2 "s71"

// r1 = range * P(entry) >> 8, range = r0 = range - r1: the outcome of a 0, in place
#define LD_SPLIT(EL)                                                                                                       \
    "v_mul_u32_u24_sdwa " LD_PROD ", " EL ", %[range] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_lshrrev_b32_e32 " LD_R1 ", 8, " LD_PROD "\n\t"                                                                      \
    "v_sub_u32_e32 %[range], %[range], " LD_R1 "\n\t"
// refill (llcomp.hpp:115-120) of the lanes in exec
#define LD_REFILL(N)                                         \
    "v_cmp_gt_u32_e32 vcc, %[c100], %[range]\n\t"            \
    "s_and_saveexec_b64 " LD_SW ", vcc\n\t"                  \
    "s_cbranch_execz .Lrf" N "_%=\n\t"                       \
    "v_lshlrev_b32_e32 %[range], 8, %[range]\n\t"            \
    "v_perm_b32 %[low], %[low], %[w32], %[sel]\n\t"          \
    "v_lshrrev_b32_e32 %[w32], 8, %[w32]\n"                   \
    ".Lrf" N "_%=:\n\t"                                      \
    "s_mov_b64 exec, " LD_SW "\n\t"
// the same for lanes that have just taken r1 as their range: r1 << 8 = the product without its low byte
#define LD_REFILL_R1(N)                                           \
    "v_cmp_gt_u32_e32 vcc, %[c100], " LD_R1 "\n\t"                \
    "s_and_saveexec_b64 " LD_SW ", vcc\n\t"                       \
    "s_cbranch_execz .Lrf" N "_%=\n\t"                            \
    "v_and_b32_e32 %[range], 0xffffff00, " LD_PROD "\n\t"         \
    "v_perm_b32 %[low], %[low], %[w32], %[sel]\n\t"               \
    "v_lshrrev_b32_e32 %[w32], 8, %[w32]\n"                        \
    ".Lrf" N "_%=:\n\t"                                           \
    "s_mov_b64 exec, " LD_SW "\n\t"
// one bin of the nested unary prefix (slots 1..3): a 1 = the lane stays
#define LD_UNARY(EL, EH, OFS, N, SAVE)                                      \
    LD_SPLIT(EL)                                                            \
    "v_sub_co_u32_e32 %[low], vcc, %[low], %[range]\n\t"                    \
    "ds_write_b16_d16_hi %[bank], " EL " offset:" OFS "\n\t"                \
    "s_andn2_b64 exec, exec, vcc\n\t"                                       \
    SAVE                                                                    \
    "s_cbranch_execz .Lx_done_%=\n\t"                                       \
    "ds_write_b16_d16_hi %[bank], " EH " offset:" OFS "\n\t"                \
    "v_add_u32_e32 " LD_EX ", 1, " LD_EX "\n\t"                                     \
    "v_mov_b32_e32 %[range], " LD_R1 "\n\t"                                 \
    LD_REFILL_R1(N)
// entries of slots 1..7 from the bank words (addresses by mask / constant right shift)
#define LD_FETCH_REST                                         \
    "v_lshrrev_b32_e32 " LD_E1L ", 16, %[w0]\n\t"             \
    "v_and_b32_e32 " LD_E2L ", 0xffff, %[w1]\n\t"             \
    "v_lshrrev_b32_e32 " LD_E3L ", 16, %[w1]\n\t"             \
    "v_and_b32_e32 " LD_E4L ", 0xffff, %[w2]\n\t"             \
    "ds_read_b64 " LD_E1 ", " LD_E1L "\n\t"                   \
    "ds_read_b64 " LD_E2 ", " LD_E2L "\n\t"                   \
    "ds_read_b64 " LD_E3 ", " LD_E3L "\n\t"                   \
    "ds_read_b64 " LD_E4 ", " LD_E4L "\n\t"                   \
    "v_lshrrev_b32_e32 " LD_E5L ", 16, %[w2]\n\t"             \
    "v_and_b32_e32 " LD_E6L ", 0xffff, %[w3]\n\t"             \
    "v_lshrrev_b32_e32 " LD_E7L ", 16, %[w3]\n\t"             \
    "ds_read_b64 " LD_E5 ", " LD_E5L "\n\t"                   \
    "ds_read_b64 " LD_E6 ", " LD_E6L "\n\t"                   \
    "ds_read_b64 " LD_E7 ", " LD_E7L "\n\t"

// in: exec = the lanes of the wavefront's slices; low / range / win as in RangeDec; w0..w3 = the context's wide row bank (read by
// the caller, who also needs it to roll back), bank = its LDS address; hot = wave-uniform "most lanes had a non-zero residual
// last time" (in: entries of slots 1..7 are requested up front; out: the new flag).  out: value = the decoded residual's
// magnitude with the sign bin applied (0 for a zero residual); w32 = what is left of the sample's private copy of the window's
// next three bytes (refills take their byte from it and shift it down by a 2-cycle 32-bit shift; the 64-bit window moves once,
// behind the sample, by the bytes that went): 0 = the sample wanted more than the three, or its exponent exceeded 31 ("Invalid
// exponent", llcomp.hpp:230-235: the lane stops behind the exponent) -- the caller replays that lane on the checked path.
__device__ __forceinline__ void dec_rows_sample_asm(uint32_t& low, uint32_t& range, unsigned long long& win, uint32_t w0, uint32_t w1,
                                                    uint32_t w2, uint32_t w3, uint32_t bank, uint32_t& hot, uint32_t& value, uint32_t& w32) {
    asm volatile(
        "s_mov_b64 " LD_SX ", exec\n\t"
        "v_and_b32_e32 %[w32], 0xffffff, " LD_WINL "\n\t"  // the sample's bytes: the window's next three under a sentinel 1
        "v_or_b32_e32 %[w32], 0x1000000, %[w32]\n\t"
        "v_and_b32_e32 " LD_OFF ", 0xffff, %[w0]\n\t"
        "ds_read_b64 " LD_E0 ", " LD_OFF "\n\t"
        "v_mov_b32_e32 %[value], 0\n\t"
        "s_cmp_lg_u32 %[hot], 0\n\t"
        "s_cbranch_scc0 .Lcold_%=\n\t"
        LD_FETCH_REST
        "s_waitcnt lgkmcnt(7)\n\t"
        "s_branch .Lzero_%=\n"
        ".Lcold_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        // ---- slot 0: a 1 = the residual is zero (the lane is done), a 0 (borrow) = it goes on
        ".Lzero_%=:\n\t"
        LD_SPLIT(LD_E0L)
        "v_sub_co_u32_e32 " LD_DIFF ", vcc, %[low], %[range]\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E0L "\n\t"
        "s_and_b64 " LD_SA ", exec, vcc\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lz1_%=\n\t"
        "v_mov_b32_e32 %[low], " LD_DIFF "\n\t"
        "v_mov_b32_e32 %[range], " LD_R1 "\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E0H "\n"
        ".Lz1_%=:\n\t"
        "s_mov_b64 exec, " LD_SX "\n\t"
        LD_REFILL("0")
        "s_and_b64 exec, " LD_SA ", " LD_SA "\n\t"
        "s_cbranch_execz .Ldone_%=\n\t"
        "s_cmp_lg_u32 %[hot], 0\n\t"
        "s_cbranch_scc1 .Lhave_%=\n\t"
        LD_FETCH_REST
        ".Lhave_%=:\n\t"
        "v_mov_b32_e32 " LD_EX ", 0\n\t"
        "v_mov_b32_e32 " LD_W ", 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        // ---- unary exponent: slots 1, 2, 3 once each, then a run on slot 4
        // (the lanes that stay behind slots 1 and 2 are the lanes of slot 5 and of the mantissa run: kept, not compared for again)
        "s_mov_b64 " LD_SM ", 0\n\t"
        LD_UNARY(LD_E1L, LD_E1H, "2", "1", "s_mov_b64 " LD_S1L ", exec\n\t")
        LD_UNARY(LD_E2L, LD_E2H, "256", "2", "s_mov_b64 " LD_SM ", exec\n\t")
        LD_UNARY(LD_E3L, LD_E3H, "258", "3", "")
        "s_mov_b64 " LD_SU ", exec\n\t"
        ".p2align 6\n"
        ".Lu_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        LD_SPLIT(LD_E4L)
        "v_sub_co_u32_e32 %[low], vcc, %[low], %[range]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lu_done_%=\n\t"
        "v_lshrrev_b32_e32 " LD_OFF ", 16, " LD_E4H "\n\t"
        "ds_read_b64 " LD_E4 ", " LD_OFF "\n\t"
        "v_add_u32_e32 " LD_EX ", 1, " LD_EX "\n\t"
        "v_mov_b32_e32 %[range], " LD_R1 "\n\t"
        LD_REFILL_R1("4")
        "s_branch .Lu_%=\n"
        ".Lu_done_%=:\n\t"
        "s_mov_b64 exec, " LD_SU "\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E4L " offset:512\n"  // slot 4: the closing 0's successor
        ".Lx_done_%=:\n\t"
        "s_mov_b64 exec, " LD_SA "\n\t"
        "v_add_u32_e32 %[low], %[low], %[range]\n\t"  // every lane left by a borrow, with low - r0: put r0 back ...
        LD_REFILL("5")                                // ... and refill behind the closing 0
        // ---- exponent > 31: "Invalid exponent" -- the lane stops here (rare; the caller replays it)
        "v_cmp_lt_u32_e32 vcc, 31, " LD_EX "\n\t"
        "s_cbranch_vccz .Lexok_%=\n\t"
        "s_and_saveexec_b64 " LD_SP ", vcc\n\t"
        "v_mov_b32_e32 %[w32], 0\n\t"  // used-up bytes (not even the sentinel left) are what makes the caller replay a sample
        "s_andn2_b64 exec, " LD_SP ", exec\n\t"
        "s_cbranch_execz .Ldone_%=\n"
        ".Lexok_%=:\n\t"
        "s_mov_b64 " LD_S5 ", exec\n\t"  // (the lanes that finish the sample: value and sign below)
        // ---- slot 5: the mantissa bit below the leading one (exponent > 0); w = 2 + inverted bit
        "s_and_b64 exec, exec, " LD_S1L "\n\t"  // exponent > 0
        "s_cbranch_execz .Lvalue_%=\n\t"
        "s_mov_b64 " LD_SU ", exec\n\t"
        LD_SPLIT(LD_E5L)
        "v_sub_co_u32_e32 " LD_DIFF ", vcc, %[low], %[range]\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E5L " offset:514\n\t"
        "s_andn2_b64 " LD_SP ", exec, vcc\n\t"                 // the lanes that decoded a 1
        "v_addc_co_u32_e32 " LD_W ", vcc, " LD_W ", " LD_W ", vcc\n\t"    // 1 + 1 + borrow
        "v_mov_b32_e32 " LD_RUN ", 0\n\t"
        "s_and_b64 exec, " LD_SP ", " LD_SP "\n\t"
        "s_cbranch_execz .Lp5_%=\n\t"
        "v_mov_b32_e32 %[low], " LD_DIFF "\n\t"
        "v_mov_b32_e32 %[range], " LD_R1 "\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E5H " offset:514\n"
        ".Lp5_%=:\n\t"
        "s_mov_b64 exec, " LD_SU "\n\t"
        LD_REFILL("6")
        // ---- the rest of the mantissa as a run on slot 6 (exponent > 1): bits gathered under a marker that leaves as a carry
        "v_add_u32_e32 " LD_T ", -1, " LD_EX "\n\t"  // m = bins of the run
        "s_and_b64 exec, exec, " LD_SM "\n\t"  // exponent > 1
        "s_cbranch_execz .Lm_skip_%=\n\t"
        "s_mov_b64 " LD_SP ", exec\n\t"
        "v_add_u32_e32 " LD_NX ", -2, " LD_EX "\n\t"
        "v_lshrrev_b32_e32 " LD_RUN ", " LD_NX ", %[sentv]\n\t"  // marker at bit 32 - m
        ".p2align 6\n"
        ".Lm_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        LD_SPLIT(LD_E6L)
        "v_sub_co_u32_e32 " LD_DIFF ", vcc, %[low], %[range]\n\t"
        "s_nop 1\n\t"  // (VALU wrote VCC, VALU reads VCC: two wait states on gfx950)
        "v_cndmask_b32_e32 %[range], " LD_R1 ", %[range], vcc\n\t"
        "v_cndmask_b32_e32 %[low], " LD_DIFF ", %[low], vcc\n\t"
        "v_cndmask_b32_e32 " LD_NX ", " LD_E6H ", " LD_E6L ", vcc\n\t"
        "v_addc_co_u32_e32 " LD_RUN ", vcc, " LD_RUN ", " LD_RUN ", vcc\n\t"  // 2 run + borrow (inverted bits); carry out: this was the last bit
        "s_mov_b64 " LD_SD ", vcc\n\t"
        "v_lshrrev_b32_e32 " LD_OFF ", 16, " LD_NX "\n\t"
        "ds_read_b64 " LD_E6 ", " LD_OFF "\n\t"
        "v_cmp_gt_u32_e32 vcc, %[c100], %[range]\n\t"
        "s_and_saveexec_b64 " LD_SW ", vcc\n\t"
        "s_cbranch_execz .Lmr_%=\n\t"
        "v_lshlrev_b32_e32 %[range], 8, %[range]\n\t"
        "v_perm_b32 %[low], %[low], %[w32], %[sel]\n\t"
        "v_lshrrev_b32_e32 %[w32], 8, %[w32]\n"
        ".Lmr_%=:\n\t"
        "s_andn2_b64 exec, " LD_SW ", " LD_SD "\n\t"  // everybody back, minus the lanes that are done
        "s_cbranch_execnz .Lm_%=\n\t"
        "s_mov_b64 exec, " LD_SP "\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_NX " offset:768\n"  // slot 6: the successor the last bin chose
        ".Lm_skip_%=:\n\t"
        "s_mov_b64 exec, " LD_SU "\n\t"
        "v_lshlrev_b32_e32 " LD_W ", " LD_T ", " LD_W "\n\t"  // what was gathered before the run moves up to make room for it
        "v_or_b32_e32 " LD_W ", " LD_W ", " LD_RUN "\n"
        // ---- the value: w with the bits below its leading one inverted back (w == 1 for exponent 0)
        ".Lvalue_%=:\n\t"
        "s_mov_b64 exec, " LD_S5 "\n\t"
        "v_lshlrev_b32_e64 " LD_T ", " LD_EX ", 1\n\t"
        "v_add_u32_e32 " LD_T ", -1, " LD_T "\n\t"
        "v_xor_b32_e32 %[value], " LD_W ", " LD_T "\n\t"
        // ---- slot 7: the sign (a 1 = negative)
        LD_SPLIT(LD_E7L)
        "v_sub_co_u32_e32 " LD_DIFF ", vcc, %[low], %[range]\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E7L " offset:770\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lp7_%=\n\t"
        "v_mov_b32_e32 %[low], " LD_DIFF "\n\t"
        "v_mov_b32_e32 %[range], " LD_R1 "\n\t"
        "ds_write_b16_d16_hi %[bank], " LD_E7H " offset:770\n\t"
        "v_sub_u32_e32 %[value], 0, %[value]\n"
        ".Lp7_%=:\n\t"
        "s_mov_b64 exec, " LD_S5 "\n\t"
        LD_REFILL("7")
        ".Ldone_%=:\n\t"
        "s_mov_b64 exec, " LD_SX "\n\t"
        "v_ffbh_u32_e32 " LD_DIFF ", %[w32]\n\t"          // the sentinel has moved down by 8 bits per byte consumed:
        "v_add_u32_e32 " LD_DIFF ", -7, " LD_DIFF "\n\t"  // clz - 7 bits  (a copy that is used up gives nonsense: that lane is replayed)
        "v_lshrrev_b64 " LD_WIN ", " LD_DIFF ", " LD_WIN "\n\t"
        "s_bcnt1_i32_b64 " LD_S1 ", " LD_SA "\n\t"
        "s_bcnt1_i32_b64 " LD_S2 ", " LD_SX "\n\t"
        "s_lshl_b32 " LD_S1 ", " LD_S1 ", 1\n\t"
        "s_cmp_ge_u32 " LD_S1 ", " LD_S2 "\n\t"
        "s_cselect_b32 %[hot], 1, 0\n\t"
        "s_mov_b64 exec, " LD_SX "\n\t"
        : [low] "+v"(low), [range] "+v"(range), "+{v[46:47]}"(win), [hot] "+s"(hot), [value] "=&v"(value), [w32] "=&v"(w32)
        : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [bank] "v"(bank), [c100] "s"(0x100u), [sel] "s"(0x06050400u),
          [sentv] "v"(0x80000000u)
        : "vcc", "scc", "memory", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45",
          "v48", "v49", "v50", "v51", "v52", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66",
          "s67", "s68", "s69", "s70", "s71");
}

}  // namespace llcomp_mi
