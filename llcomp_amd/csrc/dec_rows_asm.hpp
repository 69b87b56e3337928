// dec_rows_asm.hpp -- the mantissa run of the decoder for 1-row slices (slot 6 of getSymbol<true,4,6,7>, llcomp.hpp:236-243,
// over the range decoder, llcomp.hpp:98-121), hand-written for gfx950.  Same bins and arithmetic as the loop around
// dec_step_acc in slice_kernels.hip (which the checked replay and the other kernel families keep using).
//
// Why by hand: hipcc's loop ends on a compare of the gathered bits against a per-lane limit -- a 4-cycle operation per bin on
// top of the add-with-carry that gathers them.  Here the register that gathers the bits (at its bottom) carries a marker bit
// above them, placed so that it leaves the register as the carry of that same add-with-carry when the lane's last bit has
// come in; the carry is the loop's exit mask.  (The caller merges the run into what it had gathered before, once per sample.)
//
// LDS contract: the entries of the model table carry absolute LDS addresses of their successors (load_table in
// slice_kernels.hip), so the loop needs no base and the table may sit anywhere.
// Register contract: the window lives in v[46:47] and the current entry in v[48:49] (operands tied to those registers: the
// block needs their halves by name), v50..v54 and s56..s61 are owned by the block; with hipcc's own needs the kernel stays
// at 56 VGPRs / 64 SGPRs (eight wavefronts per SIMD with room to spare).
#pragma once
#include <cstdint>

namespace llcomp_mi {

// in: exec = the lanes with a mantissa run (exponent > 1); low / range as in RangeDec; win = the 64-bit stream window (sentinel
// scheme of RangeDec); cur = entry of slot 6's state; wl = the marker bit (bit 32 - bins of the run), the run's inverted bits
// below it on return.  Returns the
// half-entry that belongs to the last decoded bit (the caller stores the slot's new state from it).  Never looks at the fill
// level of the window (the unchecked fast path of dec_sample).
__device__ __forceinline__ uint32_t dec_rows_mantissa_asm(uint32_t& low, uint32_t& range, unsigned long long& win,
                                                          unsigned long long cur, uint32_t& wl) {
    uint32_t nx;
    asm volatile(
        "s_mov_b64 s[56:57], exec\n\t"
        ".p2align 6\n"
        ".Lm_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_u32_u24_sdwa v50, v48, %[range] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
        "v_lshrrev_b32_e32 v50, 8, v50\n\t"                   // r1
        "v_sub_u32_e32 v51, %[range], v50\n\t"                // r0
        "v_sub_co_u32_e32 v52, vcc, %[low], v51\n\t"          // borrow: the bit is 0
        "s_nop 1\n\t"                                         // (VALU wrote VCC, VALU reads VCC: two wait states on gfx950)
        "v_cndmask_b32_e32 %[range], v50, v51, vcc\n\t"
        "v_cndmask_b32_e32 %[low], v52, %[low], vcc\n\t"
        "v_cndmask_b32_e32 %[nx], v49, v48, vcc\n\t"
        "v_addc_co_u32_e32 %[wl], vcc, %[wl], %[wl], vcc\n\t"  // 2w + borrow (inverted bits); carry out: this was the last bit
        "s_mov_b64 s[58:59], vcc\n\t"
        "v_lshrrev_b32_e32 v53, 16, %[nx]\n\t"
        "ds_read_b64 v[48:49], v53\n\t"                       // successor's entry
        "v_cmp_gt_u32_e32 vcc, %[c100], %[range]\n\t"         // refill (llcomp.hpp:115-120)
        "s_and_saveexec_b64 s[60:61], vcc\n\t"
        "s_cbranch_execz .Lr_%=\n\t"
        "v_lshlrev_b32_e32 %[range], 8, %[range]\n\t"
        "v_perm_b32 %[low], %[low], v46, %[sel]\n\t"          // low << 8 | next byte of the window
        "v_lshrrev_b64 v[46:47], 8, v[46:47]\n"
        ".Lr_%=:\n\t"
        "s_andn2_b64 exec, s[60:61], s[58:59]\n\t"            // everybody back, minus the lanes that are done
        "s_cbranch_execnz .Lm_%=\n\t"
        "s_mov_b64 exec, s[56:57]\n\t"
        : [low] "+v"(low), [range] "+v"(range), [wl] "+v"(wl), [nx] "=&v"(nx), "+{v[46:47]}"(win), "+{v[48:49]}"(cur)
        : [c100] "s"(0x100u), [sel] "s"(0x06050400u)
        : "vcc", "scc", "memory", "v50", "v51", "v52", "v53", "v54", "s56", "s57", "s58", "s59", "s60", "s61");
    return nx;
}

// The whole unary exponent (llcomp.hpp:226-235: slots 1, 2, 3 once each, then a run on slot 4) of the lanes in exec (those with
// a non-zero residual).  The lane sets are nested -- the lanes that decode a 1 on slot k are the lanes that decode slot k + 1 -- so
// exec only shrinks until the phase is over.  Every lane takes the outcome of a 0 IN PLACE (range = r0, low -= r0: a borrow says
// the bin is a 0 and the lane drops out; its `low` is put back and its refill done once behind the phase -- hipcc's code keeps the
// difference apart and selects / copies per bin); the lanes that stay take r1 as the range, count the 1, refill (r1 << 8 is the
// product with its low byte masked away: no left shift) and go on.  New states go to the wide row bank as they are decided (the
// low half's first, the lanes that stay overwrite it with the high half's).  e1..e3 / cur: entries of slots 1..3 / 4.
// out: ex = exponent (ones decoded; the caller rejects > 31), low / range / win as after getSymbol's unary part INCLUDING the
// refill of the closing 0.  No limit per step: a run fed past the window ends by itself when the window's zeros come.
#define LLD_REFILL_R1(N)                                         \
    "v_cmp_gt_u32_e32 vcc, %[c100], v51\n\t"                     \
    "s_and_saveexec_b64 s[60:61], vcc\n\t"                       \
    "s_cbranch_execz .Lxr" N "_%=\n\t"                           \
    "v_and_b32_e32 %[range], 0xffffff00, v50\n\t"                \
    "v_perm_b32 %[low], %[low], v46, %[sel]\n\t"                 \
    "v_lshrrev_b64 v[46:47], 8, v[46:47]\n"                      \
    ".Lxr" N "_%=:\n\t"                                          \
    "s_mov_b64 exec, s[60:61]\n\t"
#define LLD_UNARY(EL, EH, OFS, N)                                                                                     \
    "v_mul_u32_u24_sdwa v50, " EL ", %[range] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_lshrrev_b32_e32 v51, 8, v50\n\t"                                                                               \
    "v_sub_u32_e32 %[range], %[range], v51\n\t"                                                                       \
    "v_sub_co_u32_e32 %[low], vcc, %[low], %[range]\n\t"                                                              \
    "ds_write_b16_d16_hi %[bank], " EL " offset:" OFS "\n\t"                                                          \
    "s_andn2_b64 exec, exec, vcc\n\t"                                                                                 \
    "s_cbranch_execz .Lx_done_%=\n\t"                                                                                 \
    "ds_write_b16_d16_hi %[bank], " EH " offset:" OFS "\n\t"                                                          \
    "v_add_u32_e32 %[ex], 1, %[ex]\n\t"                                                                               \
    "v_mov_b32_e32 %[range], v51\n\t"                                                                                 \
    LLD_REFILL_R1(N)
__device__ __forceinline__ uint32_t dec_rows_exponent_asm(uint32_t& low, uint32_t& range, unsigned long long& win, uint32_t bank,
                                                          unsigned long long e1, unsigned long long e2, unsigned long long e3,
                                                          unsigned long long cur) {
    uint32_t ex;
    asm volatile(
        "s_mov_b64 s[56:57], exec\n\t"
        "v_mov_b32_e32 %[ex], 0\n\t"
        LLD_UNARY("v40", "v41", "2", "1")
        LLD_UNARY("v42", "v43", "256", "2")
        LLD_UNARY("v44", "v45", "258", "3")
        "s_mov_b64 s[58:59], exec\n\t"  // the lanes of the run on slot 4
        ".p2align 6\n"
        ".Lu_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_mul_u32_u24_sdwa v50, v48, %[range] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
        "v_lshrrev_b32_e32 v51, 8, v50\n\t"            // r1
        "v_sub_u32_e32 %[range], %[range], v51\n\t"    // r0
        "v_sub_co_u32_e32 %[low], vcc, %[low], %[range]\n\t"
        "s_andn2_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz .Lu_done_%=\n\t"
        "v_lshrrev_b32_e32 v53, 16, v49\n\t"
        "ds_read_b64 v[48:49], v53\n\t"
        "v_add_u32_e32 %[ex], 1, %[ex]\n\t"
        "v_mov_b32_e32 %[range], v51\n\t"
        LLD_REFILL_R1("4")
        "s_branch .Lu_%=\n"
        ".Lu_done_%=:\n\t"
        "s_mov_b64 exec, s[58:59]\n\t"
        "ds_write_b16_d16_hi %[bank], v48 offset:512\n"  // slot 4: the closing 0's successor
        ".Lx_done_%=:\n\t"
        "s_mov_b64 exec, s[56:57]\n\t"
        "v_add_u32_e32 %[low], %[low], %[range]\n\t"     // every lane left by a borrow, with low - r0: put r0 back ...
        "v_cmp_gt_u32_e32 vcc, %[c100], %[range]\n\t"    // ... and refill behind the closing 0
        "s_and_saveexec_b64 s[60:61], vcc\n\t"
        "s_cbranch_execz .Lxr5_%=\n\t"
        "v_lshlrev_b32_e32 %[range], 8, %[range]\n\t"
        "v_perm_b32 %[low], %[low], v46, %[sel]\n\t"
        "v_lshrrev_b64 v[46:47], 8, v[46:47]\n"
        ".Lxr5_%=:\n\t"
        "s_mov_b64 exec, s[60:61]\n\t"
        : [low] "+v"(low), [range] "+v"(range), [ex] "=&v"(ex), "+{v[46:47]}"(win), "+{v[48:49]}"(cur)
        : [c100] "s"(0x100u), [sel] "s"(0x06050400u), [bank] "v"(bank), "{v[40:41]}"(e1), "{v[42:43]}"(e2), "{v[44:45]}"(e3)
        : "vcc", "scc", "memory", "v50", "v51", "v52", "v53", "v54", "s56", "s57", "s58", "s59", "s60", "s61");
    return ex;
}

}  // namespace llcomp_mi
