// tables.hpp -- bitstream constants of the llcomp coding path, in the form the kernels use.
//
// The adaptive binary model (reference: cabac::State + nextStateMps/nextStateLps/stateProbability,
// /root/reference/llcomp.hpp:252-293) is a 128-state machine, state = 2*level + mps, level 0..63:
//   P(bit==1)*256 = mps ? 254 - kLpsProb[level] : kLpsProb[level]
//   bit == mps -> level+1 (saturating at 63);  bit != mps -> level==0 ? flip mps : kLpsFall[level]
// entry_lo/entry_hi fold that into one 8-byte LDS entry per state, arranged so that ONE select on the coded bit yields
// the successor both as a state byte (what a context's bank stores) and as the byte offset of ITS entry in the table
// (pre-scaled, in the upper half: the address of the next entry is one shift by a constant, a 2-cycle operation, where a
// byte extraction + scaling is a 4-cycle SDWA shift in every bin of a run):
//   lo: byte0 next state if bit 0 | byte1 P(this state) | bits 16..31 8 * that next state
//   hi: byte0 next state if bit 1 | byte1 P(this state) | bits 16..31 8 * that next state
#pragma once
#include <cstdint>

namespace llcomp_mi {

constexpr int kContexts = 7926;      // reachable folded contexts 0..7925 (hash multipliers 1,11,121,605,3025)
constexpr int kSlotsPerContext = 8;  // llcomp.hpp:25 substates_nb
constexpr int kMaxBytesPerSample = 13;  // 19 bins * log2(256/6) bits < 104 bits (DESIGN.md "Output bound")

constexpr uint8_t kLpsProb[64] = {
    123, 117, 111, 106, 101, 96, 91, 87, 83, 79, 75, 72, 68, 66, 63, 60, 57, 54, 52, 49, 48, 45,
    43,  41,  40,  38,  36,  35, 33, 32, 30, 30, 28, 27, 26, 25, 24, 23, 22, 21, 21, 20, 19, 18,
    18,  17,  17,  16,  16,  15, 15, 14, 14, 13, 13, 13, 12, 12, 12, 11, 11, 11, 11, 7};
constexpr uint8_t kLpsFall[64] = {
    0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
    18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
    31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 38, 38, 38, 38, 39};

constexpr uint32_t state_prob(uint32_t s) { return (s & 1) ? 254u - kLpsProb[s >> 1] : kLpsProb[s >> 1]; }
constexpr uint32_t state_next(uint32_t s, uint32_t bit) {
    const uint32_t level = s >> 1, mps = s & 1;
    if (bit == mps) return 2 * (level < 63 ? level + 1 : 63) + mps;
    if (level == 0) return mps ^ 1;
    return 2 * kLpsFall[level] + mps;
}
constexpr uint32_t entry_half(uint32_t s, uint32_t bit) { return state_next(s, bit) | (state_prob(s) << 8) | (state_next(s, bit) << 19); }
constexpr uint32_t entry_lo(uint32_t s) { return entry_half(s, 0); }
constexpr uint32_t entry_hi(uint32_t s) { return entry_half(s, 1); }

}  // namespace llcomp_mi
