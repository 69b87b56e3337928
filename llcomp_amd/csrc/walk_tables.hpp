// walk_tables.hpp -- what ONE SAMPLE does to the eight adaptive states of its context, folded into lookup tables.
//
// The state of slot k of a context only ever sees the bins that putSymbol<true,4,6,7> (llcomp.hpp:166-206) codes on slot k,
// and a state never depends on the range coder.  So the sequence of states a (context, slot) pair runs through inside a slice
// can be replayed WITHOUT coding anything: that is what the snapshot pass of the 2-D encoder does (snapshot_kernels.hip),
// sample by sample in context-sorted order.  Per sample and slot the input is
//   slot 0        one bin: residual == 0                                   (always)
//   slot 1, 7     one bin: exponent > 0 / residual < 0                     (residual != 0)
//   slot 2, 5     one bin: exponent > 1 / the mantissa bit below the leading one   (exponent > 0)
//   slot 3        one bin: exponent > 2                                    (exponent > 1)
//   slot 4        (exponent - 3) ones and a closing zero                   (exponent > 2;  |residual| <= 510: exponent <= 8)
//   slot 6        the remaining exponent - 1 mantissa bits, MSB first      (exponent > 1)
// and the tables below map (state, input code) -> state after the sample in one step:
//   kWalkOnce  [state][code]   code 0 / 1 = that bin, 2 = slot not coded by this sample (identity)
//   kWalkUnary [state][code]   code j = j ones and a zero (0..5), 6 = identity
//   kWalkBits  [state][code]   code = (1 << len) - 1 + bits: `len` (0..4) bins, MSB first; a 7-bit run is two look-ups
//   kWalkCodes [residual & 1023]  the codes of all slots, packed (walk_codes below)
// All of it is constexpr arithmetic on tables.hpp's state machine (cabac::State::update, llcomp.hpp:283-293).
#pragma once
#include <cstdint>

#include "tables.hpp"

namespace llcomp_mi {

constexpr uint32_t kWalkOnceStride = 4, kWalkUnaryStride = 8, kWalkBitsStride = 32;

struct WalkTables {
    uint8_t once[128 * kWalkOnceStride];
    uint8_t unary[128 * kWalkUnaryStride];
    uint8_t bits[128 * kWalkBitsStride];
    uint32_t codes[1024];
};

// packed codes of one residual:  bits 0..1 slot 0 | 2..3 slot 1 | 4..5 slot 2 | 6..7 slot 3 | 8..10 slot 4 | 11..12 slot 5 |
//                                13..17 slot 6, first look-up | 18..22 slot 6, second look-up | 23..24 slot 7
constexpr uint32_t walk_codes(int res) {
    uint32_t a = uint32_t(res < 0 ? -res : res);
    int ex = -1;
    for (uint32_t t = a; t; t >>= 1) ++ex;  // floor(log2 a), -1 for 0 (llcomp.hpp:132-152)
    const uint32_t c0 = res == 0 ? 1u : 0u;
    const uint32_t c1 = res != 0 ? (ex > 0 ? 1u : 0u) : 2u;
    const uint32_t c2 = ex > 0 ? (ex > 1 ? 1u : 0u) : 2u;
    const uint32_t c3 = ex > 1 ? (ex > 2 ? 1u : 0u) : 2u;
    const uint32_t c4 = ex > 2 ? uint32_t(ex - 3) : 6u;
    const uint32_t c5 = ex > 0 ? ((a >> (ex - 1)) & 1u) : 2u;
    const uint32_t c7 = res != 0 ? (res < 0 ? 1u : 0u) : 2u;
    const uint32_t n6 = ex > 1 ? uint32_t(ex - 1) : 0u;            // bins of the run on slot 6: bits n6-1..0 of a
    const uint32_t m = n6 ? (a & ((1u << n6) - 1u)) : 0u;
    const uint32_t len1 = n6 < 4 ? n6 : 4u, len2 = n6 - len1;       // first look-up: the upper len1 bits, second: the rest
    const uint32_t c6a = (1u << len1) - 1u + (m >> len2);
    const uint32_t c6b = (1u << len2) - 1u + (m & ((1u << len2) - 1u));
    return c0 | c1 << 2 | c2 << 4 | c3 << 6 | c4 << 8 | c5 << 11 | c6a << 13 | c6b << 18 | c7 << 23;
}

constexpr WalkTables make_walk_tables() {
    WalkTables t{};
    for (uint32_t s = 0; s < 128; ++s) {
        t.once[s * kWalkOnceStride + 0] = uint8_t(state_next(s, 0));
        t.once[s * kWalkOnceStride + 1] = uint8_t(state_next(s, 1));
        t.once[s * kWalkOnceStride + 2] = uint8_t(s);
        t.once[s * kWalkOnceStride + 3] = uint8_t(s);
        for (uint32_t j = 0; j < kWalkUnaryStride; ++j) {
            uint32_t x = s;
            if (j < 6) {
                for (uint32_t i = 0; i < j; ++i) x = state_next(x, 1);
                x = state_next(x, 0);
            }
            t.unary[s * kWalkUnaryStride + j] = uint8_t(x);
        }
        for (uint32_t len = 0; len <= 4; ++len)
            for (uint32_t b = 0; b < (1u << len); ++b) {
                uint32_t x = s;
                for (uint32_t i = 0; i < len; ++i) x = state_next(x, (b >> (len - 1 - i)) & 1u);
                t.bits[s * kWalkBitsStride + (1u << len) - 1u + b] = uint8_t(x);
            }
        t.bits[s * kWalkBitsStride + 31] = uint8_t(s);
    }
    for (int r = -512; r < 512; ++r) t.codes[uint32_t(r) & 1023u] = walk_codes(r);
    return t;
}

}  // namespace llcomp_mi
