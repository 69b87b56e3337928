// Host-side check of llcomp_amd/csrc/walk_tables.hpp (plain constexpr C++, no HIP): for every residual -510..510 and every
// state 0..127 of every slot, the table-driven step of the 2-D encoder's snapshot walk must equal the state machine replayed
// over the bins that putSymbol<true,4,6,7> codes on that slot (llcomp.hpp:166-206, 283-293 as restated in tables.hpp).
// Prints "ok <cases>" or the first mismatch.   g++ -std=c++17 -O1 -I llcomp_amd/csrc tests/helpers/walk_tables_check.cpp
#include <cstdio>
#include <vector>

#include "walk_tables.hpp"

using namespace llcomp_mi;

struct Bin { int slot, bit; };

static std::vector<Bin> bins_of(int res) {  // the reference's binarisation, restated
    std::vector<Bin> b;
    b.push_back({0, res == 0});
    if (res == 0) return b;
    const unsigned a = unsigned(res < 0 ? -res : res);
    int ex = 0;
    while ((a >> (ex + 1)) != 0) ++ex;
    for (int i = 0; i < ex; ++i) b.push_back({i + 1 < 4 ? i + 1 : 4, 1});
    b.push_back({ex + 1 < 4 ? ex + 1 : 4, 0});
    for (int i = ex - 1; i >= 0; --i) b.push_back({(ex - 1 - i) + 5 < 6 ? (ex - 1 - i) + 5 : 6, int((a >> i) & 1u)});
    b.push_back({7, res < 0});
    return b;
}

int main() {
    static const WalkTables t = make_walk_tables();
    long cases = 0;
    for (int res = -510; res <= 510; ++res) {
        const std::vector<Bin> bins = bins_of(res);
        const uint32_t c = t.codes[uint32_t(res) & 1023u];
        if (c != walk_codes(res)) { std::printf("codes table differs at %d\n", res); return 1; }
        for (uint32_t s = 0; s < 128; ++s) {
            uint32_t want[8];
            for (int k = 0; k < 8; ++k) want[k] = s;
            for (const Bin& bn : bins) want[bn.slot] = state_next(want[bn.slot], uint32_t(bn.bit));
            uint32_t got[8];
            got[0] = t.once[s * kWalkOnceStride + (c & 3u)];
            got[1] = t.once[s * kWalkOnceStride + ((c >> 2) & 3u)];
            got[2] = t.once[s * kWalkOnceStride + ((c >> 4) & 3u)];
            got[3] = t.once[s * kWalkOnceStride + ((c >> 6) & 3u)];
            got[4] = t.unary[s * kWalkUnaryStride + ((c >> 8) & 7u)];
            got[5] = t.once[s * kWalkOnceStride + ((c >> 11) & 3u)];
            got[6] = t.bits[s * kWalkBitsStride + ((c >> 13) & 31u)];
            got[6] = t.bits[got[6] * kWalkBitsStride + ((c >> 18) & 31u)];
            got[7] = t.once[s * kWalkOnceStride + ((c >> 23) & 3u)];
            for (int k = 0; k < 8; ++k) {
                ++cases;
                if (got[k] != want[k]) {
                    std::printf("mismatch: residual %d slot %d state %u: table %u, state machine %u\n", res, k, s, got[k], want[k]);
                    return 1;
                }
            }
        }
    }
    std::printf("ok %ld\n", cases);
    return 0;
}
