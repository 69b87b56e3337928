// container.hpp -- wire-format helpers (host only, no GPU): the reference's 6-byte legacy header
// (/root/reference/llcomp.hpp:375-378, 463-470) and this project's sliced container (include/llcomp_mi.h).
#pragma once
#include <cstddef>
#include <cstdint>

#include "geometry.hpp"

namespace llcomp_mi {

inline void put_u32le(uint8_t* p, uint32_t v) {
    p[0] = uint8_t(v); p[1] = uint8_t(v >> 8); p[2] = uint8_t(v >> 16); p[3] = uint8_t(v >> 24);
}
inline uint32_t get_u32le(const uint8_t* p) {
    return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24);
}

void write_legacy_header(uint8_t* out6, uint32_t w, uint32_t h, uint32_t c);
void write_sliced_header(uint8_t* out24, const Geometry& g);  // g.frames must be 1

}  // namespace llcomp_mi
