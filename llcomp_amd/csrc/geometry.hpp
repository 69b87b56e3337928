// geometry.hpp -- slice geometry shared by host and device code.
//
// A "slice" is the unit the serial entropy coder runs over: a tile_w x tile_h rectangle of one frame, either
// with all channels interleaved in one stream (planar == 0; payload == reference stream of the cropped
// sub-image, llcomp.hpp:390-449) or one colour-transformed channel plane of it (planar == 1).  Slices have
// fresh adaptive state and slice-local border rules, so they are independent: one GPU lane each.
#pragma once
#include <cstdint>
#include <cstdlib>

#if defined(__HIPCC__)
#define LLMI_HD __host__ __device__
#else
#define LLMI_HD
#endif

namespace llcomp_mi {

struct Geometry {
    uint32_t frames, w, h, c;
    uint32_t tile_w, tile_h, planar;
    uint32_t ntx, nty;          // tiles per row / column of one frame
    uint32_t slices_per_frame;  // ntx*nty*(planar ? c : 1)
    uint32_t n_slices;          // frames * slices_per_frame
    uint32_t slice_cap;         // scratch bytes reserved per slice (worst case, multiple of 16)
    uint32_t nch;               // channels coded inside one slice: planar ? 1 : c
    uint32_t lane_shift;        // log2 of the lane-group width (64 lanes, fewer when there are fewer slices)
    uint32_t slice_samples;     // sample capacity of one slice: tile_w * tile_h * nch
};

// lane order: element k of slice `id` inside an array laid out [group][k][group width]
LLMI_HD inline size_t lane_order_index(const Geometry& g, uint32_t id, uint32_t k) {
    const uint32_t gw = 1u << g.lane_shift;
    return ((size_t(id >> g.lane_shift) * g.slice_samples + k) << g.lane_shift) + (id & (gw - 1));
}

struct SliceRect {
    uint32_t frame, x0, y0, sw, sh, ch;  // ch = first channel of the slice (planar) or 0
};

LLMI_HD inline SliceRect slice_rect(const Geometry& g, uint32_t id) {
    SliceRect r;
    r.frame = id / g.slices_per_frame;
    uint32_t s = id - r.frame * g.slices_per_frame;
    uint32_t tile = s;
    r.ch = 0;
    if (g.planar) {
        tile = s / g.c;
        r.ch = s - tile * g.c;
    }
    const uint32_t ty = tile / g.ntx, tx = tile - ty * g.ntx;
    r.x0 = tx * g.tile_w;
    r.y0 = ty * g.tile_h;
    r.sw = g.w - r.x0 < g.tile_w ? g.w - r.x0 : g.tile_w;
    r.sh = g.h - r.y0 < g.tile_h ? g.h - r.y0 : g.tile_h;
    return r;
}

// Width of a lane group = slices per wavefront.  A wavefront owns whole rows of the lane-order arrays (rows shared
// between wavefronts on different XCDs are false sharing across non-coherent L2s: measured 2x slower), so the only
// knob for "few slices" is a narrower group: full 64-lane groups as soon as that still gives >= kMinWaves wavefronts.
// LLCOMP_MI_LANE_SHIFT overrides (tuning / tests).
inline uint32_t default_lane_shift(uint32_t n_slices) {
    if (const char* e = std::getenv("LLCOMP_MI_LANE_SHIFT")) {
        const long v = std::strtol(e, nullptr, 10);
        if (v >= 0 && v <= 6) return uint32_t(v);
    }
    constexpr uint32_t kMinWaves = 96;
    uint32_t s = 6;
    while (s > 0 && (n_slices >> s) < kMinWaves) --s;
    return s;
}

inline bool make_geometry(Geometry& g, uint32_t frames, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                          uint32_t tile_h, uint32_t planar) {
    if (!frames || !w || !h || c < 1 || c > 4) return false;
    if (tile_w == 0 || tile_w > w) tile_w = w;
    if (tile_h == 0 || tile_h > h) tile_h = h;
    const uint64_t samples = uint64_t(w) * h * c;
    if (samples >= (1ull << 31)) return false;
    g.frames = frames; g.w = w; g.h = h; g.c = c;
    g.tile_w = tile_w; g.tile_h = tile_h; g.planar = planar ? 1 : 0;
    g.ntx = (w + tile_w - 1) / tile_w;
    g.nty = (h + tile_h - 1) / tile_h;
    const uint64_t spf = uint64_t(g.ntx) * g.nty * (g.planar ? c : 1);
    if (spf * frames >= (1ull << 31)) return false;
    g.slices_per_frame = uint32_t(spf);
    g.n_slices = uint32_t(spf * frames);
    g.nch = g.planar ? 1 : c;
    const uint64_t cap = (uint64_t(tile_w) * tile_h * g.nch * 13 + 32 + 15) & ~15ull;  // 13 B/sample bound + slack
    if (cap >= (1ull << 32)) return false;
    g.slice_cap = uint32_t(cap);
    g.slice_samples = tile_w * tile_h * g.nch;
    g.lane_shift = default_lane_shift(g.n_slices);
    return true;
}

}  // namespace llcomp_mi
