// container.cpp -- parsing / assembling the two wire formats and the multi-GPU band concatenator.
// Host-only logic (no kernels, no coded bytes are produced here): it moves slice tables and payloads around.
#include "container.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/llcomp_mi.h"

namespace llcomp_mi {

void write_legacy_header(uint8_t* o, uint32_t w, uint32_t h, uint32_t c) {
    o[0] = LLCOMP_MI_MAGIC_LEGACY;  // llcomp.hpp:375-378
    o[1] = uint8_t(c);
    o[2] = uint8_t(w & 0xFF); o[3] = uint8_t((w >> 8) & 0xFF);
    o[4] = uint8_t(h & 0xFF); o[5] = uint8_t((h >> 8) & 0xFF);
}

void write_sliced_header(uint8_t* o, const Geometry& g) {
    o[0] = LLCOMP_MI_MAGIC_SLICED; o[1] = 1; o[2] = uint8_t(g.c);
    o[3] = uint8_t((g.planar ? 1 : 0) | ((g.flags & kGeoSmallModel) ? 2 : 0));
    put_u32le(o + 4, g.w); put_u32le(o + 8, g.h);
    put_u32le(o + 12, g.tile_w); put_u32le(o + 16, g.tile_h);
    put_u32le(o + 20, g.slices_per_frame);
}

}  // namespace llcomp_mi

using namespace llcomp_mi;

extern "C" {

// FNV-1a-64 (offset 1469598103934665603, prime 1099511628211): the checksum the golden vectors of this project are
// recorded in; `seed` = 0 starts a new hash, the previous result continues one over several pieces.
uint64_t llcomp_mi_fnv1a64(const uint8_t* data, size_t len, uint64_t seed) {
    uint64_t h = seed ? seed : 1469598103934665603ull;
    for (size_t i = 0; i < len; ++i) h = (h ^ data[i]) * 1099511628211ull;
    return h;
}

// Width of one-row slices (tile_h == 1) for a call that codes `frames` frames at once: the widest slice that still gives
// the GPU about four wavefronts per SIMD (1024 SIMDs x 64 lanes x 4 = 262 144 slices), never narrower than 64 pixels
// (narrow slices cost compression: every slice starts with fresh models) and never wider than 480 (the throughput
// default of bench.py).  Few-frame calls are latency-bound with wide slices: one 4K frame in 480x1 planes is 51 840
// slices = 0.8 wavefronts per SIMD (2.1 ms); the width this returns, 80, gives 4 per SIMD.
uint32_t llcomp_mi_suggest_tile_w(uint32_t frames, uint32_t w, uint32_t h, uint32_t c, uint32_t planar) {
    if (!frames || !w || !h || !c) return 0;
    const uint64_t target = 262144;
    const uint64_t rows = uint64_t(frames) * h * (planar ? c : 1);  // slices per tile column
    uint64_t cols = (target + rows - 1) / rows;                     // tile columns wanted
    if (cols < 1) cols = 1;
    uint64_t tw = (w + cols - 1) / cols;
    if (tw > 480) tw = 480;
    if (tw < 64) tw = 64;
    if (tw > w) tw = w;
    // prefer a width that divides the image width when one is near (no ragged last column)
    for (uint64_t d = tw; 4 * d >= 3 * tw && d >= 64; --d)
        if (w % d == 0) return uint32_t(d);
    return uint32_t(tw);
}

// The work split of the multi-GPU paths (multidev.hip: device lists; llcomp_amd/sharding.py: ranks, through ctypes).  A chunk is a
// run of whole tile rows: slices have fresh state and slice-local borders, so a band of whole tile rows coded as an image of its own
// yields exactly the full image's slices.  Chunk i belongs to part i % n_parts (fine interleaving: the cost of a slice follows its
// entropy, not its pixels).
int llcomp_mi_plan_chunks(uint32_t height, uint32_t tile_h, uint32_t n_parts, uint32_t chunks_per_part, uint32_t* triples,
                          uint32_t cap_chunks, uint32_t* n_chunks) {
    if (!height || !n_parts || !n_chunks) return LLCOMP_MI_BAD_ARGS;
    if (tile_h == 0 || tile_h > height) tile_h = height;
    if (chunks_per_part == 0) chunks_per_part = 4;
    const uint64_t nty = (uint64_t(height) + tile_h - 1) / tile_h;
    const uint64_t want = uint64_t(n_parts) * chunks_per_part;
    const uint64_t n = nty < want ? nty : want;  // >= 1
    *n_chunks = uint32_t(n);
    if (!triples) return LLCOMP_MI_OK;
    if (n > cap_chunks) return LLCOMP_MI_OUTPUT_OVERFLOW;
    uint64_t t = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t cnt = nty / n + (i < nty % n ? 1 : 0);
        triples[3 * i + 0] = uint32_t(t);
        triples[3 * i + 1] = uint32_t(t + cnt);
        triples[3 * i + 2] = uint32_t(i % n_parts);
        t += cnt;
    }
    return LLCOMP_MI_OK;
}

uint32_t llcomp_mi_slice_count(uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar) {
    Geometry g;
    if (!make_geometry(g, 1, w, h, c, tile_w, tile_h, planar)) return 0;
    return g.slices_per_frame;
}

int llcomp_mi_probe(const uint8_t* data, size_t len, llcomp_mi_info* info) {
    if (!data || !info) return LLCOMP_MI_BAD_ARGS;
    std::memset(info, 0, sizeof(*info));
    if (len < 1) return LLCOMP_MI_TRUNCATED;
    if (data[0] == LLCOMP_MI_MAGIC_LEGACY) {
        if (len < 6) return LLCOMP_MI_TRUNCATED;  // reference reads these bytes unchecked (D5)
        info->format = LLCOMP_MI_FORMAT_LEGACY;
        info->channels = data[1];
        info->width = uint32_t(data[2]) | (uint32_t(data[3]) << 8);
        info->height = uint32_t(data[4]) | (uint32_t(data[5]) << 8);
        info->tile_w = info->width;
        info->tile_h = info->height;
        info->planar = 0;
        info->n_slices = 1;
        info->table_offset = 0;
        info->payload_offset = 6;
        return LLCOMP_MI_OK;
    }
    if (data[0] == LLCOMP_MI_MAGIC_SLICED) {
        if (len < LLCOMP_MI_SLICED_HEADER_BYTES) return LLCOMP_MI_TRUNCATED;
        if (data[1] != 1) return LLCOMP_MI_BAD_ARGS;
        info->format = LLCOMP_MI_FORMAT_SLICED;
        info->channels = data[2];
        if (data[3] & ~3u) return LLCOMP_MI_BAD_ARGS;  // unknown flag bits
        info->planar = data[3] & 1;
        info->small_model = (data[3] >> 1) & 1;
        info->width = get_u32le(data + 4);
        info->height = get_u32le(data + 8);
        info->tile_w = get_u32le(data + 12);
        info->tile_h = get_u32le(data + 16);
        info->n_slices = get_u32le(data + 20);
        Geometry g;
        if (info->tile_w == 0 || info->tile_h == 0 || info->tile_w > info->width || info->tile_h > info->height ||
            !make_geometry(g, 1, info->width, info->height, info->channels, info->tile_w, info->tile_h, info->planar) ||
            g.slices_per_frame != info->n_slices)
            return LLCOMP_MI_BAD_ARGS;
        info->table_offset = LLCOMP_MI_SLICED_HEADER_BYTES;
        info->payload_offset = uint64_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4ull * info->n_slices;
        if (len < info->payload_offset) return LLCOMP_MI_TRUNCATED;
        return LLCOMP_MI_OK;
    }
    return LLCOMP_MI_BAD_MAGIC;
}

int llcomp_mi_merge_bands(const uint8_t* const* bands, const size_t* band_lens, uint32_t n_bands, uint8_t** out,
                          size_t* out_len) {
    if (!bands || !band_lens || !n_bands || !out || !out_len) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    *out_len = 0;
    std::vector<llcomp_mi_info> infos(n_bands);
    uint64_t height = 0, n_slices = 0, payload = 0;
    for (uint32_t i = 0; i < n_bands; ++i) {
        if (int rc = llcomp_mi_probe(bands[i], band_lens[i], &infos[i])) return rc;
        const llcomp_mi_info& a = infos[i];
        const llcomp_mi_info& f = infos[0];
        if (a.format != LLCOMP_MI_FORMAT_SLICED) return LLCOMP_MI_BAD_ARGS;
        if (a.width != f.width || a.channels != f.channels || a.planar != f.planar || a.tile_w != f.tile_w || a.small_model != f.small_model)
            return LLCOMP_MI_BAD_ARGS;
        // every band but the last must be whole tile rows of the common tile height; the last may be shorter
        // (then its own tile_h was clamped to its height)
        if (i + 1 < n_bands) {
            if (a.tile_h != f.tile_h || a.height % f.tile_h != 0) return LLCOMP_MI_BAD_ARGS;
        } else if (a.tile_h != f.tile_h && !(a.height < f.tile_h && a.tile_h == a.height)) {
            return LLCOMP_MI_BAD_ARGS;
        }
        uint64_t sum = 0;
        for (uint32_t s = 0; s < a.n_slices; ++s) sum += get_u32le(bands[i] + a.table_offset + 4ull * s);
        if (a.payload_offset + sum > band_lens[i]) return LLCOMP_MI_TRUNCATED;
        height += a.height;
        n_slices += a.n_slices;
        payload += sum;
    }
    if (height >= (1ull << 31) || n_slices >= (1ull << 31)) return LLCOMP_MI_OUT_OF_RANGE;
    Geometry g;
    if (!make_geometry(g, 1, infos[0].width, uint32_t(height), infos[0].channels, infos[0].tile_w, infos[0].tile_h,
                       infos[0].planar, Tuning{}, infos[0].small_model != 0) ||
        g.slices_per_frame != n_slices)
        return LLCOMP_MI_BAD_ARGS;
    const size_t head = LLCOMP_MI_SLICED_HEADER_BYTES + 4 * size_t(n_slices);
    uint8_t* o = static_cast<uint8_t*>(std::malloc(head + payload + 1));
    if (!o) return LLCOMP_MI_NOMEM;
    write_sliced_header(o, g);
    uint8_t* tab = o + LLCOMP_MI_SLICED_HEADER_BYTES;
    uint8_t* pay = o + head;
    for (uint32_t i = 0; i < n_bands; ++i) {
        const llcomp_mi_info& a = infos[i];
        std::memcpy(tab, bands[i] + a.table_offset, 4 * size_t(a.n_slices));
        tab += 4 * size_t(a.n_slices);
        uint64_t sum = 0;
        for (uint32_t s = 0; s < a.n_slices; ++s) sum += get_u32le(bands[i] + a.table_offset + 4ull * s);
        std::memcpy(pay, bands[i] + a.payload_offset, sum);
        pay += sum;
    }
    *out = o;
    *out_len = head + payload;
    return LLCOMP_MI_OK;
}

int llcomp_mi_split_band(const uint8_t* data, size_t len, uint32_t tile_row0, uint32_t tile_row1, uint8_t** out,
                         size_t* out_len) {
    if (!data || !out || !out_len) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    *out_len = 0;
    llcomp_mi_info a;
    if (int rc = llcomp_mi_probe(data, len, &a)) return rc;
    if (a.format != LLCOMP_MI_FORMAT_SLICED) return LLCOMP_MI_BAD_ARGS;
    const uint32_t nty = (a.height + a.tile_h - 1) / a.tile_h;
    if (tile_row0 >= tile_row1 || tile_row1 > nty) return LLCOMP_MI_BAD_ARGS;
    const uint32_t per_row = a.n_slices / nty;  // slices per tile row
    const uint32_t s0 = tile_row0 * per_row, s1 = tile_row1 * per_row;
    uint64_t before = 0, inside = 0;
    for (uint32_t s = 0; s < s1; ++s) {
        const uint64_t l = get_u32le(data + a.table_offset + 4ull * s);
        (s < s0 ? before : inside) += l;
    }
    if (a.payload_offset + before + inside > len) return LLCOMP_MI_TRUNCATED;
    const uint32_t y0 = tile_row0 * a.tile_h;
    const uint32_t y1 = tile_row1 * a.tile_h < a.height ? tile_row1 * a.tile_h : a.height;
    Geometry g;
    const uint32_t band_h = y1 - y0;
    if (!make_geometry(g, 1, a.width, band_h, a.channels, a.tile_w, a.tile_h < band_h ? a.tile_h : band_h, a.planar, Tuning{}, a.small_model != 0) ||
        g.slices_per_frame != s1 - s0)
        return LLCOMP_MI_BAD_ARGS;
    const size_t head = LLCOMP_MI_SLICED_HEADER_BYTES + 4 * size_t(s1 - s0);
    uint8_t* o = static_cast<uint8_t*>(std::malloc(head + inside + 1));
    if (!o) return LLCOMP_MI_NOMEM;
    write_sliced_header(o, g);
    std::memcpy(o + LLCOMP_MI_SLICED_HEADER_BYTES, data + a.table_offset + 4ull * s0, 4 * size_t(s1 - s0));
    std::memcpy(o + head, data + a.payload_offset + before, inside);
    *out = o;
    *out_len = head + inside;
    return LLCOMP_MI_OK;
}

}  // extern "C"
