// device_common.hpp -- device-side pieces shared by the model kernels and the slice (entropy) kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "geometry.hpp"
#include "tables.hpp"

namespace llcomp_mi {

// ---- context model (llcomp.hpp:335-356, 417-436) --------------------------------------------------------
__device__ __forceinline__ int quant11(int d) {  // closed form of quant11_table (thresholds 1,2,5,12,35)
    const int a = d < 0 ? -d : d;
    const int q = (a > 0) + (a > 1) + (a > 4) + (a > 11) + (a > 34);
    return d < 0 ? -q : q;
}
__device__ __forceinline__ int quant5(int d) {  // closed form of quant5_table (thresholds 1,4)
    const int a = d < 0 ? -d : d;
    const int q = (a > 0) + (a > 3);
    return d < 0 ? -q : q;
}
__device__ __forceinline__ int median3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// raw neighbours -> border-corrected neighbours -> (context hash, prediction).  lx/ly are SLICE-local.
struct Hood {
    int l, t, L, tl, tr, T;
};
__device__ __forceinline__ Hood apply_borders(int l_raw, int L_raw, int t_raw, int tl_raw, int tr_raw, int T_raw,
                                              uint32_t lx, uint32_t ly, uint32_t sw) {
    Hood n;
    n.l = lx > 0 ? l_raw : (ly > 0 ? t_raw : 128);
    n.t = ly > 0 ? t_raw : n.l;
    n.L = lx > 1 ? L_raw : n.l;
    n.tl = (ly > 0 && lx > 0) ? tl_raw : n.t;
    n.tr = (ly > 0 && lx + 1 < sw) ? tr_raw : n.t;
    n.T = ly > 1 ? T_raw : n.t;
    return n;
}
// `small` = the reference built with LargeModel = false (llcomp.hpp:21, 427-429): the two quant5 terms are left out.
__device__ __forceinline__ int context_hash(const Hood& n, bool small = false) {
    const int h3 = quant11(n.l - n.tl) + 11 * quant11(n.tl - n.t) + 121 * quant11(n.t - n.tr);
    return small ? h3 : h3 + 605 * quant5(n.L - n.l) + 3025 * quant5(n.T - n.t);
}
// the same through byte tables in LDS: lut[d + 128] = quant11(d), lut[256 + d + 128] = quant5(d) for d = -128 .. 127 (differences
// are clamped to that range first, as the reference's tables are indexed: llcomp.hpp:335-341)
__device__ __forceinline__ int context_hash_lut(const Hood& n, const int8_t* lut, bool small = false) {
    auto at = [](int d) { return min(max(d, -128), 127) + 128; };
    const int h3 = lut[at(n.l - n.tl)] + 11 * lut[at(n.tl - n.t)] + 121 * lut[at(n.t - n.tr)];
    return small ? h3 : h3 + 605 * lut[256 + at(n.L - n.l)] + 3025 * lut[256 + at(n.T - n.t)];
}
__device__ __forceinline__ int predict(const Hood& n) { return median3(n.l, n.l + n.t - n.tl, n.t); }

// ---- generation tags of the state tables in HBM (slice_kernels.hip: "State tables in HBM"; also the snapshot pass's carry) ----------
constexpr uint64_t kTagBits = 0x8080808080808080ull;
template <bool INLDS_TABLE>
__device__ __forceinline__ uint64_t bank_fresh(uint64_t raw, uint64_t gpat) {  // what the table holds for THIS call
    if constexpr (INLDS_TABLE) return raw;  // (a table in LDS is cleared by the kernel itself and carries no tags)
    return (raw & kTagBits) == gpat ? (raw & ~kTagBits) : 0ull;
}
template <bool INLDS_TABLE>
__device__ __forceinline__ uint64_t bank_tagged(uint64_t states, uint64_t gpat) { return INLDS_TABLE ? states : (states | gpat); }


// ---- sample layouts in HBM -------------------------------------------------------------------------------
// Per-sample work arrays (encode: u32 symbols, decode: int16 reconstructed samples) are laid out so that the
// samples of one slice row are CONTIGUOUS:
//   interleaved slices : [frame][y][x][c]     (same as the pixels)
//   planar slices      : [frame][c][y][x]     (plane-major)
// slice_origin() = index of the slice's first sample, slice_row_stride() = distance between its rows.
__host__ __device__ inline size_t slice_origin(const Geometry& g, const SliceRect& r) {
    if (g.planar) return ((size_t(r.frame) * g.c + r.ch) * g.h + r.y0) * g.w + r.x0;
    return ((size_t(r.frame) * g.h + r.y0) * g.w + r.x0) * g.c;
}
__host__ __device__ inline size_t slice_row_stride(const Geometry& g) { return g.planar ? size_t(g.w) : size_t(g.w) * g.c; }
__host__ __device__ inline size_t sample_index(const Geometry& g, uint32_t frame, uint32_t y, uint32_t x, uint32_t k) {
    if (g.planar) return ((size_t(frame) * g.c + k) * g.h + y) * g.w + x;
    return ((size_t(frame) * g.h + y) * g.w + x) * g.c + k;
}

}  // namespace llcomp_mi
