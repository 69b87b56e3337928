// model_kernels.hip -- the data-parallel kernels of the llcomp coding path (gfx950, wave64).
//
//   k_model_fwd   stage A, encode side: colour transform + 6-neighbour context hash + median predictor + residual
//                 for every sample in parallel.  Coalesced row-major HBM reads, a 4-row LDS ring of
//                 colour-transformed rows (current + the two above + the row being prefetched).
//                 Reference: llcomp.hpp:396-436.
//   k_model_inv   stage A, decode side: inverse colour transform + clamp.  llcomp.hpp:532-543.
//   k_model_rows_fwd / k_model_rows_inv
//                 the same two stages fused with the lane-order transpose for planar 1-row slices.
//   k_to_lane_order / k_from_lane_order
//                 64x64 LDS transposes between image order and the [group][k][lane] order of the serial kernels.
//   k_group_sums / k_scan_groups
//                 prefix sum of the slice lengths, one value per lane group (the rest is a wave prefix in pack / stage).
//   k_pack_payload / k_stage_streams
//                 LDS-tile moves between the packed payload and the stream lane order the serial kernels use.
// None of this is GEMM-shaped; there is no MFMA here on purpose.
#include <algorithm>

#include "device_common.hpp"
#include "kernels.hpp"

namespace llcomp_mi {

namespace {

// ---- stage A, encode side -------------------------------------------------------------------------------
constexpr int kMW = 256;  // pixels per block row segment == threads per block
constexpr int kMH = 32;   // rows per block (2 halo rows above are re-read: 6% over-fetch)

template <int C>
__device__ __forceinline__ void rct_forward(const uint8_t* p, int16_t (&o)[C]) {
    if constexpr (C >= 3) {
        const int g = p[1], cb = int(p[2]) - g, cr = int(p[0]) - g;
        o[0] = int16_t(cr);
        o[1] = int16_t(g + (cb + cr) / 4);  // C++ division truncates toward zero, like llcomp.hpp:402
        o[2] = int16_t(cb);
        if constexpr (C == 4) o[3] = p[3];
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) o[k] = p[k];
    }
}

template <int C>
__global__ __launch_bounds__(kMW) void k_model_fwd(const Geometry g, const uint8_t* __restrict__ px,
                                                   uint32_t* __restrict__ sym) {
    // ring of 4 colour-transformed rows, planar per channel; column j holds image column bx0 - 2 + j
    __shared__ int16_t win[4][C][kMW + 4];
    const uint32_t nbx = (g.w + kMW - 1) / kMW;
    const uint32_t bx = blockIdx.x % nbx;
    const uint32_t by = blockIdx.x / nbx;  // strip of kMH rows, never crossing a tile row or frame
    // row strips are enumerated per (frame, tile row): strips_per_tile_row = ceil(tile_h / kMH)
    const uint32_t spt = (g.tile_h + kMH - 1) / kMH;
    const uint32_t trow = by / spt;             // global tile-row index over all frames
    const uint32_t strip = by - trow * spt;
    const uint32_t frame = trow / g.nty;
    const uint32_t ty = trow - frame * g.nty;
    const uint32_t tile_y0 = ty * g.tile_h;
    const uint32_t tile_rows = g.h - tile_y0 < g.tile_h ? g.h - tile_y0 : g.tile_h;
    const uint32_t ly0 = strip * kMH;
    if (ly0 >= tile_rows) return;  // uniform per block
    const uint32_t ly1 = ly0 + kMH < tile_rows ? ly0 + kMH : tile_rows;

    const uint32_t t = threadIdx.x;
    const uint32_t bx0 = bx * kMW;
    const uint32_t x = bx0 + t;
    const bool in_x = x < g.w;
    const uint32_t tx = (in_x ? x : g.w - 1) / g.tile_w;
    const uint32_t lx = (in_x ? x : g.w - 1) - tx * g.tile_w;
    const uint32_t sw = g.w - tx * g.tile_w < g.tile_w ? g.w - tx * g.tile_w : g.tile_w;

    const size_t row_bytes = size_t(g.w) * C;
    const uint8_t* fbase = px + size_t(frame) * g.h * row_bytes;

    // loader: thread t stages column bx0-2+t ... plus 4 extra columns by threads 0..3 (kMW+4 columns total)
    auto stage_row = [&](int ly, int16_t (&a)[C], int16_t (&b)[C]) {
        // ly may be negative (rows above the tile are never used by the border rules): stage zeros
        const bool row_ok = ly >= 0;
        const uint8_t* rowp = fbase + size_t(tile_y0 + (row_ok ? ly : 0)) * row_bytes;
        const int xa = int(bx0) - 2 + int(t);
#pragma unroll
        for (int k = 0; k < C; ++k) a[k] = b[k] = 0;
        if (row_ok && xa >= 0 && xa < int(g.w)) rct_forward<C>(rowp + size_t(xa) * C, a);
        if (t < 4) {
            const int xb = int(bx0) - 2 + kMW + int(t);
            if (row_ok && xb < int(g.w)) rct_forward<C>(rowp + size_t(xb) * C, b);
        }
    };
    auto commit_row = [&](int slot, const int16_t (&a)[C], const int16_t (&b)[C]) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            win[slot][k][t] = a[k];
            if (t < 4) win[slot][k][kMW + t] = b[k];
        }
    };

    int16_t ra[C], rb[C];
    // prologue: rows ly0-2, ly0-1, ly0 into slots (ly & 3)
    for (int ly = int(ly0) - 2; ly <= int(ly0); ++ly) {
        stage_row(ly, ra, rb);
        commit_row(ly & 3, ra, rb);
    }
    __syncthreads();
    for (uint32_t ly = ly0; ly < ly1; ++ly) {
        const bool more = ly + 1 < ly1;
        if (more) stage_row(int(ly) + 1, ra, rb);  // global loads in flight while this row is modelled
        if (in_x) {
            const int s0 = ly & 3, s1 = (ly + 3) & 3, s2 = (ly + 2) & 3;
            uint32_t out[C];
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int cur = win[s0][k][t + 2];
                const Hood n = apply_borders(win[s0][k][t + 1], win[s0][k][t], win[s1][k][t + 2], win[s1][k][t + 1],
                                             win[s1][k][t + 3], win[s2][k][t + 2], lx, ly, sw);
                int ctx = context_hash(n, (g.flags & kGeoSmallModel) != 0);
                int res = cur - predict(n);
                if (ctx < 0) {  // llcomp.hpp:433-436
                    ctx = -ctx;
                    res = -res;
                }
                out[k] = uint32_t(ctx) | (uint32_t(res) << 16);
            }
            if (g.planar) {  // plane-major: every plane row is written as full coalesced dword rows
#pragma unroll
                for (int k = 0; k < C; ++k) sym[sample_index(g, frame, tile_y0 + ly, x, k)] = out[k];
            } else {
                uint32_t* o = sym + sample_index(g, frame, tile_y0 + ly, x, 0);
#pragma unroll
                for (int k = 0; k < C; ++k) o[k] = out[k];
            }
        }
        if (more) commit_row((ly + 1) & 3, ra, rb);  // slot (ly+1)&3 == (ly-3)&3: not read this iteration
        __syncthreads();
    }
}

// ---- stage A, decode side -------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_model_inv(const Geometry g, const int16_t* __restrict__ rec,
                                                   uint8_t* __restrict__ px, size_t npix) {
    const size_t plane = size_t(g.h) * g.w;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < npix; i += size_t(gridDim.x) * blockDim.x) {
        int16_t s[C];
        if (g.planar) {
            const size_t f = i / plane, p = i - f * plane;
#pragma unroll
            for (int k = 0; k < C; ++k) s[k] = rec[(f * C + k) * plane + p];
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) s[k] = rec[i * C + k];
        }
        uint8_t* o = px + i * C;
        if constexpr (C >= 3) {
            int r = s[0], gg = s[1], b = s[2];
            gg -= (r + b) / 4;
            r += gg;
            b += gg;
            o[0] = uint8_t(min(max(r, 0), 255));
            o[1] = uint8_t(min(max(gg, 0), 255));
            o[2] = uint8_t(min(max(b, 0), 255));
            if constexpr (C == 4) o[3] = uint8_t(s[3]);
        } else {
#pragma unroll
            for (int k = 0; k < C; ++k) o[k] = uint8_t(s[k]);
        }
    }
}

// ---- stage A for any channel count (c > 4) -------------------------------------------------------------------------
// The reference's header holds the channel count in one byte and its coder passes channels beyond the third through
// untransformed (llcomp.hpp:407-409, 541-543).  stb_image never delivers more than four, so these two kernels are the
// plain form -- one thread per pixel, neighbours straight from HBM, the colour transform redone per neighbour -- kept
// for format completeness, not for speed.
__device__ __forceinline__ int sample_at(const uint8_t* p, uint32_t c, uint32_t k) {  // colour-transformed sample k of the pixel at p
    if (c >= 3 && k < 3) {
        const int g = p[1], cb = int(p[2]) - g, cr = int(p[0]) - g;
        return k == 0 ? cr : (k == 1 ? g + (cb + cr) / 4 : cb);  // llcomp.hpp:396-406
    }
    return p[k];
}
__global__ __launch_bounds__(256) void k_model_fwd_any(const Geometry g, const uint8_t* __restrict__ px, uint32_t* __restrict__ sym,
                                                       size_t npix) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= npix) return;
    const size_t plane = size_t(g.h) * g.w;
    const uint32_t frame = uint32_t(i / plane);
    const uint32_t rem = uint32_t(i - size_t(frame) * plane), y = rem / g.w, x = rem - y * g.w;
    const uint32_t tx = x / g.tile_w, ty = y / g.tile_h, lx = x - tx * g.tile_w, ly = y - ty * g.tile_h;
    const uint32_t sw = g.w - tx * g.tile_w < g.tile_w ? g.w - tx * g.tile_w : g.tile_w;
    const size_t rowb = size_t(g.w) * g.c;
    const uint8_t* p = px + i * g.c;
    for (uint32_t k = 0; k < g.c; ++k) {
        const int cur = sample_at(p, g.c, k);
        const int l = lx > 0 ? sample_at(p - g.c, g.c, k) : 0, L = lx > 1 ? sample_at(p - 2 * g.c, g.c, k) : 0;
        const int t = ly > 0 ? sample_at(p - rowb, g.c, k) : 0, tl = (ly > 0 && lx > 0) ? sample_at(p - rowb - g.c, g.c, k) : 0;
        const int tr = (ly > 0 && lx + 1 < sw) ? sample_at(p - rowb + g.c, g.c, k) : 0, T = ly > 1 ? sample_at(p - 2 * rowb, g.c, k) : 0;
        const Hood n = apply_borders(l, L, t, tl, tr, T, lx, ly, sw);
        int ctx = context_hash(n, (g.flags & kGeoSmallModel) != 0), res = cur - predict(n);
        if (ctx < 0) {  // llcomp.hpp:433-436
            ctx = -ctx;
            res = -res;
        }
        sym[sample_index(g, frame, y, x, k)] = uint32_t(ctx) | (uint32_t(res) << 16);
    }
}
__global__ __launch_bounds__(256) void k_model_inv_any(const Geometry g, const int16_t* __restrict__ rec, uint8_t* __restrict__ px,
                                                       size_t npix) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= npix) return;
    const size_t plane = size_t(g.h) * g.w;
    const uint32_t frame = uint32_t(i / plane);
    const uint32_t rem = uint32_t(i - size_t(frame) * plane), y = rem / g.w, x = rem - y * g.w;
    uint8_t* o = px + i * g.c;
    uint32_t k = 0;
    if (g.c >= 3) {  // llcomp.hpp:532-540
        int r = rec[sample_index(g, frame, y, x, 0)], gg = rec[sample_index(g, frame, y, x, 1)], b = rec[sample_index(g, frame, y, x, 2)];
        gg -= (r + b) / 4;
        r += gg;
        b += gg;
        o[0] = uint8_t(min(max(r, 0), 255));
        o[1] = uint8_t(min(max(gg, 0), 255));
        o[2] = uint8_t(min(max(b, 0), 255));
        k = 3;
    }
    for (; k < g.c; ++k) o[k] = uint8_t(rec[sample_index(g, frame, y, x, k)]);  // llcomp.hpp:541-543
}

// ---- slice length scan ------------------------------------------------------------------------------------------------
// The packed payload holds the slices back to back in slice order, so slice i starts at the sum of the lengths before
// it.  pack / stage work on whole lane groups (64 consecutive slices) and find a slice's offset as
//     group_off[group] + (wave prefix sum of the group's 64 lengths),
// so only ONE value per lane group has to be scanned globally:
//   * the encoder kernel leaves the sum of its wavefront's lengths behind for free (k_encode_slices, slice_kernels.hip);
//     k_group_sums does the same for lengths that come from outside (decode; encode with fewer lanes per wavefront
//     than the group is wide),
//   * k_scan_groups -- ONE small block -- turns the <= tens of thousands of group sums into exclusive offsets.
// (Round 1 scanned all 1.6 M slice lengths in three dependent launches; next to the other pipelines' slice kernels
// each of them waited ~0.4 ms for a free wave slot, profiles/r01_default_kernel_stats.csv.)
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(v, d, 64);
        if (lane >= uint32_t(d)) v += o;
    }
    return v;
}

// (One wavefront per workgroup, here and in k_scan_groups: beside the slice kernels, whose one-wavefront workgroups take every
// wave slot as it comes free, a workgroup of four wavefronts waits until four slots of ONE CU are free at the same moment.)
constexpr uint32_t kSumThreads = 64;
__global__ __launch_bounds__(kSumThreads) void k_group_sums(const uint32_t* __restrict__ len, uint32_t n, uint32_t lane_shift,
                                                    uint64_t* __restrict__ group_sum) {
    const uint32_t i = blockIdx.x * kSumThreads + threadIdx.x;
    unsigned long long v = i < n ? len[i] : 0;
    for (uint32_t d = 1; d < (1u << lane_shift); d <<= 1) v += __shfl_xor(v, int(d), 64);  // groups are aligned pieces of a wave
    if ((i & ((1u << lane_shift) - 1)) == 0 && i < n) group_sum[i >> lane_shift] = v;
}

// payload bytes of every frame of a batch (sum of its slices' lengths): what a host needs to cut a batch's packed
// payload into per-frame containers (stream.hip, jobs of several frames)
__global__ __launch_bounds__(256) void k_frame_bytes(const uint32_t* __restrict__ len, uint32_t slices_per_frame, uint64_t* __restrict__ out) {
    __shared__ unsigned long long part[4];
    const uint32_t* p = len + size_t(blockIdx.x) * slices_per_frame;
    unsigned long long v = 0;
    for (uint32_t i = threadIdx.x; i < slices_per_frame; i += 256) v += p[i];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

constexpr uint32_t kScanThreads = 64, kScanPerThread = 16;
// in: sums[0 .. ng)   out: sums[g] = sum of the groups before g, sums[ng] = *total = sum of all
__global__ __launch_bounds__(kScanThreads) void k_scan_groups(uint64_t* __restrict__ sums, uint32_t ng, uint64_t* total) {
    __shared__ unsigned long long wave_sum[kScanThreads / 64];
    __shared__ unsigned long long carry;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ng; base += kScanThreads * kScanPerThread) {
        const uint32_t i0 = base + threadIdx.x * kScanPerThread;
        unsigned long long v[kScanPerThread], mine = 0;
#pragma unroll
        for (uint32_t j = 0; j < kScanPerThread; ++j) {
            v[j] = i0 + j < ng ? sums[i0 + j] : 0;
            mine += v[j];
        }
        const unsigned long long inc = wave_inclusive_scan(mine, lane);
        if (lane == 63) wave_sum[wv] = inc;
        __syncthreads();
        unsigned long long run = carry + inc - mine;
        for (uint32_t k = 0; k < wv; ++k) run += wave_sum[k];
#pragma unroll
        for (uint32_t j = 0; j < kScanPerThread; ++j) {
            if (i0 + j < ng) sums[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) carry = run;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sums[ng] = carry;
        *total = carry;
    }
}

// ---- slice streams: packed payload <-> stream lane order ---------------------------------------------------------------
// The coded bytes of a slice are produced / consumed by ONE lane, a few bytes per sample.  In HBM they therefore live
// in STREAM LANE ORDER: 16-byte units laid out [group][unit][lane], so the 64 lanes of a wavefront, which advance
// through their streams at nearly the same pace, touch neighbouring units (one 1 KiB row) instead of 64 separate
// cache lines.  (With per-lane contiguous streams rocprofv3 showed 6.8 GB fetched per decode launch for 0.5 GB of
// payload.)  These two kernels move whole groups between that order and the packed payload of the container through
// a 64 x 256-byte LDS tile: unit reads/writes are 1 KiB rows, payload reads/writes are 256-byte runs per slice.
constexpr int kChunkDwords = 32;  // 128 bytes of every slice per LDS tile (64 = 16.6 KB of LDS per workgroup: measured slower beside the slice kernels)

__device__ __forceinline__ uint32_t load_bytes_le(const uint8_t* p, uint32_t n) {  // n = 1..3
    uint32_t w = p[0];
    if (n > 1) w |= uint32_t(p[1]) << 8;
    if (n > 2) w |= uint32_t(p[2]) << 16;
    return w;
}

struct GroupStreams {
    unsigned long long off[64];
    uint32_t len[64];
    uint32_t max_len;
};
__device__ __forceinline__ void load_group_streams(const Geometry& g, uint32_t group, const uint32_t* slice_len,
                                                   const uint64_t* group_off, uint64_t limit, uint32_t* status,
                                                   uint32_t err_bit, GroupStreams& gs) {
    if (threadIdx.x < 64) {
        const uint32_t id = (group << g.lane_shift) + threadIdx.x;
        const bool live = threadIdx.x < (1u << g.lane_shift) && id < g.n_slices;
        uint32_t n = live ? slice_len[id] : 0;
        // offset of the slice = offset of its lane group + the lengths of the group's slices before it
        unsigned long long o = group_off[group] + wave_inclusive_scan(n, threadIdx.x) - n;
        if (live) {
            if (o + n > limit) {  // the slice does not fit the payload (decode: table promises too much)
                atomicOr(status, err_bit);
                n = err_bit == kStOverflow ? 0u : (o < limit ? uint32_t(limit - o) : 0u);
            }
            if (n > g.slice_cap - 16) {  // no valid stream is longer than the proven bound: damaged slice table
                atomicOr(status, err_bit);
                n = g.slice_cap - 16;
            }
        }
        gs.off[threadIdx.x] = o;
        gs.len[threadIdx.x] = n;
        uint32_t m = n;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, uint32_t(__shfl_xor(int(m), d, 64)));
        if (threadIdx.x == 0) gs.max_len = m;
    }
    __syncthreads();
}

// stream lane order -> packed payload (after the encoder)
// Thread layout of the payload side: a run of kChunkDwords dwords of one slice is moved by kChunkDwords neighbouring threads;
// 256 / kChunkDwords slices per pass.
constexpr uint32_t kRunThreads = kChunkDwords, kRunsPerPass = 256 / kChunkDwords, kChunkBytes = kChunkDwords * 4, kChunkUnits = kChunkDwords / 4;
static_assert(kChunkUnits % 4 == 0 && kChunkDwords % 4 == 0 && 256 % kChunkDwords == 0, "thread layouts of pack / stage");
__global__ __launch_bounds__(256) void k_pack_payload(const Geometry g, const uint4* __restrict__ units,
                                                      const uint32_t* __restrict__ slice_len,
                                                      const uint64_t* __restrict__ off, uint8_t* __restrict__ payload,
                                                      uint64_t payload_cap, uint32_t* status) {
    __shared__ uint32_t tile[64][kChunkDwords + 1];
    __shared__ GroupStreams gs;
    const uint32_t group = blockIdx.x;
    load_group_streams(g, group, slice_len, off, payload_cap, status, kStOverflow, gs);
    const uint32_t cap16 = g.slice_cap >> 4;
    const uint32_t a = threadIdx.x & 63, b = threadIdx.x >> 6;
    const uint32_t dw = threadIdx.x % kRunThreads, jb = threadIdx.x / kRunThreads;
    // 1 KiB rows of units.  All loads of a thread are in flight before the first is stored (see k_model_rows_inv),
    // and the loads of the NEXT chunk are issued before this chunk's stores: they fly during the store phase.
    constexpr int LPT = kChunkUnits / 4;  // units per thread and chunk
    uint4 v[LPT];
    auto request = [&](uint32_t c0) {
#pragma unroll
        for (int t = 0; t < LPT; ++t) {
            const uint32_t u = c0 * kChunkUnits + b + 4 * t;
            v[t] = make_uint4(0, 0, 0, 0);
            if (u < cap16 && a < (1u << g.lane_shift)) v[t] = units[((size_t(group) * cap16 + u) << g.lane_shift) + a];
        }
    };
    if (gs.max_len) request(0);
    for (uint32_t c0 = 0; c0 * kChunkBytes < gs.max_len; ++c0) {
#pragma unroll
        for (int t = 0; t < LPT; ++t) {
            const uint32_t uu = b + 4 * t;
            tile[a][uu * 4 + 0] = v[t].x; tile[a][uu * 4 + 1] = v[t].y; tile[a][uu * 4 + 2] = v[t].z; tile[a][uu * 4 + 3] = v[t].w;
        }
        __syncthreads();
        if ((c0 + 1) * kChunkBytes < gs.max_len) request(c0 + 1);
        for (uint32_t j = jb; j < 64; j += kRunsPerPass) {  // runs of one slice
            const uint32_t n = gs.len[j], p = c0 * kChunkBytes + dw * 4;
            if (p < n) {
                uint8_t* dst = payload + gs.off[j] + p;
                const uint32_t w = tile[j][dw];
                if (p + 4 <= n) {
                    __builtin_memcpy(dst, &w, 4);  // byte offset of a slice is arbitrary: unaligned dword store
                } else {
                    for (uint32_t i = 0; i < n - p; ++i) dst[i] = uint8_t(w >> (8 * i));
                }
            }
        }
        __syncthreads();
    }
}

// packed payload -> stream lane order (before the decoder)
__global__ __launch_bounds__(256) void k_stage_streams(const Geometry g, const uint8_t* __restrict__ payload,
                                                       uint64_t payload_bytes, const uint32_t* __restrict__ slice_len,
                                                       const uint64_t* __restrict__ off, uint4* __restrict__ units,
                                                       uint32_t* status) {
    __shared__ uint32_t tile[64][kChunkDwords + 1];
    __shared__ GroupStreams gs;
    const uint32_t group = blockIdx.x;
    load_group_streams(g, group, slice_len, off, payload_bytes, status, kStTruncated, gs);
    const uint32_t capdw = g.slice_cap >> 2;
    const uint32_t a = threadIdx.x & 63, b = threadIdx.x >> 6;
    const uint32_t dwi = threadIdx.x % kRunThreads, jb = threadIdx.x / kRunThreads;
    // "+ 4": the dword right behind every stream is staged too (as zeros) -- the decoder clamps its reads to it.
    // All loads of a thread are in flight before the first is stored, and the loads of the NEXT chunk are issued
    // before this chunk's stores.  The last, partial dword of a stream is read as a whole dword and masked wherever the
    // payload has the bytes (always, except at its very end).
    constexpr int LPT = 64 / kRunsPerPass;  // slices per thread and chunk
    uint32_t w[LPT];
    auto request = [&](uint32_t c0) {
#pragma unroll
        for (int t = 0; t < LPT; ++t) {
            const uint32_t j = jb + kRunsPerPass * t, n = gs.len[j], p = c0 * kChunkBytes + dwi * 4;
            w[t] = 0;
            if (p < n) {
                const unsigned long long at = gs.off[j] + p;
                const uint8_t* src = payload + at;
                if (at + 4 <= payload_bytes) __builtin_memcpy(&w[t], src, 4);
                else w[t] = load_bytes_le(src, n - p);
            }
        }
    };
    request(0);
    for (uint32_t c0 = 0; c0 * kChunkBytes < gs.max_len + 4; ++c0) {
#pragma unroll
        for (int t = 0; t < LPT; ++t) {  // (the mask is applied here, not above: nothing waits for a load before all are issued)
            const uint32_t j = jb + kRunsPerPass * t, n = gs.len[j], p = c0 * kChunkBytes + dwi * 4;
            const uint32_t keep = n >= p + 4 ? 0xFFFFFFFFu : n > p ? 0xFFFFFFFFu >> (8 * (p + 4 - n)) : 0u;
            tile[j][dwi] = w[t] & keep;
        }
        __syncthreads();
        if ((c0 + 1) * kChunkBytes < gs.max_len + 4) request(c0 + 1);
        // DWORD lane order for the decoder, [group][dword k][lane]: a wavefront stores one 256-byte row per dword index
        uint32_t* const dw = reinterpret_cast<uint32_t*>(units);
        for (uint32_t kk = b; kk < kChunkDwords; kk += 4) {
            const uint32_t k = c0 * kChunkDwords + kk;
            if (k < capdw && a < (1u << g.lane_shift)) dw[((size_t(group) * capdw + k) << g.lane_shift) + a] = tile[a][kk];
        }
        __syncthreads();
    }
}

// ---- byte segments: the device-side concatenator of the multi-GPU path ------------------------------------------------
// A sharded image arrives on the gathering rank as one packed payload per rank; the container wants the slices in
// image order, i.e. the ranks' pieces interleaved chunk by chunk (llcomp_amd/sharding.py).  One launch copies n_seg byte
// ranges src[src_off[i] .. +len[i]) -> dst[dst_off[i] .. +len[i]); offsets and lengths live in HBM (they were computed
// there), alignment is arbitrary.  blockIdx.y = segment, blockIdx.x strides over 16 KiB pieces of it.
constexpr uint32_t kSegPieceDwords = 4096;
__global__ __launch_bounds__(256) void k_copy_segments(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                       const uint64_t* __restrict__ src_off, const uint64_t* __restrict__ dst_off,
                                                       const uint64_t* __restrict__ len) {
    const uint64_t n = len[blockIdx.y];
    const uint8_t* s = src + src_off[blockIdx.y];
    uint8_t* d = dst + dst_off[blockIdx.y];
    // dword stores want an aligned destination: up to 3 head bytes and up to 3 tail bytes go one by one
    const uint64_t head = min(n, uint64_t((4 - (reinterpret_cast<uintptr_t>(d) & 3)) & 3));
    const uint64_t ndw = (n - head) >> 2, tail = (n - head) & 3;
    for (uint64_t j0 = uint64_t(blockIdx.x) * kSegPieceDwords; j0 < ndw; j0 += uint64_t(gridDim.x) * kSegPieceDwords) {
        const uint64_t j1 = min(ndw, j0 + kSegPieceDwords);
        for (uint64_t j = j0 + threadIdx.x; j < j1; j += 256) {
            uint32_t w;
            __builtin_memcpy(&w, s + head + 4 * j, 4);  // source alignment is arbitrary
            *reinterpret_cast<uint32_t*>(d + head + 4 * j) = w;
        }
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
        if (threadIdx.x < tail) d[head + 4 * ndw + threadIdx.x] = s[head + 4 * ndw + threadIdx.x];
    }
}

// Sums of ranges of a u32 table: out[i] = sum of min(vals[j], cap) for j in [start[i], start[i] + count[i]).  The multi-GPU
// path (llcomp_amd/sharding.py) turns slice lengths into the byte counts of its (image, chunk) segments with it: one launch
// instead of a prefix sum over every slice plus a dozen gathers.  One block per range.
__global__ __launch_bounds__(256) void k_range_sums(const uint32_t* __restrict__ vals, const uint64_t* __restrict__ start,
                                                    const uint64_t* __restrict__ count, uint64_t* __restrict__ out, uint32_t cap) {
    __shared__ unsigned long long part[4];
    const uint64_t s = start[blockIdx.x], n = count[blockIdx.x];
    unsigned long long acc = 0;
    for (uint64_t j = threadIdx.x; j < n; j += 256) acc += min(vals[s + j], cap);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// ---- image order <-> lane order -------------------------------------------------------------------------------------
// The serial kernels run one slice per lane, 64 slices ("lane group") per wavefront, all lanes at the same sample
// index k.  Their per-sample arrays therefore live in LANE ORDER  [group][k][64 lanes]  so that one wavefront access
// is one contiguous 256-byte (u32) or 128-byte (int16) piece.  (In image order every lane walks its own row: rocprofv3
// showed 5-7 GB of fabric traffic per launch for 0.4-0.8 GB of useful bytes.)  These two kernels convert between the
// image-order arrays of the model kernels and lane order through a 64x64 LDS tile; both sides are coalesced.
// (With fewer than 64 slices the group is narrower, Geometry::lane_shift, so a lone whole-image slice is not padded.)
struct SliceSpan {
    unsigned long long origin;  // index of the slice's first sample in the image-order array
    uint32_t n_row;             // samples per slice row (contiguous)
    uint32_t n;                 // samples in the slice
};
__device__ __forceinline__ void load_spans(const Geometry& g, uint32_t group, SliceSpan* spans) {
    if (threadIdx.x < 64) {
        const uint32_t id = (group << g.lane_shift) + threadIdx.x;
        SliceSpan sp{0, 1, 0};
        if (threadIdx.x < (1u << g.lane_shift) && id < g.n_slices) {
            const SliceRect r = slice_rect(g, id);
            sp.origin = slice_origin(g, r);
            sp.n_row = r.sw * g.nch;
            sp.n = sp.n_row * r.sh;
        }
        spans[threadIdx.x] = sp;
    }
    __syncthreads();
}
__device__ __forceinline__ size_t span_index(const SliceSpan& sp, uint32_t k, size_t row_stride) {
    const uint32_t y = k / sp.n_row;
    return size_t(sp.origin) + size_t(y) * row_stride + (k - y * sp.n_row);
}

template <typename T>
__global__ __launch_bounds__(256) void k_to_lane_order(const Geometry g, const uint32_t max_n,
                                                       const T* __restrict__ img, T* __restrict__ lanes) {
    __shared__ SliceSpan spans[64];
    __shared__ T tile[64][65];
    const uint32_t chunks = (max_n + 63) / 64;
    const uint32_t group = blockIdx.x / chunks, k0 = (blockIdx.x - group * chunks) * 64;
    load_spans(g, group, spans);
    const size_t rs = slice_row_stride(g);
    const uint32_t a = threadIdx.x & 63, b = threadIdx.x >> 6;
    {   // read: lanes run along k (contiguous in image order); sixteen loads in flight, then sixteen LDS stores
        T v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const SliceSpan sp = spans[b + 4 * t];
            const uint32_t k = k0 + a;
            v[t] = k < sp.n ? img[span_index(sp, k, rs)] : T(0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) tile[b + 4 * t][a] = v[t];
    }
    __syncthreads();
    const uint32_t gw = 1u << g.lane_shift;
    // write: the block's 64 rows of `gw` lanes are one contiguous piece of the lane-order array; consecutive threads take
    // consecutive elements, so a wavefront always stores whole 128-byte lines whatever the group width
    T* out = lanes + ((size_t(group) * max_n + k0) << g.lane_shift);
    const uint32_t n_el = min(64u, max_n - k0) << g.lane_shift;
    for (uint32_t el = threadIdx.x; el < n_el; el += 256) out[el] = tile[el & (gw - 1)][el >> g.lane_shift];
}

template <typename T>
__global__ __launch_bounds__(256) void k_from_lane_order(const Geometry g, const uint32_t max_n,
                                                         const T* __restrict__ lanes, T* __restrict__ img) {
    __shared__ SliceSpan spans[64];
    __shared__ T tile[64][65];
    const uint32_t chunks = (max_n + 63) / 64;
    const uint32_t group = blockIdx.x / chunks, k0 = (blockIdx.x - group * chunks) * 64;
    load_spans(g, group, spans);
    const size_t rs = slice_row_stride(g);
    const uint32_t a = threadIdx.x & 63, b = threadIdx.x >> 6;
    const uint32_t gw = 1u << g.lane_shift;
    // read: the block's 64 rows of `gw` lanes are one contiguous piece (see k_to_lane_order); rows of lanes beyond the
    // group width do not exist and their tile entries are never looked at (those spans are empty)
    const T* in = lanes + ((size_t(group) * max_n + k0) << g.lane_shift);
    const uint32_t n_el = min(64u, max_n - k0) << g.lane_shift;
    {   // (sixteen loads in flight, then sixteen LDS stores)
        T v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t el = threadIdx.x + 256 * t;
            v[t] = el < n_el ? in[el] : T(0);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t el = threadIdx.x + 256 * t;
            if (el < (64u << g.lane_shift)) tile[el & (gw - 1)][el >> g.lane_shift] = v[t];
        }
    }
    __syncthreads();
    for (uint32_t j = b; j < 64; j += 4) {
        const SliceSpan sp = spans[j];
        const uint32_t k = k0 + a;
        if (k < sp.n) img[span_index(sp, k, rs)] = tile[j][a];
    }
}

// Workgroups are dealt round-robin over the 8 XCDs (workgroup i runs on XCD i % 8), each with its own L2.  The 64-sample
// chunks of one lane group touch neighbouring pieces of the same pixel rows (shared cache lines) and of the same lane-order
// rows, so they should meet in ONE L2: lane group g goes to XCD g % 8 and its chunks occupy consecutive slots there.
// Returns false for the padding workgroups of the last, incomplete round of groups.
constexpr uint32_t kXcds = 8;
__device__ __forceinline__ bool xcd_group_chunk(uint32_t n_groups, uint32_t chunks, uint32_t& group, uint32_t& chunk) {
    const uint32_t xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds;
    group = (slot / chunks) * kXcds + xcd;
    chunk = slot - (slot / chunks) * chunks;
    return group < n_groups;
}
__host__ inline uint64_t xcd_grid(uint32_t n_groups, uint32_t chunks) { return uint64_t((n_groups + kXcds - 1) / kXcds) * kXcds * chunks; }
// The other way round for the decode side: there a workgroup also reads a few lanes of the NEXT lane group's rows (the
// channel planes of a tile that straddles two groups), i.e. the same lane-order rows as workgroup (group + 1, same chunk).
// Chunk c of every group goes to XCD c % 8, so those rows are fetched into one L2 only (measured: 2x the fetch traffic
// with the group-major mapping above).
__device__ __forceinline__ bool xcd_chunk_group(uint32_t n_groups, uint32_t chunks, uint32_t& group, uint32_t& chunk) {
    const uint32_t xcd = blockIdx.x % kXcds, slot = blockIdx.x / kXcds;
    const uint32_t per_class = (chunks + kXcds - 1) / kXcds;  // chunks c with c % 8 == xcd
    group = slot / per_class;
    chunk = xcd + kXcds * (slot - group * per_class);
    return group < n_groups && chunk < chunks;
}
__host__ inline uint64_t xcd_chunk_grid(uint32_t n_groups, uint32_t chunks) { return uint64_t(n_groups) * ((chunks + kXcds - 1) / kXcds) * kXcds; }

// ---- fused stage A for planar 1-row slices (tile_h == 1, planar) -------------------------------------------------------
// The headline configuration.  One block = one lane group (64 consecutive slice ids = ~64/C tiles x C channel planes)
// x 64 consecutive samples: the pixel runs of those tiles are staged raw in LDS (coalesced reads of 66*C bytes each),
// every thread then walks 8 consecutive PIXELS of one tile (one colour transform per pixel, all C symbols of it),
// the symbols meet in an LDS tile [sample][lane] and leave as whole 128-byte rows of the lane-order array, so the
// image-order symbol array and the transpose pass disappear.  With h == 0 in llcomp.hpp:417-429 the context is
// 605*quant5(L - l) and the prediction is l.
template <int C>
__device__ __forceinline__ void rct_pixel(const uint8_t* p, int (&v)[C]) {  // llcomp.hpp:396-414, all channels of a pixel
    if constexpr (C >= 3) {
        const int g = p[1], cb = int(p[2]) - g, cr = int(p[0]) - g;
        v[0] = cr;
        v[1] = g + (cb + cr) / 4;  // truncating division, llcomp.hpp:402
        v[2] = cb;
        if constexpr (C == 4) v[3] = p[3];
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) v[k] = p[k];
    }
}
struct RowTile {
    unsigned long long base;  // byte offset of the tile's first pixel in the frame batch
    uint32_t sw;              // pixels in this tile
};
template <int C>
__device__ __forceinline__ void load_row_tiles(const Geometry& g, uint32_t first_tile, uint32_t ntiles, RowTile* tiles) {
    if (threadIdx.x < ntiles) {
        const uint32_t tile = first_tile + threadIdx.x;  // global tile index: (frame, row, column tile)
        const uint32_t per_frame = g.ntx * g.nty;        // nty == h because tile_h == 1
        const uint32_t frame = tile / per_frame, rem = tile - frame * per_frame;
        const uint32_t y = rem / g.ntx, tx = rem - y * g.ntx;
        const uint32_t x0 = tx * g.tile_w;
        tiles[threadIdx.x].base = ((size_t(frame) * g.h + y) * g.w + x0) * C;
        tiles[threadIdx.x].sw = g.w - x0 < g.tile_w ? g.w - x0 : g.tile_w;
    }
}

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// (hipcc turns min(x, 1) on 16-bit pairs into two compares, two selects and a byte permute)
__device__ __forceinline__ u16x2 pk_min_u16(u16x2 a, u16x2 b) {
    u16x2 d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int C>
__global__ __launch_bounds__(256) void k_model_rows_fwd(const Geometry g, const uint8_t* __restrict__ px,
                                                        uint16_t* __restrict__ lanes) {
    constexpr int K = 64, RUN = (K + 2) * C, TPG = 64 / C + 2;
    constexpr int RUNW = (RUN + 3) / 4 + 1;  // dwords that cover a run at any byte alignment
    // otile columns: PADL columns of padding, the group's 64 lanes, C-1 more.  The first and the last tile of a group may
    // have channel planes that belong to the neighbouring groups: their symbols land in the padding, no range test per store.
    constexpr int PADL = (C - 1 + 1) & ~1, OCOLS = (PADL + 64 + C - 1 + 1) & ~1;
    __shared__ uint32_t raw[TPG][RUNW + 1];
    __shared__ __attribute__((aligned(4))) uint16_t otile[K][OCOLS];  // symbols [sample][PADL + group-relative lane]
    __shared__ RowTile tiles[TPG];
    const uint32_t chunks = (g.tile_w + K - 1) / K;
    uint32_t group, chunk;
    if (!xcd_group_chunk((g.n_slices + (1u << g.lane_shift) - 1) >> g.lane_shift, chunks, group, chunk)) return;  // (uniform per block)
    const uint32_t k0 = chunk * K;
    const uint32_t gw = 1u << g.lane_shift;
    const uint32_t first_id = group << g.lane_shift;
    const uint32_t end_id = first_id + gw < g.n_slices ? first_id + gw : g.n_slices;
    // every tile with at least one channel plane in this group (the planes outside it belong to the neighbours)
    const uint32_t first_tile = first_id / C, ntiles = (end_id - 1) / C - first_tile + 1;
    load_row_tiles<C>(g, first_tile, ntiles, tiles);
    __syncthreads();
    // stage the pixel runs [k0-2, k0+64) of every tile as aligned dwords (coalesced); consumers add the byte skew.
    // One wavefront per tile, lane = dword of the run: the run's address is wave-uniform, so the common case (the whole
    // run inside the caller's buffer) is a 32-bit lane offset from a scalar base.
    const size_t total_bytes = size_t(g.frames) * g.h * g.w * C;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (every load of a wavefront is issued before the first result is stored to LDS: one memory latency per block instead
    // of one per tile -- the kernel is bound by the latency of its phases, not by bandwidth or arithmetic)
    constexpr int TPW = (TPG + 3) / 4, DPT = (RUNW + 63) / 64;  // tiles per wavefront, dwords per lane and tile
    uint32_t got[TPW][DPT];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const uint32_t tt = wave + 4 * t;
#pragma unroll
        for (int e = 0; e < DPT; ++e) got[t][e] = 0;
        if (tt < ntiles) {
            const long long start = (long long)tiles[tt].base + (long long)(int(k0) - 2) * C;  // may be < 0 for the first run
            const unsigned long long a0u = (unsigned long long)(start & ~3ll);
            const long long a0 = (long long)((unsigned long long)__builtin_amdgcn_readfirstlane(uint32_t(a0u)) |
                                             ((unsigned long long)__builtin_amdgcn_readfirstlane(uint32_t(a0u >> 32)) << 32));
            if (a0 >= 0 && size_t(a0) + 4 * size_t(RUNW) <= total_bytes) {
#pragma unroll
                for (int e = 0; e < DPT; ++e) {
                    const uint32_t d = lane + 64 * e;
                    if (d < uint32_t(RUNW)) __builtin_memcpy(&got[t][e], px + a0 + 4 * d, 4);  // (px itself may be unaligned)
                }
            } else {  // first / last run of the buffer: never a byte outside it, the partial dword comes from byte loads
#pragma unroll
                for (int e = 0; e < DPT; ++e) {
                    const long long al = a0 + 4ll * (lane + 64 * e);
                    if (al >= 0 && size_t(al) + 4 <= total_bytes) __builtin_memcpy(&got[t][e], px + al, 4);
                    else if (al >= 0 && size_t(al) < total_bytes) got[t][e] = load_bytes_le(px + al, uint32_t(total_bytes - size_t(al)));
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const uint32_t tt = wave + 4 * t;
#pragma unroll
        for (int e = 0; e < DPT; ++e) {
            const uint32_t d = lane + 64 * e;
            if (tt < ntiles && d < uint32_t(RUNW)) raw[tt][d] = got[t][e];
        }
    }
    __syncthreads();
    // this thread: pixels k0 + 8*q + i, i = 0..7, of tile tt.  Two pixels per register (packed 16-bit arithmetic: every
    // value of the path fits -- samples 0..255, Co/Cg and L - l in [-255, 255], residuals in [-510, 510]); pair j of a
    // channel holds pixels kb-2+2j and kb-1+2j, so the pair of current samples, of their left neighbours l and of their
    // left-left neighbours L are registers j+1, (j, j+1) shifted by one sample, and j.
    const uint32_t q = threadIdx.x & 7;
    const bool small = (g.flags & kGeoSmallModel) != 0;
    for (uint32_t tt = threadIdx.x >> 3; tt < ntiles; tt += 32) {  // one pass for 3 and 4 channels (<= 23 tiles)
        const uint32_t sw = tiles[tt].sw;
        const uint32_t kb = k0 + 8 * q;
        if (kb >= sw) continue;
        const uint32_t skew = uint32_t(((long long)tiles[tt].base + (long long)(int(k0) - 2) * C) & 3);
        const uint8_t* p = reinterpret_cast<const uint8_t*>(&raw[tt][0]) + skew + 8 * q * C;  // pixel kb - 2
        s16x2 V[C][5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint8_t* a = p + 2 * j * C;
            if constexpr (C >= 3) {  // llcomp.hpp:396-409: [r - g, g + (b - g + r - g) / 4 (truncating), b - g, extras]
                const s16x2 R = {short(a[0]), short(a[C])}, G = {short(a[1]), short(a[C + 1])}, B = {short(a[2]), short(a[C + 2])};
                const s16x2 cr = R - G, cb = B - G, sum = cb + cr;
                V[0][j] = cr;
                V[1][j] = G + ((sum + ((sum >> 15) & short(3))) >> 2);
                V[2][j] = cb;
                if constexpr (C == 4) V[3][j] = s16x2{short(a[3]), short(a[C + 3])};
            } else {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) V[ch][j] = s16x2{short(a[ch]), short(a[C + ch])};
            }
        }
        // slice start (llcomp.hpp:417-419): l of sample 0 is 128 and L = l for samples 0 and 1, i.e. L - l = 0 there
        uint32_t first_mask = ~0u;
        if (kb == 0) {
            first_mask = 0;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) V[ch][0] = s16x2{128, 128};
        }
        if (small) first_mask = 0;  // LargeModel = false: no quant5 term at all, context 0 (handled below for all pairs)
        // group-relative lane of this tile's channel 0, + PADL (never negative: see otile)
        const uint32_t colp = uint32_t(int((first_tile + tt) * C) - int(first_id) + PADL);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                const s16x2 cur = V[ch][m + 1], Lp = V[ch][m];
                const s16x2 lp = {V[ch][m].y, V[ch][m + 1].x};  // left neighbours of the two current samples
                s16x2 dq = Lp - lp;
                uint32_t dqb = __builtin_bit_cast(uint32_t, dq) & ((m == 0 || small) ? first_mask : ~0u);
                dq = __builtin_bit_cast(s16x2, dqb);
                const s16x2 neg = dq >> 15;                          // hash = 605*quant5(L - l) < 0: llcomp.hpp:433-436
                const u16x2 aq = __builtin_bit_cast(u16x2, (dq ^ neg) - neg);
                const u16x2 one = {1, 1};
                const u16x2 cidx = pk_min_u16(aq, one) + pk_min_u16(aq >> 2, one);  // |quant5|: 0 / 1 / 2
                const s16x2 res = ((cur - lp) ^ neg) - neg;
                // 16-bit symbol of the fused path: bits 12..13 = |quant5|, bits 0..11 = residual
                const u16x2 sym = (__builtin_bit_cast(u16x2, res) & (unsigned short)0xFFF) + (cidx << 12);
                otile[8 * q + 2 * m][colp + ch] = sym.x;
                otile[8 * q + 2 * m + 1][colp + ch] = sym.y;
            }
        }
    }
    __syncthreads();
    // rows of the lane-order array: sample k0 + kk of the group's lanes = one contiguous 2*gw-byte piece
    if (g.lane_shift == 6) {
        for (uint32_t i = threadIdx.x; i < uint32_t(K) * 32; i += 256) {
            const uint32_t kk = i >> 5, d = i & 31, k = k0 + kk;
            if (k < g.tile_w)
                *reinterpret_cast<uint32_t*>(lanes + lane_order_index(g, first_id + 2 * d, k)) =
                    *reinterpret_cast<const uint32_t*>(&otile[kk][PADL + 2 * d]);
        }
    } else {  // fewer than 64 slices in total: narrow group, plain element stores
        for (uint32_t i = threadIdx.x; i < uint32_t(K) * gw; i += 256) {
            const uint32_t kk = i / gw, col = i - kk * gw, k = k0 + kk;
            if (k < g.tile_w && first_id + col < end_id) lanes[lane_order_index(g, first_id + col, k)] = otile[kk][PADL + col];
        }
    }
}

// Decode side: lane-order reconstructed samples -> pixels.  One block = the tiles whose FIRST channel plane lies in one
// lane group x 64 samples; the other planes of the last tile may sit in the next group, hence 64 + C - 1 lanes.
// Each thread turns 4 consecutive pixels into 4*C bytes and stores them as C (unaligned) dwords.
template <int C>
__global__ __launch_bounds__(256) void k_model_rows_inv(const Geometry g, const int16_t* __restrict__ lanes,
                                                        uint8_t* __restrict__ px) {
    constexpr int K = 64, TPG = 64 / C + 2;
    __shared__ __attribute__((aligned(4))) int16_t tile[K][64 + C + 1 + ((C + 1) & 1)];  // even row length: dword rows
    __shared__ RowTile tiles[TPG];
    const uint32_t chunks = (g.tile_w + K - 1) / K;
    uint32_t group, chunk;
    if (!xcd_chunk_group((g.n_slices + (1u << g.lane_shift) - 1) >> g.lane_shift, chunks, group, chunk)) return;  // (uniform per block)
    const uint32_t k0 = chunk * K;
    const uint32_t gw = 1u << g.lane_shift;
    const uint32_t first_id = group << g.lane_shift;
    const uint32_t end_id = first_id + gw < g.n_slices ? first_id + gw : g.n_slices;
    const uint32_t first_tile = (first_id + C - 1) / C;          // first tile whose channel 0 is in this group
    const uint32_t end_tile = (end_id + C - 1) / C;              // one past the last such tile
    if (first_tile >= end_tile) return;
    const uint32_t ntiles = end_tile - first_tile;
    load_row_tiles<C>(g, first_tile, ntiles, tiles);
    // LDS column = lane index relative to this group (0..gw-1), columns gw.. = first C-1 lanes of the NEXT group.
    // Rows of this group are read as whole 128-byte pieces (32 dwords = 64 samples); the few extra lanes one by one.
    {   // (all eight loads of a thread are in flight before the first one is stored to LDS: the kernel is bound by the
        // latency of its phases, one memory round trip per block instead of eight)
        uint32_t w[K * 32 / 256];
#pragma unroll
        for (int it = 0; it < K * 32 / 256; ++it) {
            const uint32_t i = threadIdx.x + 256 * it, kk = i >> 5, d = i & 31, k = k0 + kk;
            w[it] = 0;
            if (k < g.tile_w && g.lane_shift == 6)
                w[it] = *reinterpret_cast<const uint32_t*>(lanes + lane_order_index(g, first_id + 2 * d, k));
        }
#pragma unroll
        for (int it = 0; it < K * 32 / 256; ++it) {
            const uint32_t i = threadIdx.x + 256 * it, kk = i >> 5, d = i & 31;
            *reinterpret_cast<uint32_t*>(&tile[kk][2 * d]) = w[it];
        }
    }
    if (g.lane_shift != 6) {  // fewer than 64 slices in total: narrow group, plain element loads
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < uint32_t(K) * gw; i += 256) {
            const uint32_t kk = i / gw, col = i - kk * gw, k = k0 + kk;
            tile[kk][col] = (first_id + col < g.n_slices && k < g.tile_w) ? lanes[lane_order_index(g, first_id + col, k)] : int16_t(0);
        }
    }
    if constexpr (C > 1) {
        for (uint32_t i = threadIdx.x; i < uint32_t(K) * (C - 1); i += 256) {
            const uint32_t kk = i / (C - 1), e = i - kk * (C - 1), k = k0 + kk;
            const uint32_t id = first_id + gw + e;
            tile[kk][gw + e] = (id < g.n_slices && k < g.tile_w) ? lanes[lane_order_index(g, id, k)] : int16_t(0);
        }
    }
    __syncthreads();
    const uint32_t col0 = first_tile * C - first_id;  // group-relative lane of the first tile's channel 0
    const uint32_t q4 = threadIdx.x & 15;  // which group of 4 pixels of the 64-sample chunk
    for (uint32_t tt = threadIdx.x >> 4; tt < ntiles; tt += 16) {
        const uint32_t sw = tiles[tt].sw, kb = k0 + 4 * q4;
        if (kb >= sw) continue;
        uint8_t bytes[4 * C];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int16_t* s = &tile[4 * q4 + i][col0 + tt * C];
            if constexpr (C >= 3) {  // llcomp.hpp:532-543
                int r = s[0], gg = s[1], bb = s[2];
                gg -= (r + bb) / 4;
                r += gg;
                bb += gg;
                bytes[i * C + 0] = uint8_t(min(max(r, 0), 255));
                bytes[i * C + 1] = uint8_t(min(max(gg, 0), 255));
                bytes[i * C + 2] = uint8_t(min(max(bb, 0), 255));
                if constexpr (C == 4) bytes[i * C + 3] = uint8_t(s[3]);
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) bytes[i * C + c] = uint8_t(s[c]);
            }
        }
        uint8_t* o = px + tiles[tt].base + size_t(kb) * C;
        const uint32_t npx = sw - kb < 4 ? sw - kb : 4;
        if (npx == 4) {
            uint32_t w[C];
#pragma unroll
            for (int c = 0; c < C; ++c)
                w[c] = uint32_t(bytes[4 * c]) | (uint32_t(bytes[4 * c + 1]) << 8) | (uint32_t(bytes[4 * c + 2]) << 16) |
                       (uint32_t(bytes[4 * c + 3]) << 24);
            __builtin_memcpy(o, w, 4 * C);  // pixel rows start at arbitrary byte offsets: unaligned dword stores
        } else {
            for (uint32_t i = 0; i < npx * C; ++i) o[i] = bytes[i];
        }
    }
}

}  // namespace

#define LLMI_DISPATCH_C(c, CALL) \
    switch (c) {                 \
        case 1: { constexpr int C = 1; CALL; } break; \
        case 2: { constexpr int C = 2; CALL; } break; \
        case 3: { constexpr int C = 3; CALL; } break; \
        case 4: { constexpr int C = 4; CALL; } break; \
        default: return hipErrorInvalidValue; \
    }

hipError_t launch_model_fwd(const Geometry& g, const uint8_t* d_px, uint32_t* d_sym, hipStream_t stream) {
    if (g.c > 4) {
        const size_t npix = size_t(g.frames) * g.h * g.w;
        k_model_fwd_any<<<dim3(uint32_t((npix + 255) / 256)), dim3(256), 0, stream>>>(g, d_px, d_sym, npix);
        return hipGetLastError();
    }
    const uint32_t nbx = (g.w + kMW - 1) / kMW;
    const uint32_t spt = (g.tile_h + kMH - 1) / kMH;
    const uint64_t blocks = uint64_t(nbx) * spt * g.nty * g.frames;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    LLMI_DISPATCH_C(g.c, (k_model_fwd<C><<<dim3(uint32_t(blocks)), dim3(kMW), 0, stream>>>(g, d_px, d_sym)));
    return hipGetLastError();
}

uint32_t lane_groups(const Geometry& g) { return (g.n_slices + (1u << g.lane_shift) - 1) >> g.lane_shift; }
uint32_t slice_capacity_samples(const Geometry& g) { return g.slice_samples; }

hipError_t launch_to_lane_order_u32(const Geometry& g, const uint32_t* d_img, uint32_t* d_lanes, hipStream_t stream) {
    const uint32_t max_n = slice_capacity_samples(g);
    const uint64_t blocks = uint64_t(lane_groups(g)) * ((max_n + 63) / 64);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    k_to_lane_order<uint32_t><<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(g, max_n, d_img, d_lanes);
    return hipGetLastError();
}

hipError_t launch_from_lane_order_i16(const Geometry& g, const int16_t* d_lanes, int16_t* d_img, hipStream_t stream) {
    const uint32_t max_n = slice_capacity_samples(g);
    const uint64_t blocks = uint64_t(lane_groups(g)) * ((max_n + 63) / 64);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    k_from_lane_order<int16_t><<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(g, max_n, d_lanes, d_img);
    return hipGetLastError();
}

bool model_is_fused(const Geometry& g) { return g.planar && rows_mode(g) && g.c <= 4; }

hipError_t launch_model_rows_fwd(const Geometry& g, const uint8_t* d_px, uint16_t* d_lanes, hipStream_t stream) {
    const uint64_t blocks = xcd_grid(lane_groups(g), (g.tile_w + 63) / 64);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    LLMI_DISPATCH_C(g.c, (k_model_rows_fwd<C><<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(g, d_px, d_lanes)));
    return hipGetLastError();
}

hipError_t launch_model_rows_inv(const Geometry& g, const int16_t* d_lanes, uint8_t* d_px, hipStream_t stream) {
    const uint64_t blocks = xcd_chunk_grid(lane_groups(g), (g.tile_w + 63) / 64);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    LLMI_DISPATCH_C(g.c, (k_model_rows_inv<C><<<dim3(uint32_t(blocks)), dim3(256), 0, stream>>>(g, d_lanes, d_px)));
    return hipGetLastError();
}

hipError_t launch_model_inv(const Geometry& g, const int16_t* d_rec, uint8_t* d_px, hipStream_t stream) {
    const size_t npix = size_t(g.frames) * g.h * g.w;
    if (g.c > 4) {
        k_model_inv_any<<<dim3(uint32_t((npix + 255) / 256)), dim3(256), 0, stream>>>(g, d_rec, d_px, npix);
        return hipGetLastError();
    }
    const uint32_t blocks = uint32_t(std::min<size_t>((npix + 255) / 256, 256 * 16));
    LLMI_DISPATCH_C(g.c, (k_model_inv<C><<<dim3(blocks), dim3(256), 0, stream>>>(g, d_rec, d_px, npix)));
    return hipGetLastError();
}

hipError_t launch_group_sums(const Geometry& g, const uint32_t* d_slice_len, uint64_t* d_group_off, hipStream_t stream) {
    k_group_sums<<<dim3((g.n_slices + kSumThreads - 1) / kSumThreads), dim3(kSumThreads), 0, stream>>>(d_slice_len, g.n_slices, g.lane_shift, d_group_off);
    return hipGetLastError();
}

hipError_t launch_frame_bytes(const Geometry& g, const uint32_t* d_slice_len, uint64_t* d_frame_bytes, hipStream_t stream) {
    k_frame_bytes<<<dim3(g.frames), dim3(256), 0, stream>>>(d_slice_len, g.slices_per_frame, d_frame_bytes);
    return hipGetLastError();
}

hipError_t launch_scan_groups(const Geometry& g, uint64_t* d_group_off, uint64_t* d_total, hipStream_t stream) {
    k_scan_groups<<<dim3(1), dim3(kScanThreads), 0, stream>>>(d_group_off, lane_groups(g), d_total);
    return hipGetLastError();
}

hipError_t launch_copy_segments(const uint8_t* d_src, uint8_t* d_dst, const uint64_t* d_src_off, const uint64_t* d_dst_off,
                                const uint64_t* d_len, uint32_t n_seg, uint64_t max_len, hipStream_t stream) {
    if (n_seg == 0) return hipSuccess;
    if (n_seg > 65535) return hipErrorInvalidValue;
    const uint64_t pieces = (max_len / 4 + kSegPieceDwords - 1) / kSegPieceDwords;
    const uint32_t gx = uint32_t(std::min<uint64_t>(std::max<uint64_t>(pieces, 1), 1024));
    k_copy_segments<<<dim3(gx, n_seg), dim3(256), 0, stream>>>(d_src, d_dst, d_src_off, d_dst_off, d_len);
    return hipGetLastError();
}

hipError_t launch_range_sums(const uint32_t* d_vals, const uint64_t* d_start, const uint64_t* d_count, uint64_t* d_out, uint32_t n,
                             uint32_t cap, hipStream_t stream) {
    if (!n) return hipSuccess;
    k_range_sums<<<dim3(n), dim3(256), 0, stream>>>(d_vals, d_start, d_count, d_out, cap);
    return hipGetLastError();
}

hipError_t launch_pack_payload(const Geometry& g, const uint8_t* d_units, const uint32_t* d_slice_len,
                               const uint64_t* d_offsets, uint8_t* d_payload, uint64_t payload_cap,
                               uint32_t* d_status, hipStream_t stream) {
    k_pack_payload<<<dim3(lane_groups(g)), dim3(256), 0, stream>>>(g, reinterpret_cast<const uint4*>(d_units), d_slice_len,
                                                                   d_offsets, d_payload, payload_cap, d_status);
    return hipGetLastError();
}

hipError_t launch_stage_streams(const Geometry& g, const uint8_t* d_payload, uint64_t payload_bytes,
                                const uint32_t* d_slice_len, const uint64_t* d_offsets, uint8_t* d_units,
                                uint32_t* d_status, hipStream_t stream) {
    k_stage_streams<<<dim3(lane_groups(g)), dim3(256), 0, stream>>>(g, d_payload, payload_bytes, d_slice_len, d_offsets,
                                                                    reinterpret_cast<uint4*>(d_units), d_status);
    return hipGetLastError();
}

}  // namespace llcomp_mi
