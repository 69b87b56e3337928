// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the llcomp coding path.
//
//   k_model_fwd   stage A, encode side: colour transform + 6-neighbour context hash + median predictor +
//                 residual, for every sample in parallel.  Coalesced row-major HBM reads, a 4-row LDS ring
//                 of colour-transformed rows (current + the two above + the row being prefetched).
//                 Reference: llcomp.hpp:396-436.
//   k_encode_slices / k_decode_slices
//                 the serial part: one LANE per slice (64 independent slices per wavefront) runs
//                 binarisation, the adaptive 128-state models and the range coder.  The 8 state bytes of a
//                 context travel as one 64-bit word; the 128-entry model table sits in LDS as one packed
//                 dword per state.  Reference: llcomp.hpp:33-127, 166-247, 283-293, 439-449, 486-530.
//   k_scan_lengths / k_pack_payload
//                 wave-prefix-sum of slice lengths and packing of the variable-length streams.
//   k_model_inv   stage A, decode side: inverse colour transform + clamp.  llcomp.hpp:532-543.
//
// None of this is GEMM-shaped; there is no MFMA here on purpose.  wave = 64 lanes everywhere.
#include "kernels.hpp"
#include "tables.hpp"

namespace llcomp_mi {

namespace {

// ---- model table (constant memory -> LDS at kernel start) ----------------------------------------------
struct PackedTable {
    uint32_t v[128];
};
constexpr PackedTable make_packed() {
    PackedTable t{};
    for (uint32_t s = 0; s < 128; ++s) t.v[s] = packed_state(s);
    return t;
}
__constant__ PackedTable c_packed = make_packed();

__device__ __forceinline__ void load_table(uint32_t* tab) {
    for (uint32_t i = threadIdx.x; i < 128; i += blockDim.x) tab[i] = c_packed.v[i];
    __syncthreads();
}

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
__device__ __forceinline__ int context_hash(const Hood& n) {
    return quant11(n.l - n.tl) + 11 * quant11(n.tl - n.t) + 121 * quant11(n.t - n.tr) + 605 * quant5(n.L - n.l) +
           3025 * quant5(n.T - n.t);
}
__device__ __forceinline__ int predict(const Hood& n) { return median3(n.l, n.l + n.t - n.tl, n.t); }

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
    uint32_t* sbase = sym + size_t(frame) * g.h * row_bytes;

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
                int ctx = context_hash(n);
                int res = cur - predict(n);
                if (ctx < 0) {  // llcomp.hpp:433-436
                    ctx = -ctx;
                    res = -res;
                }
                out[k] = uint32_t(ctx) | (uint32_t(res) << 16);
            }
            uint32_t* o = sbase + size_t(tile_y0 + ly) * row_bytes + size_t(x) * C;
#pragma unroll
            for (int k = 0; k < C; ++k) o[k] = out[k];
        }
        if (more) commit_row((ly + 1) & 3, ra, rb);  // slot (ly+1)&3 == (ly-3)&3: not read this iteration
        __syncthreads();
    }
}

// ---- stage A, decode side -------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void k_model_inv(const int16_t* __restrict__ rec, uint8_t* __restrict__ px,
                                                   size_t npix) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < npix; i += size_t(gridDim.x) * blockDim.x) {
        const int16_t* s = rec + i * C;
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

// ---- range encoder, one per lane (llcomp.hpp:33-89) -------------------------------------------------------
struct RangeEnc {
    uint32_t low, range, pend;
    int32_t held;
    uint8_t* out;
    uint32_t pos, cap;
};
__device__ __forceinline__ void enc_emit(RangeEnc& e, uint32_t b) {
    if (e.pos < e.cap) e.out[e.pos] = uint8_t(b);
    ++e.pos;  // keeps counting so an overflow is detected, never written
}
// one renormalisation step (the body of the reference's `while (range < 0x100)`; one step always suffices:
// after put() range >= 7, in finish() range == 0xFF)
__device__ __forceinline__ void enc_shift(RangeEnc& e) {
    const uint32_t carry = e.low >> 16;  // 0 or 1: low >= 0x10000
    const bool flush = e.held >= 0 && (e.low <= 0xFF00u || carry);
    if (flush) {
        enc_emit(e, uint32_t(e.held) + carry);
        const uint32_t fill = carry ? 0x00u : 0xFFu;
        for (; e.pend; --e.pend) enc_emit(e, fill);
    }
    if (e.held < 0 || flush)
        e.held = int32_t((e.low >> 8) & 0xFF);
    else
        ++e.pend;
    e.low = (e.low & 0xFF) << 8;
    e.range <<= 8;
}

template <int SLOT>
__device__ __forceinline__ void enc_bin(RangeEnc& e, uint32_t (&bank)[2], const uint32_t* tab, bool bit) {
    constexpr int W = SLOT >> 2, SH = (SLOT & 3) * 8;
    const uint32_t pk = tab[(bank[W] >> SH) & 0xFF];
    const uint32_t r1 = (e.range * (pk & 0xFF)) >> 8;
    if (bit) {
        e.low += e.range - r1;
        e.range = r1;
    } else {
        e.range -= r1;
    }
    const uint32_t ns = bit ? (pk >> 16) & 0xFF : (pk >> 8) & 0xFF;
    bank[W] = (bank[W] & ~(0xFFu << SH)) | (ns << SH);
    if (e.range < 0x100) enc_shift(e);
}

// putSymbol<true,4,6,7> (llcomp.hpp:166-206) with compile-time slots: because every lane of the wave enters
// each phase together, the slot of the i-th exponent / mantissa bin is wave-uniform.
__device__ __forceinline__ void enc_residual(RangeEnc& e, uint32_t (&bank)[2], const uint32_t* tab, int res) {
    if (res == 0) {
        enc_bin<0>(e, bank, tab, true);
        return;
    }
    enc_bin<0>(e, bank, tab, false);
    const uint32_t a = uint32_t(res < 0 ? -res : res);
    const int ex = 31 - __clz(int(a));
    enc_bin<1>(e, bank, tab, ex > 0);
    if (ex > 0) {
        enc_bin<2>(e, bank, tab, ex > 1);
        if (ex > 1) {
            enc_bin<3>(e, bank, tab, ex > 2);
            if (ex > 2) {
                int i = 3;
                bool b;
                do {
                    b = ex > i;
                    enc_bin<4>(e, bank, tab, b);
                    ++i;
                } while (b);
            }
        }
        enc_bin<5>(e, bank, tab, (a >> (ex - 1)) & 1);
        for (int i = ex - 2; i >= 0; --i) enc_bin<6>(e, bank, tab, (a >> i) & 1);
    }
    enc_bin<7>(e, bank, tab, res < 0);
}

template <int NCH>
__global__ __launch_bounds__(64) void k_encode_slices(const Geometry g, const uint32_t* __restrict__ sym,
                                                      uint64_t* __restrict__ states, uint8_t* __restrict__ scratch,
                                                      uint32_t* __restrict__ slice_len, uint32_t* status) {
    __shared__ uint32_t tab[128];
    load_table(tab);
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    if (id >= g.n_slices) return;
    const SliceRect r = slice_rect(g, id);
    uint64_t* banks = states + size_t(id) * kContexts;
    RangeEnc e;
    e.low = 0; e.range = 0xFF00; e.pend = 0; e.held = -1;  // llcomp.hpp:35
    e.out = scratch + size_t(id) * g.slice_cap; e.pos = 0; e.cap = g.slice_cap;
    const size_t row_stride = size_t(g.w) * g.c;
    const uint32_t* p0 = sym + (size_t(r.frame) * g.h + r.y0) * row_stride + size_t(r.x0) * g.c + r.ch;
    for (uint32_t y = 0; y < r.sh; ++y) {
        const uint32_t* row = p0 + size_t(y) * row_stride;
        for (uint32_t x = 0; x < r.sw; ++x) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const uint32_t s = row[size_t(x) * g.c + k];
                const uint32_t ctx = s & 0xFFFF;
                const int res = int(s) >> 16;
                const uint64_t b64 = banks[ctx];
                uint32_t bank[2] = {uint32_t(b64), uint32_t(b64 >> 32)};
                enc_residual(e, bank, tab, res);
                banks[ctx] = uint64_t(bank[0]) | (uint64_t(bank[1]) << 32);
            }
        }
    }
    // finish(), llcomp.hpp:75-81
    e.range = 0xFF; e.low += 0xFF; enc_shift(e);
    e.range = 0xFF; enc_shift(e);
    if (e.pos > e.cap) {
        atomicOr(status, kStOverflow);
        e.pos = e.cap;
    }
    slice_len[id] = e.pos;
}

// ---- range decoder, one per lane (llcomp.hpp:91-127) ------------------------------------------------------
struct RangeDec {
    uint32_t low, range;
    const uint8_t* p;
    uint32_t pos, len;
};
__device__ __forceinline__ uint32_t dec_byte(RangeDec& d) {  // llcomp.hpp:475-479: past the end reads 0
    uint32_t b = 0;
    if (d.pos < d.len) b = d.p[d.pos];
    ++d.pos;
    return b;
}
template <int SLOT>
__device__ __forceinline__ bool dec_bin(RangeDec& d, uint32_t (&bank)[2], const uint32_t* tab) {
    constexpr int W = SLOT >> 2, SH = (SLOT & 3) * 8;
    const uint32_t pk = tab[(bank[W] >> SH) & 0xFF];
    const uint32_t r1 = (d.range * (pk & 0xFF)) >> 8;
    d.range -= r1;
    const bool bit = d.low >= d.range;
    if (bit) {
        d.low -= d.range;
        d.range = r1;
    }
    const uint32_t ns = bit ? (pk >> 16) & 0xFF : (pk >> 8) & 0xFF;
    bank[W] = (bank[W] & ~(0xFFu << SH)) | (ns << SH);
    if (d.range < 0x100) {
        d.range <<= 8;
        d.low = (d.low << 8) + dec_byte(d);
    }
    return bit;
}
// getSymbol<true,4,6,7> (llcomp.hpp:219-247).  Returns false on "Invalid exponent".  Arithmetic modulo 2^32.
__device__ __forceinline__ bool dec_residual(RangeDec& d, uint32_t (&bank)[2], const uint32_t* tab, uint32_t& out) {
    if (dec_bin<0>(d, bank, tab)) {
        out = 0;
        return true;
    }
    int ex = 0;
    if (dec_bin<1>(d, bank, tab)) {
        ex = 1;
        if (dec_bin<2>(d, bank, tab)) {
            ex = 2;
            if (dec_bin<3>(d, bank, tab)) {
                ex = 3;
                while (dec_bin<4>(d, bank, tab)) {
                    if (++ex > 31) return false;
                }
            }
        }
    }
    uint32_t v = 1;
    if (ex > 0) {
        v += v + uint32_t(dec_bin<5>(d, bank, tab));
        for (int j = 1; j < ex; ++j) v += v + uint32_t(dec_bin<6>(d, bank, tab));
    }
    if (dec_bin<7>(d, bank, tab)) v = 0u - v;
    out = v;
    return true;
}

template <int NCH>
__global__ __launch_bounds__(64) void k_decode_slices(const Geometry g, const uint8_t* __restrict__ payload,
                                                      const uint64_t payload_bytes,
                                                      const uint32_t* __restrict__ slice_len,
                                                      const uint64_t* __restrict__ offsets,
                                                      uint64_t* __restrict__ states, int16_t* __restrict__ rec,
                                                      uint32_t* status) {
    __shared__ uint32_t tab[128];
    load_table(tab);
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    if (id >= g.n_slices) return;
    const SliceRect r = slice_rect(g, id);
    uint64_t* banks = states + size_t(id) * kContexts;
    RangeDec d;
    const uint64_t off = offsets[id];
    uint64_t len = slice_len[id];
    if (off + len > payload_bytes) {  // slice table promises more than the data holds
        atomicOr(status, kStTruncated);
        len = off < payload_bytes ? payload_bytes - off : 0;
    }
    d.p = payload + (off < payload_bytes ? off : 0);
    d.pos = 0;
    d.len = uint32_t(len);
    d.range = 0xFF00;  // llcomp.hpp:93-96
    d.low = dec_byte(d) << 8;
    d.low |= dec_byte(d);

    const ptrdiff_t ps = g.c;                      // pixel stride in samples
    const ptrdiff_t rs = ptrdiff_t(g.w) * g.c;     // row stride in samples
    int16_t* p0 = rec + (size_t(r.frame) * g.h + r.y0) * rs + size_t(r.x0) * ps + r.ch;
    for (uint32_t y = 0; y < r.sh; ++y) {
        int16_t* row = p0 + ptrdiff_t(y) * rs;
        int l[NCH], L[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) l[k] = L[k] = 0;
        for (uint32_t x = 0; x < r.sw; ++x) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                int16_t* q = row + ptrdiff_t(x) * ps + k;
                const int t_raw = y > 0 ? q[-rs] : 0;
                const int tl_raw = (y > 0 && x > 0) ? q[-rs - ps] : 0;
                const int tr_raw = (y > 0 && x + 1 < r.sw) ? q[-rs + ps] : 0;
                const int T_raw = y > 1 ? q[-2 * rs] : 0;
                const Hood n = apply_borders(l[k], L[k], t_raw, tl_raw, tr_raw, T_raw, x, y, r.sw);
                int ctx = context_hash(n);
                const bool neg = ctx < 0;  // llcomp.hpp:511-515
                if (neg) ctx = -ctx;
                const uint64_t b64 = banks[ctx];
                uint32_t bank[2] = {uint32_t(b64), uint32_t(b64 >> 32)};
                uint32_t v;
                if (!dec_residual(d, bank, tab, v)) {
                    atomicOr(status, kStBadExponent);
                    return;  // this lane's slice is unusable; the whole call reports the error
                }
                banks[ctx] = uint64_t(bank[0]) | (uint64_t(bank[1]) << 32);
                if (neg) v = 0u - v;
                const int val = int(int16_t(uint32_t(predict(n)) + v));
                *q = int16_t(val);
                L[k] = l[k];
                l[k] = val;
            }
        }
    }
}

// ---- slice length scan + payload packing -------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan_lengths(const uint32_t* __restrict__ len, uint32_t n,
                                                       uint64_t* __restrict__ off, uint64_t* total) {
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const unsigned long long v = i < n ? len[i] : 0;
        unsigned long long inc = v;  // wave-inclusive prefix sum, 64 lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(inc, d, 64);
            if (lane >= uint32_t(d)) inc += o;
        }
        if (lane == 63) wave_sum[wv] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (uint32_t k = 0; k < wv; ++k) before += wave_sum[k];
        if (i < n) off[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[n] = carry;
        *total = carry;
    }
}

__global__ __launch_bounds__(256) void k_pack_payload(const Geometry g, const uint8_t* __restrict__ scratch,
                                                      const uint32_t* __restrict__ slice_len,
                                                      const uint64_t* __restrict__ off, uint8_t* __restrict__ payload,
                                                      uint64_t payload_cap, uint32_t* status) {
    for (uint32_t id = blockIdx.x; id < g.n_slices; id += gridDim.x) {
        const uint32_t n = slice_len[id];
        const uint64_t o = off[id];
        if (o + n > payload_cap) {
            if (threadIdx.x == 0) atomicOr(status, kStOverflow);
            continue;
        }
        const uint8_t* src = scratch + size_t(id) * g.slice_cap;  // 16-byte aligned
        uint8_t* dst = payload + o;
        // head: bytes up to the first 4-byte boundary of dst
        const uint32_t head = min(n, uint32_t((4 - (uintptr_t(dst) & 3)) & 3));
        if (threadIdx.x < head) dst[threadIdx.x] = src[threadIdx.x];
        const uint32_t words = (n - head) >> 2;
        const uint32_t m = head & 3;  // misalignment of src + head
        const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src + head - m);
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst + head);
        for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) {
            const uint32_t lo = s32[i];
            const uint32_t hi = m ? s32[i + 1] : 0;  // stays inside the slice's scratch (cap has 16 B slack)
            d32[i] = __builtin_amdgcn_alignbyte(hi, lo, m);
        }
        const uint32_t done = head + (words << 2);
        if (threadIdx.x < n - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
    }
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------------------
#define LLMI_DISPATCH_C(c, CALL) \
    switch (c) {                 \
        case 1: { constexpr int C = 1; CALL; } break; \
        case 2: { constexpr int C = 2; CALL; } break; \
        case 3: { constexpr int C = 3; CALL; } break; \
        case 4: { constexpr int C = 4; CALL; } break; \
        default: return hipErrorInvalidValue; \
    }

hipError_t launch_model_fwd(const Geometry& g, const uint8_t* d_px, uint32_t* d_sym, hipStream_t stream) {
    const uint32_t nbx = (g.w + kMW - 1) / kMW;
    const uint32_t spt = (g.tile_h + kMH - 1) / kMH;
    const uint64_t blocks = uint64_t(nbx) * spt * g.nty * g.frames;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    LLMI_DISPATCH_C(g.c, (k_model_fwd<C><<<dim3(uint32_t(blocks)), dim3(kMW), 0, stream>>>(g, d_px, d_sym)));
    return hipGetLastError();
}

hipError_t launch_model_inv(const Geometry& g, const int16_t* d_rec, uint8_t* d_px, hipStream_t stream) {
    const size_t npix = size_t(g.frames) * g.h * g.w;
    const uint32_t blocks = uint32_t(std::min<size_t>((npix + 255) / 256, 256 * 16));
    LLMI_DISPATCH_C(g.c, (k_model_inv<C><<<dim3(blocks), dim3(256), 0, stream>>>(d_rec, d_px, npix)));
    return hipGetLastError();
}

hipError_t launch_encode_slices(const Geometry& g, const uint32_t* d_sym, uint64_t* d_states, uint8_t* d_scratch,
                                uint32_t* d_slice_len, uint32_t* d_status, hipStream_t stream) {
    const uint32_t blocks = (g.n_slices + 63) / 64;
    LLMI_DISPATCH_C(g.nch, (k_encode_slices<C><<<dim3(blocks), dim3(64), 0, stream>>>(g, d_sym, d_states, d_scratch,
                                                                                      d_slice_len, d_status)));
    return hipGetLastError();
}

hipError_t launch_scan_lengths(const uint32_t* d_slice_len, uint32_t n, uint64_t* d_offsets, uint64_t* d_total,
                               hipStream_t stream) {
    k_scan_lengths<<<dim3(1), dim3(1024), 0, stream>>>(d_slice_len, n, d_offsets, d_total);
    return hipGetLastError();
}

hipError_t launch_pack_payload(const Geometry& g, const uint8_t* d_scratch, const uint32_t* d_slice_len,
                               const uint64_t* d_offsets, uint8_t* d_payload, uint64_t payload_cap,
                               uint32_t* d_status, hipStream_t stream) {
    const uint32_t blocks = std::min<uint32_t>(g.n_slices, 256 * 32);
    k_pack_payload<<<dim3(blocks), dim3(256), 0, stream>>>(g, d_scratch, d_slice_len, d_offsets, d_payload,
                                                           payload_cap, d_status);
    return hipGetLastError();
}

hipError_t launch_decode_slices(const Geometry& g, const uint8_t* d_payload, uint64_t payload_bytes,
                                const uint32_t* d_slice_len, const uint64_t* d_offsets, uint64_t* d_states,
                                int16_t* d_rec, uint32_t* d_status, hipStream_t stream) {
    const uint32_t blocks = (g.n_slices + 63) / 64;
    LLMI_DISPATCH_C(g.nch, (k_decode_slices<C><<<dim3(blocks), dim3(64), 0, stream>>>(
                               g, d_payload, payload_bytes, d_slice_len, d_offsets, d_states, d_rec, d_status)));
    return hipGetLastError();
}

}  // namespace llcomp_mi
