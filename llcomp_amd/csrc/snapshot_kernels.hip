// snapshot_kernels.hip -- the STATE SNAPSHOT pass of the 2-D encoder (tile_h > 1, several slices per wavefront).
//
// The reference codes one sample after the other and lets every bin read and update states[hash*8 + slot]
// (llcomp.hpp:385, 439-444).  With one LANE per slice that table is 63 KB per lane: it lives in HBM, and every sample costs
// a random 8-byte read-modify-write (131 B of HBM traffic per sample measured for 3 algorithmic).
// But the ENCODER knows every (context, residual) of a slice before it codes anything, and the states do not depend on the
// range coder at all: the eight states a sample will find in its context are a function of the earlier samples OF THAT
// CONTEXT only.  So the table is replaced by three streaming steps, all of them coalesced:
//   k_snap_sort    one workgroup per slice: a stable LSD radix sort of the slice's samples by context in LDS (13-bit key, at most
//                  4096 samples) -> entries {residual, stream position, "first sample of its context"} in context-major order;
//   k_snap_walk    one lane per slice (64 slices per wavefront, like the coder): walks its slice's entries in that order with the
//                  eight states of the current context in REGISTERS -- a context's samples are consecutive now, a new context
//                  starts from zeros -- and leaves the bank as it stood BEFORE every sample; a sample's effect on the eight
//                  states is ten byte look-ups (walk_tables.hpp), no range arithmetic, no divergence;
//   k_snap_unperm  one workgroup per slice: puts the banks back into stream order through LDS.
// The coder (slice_kernels.hip, k_encode_slices<..., SNAP>) then reads one 8-byte bank and one residual per sample, in order.
// Slices ABOVE 4096 samples (64x64 tiles with the channels interleaved, 128x128 planes: the slicings that keep the reference's
// compression ratio) go through the same three kernels in CHUNKS of 4096 consecutive samples, chunk after chunk (round 6; before,
// they read-modify-wrote the table per sample: 130-146 B of HBM traffic per sample, on the GPU's random-transaction ceiling at load,
// profiles/r06_shape_load.txt).  What a chunk needs from its predecessors is, per context it meets, the eight states that context
// was left in: those are CARRIED through the slice's state table in HBM (the decoder's table, generation-tagged: a context no
// earlier chunk met reads as zeros) -- one table read per (chunk, distinct context) in the sorting kernel, which knows the chunk's
// contexts and fetches them all at once (k_snap_sort<.., true>: "init" states at the first sample of every context run), and one
// table write per (chunk, distinct context) from the walk at the end of every run (a store: nothing waits for it).  The walk's
// dependent chain never touches memory it has not been handed in order.  3 launches per chunk, in stream order.
// Every lane-per-slice <-> workgroup-per-slice hand-over uses the PIECE layout below, so that both sides move whole pieces and
// nothing is transposed in a pass of its own.  STORES reach HBM per instruction, as 32-byte sectors (a lone 16-byte store costs
// 32: WRITE_SIZE read 2x the bytes while a thread wrote a piece in several instructions), so every store instruction of these
// kernels covers contiguous memory: neighbouring threads write neighbouring 16-byte chunks, the walk transposes through LDS.
//
// Piece layout of an array: [lane group][piece][lane][P bytes], P = 32 for the entries (eight u32) and the residuals (sixteen
// i16), P = 64 for the banks (eight u64).  A wavefront of the lane-per-slice kernels reads element e of its 64 lanes from one
// run of 64 * P bytes (re-used by the samples that share the piece) and writes whole pieces; a workgroup that owns one slice moves
// pieces 64 * P bytes apart, and the workgroups of the lanes that share a 128-byte line run on the same XCD at about the same
// time (block -> slice mapping below), so the line meets in one L2.  (tools/ubench/piece_access.hip: 32-byte pieces move at
// 2.8 / 4.0 TB/s write / read with that mapping, 0.8 / 1.4 without; a slice's contiguous 32 KB at 5.8 / 6.2.)
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_common.hpp"
#include "kernels.hpp"
#include "snapshot.hpp"
#include "walk_tables.hpp"

namespace llcomp_mi {

namespace {

__constant__ WalkTables c_walk = make_walk_tables();

constexpr uint32_t kSortThreads = 256;
// A thread owns OWN consecutive sort positions: 16 for slices of up to 4096 samples, 8 / 4 for up to 2048 / 1024 (the passes cost
// by the capacity, not by the slice: a 480x2 slice does not pay for 4096 keys).
constexpr uint32_t kMaxOwn = kSnapMaxSamples / kSortThreads;
constexpr uint32_t kRowPad = kSortThreads + 1;  // position p lives at (p % OWN) * 257 + p / OWN: a thread's own positions and 64
                                                // consecutive positions both spread over the LDS banks
template <uint32_t OWN>
__device__ __forceinline__ uint32_t phys(uint32_t p) {
    static_assert((OWN == 4 || OWN == 8 || OWN == 16) && OWN <= kMaxOwn, "a power of two");
    return (p & (OWN - 1)) * kRowPad + p / OWN;
}

// workgroup -> slice, XCD-aware (workgroup i runs on XCD i % 8): the slices of one lane group go to ONE XCD, neighbouring lanes
// to neighbouring slots, so the 32-byte pieces that share a cache line are moved through the same L2 at about the same time
__device__ __forceinline__ bool block_slice(const Geometry& g, uint32_t& id, uint32_t& group, uint32_t& lane) {
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    group = ((slot >> g.lane_shift) << 3) + xcd;
    lane = slot & ((1u << g.lane_shift) - 1);
    id = (group << g.lane_shift) + lane;
    return id < g.n_slices;
}

struct Span {
    size_t origin;   // index of the slice's first sample in the image-order symbol array
    uint32_t n_row;  // samples per slice row (contiguous)
    uint32_t n;      // samples in the slice
};
__device__ __forceinline__ Span slice_span(const Geometry& g, uint32_t id) {
    const SliceRect r = slice_rect(g, id);
    Span s;
    s.origin = slice_origin(g, r);
    s.n_row = r.sw * g.nch;
    s.n = s.n_row * r.sh;
    return s;
}

// inclusive prefix sum over the 64 lanes of a wavefront (DPP: four shifts inside the rows of 16, two row broadcasts)
__device__ __forceinline__ uint32_t wave_scan(uint32_t x) {
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xF, 0xF, false));  // row_shr:1
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xF, 0xF, false));  // row_shr:2
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xF, 0xF, false));  // row_shr:4
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xF, 0xF, false));  // row_shr:8
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
    x += uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
    return x;
}

// One stable counting pass of the radix sort on key bits [SH, SH + BITS).  Thread t owns positions 16t .. 16t+15 of `src` (in
// order) and counts its digits in PRIVATE counters cnt[digit / 2][t] (two 16-bit counters per word: all sums stay below 2^16).
// The counters become start positions in (digit, thread) order -- every wavefront scans whole rows cnt[j][0..255], four
// threads' counters per lane -- and every thread scatters its keys in order.
template <uint32_t kOwn, int SH, int BITS>
__device__ __forceinline__ void sort_pass(const uint32_t* src, uint32_t* dst, uint32_t* cnt, uint32_t* tot) {
    constexpr uint32_t ND2 = (1u << BITS) / 2;  // packed counters per thread = rows
    static_assert(ND2 == 4 || ND2 == 8, "rows per wavefront");
    constexpr uint32_t RPW = ND2 / 4;           // rows each of the four wavefronts scans
    const uint32_t t = threadIdx.x, wave = t >> 6, l = t & 63;
#pragma unroll
    for (uint32_t j = 0; j < ND2; ++j) cnt[j * kSortThreads + t] = 0;
    uint32_t keys[kOwn], slot[kOwn], inc[kOwn];
#pragma unroll
    for (uint32_t i = 0; i < kOwn; ++i) keys[i] = src[i * kRowPad + t];
#pragma unroll
    for (uint32_t i = 0; i < kOwn; ++i) {
        slot[i] = ((keys[i] >> (SH + 1)) & (ND2 - 1)) * kSortThreads + t;
        inc[i] = 1u + ((keys[i] >> SH) & 1u) * 0xFFFFu;  // +1 in the low or in the high half
        atomicAdd(&cnt[slot[i]], inc[i]);                // (private counter: ds_add without return)
    }
    __syncthreads();
    uint4 c[RPW];
    uint32_t excl[RPW];
#pragma unroll
    for (uint32_t u = 0; u < RPW; ++u) {
        const uint32_t j = wave + 4 * u;
        c[u] = *reinterpret_cast<const uint4*>(&cnt[j * kSortThreads + 4 * l]);
        const uint32_t sum = c[u].x + c[u].y + c[u].z + c[u].w;
        const uint32_t incl = wave_scan(sum);
        excl[u] = incl - sum;
        if (l == 63) tot[j] = incl;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < RPW; ++u) {
        const uint32_t j = wave + 4 * u;
        uint32_t run = 0;  // keys with a smaller digit than 2j
#pragma unroll
        for (uint32_t jj = 0; jj < ND2; ++jj) {
            const uint32_t tj = tot[jj];
            run += jj < j ? (tj & 0xFFFFu) + (tj >> 16) : 0u;
        }
        const uint32_t mine = tot[j];
        uint4 o;
        o.x = excl[u] + (run | ((run + (mine & 0xFFFFu)) << 16));  // digit 2j starts at `run`, digit 2j+1 behind it
        o.y = o.x + c[u].x;
        o.z = o.y + c[u].y;
        o.w = o.z + c[u].z;
        *reinterpret_cast<uint4*>(&cnt[j * kSortThreads + 4 * l]) = o;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kOwn; ++i) {
        const uint32_t old = atomicAdd(&cnt[slot[i]], inc[i]);
        const uint32_t pos = (old >> (((keys[i] >> SH) & 1u) * 16)) & 0xFFFFu;
        dst[phys<kOwn>(pos)] = keys[i];
    }
    __syncthreads();
}

// entry of the sorted list: residual (10 bits, two's complement) | stream position << 10 | first-of-its-context << 22
constexpr uint32_t kEntryPosShift = 10, kEntryFirstBit = 22;

// CHUNKED: the workgroup sorts chunk `chunk` (samples chunk * 4096 ...) of its slice; beside the entries it leaves the context of
// every sorted position (u16, pieces of sixteen: the walk stores a finished run's states under its context) and, from the second
// chunk on, the states every context run starts from, fetched from the slice's carry table (`io`, u64 at the run's first position).
template <uint32_t kOwn, bool CHUNKED = false>
__global__ __launch_bounds__(kSortThreads) void k_snap_sort(const Geometry g, const uint32_t cap, const uint32_t* __restrict__ sym,
                                                           uint8_t* __restrict__ entries, const uint32_t chunk, uint8_t* __restrict__ ctx16,
                                                           uint8_t* __restrict__ io, const uint64_t* __restrict__ states, const uint64_t gpat) {
    // ONE key buffer: a pass reads its thread's keys into registers first and scatters them behind two barriers, so source and
    // destination may be the same array.  (With two buffers the kernel held 49 KB of LDS per workgroup; beside the 2-D decoder,
    // whose wavefronts keep 19.5 KB each for their bank cache, a CU then rarely had room for a sorting workgroup at all: 10.1 ms
    // per launch at 48 frames x 3 pipelines where the kernel alone takes 1.6.)
    __shared__ uint32_t key_a[kOwn * kRowPad];
    uint32_t* const key_b = key_a;
    __shared__ uint32_t cnt[8 * kSortThreads];
    __shared__ uint32_t tot[8];
    __shared__ int16_t res_of[kOwn * kSortThreads];
    static_assert(!CHUNKED || kOwn * kSortThreads == kSnapMaxSamples, "a chunk is one full sorting capacity");
    uint32_t id, group, lane;
    if (!block_slice(g, id, group, lane)) return;
    Span sp = slice_span(g, id);
    const uint32_t start = CHUNKED ? chunk * kSnapMaxSamples : 0u;  // first sample of this workgroup's chunk
    if (start >= sp.n) return;                                       // (a ragged slice with fewer chunks; uniform per workgroup)
    const uint32_t n_here = CHUNKED ? min(sp.n - start, kSnapMaxSamples) : sp.n;
    const size_t rs = slice_row_stride(g);
    const uint32_t t = threadIdx.x;
    // keys: context << 12 | position inside the chunk (the low 12 bits ride along; positions beyond the chunk sort to the end)
    {
        uint32_t v[kOwn];
        // sample k = i * 256 + t of the chunk sits in slice row (start + k) / n_row: one division per thread, then steps of 256
        // (32-bit offsets from the slice's first sample)
        const uint32_t rs32 = uint32_t(rs), dy = kSortThreads / sp.n_row, dx = kSortThreads - dy * sp.n_row;
        const uint32_t step_off = dy * rs32 + dx, wrap_off = rs32 - sp.n_row;
        const uint32_t* const base = sym + sp.origin;
        const uint32_t y0 = (start + t) / sp.n_row;
        uint32_t x = (start + t) - y0 * sp.n_row, off = y0 * rs32 + x;
#pragma unroll
        for (uint32_t i = 0; i < kOwn; ++i) {
            v[i] = i * kSortThreads + t < n_here ? base[off] : 0x1FFFu;
            x += dx;
            off += step_off;
            if (x >= sp.n_row) {
                x -= sp.n_row;
                off += wrap_off;
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < kOwn; ++i) {
            const uint32_t k = i * kSortThreads + t;
            key_a[phys<kOwn>(k)] = ((v[i] & 0xFFFFu) << 12) | k;
            res_of[k] = int16_t(v[i] >> 16);
        }
    }
    __syncthreads();
    sort_pass<kOwn, 12, 4>(key_a, key_b, cnt, tot);
    sort_pass<kOwn, 16, 3>(key_b, key_a, cnt, tot);
    sort_pass<kOwn, 19, 3>(key_a, key_b, cnt, tot);
    sort_pass<kOwn, 22, 3>(key_b, key_a, cnt, tot);
    // sorted: key_a.  Entries leave as 32-byte pieces of eight (chunk c's pieces start at piece c * 512).
    uint8_t* const out = entries + size_t(group) * (size_t(cap) * 4 << g.lane_shift) + lane * 32u + (size_t(start >> 3) << (g.lane_shift + 5));
    [[maybe_unused]] uint8_t* const ctx_out = ctx16 + size_t(group) * (size_t(cap) * 2 << g.lane_shift) + lane * 32u + (size_t(start >> 4) << (g.lane_shift + 5));
    [[maybe_unused]] uint8_t* const io_out = io + size_t(group) * (size_t(cap) * 8 << g.lane_shift) + lane * 64u + (size_t(start >> 3) << (g.lane_shift + 6));
    [[maybe_unused]] const uint64_t* const carry = states + ((size_t(group) * kContexts) << g.lane_shift) + lane;  // this slice's table: [context][lane]
    // Four entries (16 bytes) per thread and turn, neighbouring threads the neighbouring chunks of a piece: a store instruction
    // then covers whole 32-byte sectors.  (Stores reach HBM per instruction, as 32-byte sectors: a lone 16-byte store costs 32 --
    // WRITE_SIZE of this kernel read 2x its bytes while one thread wrote both halves of a piece in two instructions.)
    for (uint32_t c = t; c * 4 < n_here; c += kSortThreads) {
        uint32_t e[4], cx[4];
        uint32_t prev = c ? key_a[phys<kOwn>(c * 4 - 1)] >> 12 : ~0u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t key = key_a[phys<kOwn>(c * 4 + j)];
            const uint32_t k = key & 0xFFFu, ctx = key >> 12;
            e[j] = (uint32_t(res_of[k]) & 0x3FFu) | (k << kEntryPosShift) | (uint32_t(ctx != prev) << kEntryFirstBit);
            cx[j] = ctx;
            prev = ctx;
        }
        *reinterpret_cast<uint4*>(out + (size_t(c >> 1) << (g.lane_shift + 5)) + ((c & 1u) << 4)) = make_uint4(e[0], e[1], e[2], e[3]);
        if constexpr (CHUNKED) {
            // contexts of the four positions: 8 bytes, four neighbouring threads fill one 32-byte piece of sixteen
            *reinterpret_cast<uint2*>(ctx_out + (size_t(c >> 2) << (g.lane_shift + 5)) + ((c & 3u) << 3)) = make_uint2(cx[0] | (cx[1] << 16), cx[2] | (cx[3] << 16));
            if (chunk > 0) {  // what every run of this chunk starts from: the states its context was left in by the chunks before
                uint64_t init[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)  // (all requested before the first is looked at; padding positions carry no first flag worth a fetch)
                    init[j] = ((e[j] >> kEntryFirstBit) & 1u) && c * 4 + j < n_here ? carry[size_t(cx[j]) << g.lane_shift] : 0ull;
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    if (((e[j] >> kEntryFirstBit) & 1u) && c * 4 + j < n_here) {
                        const uint32_t p = c * 4 + j;
                        *reinterpret_cast<uint64_t*>(io_out + (size_t(p >> 3) << (g.lane_shift + 6)) + ((p & 7u) << 3)) = bank_fresh<false>(init[j], gpat);
                    }
            }
        }
    }
}

// ---- the walk ----------------------------------------------------------------------------------------------------------------
constexpr uint32_t kWalkThreads = 256;

// (Measured and dropped, round 4: cutting a slice's sorted list at context boundaries into four parts walked by four wavefronts.
// The walk is bound by the CU's vector + LDS throughput -- ten byte look-ups with random bank conflicts per step -- not by the
// length of one lane's chain: four times the wavefronts over ranges that start at different places in every lane took 4.2 ms
// where one wavefront per 64 slices takes 2.3.)
// CARRY (slices above 4096 samples, chunk after chunk): a context run does not start from zeros but from the states the chunks
// before left that context in -- `io` holds them at the run's first position, put there by this chunk's k_snap_sort -- and the run's
// final states go into the slice's carry table under the run's context (`ctx16`), for the chunks to come.  Both ride on data the
// lane is handed in order (prefetched with the entries); the table writes are stores nobody waits for.
template <bool CARRY>
__global__ __launch_bounds__(kWalkThreads) void k_snap_walk(const Geometry g, const uint32_t lpw, const uint32_t cap,
                                                           const uint8_t* __restrict__ entries, uint8_t* __restrict__ sorted_banks,
                                                           const uint32_t chunk, const uint8_t* __restrict__ ctx16, const uint8_t* __restrict__ io,
                                                           uint64_t* __restrict__ states, const uint64_t gpat) {
    __shared__ WalkTables tab;
    // a wavefront's eight banks per lane and round (64 lanes x 64 bytes) on their way out: written lane by lane, read back as the
    // 4 KB they are in HBM, so that every store instruction covers 1 KB of contiguous memory (whole sectors)
    __shared__ __attribute__((aligned(16))) uint4 outbuf[kWalkThreads / 64][4 * 64];
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&c_walk);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&tab);
        for (uint32_t i = threadIdx.x; i < sizeof(WalkTables) / 4; i += kWalkThreads) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t first = (blockIdx.x * (kWalkThreads / 64) + wave) * lpw, id = first + l;
    if (first >= g.n_slices) return;
    const uint32_t start = CARRY ? chunk * kSnapMaxSamples : 0u;  // this launch's chunk of every slice
    const uint32_t n_slice = (l < lpw && id < g.n_slices) ? slice_span(g, id).n : 0;
    const bool active = n_slice > start;
    const uint32_t n = active ? (CARRY ? min(n_slice - start, kSnapMaxSamples) : n_slice) : 0;
    [[maybe_unused]] const bool more_chunks = n_slice > start + n;  // this slice goes on behind this chunk: its runs' final states are wanted
    // (all lanes of a wavefront belong to one lane group: lpw divides the group width)
    const uint32_t grp = first >> g.lane_shift;
    const uint32_t lane_in_group = (first & ((1u << g.lane_shift) - 1)) + l;
    const uint32_t first_in_group = first & ((1u << g.lane_shift) - 1);
    // (idle lanes of a narrow or ragged wavefront walk along on the first lane's entries -- their addresses stay inside the
    // arrays -- and help with the stores below)
    const uint32_t lofs = (active ? lane_in_group : first_in_group) * 32u;
    const uint32_t row = 32u << g.lane_shift;  // bytes from one 32-byte piece of a lane to its next
    const uint8_t* const ebase = entries + size_t(grp) * (size_t(cap) * 4 << g.lane_shift) + size_t(start >> 3) * row;
    uint8_t* const bbase = sorted_banks + size_t(grp) * (size_t(cap) * 8 << g.lane_shift) + size_t(start >> 3) * (2 * row);
    [[maybe_unused]] const uint8_t* const iobase = io + size_t(grp) * (size_t(cap) * 8 << g.lane_shift) + size_t(start >> 3) * (2 * row);
    [[maybe_unused]] const uint8_t* const cbase = ctx16 + size_t(grp) * (size_t(cap) * 2 << g.lane_shift) + size_t(start >> 4) * row;
    [[maybe_unused]] uint64_t* const carry = states + ((size_t(grp) * kContexts) << g.lane_shift) + (active ? lane_in_group : first_in_group);
    uint32_t n_max = 0;
    for (unsigned long long m = __ballot(active); m; m &= m - 1)
        n_max = max(n_max, uint32_t(__builtin_amdgcn_readlane(int(n), __builtin_ctzll(m))));
    const uint32_t lanes_here = min(lpw, g.n_slices - first);  // lanes of this wavefront that own a slice
    // Eight entries (one 32-byte piece per lane) per round, requested TWO ROUNDS ahead.  Banks leave as whole 64-byte pieces.
    // Entries and banks beyond a slice's last sample are never looked at by anybody (the capacity is a multiple of 16: every
    // address stays inside the arrays).
    const uint32_t rounds = (n_max + 7) >> 3, cap_rounds = (CARRY ? kSnapMaxSamples : cap) >> 3;
    struct Piece {
        uint4 a, b;                     // eight entries
        uint4 i0, i1, i2, i3, cx;       // CARRY: the eight positions' init states (u64 each) and contexts (u16 each)
    };
    auto load_piece = [&](uint32_t r, Piece& p) {  // (r is wave-uniform: scalar base + the lane's offset)
        r = min(r, cap_rounds - 1);
        const uint4* q = reinterpret_cast<const uint4*>(ebase + size_t(r) * row + lofs);
        p.a = q[0];
        p.b = q[1];
        if constexpr (CARRY) {
            const uint4* qi = reinterpret_cast<const uint4*>(iobase + size_t(r) * (2 * row) + 2 * lofs);
            p.i0 = qi[0]; p.i1 = qi[1]; p.i2 = qi[2]; p.i3 = qi[3];
            p.cx = *reinterpret_cast<const uint4*>(cbase + size_t(r >> 1) * row + lofs + ((r & 1u) << 4));
        }
    };
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;  // the eight states of the context being walked
    [[maybe_unused]] uint32_t cur_ctx = 0;                                      // CARRY: ... and that context
    auto pack_lo = [&]() { return s0 | (s1 << 8) | (s2 << 16) | (s3 << 24); };
    auto pack_hi = [&]() { return s4 | (s5 << 8) | (s6 << 16) | (s7 << 24); };
    // p = position of the entry inside the chunk (wave-uniform), ilo / ihi / ctx: CARRY only
    auto step = [&](uint32_t e, uint32_t p, uint32_t ilo, uint32_t ihi, uint32_t ctx, uint32_t& lo, uint32_t& hi) {
        const uint32_t codes = tab.codes[e & 0x3FFu];
        if constexpr (!CARRY) {
            const uint32_t keep = ((e >> kEntryFirstBit) & 1u) - 1u;  // a new context starts from zeros (llcomp.hpp:385)
            s0 &= keep; s1 &= keep; s2 &= keep; s3 &= keep; s4 &= keep; s5 &= keep; s6 &= keep; s7 &= keep;
        } else {
            const bool live = p < n;  // (a lane behind its chunk's last sample walks on through whatever lies there: no effects)
            const bool fresh = live && ((e >> kEntryFirstBit) & 1u);
            // the run that ends here (a new context begins, or the chunk is over) leaves its states under its context
            if (more_chunks && p != 0 && (fresh || p == n))
                carry[size_t(cur_ctx) << g.lane_shift] = bank_tagged<false>(uint64_t(pack_lo()) | (uint64_t(pack_hi()) << 32), gpat);
            if (fresh) {  // ... and the new run starts from what the chunks before left its context in (zeros in the first chunk)
                cur_ctx = ctx;
                const uint32_t a = chunk ? ilo : 0u, b = chunk ? ihi : 0u;
                s0 = a & 0xFFu; s1 = (a >> 8) & 0xFFu; s2 = (a >> 16) & 0xFFu; s3 = a >> 24;
                s4 = b & 0xFFu; s5 = (b >> 8) & 0xFFu; s6 = (b >> 16) & 0xFFu; s7 = b >> 24;
            }
        }
        lo = pack_lo();
        hi = pack_hi();
        s0 = tab.once[s0 * kWalkOnceStride + (codes & 3u)];
        s1 = tab.once[s1 * kWalkOnceStride + ((codes >> 2) & 3u)];
        s2 = tab.once[s2 * kWalkOnceStride + ((codes >> 4) & 3u)];
        s3 = tab.once[s3 * kWalkOnceStride + ((codes >> 6) & 3u)];
        s4 = tab.unary[s4 * kWalkUnaryStride + ((codes >> 8) & 7u)];
        s5 = tab.once[s5 * kWalkOnceStride + ((codes >> 11) & 3u)];
        s6 = tab.bits[s6 * kWalkBitsStride + ((codes >> 13) & 31u)];
        s7 = tab.once[s7 * kWalkOnceStride + ((codes >> 23) & 3u)];
        s6 = tab.bits[s6 * kWalkBitsStride + ((codes >> 18) & 31u)];
    };
    Piece p0{}, p1{};
    load_piece(0, p0);
    load_piece(1, p1);
    for (uint32_t r = 0; r < rounds; ++r) {
        const Piece c = p0;
        p0 = p1;
        load_piece(r + 2, p1);
        uint4 o0, o1, o2, o3;
        const uint32_t p = r * 8;
        step(c.a.x, p + 0, c.i0.x, c.i0.y, c.cx.x & 0xFFFFu, o0.x, o0.y);
        step(c.a.y, p + 1, c.i0.z, c.i0.w, c.cx.x >> 16, o0.z, o0.w);
        step(c.a.z, p + 2, c.i1.x, c.i1.y, c.cx.y & 0xFFFFu, o1.x, o1.y);
        step(c.a.w, p + 3, c.i1.z, c.i1.w, c.cx.y >> 16, o1.z, o1.w);
        step(c.b.x, p + 4, c.i2.x, c.i2.y, c.cx.z & 0xFFFFu, o2.x, o2.y);
        step(c.b.y, p + 5, c.i2.z, c.i2.w, c.cx.z >> 16, o2.z, o2.w);
        step(c.b.z, p + 6, c.i3.x, c.i3.y, c.cx.w & 0xFFFFu, o3.x, o3.y);
        step(c.b.w, p + 7, c.i3.z, c.i3.w, c.cx.w >> 16, o3.z, o3.w);
        // lane l's piece = chunks 4l .. 4l+3 of the wavefront's 4 KB; store instruction j takes chunks 64j .. 64j+63.  (Chunk
        // c sits at [c % 4][c / 4]: the lane-wise writes and the chunk-wise reads both spread over the LDS banks.)
        uint4* const ob = outbuf[wave];
        ob[0 * 64 + l] = o0;
        ob[1 * 64 + l] = o1;
        ob[2 * 64 + l] = o2;
        ob[3 * 64 + l] = o3;
        // (the transpose goes from lane to lane inside one wavefront: the fence + wave barrier pin the order of its LDS accesses
        // for the compiler and compile to nothing)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint4* const q = reinterpret_cast<uint4*>(bbase + size_t(r) * (2 * row) + first_in_group * 64u) + l;  // the wavefront's first piece, chunk l
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t cc = j * 64 + l;  // chunk cc of the 4 KB belongs to lane cc / 4, part cc % 4
            if ((cc >> 2) < lanes_here) q[j * 64] = ob[(cc & 3u) * 64 + (cc >> 2)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (CARRY) {  // a chunk that ends exactly with the wavefront's last round never met "p == n" inside the loop
        if (more_chunks && n != 0 && n == rounds * 8)
            carry[size_t(cur_ctx) << g.lane_shift] = bank_tagged<false>(uint64_t(pack_lo()) | (uint64_t(pack_hi()) << 32), gpat);
    }
}

// ---- back into stream order --------------------------------------------------------------------------------------------------
// One workgroup per slice, 256 threads, HALF a slice's positions at a time (kUnpermSpan = 2048): a thread keeps its two pieces of
// eight entries and banks in registers, scatters the ones whose stream position falls into the current half into LDS, and the
// workgroup writes that half out in order.  (Round 4: 512 threads and all 4096 positions at once = 41 KB of LDS and eight wave
// slots on one CU per workgroup; beside the cached decoder's wavefronts such a workgroup waits long for its place.)
constexpr uint32_t kUnpermThreads = 256, kUnpermSpan = 2048;
// (`start` = first sample of the chunk this launch handles: 0 for slices of up to 4096 samples, chunk * 4096 above; positions inside
// the entries are chunk-local, a chunk's pieces of every array start at its first sample)
__global__ __launch_bounds__(kUnpermThreads) void k_snap_unperm(const Geometry g, const uint32_t cap, const uint8_t* __restrict__ entries,
                                                               const uint8_t* __restrict__ sorted_banks, uint8_t* __restrict__ banks,
                                                               uint8_t* __restrict__ residuals, const uint32_t start) {
    __shared__ __attribute__((aligned(16))) uint2 bank_of[kUnpermSpan];
    __shared__ __attribute__((aligned(16))) int16_t res_of[kUnpermSpan];
    uint32_t id, group, lane;
    if (!block_slice(g, id, group, lane)) return;
    const uint32_t n_slice = slice_span(g, id).n;
    if (start >= n_slice) return;
    const uint32_t n = min(n_slice - start, kSnapMaxSamples);
    const uint32_t t = threadIdx.x;
    const size_t row = size_t(32) << g.lane_shift;
    const uint8_t* const ein = entries + size_t(group) * (size_t(cap) * 4 << g.lane_shift) + lane * 32u + size_t(start >> 3) * row;
    const uint8_t* const bin = sorted_banks + size_t(group) * (size_t(cap) * 8 << g.lane_shift) + lane * 64u + size_t(start >> 3) * (2 * row);
    uint8_t* const bout = banks + size_t(group) * (size_t(cap) * 8 << g.lane_shift) + lane * 64u + size_t(start >> 3) * (2 * row);
    uint8_t* const rout = residuals + size_t(group) * (size_t(cap) * 2 << g.lane_shift) + lane * 32u + size_t(start >> 4) * row;
    // this thread's pieces: q = t and q = t + 256 (a piece = eight entries + their eight banks; twelve loads in flight)
    uint32_t e[2][8], bx[2][8], by[2][8];
#pragma unroll
    for (uint32_t h = 0; h < 2; ++h) {
        const uint32_t q = t + kUnpermThreads * h;
        if (q * 8 < n) {
            const uint4* ep = reinterpret_cast<const uint4*>(ein + q * row);
            const uint4* bp = reinterpret_cast<const uint4*>(bin + q * (2 * row));
            const uint4 ea = ep[0], eb = ep[1], b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
            e[h][0] = ea.x; e[h][1] = ea.y; e[h][2] = ea.z; e[h][3] = ea.w; e[h][4] = eb.x; e[h][5] = eb.y; e[h][6] = eb.z; e[h][7] = eb.w;
            bx[h][0] = b0.x; bx[h][1] = b0.z; bx[h][2] = b1.x; bx[h][3] = b1.z; bx[h][4] = b2.x; bx[h][5] = b2.z; bx[h][6] = b3.x; bx[h][7] = b3.z;
            by[h][0] = b0.y; by[h][1] = b0.w; by[h][2] = b1.y; by[h][3] = b1.w; by[h][4] = b2.y; by[h][5] = b2.w; by[h][6] = b3.y; by[h][7] = b3.w;
        }
    }
    for (uint32_t base = 0; base < n; base += kUnpermSpan) {  // (uniform per workgroup)
#pragma unroll
        for (uint32_t h = 0; h < 2; ++h) {
            const uint32_t q = t + kUnpermThreads * h;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                if (q * 8 + j < n) {
                    const uint32_t k = ((e[h][j] >> kEntryPosShift) & 0xFFFu) - base;  // (wraps for positions below the half)
                    if (k < kUnpermSpan) {
                        bank_of[k] = make_uint2(bx[h][j], by[h][j]);
                        res_of[k] = int16_t(int32_t(e[h][j] << 22) >> 22);
                    }
                }
            }
        }
        __syncthreads();
        // out: 16-byte chunks, neighbouring threads the neighbouring chunks of a piece (whole sectors per store instruction, see
        // k_snap_sort); chunk c of the half = banks 2c, 2c+1 / residuals 8c .. 8c+7
        const uint32_t left = n - base;  // samples from `base` on (more than the span: this half is full)
#pragma unroll
        for (uint32_t u = 0; u < kUnpermSpan / 2 / kUnpermThreads; ++u) {
            const uint32_t c = t + kUnpermThreads * u;
            if (c * 2 < left) {
                const uint32_t cg = c + base / 2;  // chunk index in the slice
                *reinterpret_cast<uint4*>(bout + size_t(cg >> 2) * (2 * row) + ((cg & 3u) << 4)) = *reinterpret_cast<const uint4*>(&bank_of[c * 2]);
            }
        }
        if (t * 8 < left && t < kUnpermSpan / 8) {
            const uint32_t qg = t + base / 8;
            *reinterpret_cast<uint4*>(rout + size_t(qg >> 1) * row + ((qg & 1u) << 4)) = *reinterpret_cast<const uint4*>(&res_of[t * 8]);
        }
        __syncthreads();
    }
}

}  // namespace

bool snapshot_mode(const Geometry& g) { return (g.flags & kGeoSnapshot) != 0; }
bool snapshot_chunked(const Geometry& g) { return snapshot_mode(g) && snapshot_chunks(g) > 1; }
uint64_t snapshot_elems(const Geometry& g) { return (uint64_t(lane_groups(g)) * snapshot_cap(g)) << g.lane_shift; }

hipError_t launch_snapshot(const Geometry& g, const uint32_t* d_sym, void* d_entries, void* d_sorted, void* d_banks, void* d_residuals,
                           hipStream_t stream) {
    if (snapshot_chunked(g)) return hipErrorInvalidValue;  // (chunk after chunk: launch_snapshot_chunk)
    const uint32_t cap = snapshot_cap(g);
    const uint32_t groups = lane_groups(g);
    const uint32_t blocks = (((groups + 7u) >> 3) << 3) << g.lane_shift;  // whole rounds of eight lane groups (one per XCD)
    const uint32_t waves = (g.n_slices + g.lpw - 1) / g.lpw;
    const dim3 walk_grid((waves + kWalkThreads / 64 - 1) / (kWalkThreads / 64));
    uint8_t* const entries = static_cast<uint8_t*>(d_entries);
    if (cap <= 4 * kSortThreads) k_snap_sort<4><<<dim3(blocks), dim3(kSortThreads), 0, stream>>>(g, cap, d_sym, entries, 0, nullptr, nullptr, nullptr, 0);
    else if (cap <= 8 * kSortThreads) k_snap_sort<8><<<dim3(blocks), dim3(kSortThreads), 0, stream>>>(g, cap, d_sym, entries, 0, nullptr, nullptr, nullptr, 0);
    else k_snap_sort<16><<<dim3(blocks), dim3(kSortThreads), 0, stream>>>(g, cap, d_sym, entries, 0, nullptr, nullptr, nullptr, 0);
    k_snap_walk<false><<<walk_grid, dim3(kWalkThreads), 0, stream>>>(g, g.lpw, cap, entries, static_cast<uint8_t*>(d_sorted), 0, nullptr, nullptr, nullptr, 0);
    k_snap_unperm<<<dim3(blocks), dim3(kUnpermThreads), 0, stream>>>(g, cap, entries, static_cast<const uint8_t*>(d_sorted),
                                                                     static_cast<uint8_t*>(d_banks), static_cast<uint8_t*>(d_residuals), 0);
    return hipGetLastError();
}

// Slices above 4096 samples: chunk c of every slice.  The chunks go in stream order (chunk c + 1 sorts -- and fetches its runs'
// starting states -- behind chunk c's walk, which left them in the table); the coder of chunk c needs nothing of chunk c + 1.
hipError_t launch_snapshot_chunk(const Geometry& g, uint32_t c, const uint32_t* d_sym, void* d_entries, void* d_sorted, void* d_banks,
                                 void* d_residuals, void* d_ctx16, void* d_io, uint64_t* d_states, uint64_t gpat, hipStream_t stream) {
    if (!snapshot_chunked(g) || c >= snapshot_chunks(g) || !d_ctx16 || !d_io || !d_states) return hipErrorInvalidValue;
    const uint32_t cap = snapshot_cap(g);
    const uint32_t groups = lane_groups(g);
    const uint32_t blocks = (((groups + 7u) >> 3) << 3) << g.lane_shift;
    const uint32_t waves = (g.n_slices + g.lpw - 1) / g.lpw;
    const dim3 walk_grid((waves + kWalkThreads / 64 - 1) / (kWalkThreads / 64));
    uint8_t* const entries = static_cast<uint8_t*>(d_entries);
    k_snap_sort<16, true><<<dim3(blocks), dim3(kSortThreads), 0, stream>>>(g, cap, d_sym, entries, c, static_cast<uint8_t*>(d_ctx16),
                                                                           static_cast<uint8_t*>(d_io), d_states, gpat);
    k_snap_walk<true><<<walk_grid, dim3(kWalkThreads), 0, stream>>>(g, g.lpw, cap, entries, static_cast<uint8_t*>(d_sorted), c,
                                                                    static_cast<const uint8_t*>(d_ctx16), static_cast<const uint8_t*>(d_io), d_states, gpat);
    k_snap_unperm<<<dim3(blocks), dim3(kUnpermThreads), 0, stream>>>(g, cap, entries, static_cast<const uint8_t*>(d_sorted),
                                                                     static_cast<uint8_t*>(d_banks), static_cast<uint8_t*>(d_residuals), c * kSnapMaxSamples);
    return hipGetLastError();
}

}  // namespace llcomp_mi
