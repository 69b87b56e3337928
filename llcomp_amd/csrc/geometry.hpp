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

constexpr uint32_t kMaxChannels = 255;  // one byte in both headers; 1..4 run the specialised kernels, more the generic ones

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
    uint32_t lpw;               // slices per wavefront = active lanes (a power of two <= the group width)
    uint32_t flags;             // kernel family, fixed when the codec object is created (kGeo* below)
};
enum : uint32_t {
    kGeoRows = 1u,         // 1-row slices: the three reachable contexts' states live in LDS
    kGeoLdsTable = 2u,     // one slice per wavefront: its 63 KB state table lives in LDS
    kGeoForceReplay = 4u,  // test hook: every decoded sample also goes through rollback + checked replay
    kGeoSmallModel = 8u,   // bitstream variant: the reference built with LargeModel = false (llcomp.hpp:21, 26-32, 427-429)
    kGeoSnapshot = 16u,    // 2-D slices of at most kSnapMaxSamples * kSnapMaxChunks samples: the ENCODER streams state snapshots
                           // (snapshot_kernels.hip) instead of read-modify-writing a 63 KB table per slice in HBM per SAMPLE; the decoder
                           // still needs that table, and so does the pass itself between the chunks of a slice above 4096 samples
    kGeoBankCache = 32u,   // 2-D decoder with tables in HBM: per-lane write-back cache of 32 state banks in LDS (slice_kernels.hip)
};
constexpr uint32_t kSnapMaxSamples = 4096;  // a slice's samples are sorted by context inside one workgroup's LDS, 4096 at a time ...
constexpr uint32_t kSnapMaxChunks = 4;      // ... and a bigger slice (up to 16384 samples: 64x64 interleaved RGB, 128x128 planes) goes through the
                                            // pass in chunks of 4096 consecutive samples whose contexts' states are carried from chunk to chunk
                                            // through the slice's table in HBM (snapshot_kernels.hip).  Beyond that the table encoder stays:
                                            // 256x256 planes (16 chunks, a few thousand slices per launch) are one wavefront's dependent chain
                                            // whatever feeds it, and 48 more launches in front of it cost 9 % (profiles/r06_chunked_snapshot_ab.txt)

// Test / tuning hooks.  They are read from the environment ONCE per process (codec.hip: current_tuning; a test that
// changes them calls llcomp_mi_reload_tuning), they select the kernel family when a codec object is created, and none
// of them changes a single output byte.
struct Tuning {
    int lane_shift = -1;       // LLCOMP_MI_LANE_SHIFT: force the lane-group width (0..6)
    int lpw = 0;               // LLCOMP_MI_LPW: fewer active lanes per wavefront
    bool norows = false;       // LLCOMP_MI_NOROWS=1: 1-row slices through the general table-per-slice kernels
    bool noldstab = false;     // LLCOMP_MI_NOLDSTAB=1: single-slice launches keep their table in HBM
    bool force_replay = false; // LLCOMP_MI_FORCE_REPLAY=1
    bool nosnap = false;       // LLCOMP_MI_NOSNAP=1: the 2-D encoder keeps its state tables in HBM (the path before round 4)
    bool nocache = false;      // LLCOMP_MI_NOCACHE=1: the 2-D decoder fetches and writes every state bank in HBM (the path before round 5)
    int overlap = 2;           // LLCOMP_MI_OVERLAP: slices above 4096 samples -- 2 (default): the snapshot pass of chunk c + 1 runs beside the coding of
                               // chunk c on ONE second stream per device, shared by all codec objects; 1: on a second stream of the codec's own
                               // (fine for one or two pipelines, 25 % below in-order with three: too many queues); 0: in order on the caller's
                               // stream (profiles/r06_chunked_snapshot_ab.txt)
    bool nofeedback = false;   // LLCOMP_MI_NOFEEDBACK=1: the bank cache stays on in every launch, whatever the last one's wavefronts did with it (A/B)
};
inline Tuning tuning_from_env() {
    Tuning t;
    auto num = [](const char* name, long lo, long hi, int dflt) {
        const char* e = std::getenv(name);
        if (!e || !*e) return dflt;
        const long v = std::strtol(e, nullptr, 10);
        return (v >= lo && v <= hi) ? int(v) : dflt;
    };
    auto flag = [](const char* name) { const char* e = std::getenv(name); return e && e[0] == '1'; };
    t.lane_shift = num("LLCOMP_MI_LANE_SHIFT", 0, 6, -1);
    t.lpw = num("LLCOMP_MI_LPW", 1, 64, 0);
    t.norows = flag("LLCOMP_MI_NOROWS");
    t.noldstab = flag("LLCOMP_MI_NOLDSTAB");
    t.force_replay = flag("LLCOMP_MI_FORCE_REPLAY");
    t.nosnap = flag("LLCOMP_MI_NOSNAP");
    t.nocache = flag("LLCOMP_MI_NOCACHE");
    t.nofeedback = flag("LLCOMP_MI_NOFEEDBACK");
    t.overlap = num("LLCOMP_MI_OVERLAP", 0, 2, 2);
    return t;
}

// snapshot pass (2-D encoder): chunks of a slice, and its sample capacity in the piece-layout arrays (a multiple of 16; whole chunks
// when there are several, so that chunk c starts at sample c * 4096 = piece boundaries of every array)
LLMI_HD inline uint32_t snapshot_chunks(const Geometry& g) { return (g.slice_samples + kSnapMaxSamples - 1) / kSnapMaxSamples; }
LLMI_HD inline uint32_t snapshot_cap(const Geometry& g) {
    return g.slice_samples <= kSnapMaxSamples ? (g.slice_samples + 15u) & ~15u : snapshot_chunks(g) * kSnapMaxSamples;
}

constexpr int kBankCacheLog2 = 5;  // entries per lane of the 2-D decoder's bank cache (slice_kernels.hip): ONE constant for flag, launcher, kernel
inline int bank_cache_log2(const Geometry& g) { return (g.flags & kGeoBankCache) ? kBankCacheLog2 : 0; }

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
inline uint32_t default_lane_shift(uint32_t n_slices) {
    constexpr uint32_t kMinWaves = 96;
    uint32_t s = 6;
    while (s > 0 && (n_slices >> s) < kMinWaves) --s;
    return s;
}

inline bool make_geometry(Geometry& g, uint32_t frames, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                          uint32_t tile_h, uint32_t planar, const Tuning& tune = Tuning{}, bool small_model = false) {
    if (!frames || !w || !h || c < 1 || c > kMaxChannels) return false;
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
    // 13 B/sample bound + slack.  A slice beyond 165 M samples (only a legacy whole-image stream can be that big) gets
    // the largest capacity the kernels' 32-bit stream positions allow; real streams stay below 1.3 B/sample and an
    // overflow would be reported, never written.
    const uint64_t cap = (uint64_t(tile_w) * tile_h * g.nch * 13 + 32 + 15) & ~15ull;
    g.slice_cap = uint32_t(cap < 0x7FFFFFF0ull ? cap : 0x7FFFFFF0ull);
    g.slice_samples = tile_w * tile_h * g.nch;
    g.lane_shift = tune.lane_shift >= 0 ? uint32_t(tune.lane_shift) : default_lane_shift(g.n_slices);
    // A few hundred BIG slices (whole-image streams in bulk, one frame in 256x256 tiles): one slice per wavefront, so that every
    // slice's 63 KB state table sits in LDS (two per CU: 512 fit the GPU at once) instead of HBM -- +13 % measured on 512 legacy
    // streams and on one 4K frame in 256x256 tiles; smaller slices have the snapshot encoder and stay several to a wavefront.
    if (tune.lane_shift < 0 && g.tile_h > 1 && g.slice_samples > kSnapMaxSamples && g.n_slices <= 512) g.lane_shift = 0;
    const uint32_t gw = 1u << g.lane_shift;
    g.lpw = gw;
    if (tune.lpw >= 1) {  // rounded down to a power of two: a wavefront never straddles lane groups
        uint32_t p = 1;
        while (p * 2 <= uint32_t(tune.lpw) && p * 2 <= gw) p *= 2;
        g.lpw = p;
    }
    g.flags = 0;
    if (g.tile_h == 1 && !tune.norows && g.nch <= 4) g.flags |= kGeoRows;  // (the register-resident row kernels exist for 1..4 channels)
    else if (g.lpw == 1 && !tune.noldstab) g.flags |= kGeoLdsTable;
    else if (g.slice_samples <= kSnapMaxSamples * kSnapMaxChunks && !tune.nosnap) g.flags |= kGeoSnapshot;
    // The decoder of the families with tables in HBM (1..4 channels per slice): bank cache in LDS.
    if (!(g.flags & (kGeoRows | kGeoLdsTable)) && g.nch <= 4 && !tune.nocache) g.flags |= kGeoBankCache;
    if (tune.force_replay) g.flags |= kGeoForceReplay;
    if (small_model) g.flags |= kGeoSmallModel;
    return true;
}

}  // namespace llcomp_mi
