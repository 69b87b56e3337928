// codec.hip -- host side of libllcomp_mi.so: the device-resident batch codec, the host-buffer drop-in calls
// (mirrors of llcomp::compressImage / decompressImage, /root/reference/llcomp.hpp:358, 461) and the C ABI.
// Every byte of coded data is produced by the kernels in kernels.hip; there is no CPU coding path here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/llcomp_mi.h"
#include "container.hpp"
#include "geometry.hpp"
#include "kernels.hpp"
#include "tables.hpp"

using namespace llcomp_mi;

struct llcomp_mi_codec {
    Geometry g{};
    int device = 0;
    // workspace (all on `device`)
    void* d_sym_or_rec = nullptr;   // image order: encode u32 symbols per sample / decode int16 reconstructed samples
    void* d_lane_order = nullptr;   // the same data in lane order [group][k][64] for the serial kernels
    uint64_t* d_states = nullptr;   // u64[lane group][kContexts][lanes of the group]
    uint8_t* d_scratch = nullptr;   // slice streams in stream lane order: 16-byte units [group][unit][lane]
    uint64_t* d_offsets = nullptr;  // u64[n_slices + 1]
    uint64_t* d_total_tmp = nullptr;
    uint64_t* d_block_sums = nullptr;  // scan scratch
    uint64_t workspace_bytes = 0;
    // staging for the host-buffer calls (llcomp_mi_encode / llcomp_mi_decode), kept with the cached object so that a call
    // does not pay for five hipMalloc / hipFree pairs
    uint8_t* io_px = nullptr;
    uint8_t* io_payload = nullptr;
    uint32_t* io_len = nullptr;
    uint64_t* io_total = nullptr;
    uint32_t* io_status = nullptr;
    uint64_t io_payload_cap = 0;
    uint64_t io_bytes = 0;  // all of the above, for the idle-cache budget
    bool need_states = true;  // false when the states live in LDS (1-row slices; one slice per wavefront)
    // optional per-kernel timing (hipEvents on the caller's stream)
    bool profiling = false;
    struct Span { hipEvent_t a, b; int slot; };
    std::vector<Span> spans;
    uint32_t n_encode = 0, n_decode = 0;
};

namespace {

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return LLCOMP_MI_HIP_ERROR; \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) return;
        ok = (dev == prev) || hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (ok && prev >= 0) (void)hipSetDevice(prev);
    }
};

// brackets a group of launches with two events when profiling is on
struct Timed {
    llcomp_mi_codec* k;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    int slot;
    Timed(llcomp_mi_codec* k_, hipStream_t s_, int slot_) : k(k_), s(s_), slot(slot_) {
        if (!k->profiling) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, s);
    }
    ~Timed() {
        if (!a) return;
        (void)hipEventRecord(b, s);
        k->spans.push_back({a, b, slot});
    }
};

int status_from_bits(uint32_t bits) {
    if (bits & kStBadExponent) return LLCOMP_MI_BAD_EXPONENT;
    if (bits & kStTruncated) return LLCOMP_MI_TRUNCATED;
    if (bits & kStOverflow) return LLCOMP_MI_OUTPUT_OVERFLOW;
    return LLCOMP_MI_OK;
}

int resolve_device(int32_t device, int* out) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return LLCOMP_MI_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(out) != hipSuccess) return LLCOMP_MI_NO_DEVICE;
        return LLCOMP_MI_OK;
    }
    if (device >= n) return LLCOMP_MI_BAD_ARGS;
    *out = device;
    return LLCOMP_MI_OK;
}

}  // namespace

extern "C" {

int llcomp_mi_abi_version(void) { return LLCOMP_MI_ABI_VERSION; }

int llcomp_mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

const char* llcomp_mi_strerror(int status) {
    switch (status) {
        case LLCOMP_MI_OK: return "ok";
        case LLCOMP_MI_BAD_MAGIC: return "Invalid magic number";  // llcomp.hpp:466, verbatim
        case LLCOMP_MI_BAD_EXPONENT: return "Invalid exponent";   // llcomp.hpp:233, verbatim
        case LLCOMP_MI_TRUNCATED: return "stream shorter than its header or slice table";
        case LLCOMP_MI_BAD_ARGS: return "bad arguments";
        case LLCOMP_MI_OUT_OF_RANGE: return "image dimensions out of range for this format";
        case LLCOMP_MI_OUTPUT_OVERFLOW: return "output capacity too small";
        case LLCOMP_MI_HIP_ERROR: return "HIP runtime error";
        case LLCOMP_MI_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case LLCOMP_MI_NOMEM: return "out of memory";
        default: return "unknown status";
    }
}

void llcomp_mi_free(void* p) { std::free(p); }

// ---- device-resident codec ------------------------------------------------------------------------------------
int llcomp_mi_codec_create(llcomp_mi_codec** out, int32_t device, uint32_t frames, uint32_t w, uint32_t h, uint32_t c,
                           uint32_t tile_w, uint32_t tile_h, uint32_t planar) {
    if (!out) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    Geometry g;
    if (!make_geometry(g, frames, w, h, c, tile_w, tile_h, planar)) return LLCOMP_MI_BAD_ARGS;
    int dev = 0;
    if (int rc = resolve_device(device, &dev)) return rc;
    DeviceGuard guard(dev);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    llcomp_mi_codec* k = new (std::nothrow) llcomp_mi_codec;
    if (!k) return LLCOMP_MI_NOMEM;
    k->g = g;
    k->device = dev;
    const uint64_t samples = uint64_t(frames) * w * h * c;
    k->need_states = slices_need_state_tables(g);
    const uint64_t b_sym = samples * 4, b_states = k->need_states ? (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8 : 8,
                   b_scratch = (uint64_t(lane_groups(g)) << g.lane_shift) * g.slice_cap, b_off = (uint64_t(g.n_slices) + 1) * 8;
    const uint64_t b_lanes = (uint64_t(lane_groups(g)) * slice_capacity_samples(g) << g.lane_shift) * 4;
    k->workspace_bytes = b_sym + b_lanes + b_states + b_scratch + b_off + 8;
    bool ok = hipMalloc(&k->d_sym_or_rec, b_sym) == hipSuccess && hipMalloc(&k->d_lane_order, b_lanes) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&k->d_states), b_states) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&k->d_scratch), b_scratch) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&k->d_offsets), b_off) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&k->d_total_tmp), 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&k->d_block_sums), 8ull * (scan_block_count(g.n_slices) + 1)) == hipSuccess;
    if (!ok) {
        llcomp_mi_codec_destroy(k);
        return LLCOMP_MI_NOMEM;
    }
    *out = k;
    return LLCOMP_MI_OK;
}

void llcomp_mi_codec_destroy(llcomp_mi_codec* k) {
    if (!k) return;
    DeviceGuard guard(k->device);
    (void)hipFree(k->d_sym_or_rec);
    (void)hipFree(k->d_lane_order);
    (void)hipFree(k->d_states);
    (void)hipFree(k->d_scratch);
    (void)hipFree(k->d_offsets);
    (void)hipFree(k->d_total_tmp);
    (void)hipFree(k->d_block_sums);
    (void)hipFree(k->io_px);
    (void)hipFree(k->io_payload);
    (void)hipFree(k->io_len);
    (void)hipFree(k->io_total);
    (void)hipFree(k->io_status);
    for (auto& sp : k->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    delete k;
}

uint32_t llcomp_mi_codec_slices(const llcomp_mi_codec* k) { return k ? k->g.n_slices : 0; }
uint64_t llcomp_mi_codec_workspace_bytes(const llcomp_mi_codec* k) { return k ? k->workspace_bytes : 0; }
uint64_t llcomp_mi_codec_max_payload_bytes(const llcomp_mi_codec* k) {
    return k ? uint64_t(k->g.n_slices) * k->g.slice_cap : 0;
}

int llcomp_mi_codec_model(llcomp_mi_codec* k, const void* d_px, void* d_sym, void* stream) {
    if (!k || !d_px || !d_sym) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    HIP_TRY(launch_model_fwd(k->g, static_cast<const uint8_t*>(d_px), static_cast<uint32_t*>(d_sym),
                             static_cast<hipStream_t>(stream)));
    return LLCOMP_MI_OK;
}

int llcomp_mi_codec_encode(llcomp_mi_codec* k, const void* d_px, void* d_payload, uint64_t payload_cap, void* d_slice_len,
                           void* d_total, void* d_status, void* stream) {
    if (!k || !d_px || !d_payload || !d_slice_len || !d_total || !d_status) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geometry& g = k->g;
    if (slices_need_state_tables(g) && !k->need_states) return LLCOMP_MI_BAD_ARGS;  // kernel family changed under us
    HIP_TRY(hipMemsetAsync(d_status, 0, 4, s));
    {
        Timed t(k, s, 0);
        if (k->need_states) HIP_TRY(hipMemsetAsync(k->d_states, 0, (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8, s));
    }
    {
        Timed t(k, s, 1);
        if (model_is_fused(g)) {
            HIP_TRY(launch_model_rows_fwd(g, static_cast<const uint8_t*>(d_px), static_cast<uint16_t*>(k->d_lane_order), s));
        } else {
            HIP_TRY(launch_model_fwd(g, static_cast<const uint8_t*>(d_px), static_cast<uint32_t*>(k->d_sym_or_rec), s));
            HIP_TRY(launch_to_lane_order_u32(g, static_cast<const uint32_t*>(k->d_sym_or_rec),
                                             static_cast<uint32_t*>(k->d_lane_order), s));
        }
    }
    {
        Timed t(k, s, 2);
        HIP_TRY(launch_encode_slices(g, k->d_lane_order, k->d_states, k->d_scratch,
                                     static_cast<uint32_t*>(d_slice_len), static_cast<uint32_t*>(d_status), s));
    }
    {
        Timed t(k, s, 3);
        HIP_TRY(launch_scan_lengths(static_cast<const uint32_t*>(d_slice_len), g.n_slices, k->d_offsets,
                                    static_cast<uint64_t*>(d_total), k->d_block_sums, s));
        HIP_TRY(launch_pack_payload(g, k->d_scratch, static_cast<const uint32_t*>(d_slice_len), k->d_offsets,
                                    static_cast<uint8_t*>(d_payload), payload_cap, static_cast<uint32_t*>(d_status), s));
    }
    ++k->n_encode;
    return LLCOMP_MI_OK;
}

int llcomp_mi_codec_decode(llcomp_mi_codec* k, const void* d_payload, uint64_t payload_bytes, const void* d_slice_len,
                           void* d_px, void* d_status, void* stream) {
    if (!k || !d_payload || !d_slice_len || !d_px || !d_status) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Geometry& g = k->g;
    if (slices_need_state_tables(g) && !k->need_states) return LLCOMP_MI_BAD_ARGS;  // kernel family changed under us
    HIP_TRY(hipMemsetAsync(d_status, 0, 4, s));
    {
        Timed t(k, s, 7);
        if (k->need_states) HIP_TRY(hipMemsetAsync(k->d_states, 0, (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8, s));
    }
    {
        Timed t(k, s, 4);
        HIP_TRY(launch_scan_lengths(static_cast<const uint32_t*>(d_slice_len), g.n_slices, k->d_offsets, k->d_total_tmp, k->d_block_sums, s));
    }
    {
        Timed t(k, s, 4);
        HIP_TRY(launch_stage_streams(g, static_cast<const uint8_t*>(d_payload), payload_bytes,
                                     static_cast<const uint32_t*>(d_slice_len), k->d_offsets, k->d_scratch,
                                     static_cast<uint32_t*>(d_status), s));
    }
    {
        Timed t(k, s, 5);
        HIP_TRY(launch_decode_slices(g, k->d_scratch, static_cast<const uint32_t*>(d_slice_len), k->d_states,
                                     static_cast<int16_t*>(k->d_lane_order), static_cast<uint32_t*>(d_status), s));
    }
    {
        Timed t(k, s, 6);
        if (model_is_fused(g)) {
            HIP_TRY(launch_model_rows_inv(g, static_cast<const int16_t*>(k->d_lane_order), static_cast<uint8_t*>(d_px), s));
        } else {
            HIP_TRY(launch_from_lane_order_i16(g, static_cast<const int16_t*>(k->d_lane_order),
                                               static_cast<int16_t*>(k->d_sym_or_rec), s));
            HIP_TRY(launch_model_inv(g, static_cast<const int16_t*>(k->d_sym_or_rec), static_cast<uint8_t*>(d_px), s));
        }
    }
    ++k->n_decode;
    return LLCOMP_MI_OK;
}

uint32_t llcomp_mi_status_from_bits(uint32_t bits) { return uint32_t(status_from_bits(bits)); }

int llcomp_mi_codec_set_profiling(llcomp_mi_codec* k, int enable) {
    if (!k) return LLCOMP_MI_BAD_ARGS;
    k->profiling = enable != 0;
    return LLCOMP_MI_OK;
}

int llcomp_mi_codec_get_profile(llcomp_mi_codec* k, double* ms8, uint32_t* n_encode, uint32_t* n_decode) {
    if (!k || !ms8) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    for (int i = 0; i < 8; ++i) ms8[i] = 0.0;
    int rc = LLCOMP_MI_OK;
    for (auto& sp : k->spans) {
        float ms = 0.f;
        if (hipEventSynchronize(sp.b) != hipSuccess || hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) rc = LLCOMP_MI_HIP_ERROR;
        ms8[sp.slot] += ms;
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    k->spans.clear();
    if (n_encode) *n_encode = k->n_encode;
    if (n_decode) *n_decode = k->n_decode;
    k->n_encode = k->n_decode = 0;
    return rc;
}

}  // extern "C"

namespace {
// The host-buffer calls need a codec object (GBs of workspace for a 4K frame) per call; allocating it every time costs
// more than the coding.  A few idle ones are kept, keyed by device + geometry.  Never torn down at exit on purpose (the
// HIP runtime may already be gone by then).
struct CodecCache {
    struct Item { llcomp_mi_codec* k; uint64_t stamp; };
    std::mutex mu;
    std::vector<Item> idle;
    uint64_t clock = 0;
    static constexpr size_t kMaxIdle = 4;
    static constexpr uint64_t kMaxIdleBytes = 12ull << 30;

    llcomp_mi_codec* take(int dev, const Geometry& g) {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].k->device == dev && std::memcmp(&idle[i].k->g, &g, sizeof(Geometry)) == 0 &&
                idle[i].k->need_states == slices_need_state_tables(g)) {
                llcomp_mi_codec* k = idle[i].k;
                idle.erase(idle.begin() + long(i));
                return k;
            }
        return nullptr;
    }
    void give(llcomp_mi_codec* k) {
        std::vector<llcomp_mi_codec*> drop;
        {
            std::lock_guard<std::mutex> lock(mu);
            idle.push_back({k, ++clock});
            auto bytes = [&]() { uint64_t b = 0; for (auto& it : idle) b += it.k->workspace_bytes + it.k->io_bytes; return b; };
            while (idle.size() > kMaxIdle || (idle.size() > 1 && bytes() > kMaxIdleBytes)) {
                size_t oldest = 0;
                for (size_t i = 1; i < idle.size(); ++i) if (idle[i].stamp < idle[oldest].stamp) oldest = i;
                drop.push_back(idle[oldest].k);
                idle.erase(idle.begin() + long(oldest));
            }
        }
        for (auto* d : drop) llcomp_mi_codec_destroy(d);
    }
};
CodecCache& codec_cache() {
    static CodecCache* c = new CodecCache;  // leaked deliberately
    return *c;
}
int acquire_codec(llcomp_mi_codec** out, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h,
                  uint32_t planar) {
    Geometry g;
    if (!make_geometry(g, 1, w, h, c, tile_w, tile_h, planar)) return LLCOMP_MI_BAD_ARGS;
    int dev = 0;
    if (int rc = resolve_device(device, &dev)) return rc;
    if ((*out = codec_cache().take(dev, g))) return LLCOMP_MI_OK;
    return llcomp_mi_codec_create(out, dev, 1, w, h, c, tile_w, tile_h, planar);
}
}  // namespace

extern "C" {

// ---- host-buffer API --------------------------------------------------------------------------------------------
// staging buffers of the host-buffer calls: allocated on first use, the payload buffer grown on demand
static bool ensure_io(llcomp_mi_codec* k, uint64_t payload_cap) {
    const Geometry& g = k->g;
    const uint64_t raw = uint64_t(g.w) * g.h * g.c * g.frames;
    if (!k->io_px && hipMalloc(reinterpret_cast<void**>(&k->io_px), raw) != hipSuccess) return false;
    if (!k->io_len && hipMalloc(reinterpret_cast<void**>(&k->io_len), uint64_t(g.n_slices) * 4) != hipSuccess) return false;
    if (!k->io_total && hipMalloc(reinterpret_cast<void**>(&k->io_total), 8) != hipSuccess) return false;
    if (!k->io_status && hipMalloc(reinterpret_cast<void**>(&k->io_status), 4) != hipSuccess) return false;
    if (k->io_payload_cap < payload_cap) {
        (void)hipFree(k->io_payload);
        k->io_payload = nullptr;
        k->io_payload_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&k->io_payload), payload_cap) != hipSuccess) return false;
        k->io_payload_cap = payload_cap;
    }
    k->io_bytes = raw + uint64_t(g.n_slices) * 4 + 12 + k->io_payload_cap;
    return true;
}

int llcomp_mi_encode(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts, uint8_t** out,
                     size_t* out_len) {
    if (!px || !out || !out_len) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    *out_len = 0;
    llcomp_mi_opts o{};
    o.struct_size = sizeof(o);
    o.format = LLCOMP_MI_FORMAT_LEGACY;
    o.device = -1;
    if (opts) {
        if (opts->struct_size != sizeof(llcomp_mi_opts)) return LLCOMP_MI_BAD_ARGS;
        o = *opts;
    }
    if (o.format != LLCOMP_MI_FORMAT_LEGACY && o.format != LLCOMP_MI_FORMAT_SLICED) return LLCOMP_MI_BAD_ARGS;
    if (!w || !h || c < 1 || c > 4) return LLCOMP_MI_BAD_ARGS;
    if (uint64_t(w) * h * c >= (1ull << 31)) return LLCOMP_MI_OUT_OF_RANGE;
    const bool legacy = o.format == LLCOMP_MI_FORMAT_LEGACY;
    if (legacy && (w > 65535 || h > 65535)) return LLCOMP_MI_OUT_OF_RANGE;  // u16 header fields, llcomp.hpp:377-378
    const uint32_t tile_w = legacy ? w : (o.tile_w == 0 || o.tile_w > w ? w : o.tile_w);
    const uint32_t tile_h = legacy ? h : (o.tile_h == 0 || o.tile_h > h ? h : o.tile_h);
    const uint32_t planar = legacy ? 0 : (o.planar ? 1 : 0);

    llcomp_mi_codec* k = nullptr;
    if (int rc = acquire_codec(&k, o.device, w, h, c, tile_w, tile_h, planar)) return rc;
    DeviceGuard guard(k->device);
    const Geometry& g = k->g;
    const uint64_t raw = uint64_t(w) * h * c;
    const uint64_t max_payload = llcomp_mi_codec_max_payload_bytes(k);
    uint8_t* host = nullptr;
    int rc = LLCOMP_MI_OK;
    auto fail = [&](int code) { codec_cache().give(k); std::free(host); return code; };
    // first try with room for 2x raw (incompressible noise needs ~1.25x), then the proven worst case
    uint64_t cap = std::min<uint64_t>(max_payload, 2 * raw + 64ull * g.n_slices + 4096);
    if (!ensure_io(k, cap)) return fail(LLCOMP_MI_NOMEM);
    uint8_t* const d_px = k->io_px;
    uint32_t* const d_len = k->io_len;
    uint64_t* const d_total = k->io_total;
    uint32_t* const d_status = k->io_status;
    if (hipMemcpy(d_px, px, raw, hipMemcpyHostToDevice) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    uint64_t total = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        rc = llcomp_mi_codec_encode(k, d_px, k->io_payload, cap, d_len, d_total, d_status, nullptr);
        if (rc) return fail(rc);
        uint32_t bits = 0;
        if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
        if (hipMemcpy(&bits, d_status, 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(LLCOMP_MI_HIP_ERROR);
        rc = status_from_bits(bits);
        if (rc == LLCOMP_MI_OUTPUT_OVERFLOW && cap < max_payload) {
            cap = max_payload;
            if (!ensure_io(k, cap)) return fail(LLCOMP_MI_NOMEM);
            continue;
        }
        break;
    }
    uint8_t* const d_payload = k->io_payload;
    if (rc) return fail(rc);
    const size_t head = legacy ? 6 : size_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4 * size_t(g.n_slices);
    host = static_cast<uint8_t*>(std::malloc(head + total + 1));
    if (!host) return fail(LLCOMP_MI_NOMEM);
    if (legacy) {
        write_legacy_header(host, w, h, c);
    } else {
        write_sliced_header(host, g);
        if (hipMemcpy(host + LLCOMP_MI_SLICED_HEADER_BYTES, d_len, 4 * size_t(g.n_slices), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(LLCOMP_MI_HIP_ERROR);  // slice table is little-endian u32 on both sides
    }
    if (total && hipMemcpy(host + head, d_payload, total, hipMemcpyDeviceToHost) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    codec_cache().give(k);
    *out = host;
    *out_len = head + total;
    return LLCOMP_MI_OK;
}

int llcomp_mi_decode(const uint8_t* data, size_t len, int32_t device, uint8_t** px, uint32_t* w, uint32_t* h, uint32_t* c) {
    if (!data || !px || !w || !h || !c) return LLCOMP_MI_BAD_ARGS;
    *px = nullptr;
    llcomp_mi_info info;
    if (int rc = llcomp_mi_probe(data, len, &info)) return rc;
    if (!info.width || !info.height || info.channels < 1 || info.channels > 4) return LLCOMP_MI_BAD_ARGS;
    llcomp_mi_codec* k = nullptr;
    if (int rc = acquire_codec(&k, device, info.width, info.height, info.channels, info.tile_w, info.tile_h, info.planar)) return rc;
    DeviceGuard guard(k->device);
    const Geometry& g = k->g;
    const uint64_t raw = uint64_t(info.width) * info.height * info.channels;
    const uint64_t payload_bytes = len - info.payload_offset;
    uint8_t* host = nullptr;
    auto fail = [&](int code) { codec_cache().give(k); std::free(host); return code; };
    if (!ensure_io(k, payload_bytes + 16)) return fail(LLCOMP_MI_NOMEM);
    uint8_t* const d_px = k->io_px;
    uint8_t* const d_payload = k->io_payload;
    uint32_t* const d_len = k->io_len;
    uint32_t* const d_status = k->io_status;
    if (payload_bytes && hipMemcpy(d_payload, data + info.payload_offset, payload_bytes, hipMemcpyHostToDevice) != hipSuccess)
        return fail(LLCOMP_MI_HIP_ERROR);
    if (info.format == LLCOMP_MI_FORMAT_LEGACY) {
        const uint32_t one = uint32_t(std::min<uint64_t>(payload_bytes, 0xFFFFFFFFull));
        if (hipMemcpy(d_len, &one, 4, hipMemcpyHostToDevice) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    } else {
        if (hipMemcpy(d_len, data + info.table_offset, 4 * size_t(g.n_slices), hipMemcpyHostToDevice) != hipSuccess)
            return fail(LLCOMP_MI_HIP_ERROR);
    }
    if (int rc = llcomp_mi_codec_decode(k, d_payload, payload_bytes, d_len, d_px, d_status, nullptr)) return fail(rc);
    if (hipStreamSynchronize(nullptr) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    uint32_t bits = 0;
    if (hipMemcpy(&bits, d_status, 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    if (int rc = status_from_bits(bits)) return fail(rc);
    host = static_cast<uint8_t*>(std::malloc(raw ? raw : 1));
    if (!host) return fail(LLCOMP_MI_NOMEM);
    if (hipMemcpy(host, d_px, raw, hipMemcpyDeviceToHost) != hipSuccess) return fail(LLCOMP_MI_HIP_ERROR);
    codec_cache().give(k);
    *px = host;
    *w = info.width;
    *h = info.height;
    *c = info.channels;
    return LLCOMP_MI_OK;
}

}  // extern "C"
