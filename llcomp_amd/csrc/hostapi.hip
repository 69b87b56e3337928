// hostapi.hip -- host-buffer calls of libllcomp_mi.so: the drop-in for llcomp::compressImage / decompressImage
// (/root/reference/llcomp.hpp:358, 461; callers llcompc.cpp:33, llcompd.cpp:26).
//
// A call takes a "lane" (codec_internal.hpp: codec object for one frame + private HIP stream + the frame and the container
// in HBM in wire layout) from a small cache keyed by device + geometry, copies the input over PCIe, enqueues the kernels,
// reads back {payload bytes, status} through a pinned mailbox and then the container / the pixels in ONE copy of the
// exact size.  Nothing runs on the NULL stream and nothing synchronises the device: concurrent callers overlap.
// With buffers from llcomp_mi_host_alloc (pinned) on both sides the copies are plain DMA; with pageable buffers the HIP
// runtime stages them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/llcomp_mi.h"
#include "codec_internal.hpp"
#include "container.hpp"

using namespace llcomp_mi;

namespace llcomp_mi {

void lane_destroy(HostLane* l) {
    if (!l) return;
    {
        DeviceGuard guard(l->k ? l->k->device : 0);
        if (l->stream) (void)hipStreamSynchronize(l->stream);
        if (l->k) l->k->done.reset();  // the lane's stream is drained and about to be destroyed: no event may refer to it afterwards
        dev_free(l->d_px);
        dev_free(l->d_container);
        dev_free(l->d_len_legacy);
        dev_free(l->d_meta);
        if (l->h_meta) (void)hipHostFree(l->h_meta);
        if (l->stream) (void)hipStreamDestroy(l->stream);
    }
    codec_release(l->k);  // the lane's stream was synchronised above; nothing else runs on this codec
    delete l;
}

static int lane_write_header(HostLane* l) {
    uint8_t head[LLCOMP_MI_SLICED_HEADER_BYTES];
    const Geometry& g = l->k->g;
    if (l->legacy) write_legacy_header(head, g.w, g.h, g.c);
    else write_sliced_header(head, g);
    LLMI_HIP_TRY(hipMemcpyAsync(l->d_container, head, l->legacy ? 6 : LLCOMP_MI_SLICED_HEADER_BYTES, hipMemcpyHostToDevice, l->stream));
    LLMI_HIP_TRY(hipStreamSynchronize(l->stream));  // `head` lives on this stack frame
    return LLCOMP_MI_OK;
}

int lane_grow(HostLane* l, uint64_t payload_cap) {
    if (payload_cap <= l->payload_cap) return LLCOMP_MI_OK;
    DeviceGuard guard(l->k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    (void)hipStreamSynchronize(l->stream);
    dev_free(l->d_container);
    l->d_container = nullptr;
    l->bytes -= l->head_bytes + l->payload_cap;
    l->payload_cap = 0;
    if (dev_alloc(reinterpret_cast<void**>(&l->d_container), l->head_bytes + payload_cap + 16) != hipSuccess) return LLCOMP_MI_NOMEM;
    l->payload_cap = payload_cap;
    l->bytes += l->head_bytes + payload_cap;
    return lane_write_header(l);
}

int lane_create(HostLane** out, int dev, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar,
                bool legacy, uint64_t payload_cap, bool small_model, uint32_t frames) {
    *out = nullptr;
    HostLane* l = new (std::nothrow) HostLane;
    if (!l) return LLCOMP_MI_NOMEM;
    l->legacy = legacy;
    l->frames = frames;
    if (int rc = llcomp_mi_codec_create_ex(&l->k, dev, frames, w, h, c, tile_w, tile_h, planar, small_model ? LLCOMP_MI_FLAG_SMALL_MODEL : 0)) {
        delete l;
        return rc;
    }
    DeviceGuard guard(dev);
    const Geometry& g = l->k->g;
    l->head_bytes = legacy ? 6u : uint32_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4u * g.n_slices;
    const uint64_t raw = l->raw_bytes();
    const bool ok = guard.ok && hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&l->d_px), raw + 4) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&l->d_len_legacy), 4) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&l->d_meta), l->meta_bytes()) == hipSuccess &&
                    hipHostMalloc(reinterpret_cast<void**>(&l->h_meta), l->meta_bytes(), hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        lane_destroy(l);
        return LLCOMP_MI_NOMEM;
    }
    l->bytes = raw + 24;
    if (int rc = lane_grow(l, payload_cap)) {
        lane_destroy(l);
        return rc;
    }
    *out = l;
    return LLCOMP_MI_OK;
}

int lane_enqueue_encode(HostLane* l) {
    if (int rc = llcomp_mi_codec_encode(l->k, l->d_px, l->d_payload(), l->payload_cap, l->d_len(), l->d_meta, l->d_meta + 1, l->stream))
        return rc;
    if (l->frames > 1) LLMI_HIP_TRY(launch_frame_bytes(l->k->g, l->d_len(), l->d_meta + 2, l->stream));
    LLMI_HIP_TRY(hipMemcpyAsync(l->h_meta, l->d_meta, l->meta_bytes(), hipMemcpyDeviceToHost, l->stream));
    return LLCOMP_MI_OK;
}

int lane_enqueue_decode(HostLane* l, uint64_t payload_bytes) {
    if (l->legacy) {
        const uint32_t one = uint32_t(std::min<uint64_t>(payload_bytes, 0xFFFFFFFFull));
        LLMI_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(l->d_len_legacy), int(one), 1, l->stream));
    }
    if (int rc = llcomp_mi_codec_decode(l->k, l->d_payload(), payload_bytes, l->d_len(), l->d_px, l->d_meta + 1, l->stream)) return rc;
    LLMI_HIP_TRY(hipMemcpyAsync(l->h_meta, l->d_meta, 16, hipMemcpyDeviceToHost, l->stream));
    return LLCOMP_MI_OK;
}

}  // namespace llcomp_mi

namespace {

// A lane holds GBs of workspace for a 4K frame; allocating it per call costs more than the coding.  A few idle ones are
// kept, keyed by device + geometry (which includes the kernel family the tuning hooks selected when the lane was
// made).  Never torn down at exit on purpose (the HIP runtime may already be gone by then).
struct LaneCache {
    struct Item { HostLane* l; uint64_t stamp; };
    std::mutex mu;
    std::vector<Item> idle;
    uint64_t clock = 0;
    // per device: a device list parks a lane on every GPU, concurrent callers one each (an 8192 x 8192 lane holds about 10 GB of the
    // GPU's 288)
    static constexpr size_t kMaxIdle = 6;
    static constexpr uint64_t kMaxIdleBytes = 40ull << 30;

    HostLane* take(int dev, const Geometry& g, bool legacy) {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].l->k->device == dev && idle[i].l->legacy == legacy && std::memcmp(&idle[i].l->k->g, &g, sizeof(Geometry)) == 0) {
                HostLane* l = idle[i].l;
                idle.erase(idle.begin() + long(i));
                return l;
            }
        return nullptr;
    }
    void give(HostLane* l) {
        std::vector<HostLane*> drop;
        {
            std::lock_guard<std::mutex> lock(mu);
            idle.push_back({l, ++clock});
            // the limits hold PER DEVICE (a device list parks one lane on every GPU); bytes = what the lane holds right now (the
            // codec's state tables / snapshot arrays only once a call has allocated them)
            const int dev = l->k->device;
            auto count = [&]() { size_t n = 0; for (auto& it : idle) n += it.l->k->device == dev; return n; };
            auto bytes = [&]() { uint64_t b = 0; for (auto& it : idle) if (it.l->k->device == dev) b += it.l->k->allocated_bytes + it.l->bytes; return b; };
            while (count() > kMaxIdle || (count() > 1 && bytes() > kMaxIdleBytes)) {
                size_t oldest = idle.size();
                for (size_t i = 0; i < idle.size(); ++i)
                    if (idle[i].l->k->device == dev && (oldest == idle.size() || idle[i].stamp < idle[oldest].stamp)) oldest = i;
                drop.push_back(idle[oldest].l);
                idle.erase(idle.begin() + long(oldest));
            }
        }
        for (auto* d : drop) lane_destroy(d);
    }
};
void drop_idle_lanes(LaneCache& c) {
    std::vector<HostLane*> drop;
    {
        std::lock_guard<std::mutex> lock(c.mu);
        for (auto& it : c.idle) drop.push_back(it.l);
        c.idle.clear();
    }
    for (auto* d : drop) lane_destroy(d);
}
LaneCache& lane_cache() {
    static LaneCache* c = new LaneCache;  // leaked deliberately
    return *c;
}

}  // namespace

namespace llcomp_mi {
int lane_acquire(HostLane** out, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar,
                 bool legacy, uint64_t min_cap, bool small_model) {
    Geometry g;
    std::memset(&g, 0, sizeof(g));
    if (!make_geometry(g, 1, w, h, c, tile_w, tile_h, planar, current_tuning(), small_model)) return LLCOMP_MI_OUT_OF_RANGE;
    int dev = 0;
    if (int rc = resolve_device(device, &dev)) return rc;
    if ((*out = lane_cache().take(dev, g, legacy))) {
        if (int rc = lane_grow(*out, min_cap)) {  // (a lane whose container buffer could not be reallocated is of no use to anybody)
            lane_destroy(*out);
            *out = nullptr;
            return rc;
        }
        return LLCOMP_MI_OK;
    }
    return lane_create(out, dev, w, h, c, tile_w, tile_h, planar, legacy, min_cap, small_model);
}
void lane_release(HostLane* l) {
    if (l) lane_cache().give(l);
}
}  // namespace llcomp_mi

namespace {

// Copies between a CALLER's buffer and HBM, stream-ordered on the lane's private stream and complete on return (staged by
// the runtime for pageable memory, plain DMA for pinned memory).  Measured on a 4K noise frame, buffers reused across
// calls: 2.1 ms encode + 2.1 ms decode with pageable and with pinned buffers alike; what makes the allocating calls
// slower is the first touch of a fresh 30 MB malloc per call, not the copy.
inline hipError_t copy_user(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    return n ? hipMemcpyWithStream(dst, src, n, kind, s) : hipSuccess;
}

struct LaneLease {  // returns the lane to the cache on every exit path
    HostLane* l = nullptr;
    ~LaneLease() { lane_release(l); }
};

// encode into `out` (capacity out_cap) when out != nullptr, else into a malloc'ed buffer returned through *out_alloc
int encode_common(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts, uint8_t* out, size_t out_cap,
                  uint8_t** out_alloc, size_t* out_len) {
    *out_len = 0;
    llcomp_mi_opts o{};
    o.struct_size = sizeof(o);
    o.format = LLCOMP_MI_FORMAT_LEGACY;
    o.device = -1;
    if (opts) {  // one layout per ABI version: a caller built against another header is refused, not half-read
        if (opts->struct_size != sizeof(llcomp_mi_opts)) return LLCOMP_MI_BAD_ARGS;
        std::memcpy(&o, opts, sizeof(o));
        if (o.small_model > 1 || o.reserved) return LLCOMP_MI_BAD_ARGS;
        if (o.n_devices && (!o.devices || o.n_devices > LLCOMP_MI_MAX_DEVICES)) return LLCOMP_MI_BAD_ARGS;
    }
    if (o.format != LLCOMP_MI_FORMAT_LEGACY && o.format != LLCOMP_MI_FORMAT_SLICED) return LLCOMP_MI_BAD_ARGS;
    const bool legacy = o.format == LLCOMP_MI_FORMAT_LEGACY;
    if (int rc = check_shape(w, h, c, legacy)) return rc;
    const uint32_t tile_w = legacy ? w : (o.tile_w == 0 || o.tile_w > w ? w : o.tile_w);
    const uint32_t tile_h = legacy ? h : (o.tile_h == 0 || o.tile_h > h ? h : o.tile_h);
    const uint32_t planar = legacy ? 0 : (o.planar ? 1 : 0);
    const uint64_t raw = uint64_t(w) * h * c;
    clear_device_error();
    if (o.n_devices) {
        // a device list: the tile rows are dealt over the devices (multidev.hip).  One serial stream does not shard, and a list of
        // one device is the plain call on that device.
        if (!legacy && o.n_devices > 1)
            return encode_multi(px, w, h, c, tile_w, tile_h, planar, o.small_model != 0, DeviceList{o.devices, o.n_devices, o.chunks_per_device}, out,
                                out_cap, out_alloc, out_len);
        o.device = o.devices[0];
        if (o.device < 0) return LLCOMP_MI_BAD_ARGS;
    }

    LaneLease lease;
    // first try with room for 2x raw (incompressible noise needs ~1.25x), then the proven worst case
    const uint64_t first_cap = 2 * raw + 64ull * llcomp_mi_slice_count(w, h, c, tile_w, tile_h, planar) + 4096;
    if (int rc = lane_acquire(&lease.l, o.device, w, h, c, tile_w, tile_h, planar, legacy, 0, o.small_model != 0)) return rc;
    HostLane* l = lease.l;
    const uint64_t max_payload = llcomp_mi_codec_max_payload_bytes(l->k);
    if (int rc = lane_grow(l, std::min(first_cap, max_payload))) return rc;
    DeviceGuard guard(l->k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    LLMI_HIP_TRY(copy_user(l->d_px, px, raw, hipMemcpyHostToDevice, l->stream));
    int rc = LLCOMP_MI_OK;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = lane_enqueue_encode(l))) return rc;
        LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
        rc = status_from_bits(uint32_t(l->h_meta[1]));
        if (rc == LLCOMP_MI_OUTPUT_OVERFLOW && l->payload_cap < max_payload) {
            if (int rc2 = lane_grow(l, max_payload)) return rc2;
            continue;
        }
        break;
    }
    if (rc) return rc;
    const size_t n = size_t(l->head_bytes) + size_t(l->h_meta[0]);
    *out_len = n;
    uint8_t* dst = out;
    if (!dst) {
        dst = static_cast<uint8_t*>(std::malloc(n + 1));
        if (!dst) return LLCOMP_MI_NOMEM;
    } else if (n > out_cap) {
        return LLCOMP_MI_OUTPUT_OVERFLOW;  // *out_len tells the caller what it takes
    }
    // header, slice table and payload sit in HBM exactly as on the wire (little-endian u32 on both sides): one copy
    if (copy_user(dst, l->d_container, n, hipMemcpyDeviceToHost, l->stream) != hipSuccess) {
        if (!out) std::free(dst);
        return LLCOMP_MI_HIP_ERROR;
    }
    if (out_alloc) *out_alloc = dst;
    return LLCOMP_MI_OK;
}

int decode_common(const uint8_t* data, size_t len, int32_t device, uint32_t flags, uint8_t* px, size_t px_cap, uint8_t** px_alloc,
                  uint32_t* w, uint32_t* h, uint32_t* c) {
    if (flags & ~LLCOMP_MI_FLAG_SMALL_MODEL) return LLCOMP_MI_BAD_ARGS;
    llcomp_mi_info info;
    if (int rc = llcomp_mi_probe(data, len, &info)) return rc;
    const bool legacy = info.format == LLCOMP_MI_FORMAT_LEGACY;
    if (int rc = check_shape(info.width, info.height, info.channels, legacy)) return rc;
    const uint64_t raw = uint64_t(info.width) * info.height * info.channels;
    *w = info.width;
    *h = info.height;
    *c = info.channels;
    if (px && raw > px_cap) return LLCOMP_MI_OUTPUT_OVERFLOW;  // dimensions are reported: the caller can size its buffer
    LaneLease lease;
    if (int rc = lane_acquire(&lease.l, device, info.width, info.height, info.channels, info.tile_w, info.tile_h, info.planar, legacy,
                              len - info.payload_offset + 16, legacy ? (flags & LLCOMP_MI_FLAG_SMALL_MODEL) != 0 : info.small_model != 0))
        return rc;
    HostLane* l = lease.l;
    DeviceGuard guard(l->k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    // the container goes to HBM as it is: the slice table is read where it lies (offset 24, dword aligned)
    LLMI_HIP_TRY(copy_user(l->d_container, data, len, hipMemcpyHostToDevice, l->stream));
    if (int rc = lane_enqueue_decode(l, len - l->head_bytes)) return rc;
    LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
    if (int rc = status_from_bits(uint32_t(l->h_meta[1]))) return rc;
    uint8_t* dst = px;
    if (!dst) {
        dst = static_cast<uint8_t*>(std::malloc(raw ? raw : 1));
        if (!dst) return LLCOMP_MI_NOMEM;
    }
    if (copy_user(dst, l->d_px, raw, hipMemcpyDeviceToHost, l->stream) != hipSuccess) {
        if (!px) std::free(dst);
        return LLCOMP_MI_HIP_ERROR;
    }
    if (px_alloc) *px_alloc = dst;
    return LLCOMP_MI_OK;
}

// decode over a device list: sliced containers whose table fits their payload are dealt over the devices (multidev.hip); a legacy
// stream, a list of one device and damaged containers (the one-device path forms their verdict) go to devices[0]
int decode_devices_common(const uint8_t* data, size_t len, const DeviceList& dl, uint32_t flags, uint8_t* px, size_t px_cap, uint8_t** px_alloc,
                          uint32_t* w, uint32_t* h, uint32_t* c) {
    if (flags & ~LLCOMP_MI_FLAG_SMALL_MODEL) return LLCOMP_MI_BAD_ARGS;
    clear_device_error();
    for (uint32_t i = 0; i < dl.n; ++i)
        if (dl.devices[i] < 0) return LLCOMP_MI_BAD_ARGS;
    llcomp_mi_info info;
    if (int rc = llcomp_mi_probe(data, len, &info)) return rc;
    if (info.format == LLCOMP_MI_FORMAT_SLICED && dl.n > 1) {
        bool handled = false;
        const int rc = decode_multi(data, len, info, dl, px, px_cap, px_alloc, w, h, c, &handled);
        if (handled) return rc;
    }
    return decode_common(data, len, dl.devices[0], flags, px, px_cap, px_alloc, w, h, c);
}

}  // namespace

extern "C" {

int llcomp_mi_encode(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts, uint8_t** out, size_t* out_len) {
    if (!px || !out || !out_len) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    const int rc = encode_common(px, w, h, c, opts, nullptr, 0, out, out_len);
    if (rc) *out_len = 0;
    return rc;
}

int llcomp_mi_encode_into(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, const llcomp_mi_opts* opts, uint8_t* out,
                          size_t out_cap, size_t* out_len) {
    if (!px || !out || !out_len) return LLCOMP_MI_BAD_ARGS;
    return encode_common(px, w, h, c, opts, out, out_cap, nullptr, out_len);
}

int llcomp_mi_decode(const uint8_t* data, size_t len, int32_t device, uint8_t** px, uint32_t* w, uint32_t* h, uint32_t* c) {
    if (!data || !px || !w || !h || !c) return LLCOMP_MI_BAD_ARGS;
    *px = nullptr;
    return decode_common(data, len, device, 0, nullptr, 0, px, w, h, c);
}

int llcomp_mi_decode_flags(const uint8_t* data, size_t len, int32_t device, uint32_t flags, uint8_t** px, uint32_t* w, uint32_t* h,
                           uint32_t* c) {
    if (!data || !px || !w || !h || !c) return LLCOMP_MI_BAD_ARGS;
    *px = nullptr;
    return decode_common(data, len, device, flags, nullptr, 0, px, w, h, c);
}

int llcomp_mi_decode_into(const uint8_t* data, size_t len, int32_t device, uint8_t* px, size_t px_cap, uint32_t* w, uint32_t* h,
                          uint32_t* c) {
    return llcomp_mi_decode_into_flags(data, len, device, 0, px, px_cap, w, h, c);
}

int llcomp_mi_decode_into_flags(const uint8_t* data, size_t len, int32_t device, uint32_t flags, uint8_t* px, size_t px_cap, uint32_t* w,
                                uint32_t* h, uint32_t* c) {
    if (!data || !px || !w || !h || !c || (flags & ~LLCOMP_MI_FLAG_SMALL_MODEL)) return LLCOMP_MI_BAD_ARGS;
    return decode_common(data, len, device, flags, px, px_cap, nullptr, w, h, c);
}

int llcomp_mi_decode_devices(const uint8_t* data, size_t len, const int32_t* devices, uint32_t n_devices, uint32_t chunks_per_device,
                             uint32_t flags, uint8_t** px, uint32_t* w, uint32_t* h, uint32_t* c) {
    if (!data || !px || !w || !h || !c || !devices || !n_devices || n_devices > LLCOMP_MI_MAX_DEVICES) return LLCOMP_MI_BAD_ARGS;
    *px = nullptr;
    return decode_devices_common(data, len, DeviceList{devices, n_devices, chunks_per_device}, flags, nullptr, 0, px, w, h, c);
}

int llcomp_mi_decode_into_devices(const uint8_t* data, size_t len, const int32_t* devices, uint32_t n_devices, uint32_t chunks_per_device,
                                  uint32_t flags, uint8_t* px, size_t px_cap, uint32_t* w, uint32_t* h, uint32_t* c) {
    if (!data || !px || !w || !h || !c || !devices || !n_devices || n_devices > LLCOMP_MI_MAX_DEVICES) return LLCOMP_MI_BAD_ARGS;
    return decode_devices_common(data, len, DeviceList{devices, n_devices, chunks_per_device}, flags, px, px_cap, nullptr, w, h, c);
}

void llcomp_mi_trim(void) {
    drop_idle_lanes(lane_cache());
    dev_release_idle();
}

void llcomp_mi_set_pool_limit(uint64_t bytes_per_device) { dev_set_limit(bytes_per_device); }
uint64_t llcomp_mi_pool_limit(void) { return dev_limit(); }
uint64_t llcomp_mi_pool_idle_bytes(void) { return dev_idle_bytes(); }

void* llcomp_mi_host_alloc(size_t bytes) {
    // portable: a buffer from here is handed to EVERY device of a device list (multidev.hip: each GPU copies its rows / its payload
    // straight out of / into the caller's buffer), not only to the device that was current when it was allocated
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return nullptr;
    return p;
}

void llcomp_mi_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
