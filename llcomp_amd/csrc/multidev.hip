// multidev.hip -- ONE image over a list of GPUs, inside one process (BASELINE config 4 behind the C ABI: llcomp_mi_opts.devices,
// llcomp_mi_decode_devices; SURVEY 8b "device list").  The reference's callers are in-process C++ (llcompc.cpp:33, llcompd.cpp:26)
// and code one image with one call; this is that call with N devices behind it.  (llcomp_amd/sharding.py is the other multi-GPU
// shape: one process per GPU under torch.distributed, containers assembled in HBM over RCCL.)
//
// Slices have fresh adaptive state and slice-local borders, so a band of whole tile rows coded as an image of its own yields exactly
// the full image's slices.  llcomp_mi_plan_chunks deals the tile rows in chunks round-robin over the devices ("parts"); a part stacks
// its chunks into one local image and codes it with one cached lane (hostapi.hip) on its device.  The host buffers are the meeting
// point, so nothing is exchanged between the GPUs and there is no collective:
//
//   encode   every part: H2D of ITS rows only (one copy per chunk, its own PCIe link) -> kernels -> D2H of its slice-table pieces
//            straight to their place in the container's table
//            host: byte count of every chunk from the table (a few 100 K additions) -> where every chunk's payload goes
//            every part: D2H of its chunks' payload straight to their place in the container
//   decode   host: the same sums from the container's table
//            every part: H2D of its table pieces + payload pieces -> kernels -> status
//            all parts OK: D2H of every chunk's rows straight to their place in the picture
//
// A part runs on a thread of its own (the first on the caller's): pageable host memory is staged by the HIP runtime inside the copy
// call, so N threads are what keeps N links busy.  Nothing is published before every part has succeeded; the first failing part in
// list order decides the status -- data verdicts (BAD_EXPONENT, OUTPUT_OVERFLOW) as themselves, a failing device as
// LLCOMP_MI_DEVICE_FAILED with the device and its own status in llcomp_mi_last_device_error.
// Every coded byte still comes out of the kernels of slice_kernels.hip / model_kernels.hip: this file moves rows, table entries and
// payload bytes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/llcomp_mi.h"
#include "codec_internal.hpp"
#include "container.hpp"

using namespace llcomp_mi;

namespace {

thread_local struct { bool set; int32_t device; uint32_t index; int status; } t_dev_err = {false, 0, 0, 0};

struct Chunk {
    uint32_t y0, y1;      // pixel rows of the full image
    uint32_t s0, s1;      // slices of the full image (container order)
    uint32_t part;        // owner
    uint32_t local_y0;    // first row inside the owner's stacked image
    uint32_t local_s0;    // first slice inside the owner's table
    uint64_t bytes = 0;   // payload bytes of the chunk's slices
    uint64_t local_off = 0, final_off = 0;  // ... where they start in the owner's payload / in the container's payload
};

struct Part {
    int32_t device = 0;   // HIP ordinal as the caller wrote it
    uint32_t index = 0;   // position in the device list
    std::vector<uint32_t> chunks;
    uint32_t local_h = 0, local_slices = 0;
    uint64_t payload_bytes = 0;
    HostLane* lane = nullptr;
    int rc = LLCOMP_MI_OK;
};

struct Plan {
    uint32_t w = 0, h = 0, c = 0, tile_w = 0, tile_h = 0, planar = 0;
    uint32_t per_row = 0, spf = 0;  // slices per tile row / per image
    size_t row_bytes = 0;
    std::vector<Chunk> chunks;
    std::vector<Part> parts;  // only parts that own rows
};

int make_plan(Plan& p, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar, const DeviceList& dl) {
    p.w = w; p.h = h; p.c = c; p.tile_w = tile_w; p.tile_h = tile_h; p.planar = planar;
    p.row_bytes = size_t(w) * c;
    const uint32_t ntx = (w + tile_w - 1) / tile_w;
    p.per_row = ntx * (planar ? c : 1);
    p.spf = llcomp_mi_slice_count(w, h, c, tile_w, tile_h, planar);
    if (!p.spf) return LLCOMP_MI_OUT_OF_RANGE;
    for (uint32_t i = 0; i < dl.n; ++i)
        if (dl.devices[i] < 0) return LLCOMP_MI_BAD_ARGS;  // a list names its devices ("current device" means nothing in it)
    uint32_t n_chunks = 0;
    if (int rc = llcomp_mi_plan_chunks(h, tile_h, dl.n, dl.chunks_per_device, nullptr, 0, &n_chunks)) return rc;
    std::vector<uint32_t> tri(size_t(n_chunks) * 3);
    if (int rc = llcomp_mi_plan_chunks(h, tile_h, dl.n, dl.chunks_per_device, tri.data(), n_chunks, &n_chunks)) return rc;
    std::vector<Part> all(dl.n);
    for (uint32_t i = 0; i < dl.n; ++i) { all[i].device = dl.devices[i]; all[i].index = i; }
    p.chunks.resize(n_chunks);
    for (uint32_t i = 0; i < n_chunks; ++i) {
        Chunk& ch = p.chunks[i];
        const uint32_t t0 = tri[3 * i], t1 = tri[3 * i + 1];
        Part& owner = all[tri[3 * i + 2]];
        ch.y0 = t0 * tile_h;
        ch.y1 = std::min<uint64_t>(h, uint64_t(t1) * tile_h);
        ch.s0 = t0 * p.per_row;
        ch.s1 = t1 * p.per_row;
        ch.local_y0 = owner.local_h;
        ch.local_s0 = owner.local_slices;
        owner.local_h += ch.y1 - ch.y0;
        owner.local_slices += ch.s1 - ch.s0;
        owner.chunks.push_back(i);
    }
    // (the image's last tile row may be short: it is the last chunk, hence the last rows of its owner's stack -- a short last tile
    // row there as well)
    uint32_t k = 0;
    for (Part& a : all)
        if (a.local_h) {
            p.parts.push_back(a);
            for (uint32_t ci : a.chunks) p.chunks[ci].part = k;
            ++k;
        }
    return LLCOMP_MI_OK;
}

// runs fn(part) for every part, part 0 on this thread and the others on a thread each; the parts' results are in part.rc
template <typename Fn>
void for_each_part(std::vector<Part>& parts, Fn fn) {
    std::vector<std::thread> threads;
    threads.reserve(parts.size());
    for (size_t i = 1; i < parts.size(); ++i) {
        Part* p = &parts[i];
        if (p->rc) continue;
        try {
            threads.emplace_back([p, &fn] { p->rc = fn(*p); });
        } catch (...) {  // no thread to be had: the part runs here
            p->rc = fn(*p);
        }
    }
    if (!parts.empty() && !parts[0].rc) parts[0].rc = fn(parts[0]);
    for (auto& t : threads) t.join();
}

bool is_data_verdict(int rc) { return rc == LLCOMP_MI_BAD_EXPONENT || rc == LLCOMP_MI_TRUNCATED || rc == LLCOMP_MI_OUTPUT_OVERFLOW; }

// the first failing part in list order decides.  Verdicts about the data come back as themselves, and so does "this machine has no
// HIP device at all" (there is no list member to blame); everything else is the failure of ONE device of the list.
int verdict(const std::vector<Part>& parts) {
    for (const Part& p : parts)
        if (p.rc) return (is_data_verdict(p.rc) || p.rc == LLCOMP_MI_NO_DEVICE) ? p.rc : device_failed(p.device, p.index, p.rc);
    return LLCOMP_MI_OK;
}

struct LaneReturn {  // every lane goes back to the cache on every exit path
    std::vector<Part>& parts;
    ~LaneReturn() {
        for (Part& p : parts) {
            if (!p.lane) continue;
            if (p.rc && !is_data_verdict(p.rc)) lane_destroy(p.lane);  // a lane that saw a HIP error is not kept
            else lane_release(p.lane);
            p.lane = nullptr;
        }
    }
};

uint32_t part_tile_h(const Plan& pl, const Part& p) { return std::min(pl.tile_h, p.local_h); }

}  // namespace

namespace llcomp_mi {

void clear_device_error() { t_dev_err.set = false; }
int device_failed(int32_t device, uint32_t index, int status) {
    t_dev_err = {true, device, index, status};
    return LLCOMP_MI_DEVICE_FAILED;
}

int encode_multi(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar, bool small_model,
                 const DeviceList& dl, uint8_t* out, size_t out_cap, uint8_t** out_alloc, size_t* out_len) {
    *out_len = 0;
    Plan pl;
    if (int rc = make_plan(pl, w, h, c, tile_w, tile_h, planar, dl)) return rc;
    const size_t head_bytes = size_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4 * size_t(pl.spf);
    std::vector<uint8_t> head(head_bytes);  // header + the whole image's slice table; the parts fill in their pieces
    {
        Geometry g;
        if (!make_geometry(g, 1, w, h, c, tile_w, tile_h, planar, Tuning{}, small_model)) return LLCOMP_MI_OUT_OF_RANGE;
        write_sliced_header(head.data(), g);
    }
    LaneReturn lanes{pl.parts};

    // phase 1: rows in, kernels, slice-table pieces out
    for_each_part(pl.parts, [&](Part& p) -> int {
        const uint32_t th = part_tile_h(pl, p);
        if (int rc = lane_acquire(&p.lane, p.device, w, p.local_h, c, tile_w, th, planar, false, 0, small_model)) return rc;
        HostLane* l = p.lane;
        if (l->k->g.n_slices != p.local_slices) return LLCOMP_MI_HIP_ERROR;  // (the band's tiling is not the image's: cannot happen)
        const uint64_t raw = uint64_t(p.local_h) * pl.row_bytes;
        const uint64_t max_payload = llcomp_mi_codec_max_payload_bytes(l->k);
        if (int rc = lane_grow(l, std::min<uint64_t>(2 * raw + 64ull * p.local_slices + 4096, max_payload))) return rc;
        DeviceGuard guard(l->k->device);
        if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
        for (uint32_t ci : p.chunks) {
            const Chunk& ch = pl.chunks[ci];
            LLMI_HIP_TRY(hipMemcpyAsync(l->d_px + size_t(ch.local_y0) * pl.row_bytes, px + size_t(ch.y0) * pl.row_bytes,
                                        size_t(ch.y1 - ch.y0) * pl.row_bytes, hipMemcpyHostToDevice, l->stream));
        }
        int rc = LLCOMP_MI_OK;
        for (int attempt = 0; attempt < 2; ++attempt) {
            if ((rc = lane_enqueue_encode(l))) return rc;
            LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
            rc = status_from_bits(uint32_t(l->h_meta[1]));
            if (rc == LLCOMP_MI_OUTPUT_OVERFLOW && l->payload_cap < max_payload) {  // (the frame is still in d_px)
                if (int rc2 = lane_grow(l, max_payload)) return rc2;
                continue;
            }
            break;
        }
        if (rc) return rc;
        p.payload_bytes = l->h_meta[0];
        for (uint32_t ci : p.chunks) {
            const Chunk& ch = pl.chunks[ci];
            LLMI_HIP_TRY(hipMemcpyAsync(head.data() + LLCOMP_MI_SLICED_HEADER_BYTES + 4 * size_t(ch.s0), l->d_len() + ch.local_s0,
                                        4 * size_t(ch.s1 - ch.s0), hipMemcpyDeviceToHost, l->stream));
        }
        LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
        return LLCOMP_MI_OK;
    });
    if (int rc = verdict(pl.parts)) return rc;

    // host: where every chunk's payload goes
    uint64_t total = 0;
    std::vector<uint64_t> part_at(pl.parts.size(), 0);
    for (Chunk& ch : pl.chunks) {
        uint64_t sum = 0;
        const uint8_t* tab = head.data() + LLCOMP_MI_SLICED_HEADER_BYTES;
        for (uint32_t s = ch.s0; s < ch.s1; ++s) sum += get_u32le(tab + 4 * size_t(s));
        ch.bytes = sum;
        ch.final_off = total;
        ch.local_off = part_at[ch.part];
        total += sum;
        part_at[ch.part] += sum;
    }
    for (size_t i = 0; i < pl.parts.size(); ++i)
        if (part_at[i] != pl.parts[i].payload_bytes)  // table and payload of a part disagree: never publish that
            return device_failed(pl.parts[i].device, pl.parts[i].index, LLCOMP_MI_HIP_ERROR);
    const size_t n = head_bytes + size_t(total);
    *out_len = n;
    uint8_t* dst = out;
    if (!dst) {
        dst = static_cast<uint8_t*>(std::malloc(n + 1));
        if (!dst) return LLCOMP_MI_NOMEM;
    } else if (n > out_cap) {
        return LLCOMP_MI_OUTPUT_OVERFLOW;  // *out_len tells the caller what it takes; nothing was written
    }

    // phase 2: every part's payload straight to its place behind the table (N links, no gather)
    for_each_part(pl.parts, [&](Part& p) -> int {
        HostLane* l = p.lane;
        DeviceGuard guard(l->k->device);
        if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
        for (uint32_t ci : p.chunks) {
            const Chunk& ch = pl.chunks[ci];
            if (ch.bytes)
                LLMI_HIP_TRY(hipMemcpyAsync(dst + head_bytes + ch.final_off, l->d_payload() + ch.local_off, ch.bytes, hipMemcpyDeviceToHost, l->stream));
        }
        LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
        return LLCOMP_MI_OK;
    });
    if (int rc = verdict(pl.parts)) {
        if (!out) std::free(dst);
        *out_len = 0;
        return rc;
    }
    std::memcpy(dst, head.data(), head_bytes);  // the header last: a buffer without it is not a container
    if (out_alloc) *out_alloc = dst;
    return LLCOMP_MI_OK;
}

int decode_multi(const uint8_t* data, size_t len, const llcomp_mi_info& info, const DeviceList& dl, uint8_t* px, size_t px_cap,
                 uint8_t** px_alloc, uint32_t* w, uint32_t* h, uint32_t* c, bool* handled) {
    *handled = false;
    if (int rc = check_shape(info.width, info.height, info.channels, false)) { *handled = true; return rc; }
    Plan pl;
    if (int rc = make_plan(pl, info.width, info.height, info.channels, info.tile_w, info.tile_h, info.planar, dl)) { *handled = true; return rc; }
    if (pl.spf != info.n_slices) return LLCOMP_MI_OK;  // (probe has checked this: not handled)
    // The container's table on the host: every chunk's bytes and where they lie.  A table that promises more than the payload holds,
    // or a slice longer than any valid stream of its tile, is damaged input: the one-device path forms the verdict for it.
    const uint8_t* tab = data + info.table_offset;
    const uint64_t payload_len = len - info.payload_offset;
    uint64_t total = 0;
    std::vector<uint64_t> part_at(pl.parts.size(), 0);
    for (Chunk& ch : pl.chunks) {
        const Part& owner = pl.parts[ch.part];
        Geometry g;
        if (!make_geometry(g, 1, pl.w, owner.local_h, pl.c, pl.tile_w, part_tile_h(pl, owner), pl.planar)) return LLCOMP_MI_OK;
        uint64_t sum = 0;
        for (uint32_t s = ch.s0; s < ch.s1; ++s) {
            const uint32_t n = get_u32le(tab + 4 * size_t(s));
            if (n > g.slice_cap - 16) return LLCOMP_MI_OK;
            sum += n;
        }
        ch.bytes = sum;
        ch.final_off = total;
        ch.local_off = part_at[ch.part];
        total += sum;
        part_at[ch.part] += sum;
    }
    if (total > payload_len) return LLCOMP_MI_OK;
    *handled = true;
    const uint64_t raw = uint64_t(pl.h) * pl.row_bytes;
    *w = pl.w;
    *h = pl.h;
    *c = pl.c;
    if (px && raw > px_cap) return LLCOMP_MI_OUTPUT_OVERFLOW;  // dimensions are reported: the caller can size its buffer
    LaneReturn lanes{pl.parts};
    const uint8_t* payload = data + info.payload_offset;

    // phase 1: every part decodes its chunks
    for_each_part(pl.parts, [&](Part& p) -> int {
        const uint64_t bytes = part_at[&p - pl.parts.data()];
        if (int rc = lane_acquire(&p.lane, p.device, pl.w, p.local_h, pl.c, pl.tile_w, part_tile_h(pl, p), pl.planar, false, bytes + 16,
                                  info.small_model != 0))
            return rc;
        HostLane* l = p.lane;
        if (l->k->g.n_slices != p.local_slices) return LLCOMP_MI_HIP_ERROR;
        DeviceGuard guard(l->k->device);
        if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
        for (uint32_t ci : p.chunks) {
            const Chunk& ch = pl.chunks[ci];
            LLMI_HIP_TRY(hipMemcpyAsync(l->d_len() + ch.local_s0, tab + 4 * size_t(ch.s0), 4 * size_t(ch.s1 - ch.s0), hipMemcpyHostToDevice, l->stream));
            if (ch.bytes)
                LLMI_HIP_TRY(hipMemcpyAsync(l->d_payload() + ch.local_off, payload + ch.final_off, ch.bytes, hipMemcpyHostToDevice, l->stream));
        }
        if (int rc = lane_enqueue_decode(l, bytes)) return rc;
        LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
        return status_from_bits(uint32_t(l->h_meta[1]));
    });
    if (int rc = verdict(pl.parts)) return rc;

    uint8_t* dst = px;
    if (!dst) {
        dst = static_cast<uint8_t*>(std::malloc(raw ? raw : 1));
        if (!dst) return LLCOMP_MI_NOMEM;
    }
    // phase 2: every chunk's rows straight to their place in the picture
    for_each_part(pl.parts, [&](Part& p) -> int {
        HostLane* l = p.lane;
        DeviceGuard guard(l->k->device);
        if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
        for (uint32_t ci : p.chunks) {
            const Chunk& ch = pl.chunks[ci];
            LLMI_HIP_TRY(hipMemcpyAsync(dst + size_t(ch.y0) * pl.row_bytes, l->d_px + size_t(ch.local_y0) * pl.row_bytes,
                                        size_t(ch.y1 - ch.y0) * pl.row_bytes, hipMemcpyDeviceToHost, l->stream));
        }
        LLMI_HIP_TRY(hipStreamSynchronize(l->stream));
        return LLCOMP_MI_OK;
    });
    if (int rc = verdict(pl.parts)) {
        if (!px) std::free(dst);
        return rc;
    }
    if (px_alloc) *px_alloc = dst;
    return LLCOMP_MI_OK;
}

}  // namespace llcomp_mi

extern "C" int llcomp_mi_last_device_error(int32_t* device, uint32_t* index, int* status) {
    if (!t_dev_err.set) return 0;
    if (device) *device = t_dev_err.device;
    if (index) *index = t_dev_err.index;
    if (status) *status = t_dev_err.status;
    return 1;
}
