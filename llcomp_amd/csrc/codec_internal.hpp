// codec_internal.hpp -- the codec object behind llcomp_mi_codec_* and the helpers the host-side translation units share
// (codec.hip: device-resident batch codec; hostapi.hip: host-buffer drop-in calls; stream.hip: streaming pipeline).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/llcomp_mi.h"
#include "geometry.hpp"
#include "kernels.hpp"

namespace llcomp_mi {
struct DoneEvent {  // devmem.hip: completion event that travels with parked device blocks
    hipEvent_t ev = nullptr;
    ~DoneEvent();
};
std::shared_ptr<DoneEvent> make_done_event();
}  // namespace llcomp_mi

struct llcomp_mi_codec {
    llcomp_mi::Geometry g{};
    int device = 0;
    // workspace (all on `device`)
    void* d_sym_or_rec = nullptr;    // image order: encode u32 symbols per sample / decode int16 reconstructed samples
    void* d_lane_order = nullptr;    // the same data in lane order [group][k][64] for the serial kernels
    uint64_t* d_states = nullptr;    // u64[lane group][kContexts][lanes of the group]
    uint8_t* d_scratch = nullptr;    // slice streams in stream lane order: 16-byte units [group][unit][lane]
    uint64_t* d_group_off = nullptr; // u64[lane groups + 1]: payload offset of every lane group's first slice
    uint64_t* d_total_tmp = nullptr;
    void* d_snap_sorted = nullptr;   // snapshot pass of the 2-D encoder (snapshot.hpp): banks in context-sorted order,
    void* d_snap_banks = nullptr;    // banks in stream order, residuals in stream order; null unless snapshot_mode(g)
    void* d_snap_res = nullptr;
    void* d_snap_ctx = nullptr;      // ... slices above 4096 samples (snapshot_chunked): the context of every sorted position (u16) and the
    void* d_snap_io = nullptr;       // states every context run of a chunk starts from (u64), both scratch of the pass
    // ... their coder runs in segments of 4096 samples (a lane parks its coder in 64 bytes per slice in between), each behind ITS chunk
    // of the pass only: the pass works ahead of the coder on a second stream (one per device, shared by all codecs; fork / join with
    // events, so the caller's stream still orders everything; LLCOMP_MI_OVERLAP=0: in order, =1: a second stream of the codec's own)
    uint32_t* d_seg_state = nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_chunk[llcomp_mi::kSnapMaxChunks] = {};
    bool overlap = true;             // (LLCOMP_MI_OVERLAP when the codec was made)
    bool aux_shared = true;          // (`aux` is the device's shared second stream, not this codec's to destroy)
    uint64_t workspace_bytes = 0;    // what the codec can hold at most
    uint64_t allocated_bytes = 0;    // what it holds right now (state tables / snapshot arrays come with the first call that needs them)
    // Event counters (kernels.hpp kCtr*): kernel-side u64[kCtrCount] in HBM, host-side additions, and the feedback that takes the bank
    // cache away from a codec whose 2-D decode wavefronts ALL gave it up: a cached launch is followed by a 16-byte copy of
    // {cached wavefronts, bypassed wavefronts} into a pinned mailbox + an event; the next decode call looks at the event WITHOUT waiting
    // (no result yet = no change) and runs the plain kernel -- no LDS held for a cache nobody uses, so the helper kernels beside it find
    // room on the CUs -- for kPlainRun calls before it probes with the cache again.
    unsigned long long* d_counters = nullptr;
    uint64_t host_counters[llcomp_mi::kCtrCount] = {};
    uint64_t* h_feedback = nullptr;    // pinned: [0] cached wavefronts, [1] bypassed wavefronts (cumulative), as of fb_event
    hipEvent_t fb_event = nullptr;
    bool fb_pending = false;           // a mailbox copy has been queued and not looked at yet
    uint64_t fb_seen[2] = {0, 0};      // the mailbox's values at the last look
    uint32_t plain_calls_left = 0;     // > 0: this many decode calls run without the bank cache
    bool feedback = true;              // (LLCOMP_MI_NOFEEDBACK=1 when the codec was made: the cache stays on in every launch)
    static constexpr uint32_t kPlainRun = 15;
    bool need_states = true;  // false when the states live in LDS (1-row slices; one slice per wavefront)
    uint32_t state_generation = 0;  // tag of the last call that used d_states (kernels.hpp); 0 = the table has not been cleared yet
    // optional per-kernel timing (hipEvents on the caller's stream)
    bool profiling = false;
    struct Span { hipEvent_t a, b; int slot; };
    std::vector<Span> spans;
    uint32_t n_encode = 0, n_decode = 0;
    // recorded on the caller's stream behind the last launch of every encode / decode: the codec's device blocks are
    // parked with it when the codec is destroyed (no device-wide wait on destroy)
    std::shared_ptr<llcomp_mi::DoneEvent> done;
};

namespace llcomp_mi {

#define LLMI_HIP_TRY(expr)                                \
    do {                                                  \
        hipError_t _e = (expr);                           \
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
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

int status_from_bits(uint32_t bits);
Tuning current_tuning();  // environment hooks, read once per process (llcomp_mi_reload_tuning re-reads them)
int resolve_device(int32_t device, int* out);
// argument checks shared by every entry point that takes an image shape: BAD_ARGS for nonsense, OUT_OF_RANGE for sizes
// the formats cannot express (reference: silent truncation / int overflow)
int check_shape(uint32_t w, uint32_t h, uint32_t c, bool legacy);

// One coding lane of the host-side calls: a codec object for ONE frame plus everything a host buffer needs on its way
// through the GPU -- a private HIP stream (never the NULL stream: concurrent callers do not serialise device-wide),
// the frame and the container in HBM laid out exactly as on the wire ([header][slice table][payload], so a container
// crosses PCIe in ONE copy), and a pinned 16-byte mailbox for {payload bytes, status}.
struct HostLane {
    llcomp_mi_codec* k = nullptr;
    hipStream_t stream = nullptr;
    uint8_t* d_px = nullptr;
    uint8_t* d_container = nullptr;  // [header][u32 length table (sliced only)][payload capacity]
    uint32_t* d_len_legacy = nullptr;  // the legacy format has no table on the wire: its single length lives here
    uint64_t* d_meta = nullptr;      // [0] payload bytes (u64)  [1] low dword: status bits of the last call  [2 + f] payload bytes of frame f
    uint64_t* h_meta = nullptr;      // pinned mirror of d_meta
    uint32_t frames = 1;             // frames per call (> 1 only in the streaming pipeline)
    uint64_t payload_cap = 0;
    uint32_t head_bytes = 0;         // 6 (legacy) or 24 + 4 * slices (all frames' tables back to back)
    bool legacy = false;
    uint64_t bytes = 0;              // device bytes held (idle-cache budget)

    uint32_t* d_len() const { return legacy ? d_len_legacy : reinterpret_cast<uint32_t*>(d_container + LLCOMP_MI_SLICED_HEADER_BYTES); }
    uint8_t* d_payload() const { return d_container + head_bytes; }
    uint64_t raw_bytes() const { return uint64_t(k->g.w) * k->g.h * k->g.c * frames; }
    size_t meta_bytes() const { return 16 + 8 * size_t(frames); }
};
// builds a lane for this shape on `dev` (geometry + tuning hooks are fixed here); payload capacity `cap` bytes
int lane_create(HostLane** out, int dev, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar,
                bool legacy, uint64_t payload_cap, bool small_model = false, uint32_t frames = 1);
int lane_grow(HostLane* l, uint64_t payload_cap);  // reallocates the container buffer (contents lost)
void lane_destroy(HostLane* l);
// enqueue on the lane's stream (asynchronous): frame in d_px -> container in d_container, {bytes, status} -> h_meta
int lane_enqueue_encode(HostLane* l);
// container in d_container (header + tables + `payload_bytes` of payload) -> frame(s) in d_px, status -> h_meta
int lane_enqueue_decode(HostLane* l, uint64_t payload_bytes);

// hostapi.hip: a lane for this shape on `device` (-1 = current) from the cache of idle lanes, or a new one; lane_release parks it again
int lane_acquire(HostLane** out, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar,
                 bool legacy, uint64_t min_cap, bool small_model);
void lane_release(HostLane* l);

// multidev.hip: one image over a device list, in this process (llcomp_mi_opts.devices / llcomp_mi_decode_devices)
struct DeviceList {
    const int32_t* devices;
    uint32_t n;
    uint32_t chunks_per_device;  // 0 = 4
};
// `out` != nullptr: caller's buffer of out_cap bytes; else the container is malloc'ed and returned through *out_alloc
int encode_multi(const uint8_t* px, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w, uint32_t tile_h, uint32_t planar, bool small_model,
                 const DeviceList& dl, uint8_t* out, size_t out_cap, uint8_t** out_alloc, size_t* out_len);
// *handled = false (and nothing done) when the container has to go through the one-device path (its table does not fit its payload)
int decode_multi(const uint8_t* data, size_t len, const llcomp_mi_info& info, const DeviceList& dl, uint8_t* px, size_t px_cap,
                 uint8_t** px_alloc, uint32_t* w, uint32_t* h, uint32_t* c, bool* handled);
// the thread's record behind LLCOMP_MI_DEVICE_FAILED (llcomp_mi_last_device_error)
void clear_device_error();
int device_failed(int32_t device, uint32_t index, int status);  // records it and returns LLCOMP_MI_DEVICE_FAILED

void codec_release(llcomp_mi_codec* k);  // codec.hip: destroy without the device-wide wait (its work is known to be done)

// devmem.hip: every device buffer of the library comes from here.  dev_alloc is hipMalloc on the current device through a
// cache of parked blocks; dev_free parks a block -- either the caller has made sure nothing in flight still uses it (the
// lanes drain their private stream), or it passes the event behind the last use and the block is handed out again only
// after that event; dev_release_idle hands the parked blocks back to the driver.
hipError_t dev_alloc(void** p, uint64_t bytes);
void dev_free(void* p, const std::shared_ptr<DoneEvent>& done = nullptr);  // done: reuse only after this event
void dev_release_idle();
uint64_t dev_idle_bytes();
void dev_set_limit(uint64_t bytes_per_device);  // parked bytes allowed per device (0: every free goes to the driver)
uint64_t dev_limit();

}  // namespace llcomp_mi
