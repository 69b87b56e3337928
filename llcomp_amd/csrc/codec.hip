// codec.hip -- the device-resident batch codec of libllcomp_mi.so (llcomp_mi_codec_*: frames stay in HBM, work is
// enqueued on the caller's HIP stream) and the small status / version entry points of the C ABI.
// Every byte of coded data is produced by the kernels in model_kernels.hip / slice_kernels.hip; there is no CPU coding
// path anywhere in this library.  Host-buffer drop-in calls: hostapi.hip.  Streaming pipeline: stream.hip.
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
#include "geometry.hpp"
#include "kernels.hpp"
#include "snapshot.hpp"
#include "tables.hpp"

using namespace llcomp_mi;

namespace llcomp_mi {

int status_from_bits(uint32_t bits) {
    if (bits & kStInternal) return LLCOMP_MI_HIP_ERROR;
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

// The test / tuning hooks of the environment are read ONCE per process (first use) -- never per call or per launch;
// llcomp_mi_reload_tuning() (tests) reads them again.
static std::mutex g_tuning_mu;
static bool g_tuning_loaded = false;
static Tuning g_tuning;
Tuning current_tuning() {
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    if (!g_tuning_loaded) {
        g_tuning = tuning_from_env();
        g_tuning_loaded = true;
    }
    return g_tuning;
}

int check_shape(uint32_t w, uint32_t h, uint32_t c, bool legacy) {
    if (!w || !h || c < 1 || c > kMaxChannels) return LLCOMP_MI_BAD_ARGS;
    if (uint64_t(w) * h * c >= (1ull << 31)) return LLCOMP_MI_OUT_OF_RANGE;  // llcomp.hpp:359 `int size`
    if (legacy && (w > 65535 || h > 65535)) return LLCOMP_MI_OUT_OF_RANGE;  // u16 header fields, llcomp.hpp:377-378
    return LLCOMP_MI_OK;
}

}  // namespace llcomp_mi

namespace {

#define HIP_TRY(expr) LLMI_HIP_TRY(expr)

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

// State tables in HBM are tagged with the generation of the call that wrote them instead of being cleared per call
// (kernels.hpp): a real clear happens before the first call and whenever the 8-bit generation would repeat.
int ensure_state_tables(llcomp_mi_codec* k) {
    if (!k->need_states || k->d_states) return LLCOMP_MI_OK;
    const Geometry& g = k->g;
    if (dev_alloc(reinterpret_cast<void**>(&k->d_states), (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8) != hipSuccess) {
        k->d_states = nullptr;
        return LLCOMP_MI_NOMEM;
    }
    k->allocated_bytes += (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8;
    k->state_generation = 0;
    return LLCOMP_MI_OK;
}
int next_state_generation(llcomp_mi_codec* k, hipStream_t s) {
    if (!k->need_states) return LLCOMP_MI_OK;
    // first call that needs the tables (an encode-only codec with the snapshot pass never gets here; llcomp_mi_codec_prepare allocates them ahead)
    if (int rc = ensure_state_tables(k)) return rc;
    if (k->state_generation == 0 || k->state_generation >= 255) {
        const Geometry& g = k->g;
        if (k->state_generation >= 255) ++k->host_counters[kCtrGenerationWraps];
        HIP_TRY(hipMemsetAsync(k->d_states, 0, (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8, s));
        k->state_generation = 0;
    }
    ++k->state_generation;
    return LLCOMP_MI_OK;
}

// LLCOMP_MI_OVERLAP=2: one second stream per device for the snapshot passes of ALL codec objects (created once, never destroyed)
hipStream_t shared_second_stream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[device]) {
        // the highest priority: priority streams sit on hardware queues of their own.  An ordinary stream can land on the queue the
        // caller's stream uses (the NULL stream of a one-pipeline caller did: no overlap at all, 2 419 instead of 2 731 MPix/s at 16
        // frames of 128x128 planes, 3 844 instead of 4 410 at 32; profiles/r06_chunked_snapshot_ab.txt)
        int least = 0, greatest = 0;
        const bool prio = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
        const hipError_t rc = prio ? hipStreamCreateWithPriority(&streams[device], hipStreamNonBlocking, greatest)
                                   : hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking);
        if (rc != hipSuccess) {
            (void)hipGetLastError();
            streams[device] = nullptr;
        }
    }
    return streams[device];
}

// the snapshot pass's arrays (2-D encoder): allocated by the first encode -- a decode-only codec never pays for them
int ensure_snapshot_arrays(llcomp_mi_codec* k) {
    if (k->d_snap_sorted) return LLCOMP_MI_OK;
    const uint64_t el = snapshot_elems(k->g);
    const bool chunked = snapshot_chunked(k->g);
    if (dev_alloc(&k->d_snap_sorted, el * 8) != hipSuccess || dev_alloc(&k->d_snap_banks, el * 8) != hipSuccess ||
        dev_alloc(&k->d_snap_res, el * 2) != hipSuccess ||
        (chunked && (dev_alloc(&k->d_snap_ctx, el * 2) != hipSuccess || dev_alloc(&k->d_snap_io, el * 8) != hipSuccess))) {
        dev_free(k->d_snap_sorted);
        dev_free(k->d_snap_banks);
        dev_free(k->d_snap_res);
        dev_free(k->d_snap_ctx);
        dev_free(k->d_snap_io);
        k->d_snap_sorted = k->d_snap_banks = k->d_snap_res = k->d_snap_ctx = k->d_snap_io = nullptr;
        return LLCOMP_MI_NOMEM;
    }
    k->allocated_bytes += el * (chunked ? 28 : 18);
    if (chunked) {  // the coder's parking records, the second stream and the events of the fork / join
        if (dev_alloc(reinterpret_cast<void**>(&k->d_seg_state), uint64_t(k->g.n_slices) * 64) != hipSuccess) { k->d_seg_state = nullptr; return LLCOMP_MI_NOMEM; }
        k->allocated_bytes += uint64_t(k->g.n_slices) * 64;
        if (k->overlap) {
            bool ok = (k->aux_shared ? (k->aux = shared_second_stream(k->device)) != nullptr
                                     : hipStreamCreateWithFlags(&k->aux, hipStreamNonBlocking) == hipSuccess) &&
                      hipEventCreateWithFlags(&k->ev_fork, hipEventDisableTiming) == hipSuccess;
            for (uint32_t c = 0; ok && c < snapshot_chunks(k->g); ++c) ok = hipEventCreateWithFlags(&k->ev_chunk[c], hipEventDisableTiming) == hipSuccess;
            if (!ok) {  // no second stream to be had: the pass runs on the caller's (slower at few frames in flight, same bytes)
                (void)hipGetLastError();
                k->overlap = false;
            }
        }
    }
    return LLCOMP_MI_OK;
}

// The 2-D decoder's bank cache, per LAUNCH (codec_internal.hpp).  A wavefront that finds fewer than one hit in eight gives the cache up
// by itself, but it goes on holding its 18 KB of LDS to the end of the kernel: content that makes EVERY wavefront give up (a dithered
// gradient) paid the occupancy cap and the helper kernels' waits for nothing (round 5: 5 296 -> 4 622 MPix/s at 48 frames x 3).  The
// kernels count {wavefronts, wavefronts that gave up}; when (nearly) all of the last cached launch did, the codec's next kPlainRun
// decode calls run the plain kernel, then one call probes with the cache again.  Nothing waits: a result that has not arrived yet
// leaves things as they are.  Same bytes either way (the tables are per call).
bool use_bank_cache(llcomp_mi_codec* k) {
    if (bank_cache_log2(k->g) == 0) return false;
    if (k->fb_pending && k->fb_event && hipEventQuery(k->fb_event) == hipSuccess) {
        k->fb_pending = false;
        const uint64_t waves = k->h_feedback[0] - k->fb_seen[0], gave_up = k->h_feedback[1] - k->fb_seen[1];
        k->fb_seen[0] = k->h_feedback[0];
        k->fb_seen[1] = k->h_feedback[1];
        if (k->feedback && waves && gave_up * 16 >= waves * 15) k->plain_calls_left = llcomp_mi_codec::kPlainRun;
    } else {
        (void)hipGetLastError();
    }
    if (k->plain_calls_left) {
        --k->plain_calls_left;
        ++k->host_counters[kCtrDecLaunchesPlainByFeedback];
        return false;
    }
    ++k->host_counters[kCtrDecLaunchesCached];
    return true;
}
// ... behind a cached launch: {cached wavefronts, bypassed wavefronts} -> the pinned mailbox, and the event that says it has arrived
void queue_feedback(llcomp_mi_codec* k, hipStream_t s) {
    if (!k->h_feedback) {
        if (hipHostMalloc(reinterpret_cast<void**>(&k->h_feedback), 16, hipHostMallocDefault) != hipSuccess) { k->h_feedback = nullptr; (void)hipGetLastError(); return; }
        k->h_feedback[0] = k->h_feedback[1] = 0;
        if (hipEventCreateWithFlags(&k->fb_event, hipEventDisableTiming) != hipSuccess) { k->fb_event = nullptr; (void)hipGetLastError(); return; }
    }
    if (!k->fb_event || k->fb_pending) return;  // (the previous answer has not been looked at: its copy may still be in flight)
    if (hipMemcpyAsync(k->h_feedback, k->d_counters + kCtrDecCachedWaves, 16, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipEventRecord(k->fb_event, s) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    k->fb_pending = true;
}

// the codec's completion event: recorded behind the last launch of a call, on the caller's stream
void mark_done(llcomp_mi_codec* k, hipStream_t s) {
    if (!k->done) k->done = llcomp_mi::make_done_event();
    if (k->done && k->done->ev && hipEventRecord(k->done->ev, s) != hipSuccess) (void)hipGetLastError();
}
// ... from a scope guard, so that a call which fails AFTER it has launched kernels leaves the event behind them too (its
// blocks must not be reused while those kernels still run)
struct DoneGuard {
    llcomp_mi_codec* k;
    hipStream_t s;
    ~DoneGuard() { mark_done(k, s); }
};

}  // namespace

namespace llcomp_mi {
// parks the codec's blocks behind the event of its last call (nothing waits here: whoever takes a block out of the cache
// waits for that event; the lanes have drained their private stream before they get here anyway)
void codec_release(llcomp_mi_codec* k) {
    if (!k) return;
    DeviceGuard guard(k->device);
    // The event travels with the blocks only while the codec's last call is still running.  Usually it has long finished:
    // then the blocks are parked without it (an event must not outlive the stream it was recorded on -- a lane's private
    // stream is destroyed right after this -- and a finished event has nothing left to say).  A query that fails (the
    // caller destroyed its stream, which drains it) counts as finished.
    if (k->done && k->done->ev && hipEventQuery(k->done->ev) != hipErrorNotReady) {
        (void)hipGetLastError();
        k->done.reset();
    }
    dev_free(k->d_sym_or_rec, k->done);
    dev_free(k->d_lane_order, k->done);
    dev_free(k->d_states, k->done);
    dev_free(k->d_scratch, k->done);
    dev_free(k->d_group_off, k->done);
    dev_free(k->d_total_tmp, k->done);
    dev_free(k->d_snap_sorted, k->done);
    dev_free(k->d_snap_banks, k->done);
    dev_free(k->d_snap_res, k->done);
    dev_free(k->d_snap_ctx, k->done);
    dev_free(k->d_snap_io, k->done);
    dev_free(k->d_seg_state, k->done);
    // (the second stream's work of a call is joined into the caller's stream before the call's last kernels: behind k->done it is idle)
    if (k->ev_fork) (void)hipEventDestroy(k->ev_fork);
    for (auto& ev : k->ev_chunk) if (ev) (void)hipEventDestroy(ev);
    if (k->aux && !k->aux_shared) (void)hipStreamDestroy(k->aux);
    dev_free(k->d_counters, k->done);
    if (k->h_feedback) {
        // the 16-byte feedback copy of the last cached decode may still be queued on the caller's stream: it must not land in freed
        // memory.  Its own event says when it has arrived (usually long ago); only a codec destroyed right behind such a call waits.
        if (k->fb_pending && k->fb_event && hipEventQuery(k->fb_event) == hipErrorNotReady) (void)hipEventSynchronize(k->fb_event);
        (void)hipGetLastError();
        (void)hipHostFree(k->h_feedback);
    }
    if (k->fb_event) (void)hipEventDestroy(k->fb_event);
    for (auto& sp : k->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    delete k;
}
}  // namespace llcomp_mi

extern "C" {

int llcomp_mi_abi_version(void) { return LLCOMP_MI_ABI_VERSION; }

void llcomp_mi_reload_tuning(void) {
    std::lock_guard<std::mutex> lock(g_tuning_mu);
    g_tuning = tuning_from_env();
    g_tuning_loaded = true;
}

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
        case LLCOMP_MI_BUSY: return "all pipeline slots are in flight (take a finished job first)";
        case LLCOMP_MI_DEVICE_FAILED: return "a device of the device list failed (llcomp_mi_last_device_error tells which); nothing was published";
        default: return "unknown status";
    }
}

void llcomp_mi_free(void* p) { std::free(p); }

// ---- device-resident codec ------------------------------------------------------------------------------------
int llcomp_mi_codec_create(llcomp_mi_codec** out, int32_t device, uint32_t frames, uint32_t w, uint32_t h, uint32_t c,
                           uint32_t tile_w, uint32_t tile_h, uint32_t planar) {
    return llcomp_mi_codec_create_ex(out, device, frames, w, h, c, tile_w, tile_h, planar, 0);
}

int llcomp_mi_codec_create_ex(llcomp_mi_codec** out, int32_t device, uint32_t frames, uint32_t w, uint32_t h, uint32_t c,
                              uint32_t tile_w, uint32_t tile_h, uint32_t planar, uint32_t flags) {
    if (!out || (flags & ~LLCOMP_MI_FLAG_SMALL_MODEL)) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    if (!frames) return LLCOMP_MI_BAD_ARGS;
    if (int rc = check_shape(w, h, c, false)) return rc;
    Geometry g;
    // kernel family and lane-group width are fixed here, for the life of the codec object
    if (!make_geometry(g, frames, w, h, c, tile_w, tile_h, planar, current_tuning(), (flags & LLCOMP_MI_FLAG_SMALL_MODEL) != 0))
        return LLCOMP_MI_OUT_OF_RANGE;
    int dev = 0;
    if (int rc = resolve_device(device, &dev)) return rc;
    DeviceGuard guard(dev);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    llcomp_mi_codec* k = new (std::nothrow) llcomp_mi_codec;
    if (!k) return LLCOMP_MI_NOMEM;
    k->g = g;
    k->device = dev;
    k->feedback = !current_tuning().nofeedback;
    k->overlap = current_tuning().overlap != 0;
    k->aux_shared = current_tuning().overlap == 2;
    const uint64_t samples = uint64_t(frames) * w * h * c;
    k->need_states = slices_need_state_tables(g);
    // the fused row path (planar 1-row slices) has no image-order intermediate and 16-bit lane-order arrays in both directions
    const bool fused = model_is_fused(g);
    const uint64_t b_sym = fused ? 8 : samples * 4, b_states = k->need_states ? (uint64_t(lane_groups(g)) * kContexts << g.lane_shift) * 8 : 8,
                   b_scratch = (uint64_t(lane_groups(g)) << g.lane_shift) * g.slice_cap, b_off = (uint64_t(lane_groups(g)) + 1) * 8;
    // the 2-D encoder's snapshot pass: sorted entries (u32, they take the place of the lane-order symbols), banks in sorted and
    // in stream order (u64 each), residuals (i16) -- piece layout, snapshot.hpp
    const bool snap = snapshot_mode(g);
    const uint64_t snap_el = snap ? snapshot_elems(g) : 0;
    const uint64_t b_lanes = std::max((uint64_t(lane_groups(g)) * slice_capacity_samples(g) << g.lane_shift) * (fused ? 2 : 4), snap_el * 4);
    // What the codec can hold at most.  The state tables (decode, and encode without the snapshot pass) and the snapshot arrays
    // (encode) are allocated by the first call that needs them: a codec that only ever encodes, or only ever decodes, 64x64 tiles
    // holds 8.8 GB resp. 6.2 GB less per 16 frames of 4K than this figure.
    k->workspace_bytes = b_sym + b_lanes + b_states + b_scratch + b_off + 8 + snap_el * (snapshot_chunked(g) ? 28 : 18);
    const bool ok = dev_alloc(&k->d_sym_or_rec, b_sym) == hipSuccess && dev_alloc(&k->d_lane_order, b_lanes) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&k->d_scratch), b_scratch) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&k->d_group_off), b_off) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&k->d_total_tmp), 8) == hipSuccess &&
                    dev_alloc(reinterpret_cast<void**>(&k->d_counters), kCtrCount * 8) == hipSuccess &&
                    hipMemset(k->d_counters, 0, kCtrCount * 8) == hipSuccess;
    if (!ok) {
        llcomp_mi_codec_destroy(k);
        return LLCOMP_MI_NOMEM;
    }
    k->allocated_bytes = b_sym + b_lanes + b_scratch + b_off + 8;
    *out = k;
    return LLCOMP_MI_OK;
}

void llcomp_mi_codec_destroy(llcomp_mi_codec* k) {
    // No device-wide wait: the blocks go back to the library's cache (devmem.hip) together with the event recorded behind
    // the codec's last encode / decode, and are handed out again only after it.  Profiling events that were never read
    // are destroyed by codec_release (hipEventDestroy of a pending event is legal: it is released when it completes).
    llcomp_mi::codec_release(k);
}

int llcomp_mi_codec_prepare(llcomp_mi_codec* k, uint32_t what) {
    if (!k || (what & ~(LLCOMP_MI_PREPARE_ENCODE | LLCOMP_MI_PREPARE_DECODE))) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    if (what & LLCOMP_MI_PREPARE_ENCODE) {
        if (snapshot_mode(k->g)) {
            if (int rc = ensure_snapshot_arrays(k)) return rc;
        }
        if (!snapshot_mode(k->g) || snapshot_chunked(k->g))
            if (int rc = ensure_state_tables(k)) return rc;
    }
    if (what & LLCOMP_MI_PREPARE_DECODE)
        if (int rc = ensure_state_tables(k)) return rc;
    return LLCOMP_MI_OK;
}

uint32_t llcomp_mi_codec_slices(const llcomp_mi_codec* k) { return k ? k->g.n_slices : 0; }
uint32_t llcomp_mi_codec_kernel_family(const llcomp_mi_codec* k) { return k ? (k->g.flags & 0xFFu) | (k->g.lane_shift << 8) | (k->g.lpw << 16) : 0; }
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
    DoneGuard done_guard{k, s};
    HIP_TRY(hipMemsetAsync(d_status, 0, 4, s));
    // (the snapshot encoder of slices up to 4096 samples never touches the state tables: they are the decoder's alone; above that
    // the pass carries a context's states from chunk to chunk through them, under a generation of its own)
    if (!snapshot_mode(g) || snapshot_chunked(g)) {
        Timed t(k, s, 0);
        if (int rc = next_state_generation(k, s)) return rc;
    }
    {
        Timed t(k, s, 1);
        if (model_is_fused(g)) {
            HIP_TRY(launch_model_rows_fwd(g, static_cast<const uint8_t*>(d_px), static_cast<uint16_t*>(k->d_lane_order), s));
        } else {
            HIP_TRY(launch_model_fwd(g, static_cast<const uint8_t*>(d_px), static_cast<uint32_t*>(k->d_sym_or_rec), s));
            if (!snapshot_mode(g))
                HIP_TRY(launch_to_lane_order_u32(g, static_cast<const uint32_t*>(k->d_sym_or_rec),
                                                 static_cast<uint32_t*>(k->d_lane_order), s));
        }
    }
    if (snapshot_mode(g) && snapshot_chunked(g)) {
        // Slices above 4096 samples: pass and coder chunk by chunk.  The pass of chunk c + 1 needs the WALK of chunk c (the contexts'
        // states travel through the table), the coder of chunk c needs the pass of chunk c only: so the pass runs AHEAD on a second
        // stream and the coder follows on the caller's, each segment behind its chunk's event (fork / join: the caller's stream still
        // orders everything).  A launch of such slices is a few hundred wavefronts -- one wavefront's dependent chain -- and the
        // pass's nine or twelve launches in FRONT of it cost 8-11 % against the table encoder at few frames in flight; beside it they
        // are hidden.  The second stream is ONE PER DEVICE, shared by all codec objects (the passes are throughput kernels: they may
        // queue behind each other): a second stream per codec left the GPU idle as soon as three pipelines made six streams (25 %
        // below in-order; more hardware queues change nothing) -- profiles/r06_chunked_snapshot_ab.txt.  LLCOMP_MI_OVERLAP=0: in order.
        if (int rc = ensure_snapshot_arrays(k)) return rc;
        const uint64_t gpat = state_generation_tag(k->state_generation);
        const uint32_t chunks = snapshot_chunks(g);
        hipStream_t ps = k->overlap ? k->aux : s;
        if (k->overlap) {  // fork: the pass starts behind stage A
            HIP_TRY(hipEventRecord(k->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(k->aux, k->ev_fork, 0));
        }
        auto pass = [&](uint32_t c) -> int {
            Timed t(k, ps, 0);
            HIP_TRY(launch_snapshot_chunk(g, c, static_cast<const uint32_t*>(k->d_sym_or_rec), k->d_lane_order, k->d_snap_sorted, k->d_snap_banks,
                                          k->d_snap_res, k->d_snap_ctx, k->d_snap_io, k->d_states, gpat, ps));
            return LLCOMP_MI_OK;
        };
        auto coder = [&](uint32_t c) -> int {
            Timed t(k, s, 2);
            HIP_TRY(launch_encode_segment(g, k->d_snap_res, static_cast<uint64_t*>(k->d_snap_banks), k->d_scratch, static_cast<uint32_t*>(d_slice_len),
                                          static_cast<uint32_t*>(d_status), k->d_counters, c * kSnapMaxSamples, k->d_seg_state, s));
            return LLCOMP_MI_OK;
        };
        int rc = LLCOMP_MI_OK;
        if (k->overlap) {
            for (uint32_t c = 0; c < chunks && !rc; ++c) {
                rc = pass(c);
                if (!rc && hipEventRecord(k->ev_chunk[c], k->aux) != hipSuccess) rc = LLCOMP_MI_HIP_ERROR;
            }
            // join: every segment waits for its chunk -- also when something failed above: whatever was queued on the second stream
            // has to be behind the caller's stream before this call returns its buffers to anybody
            for (uint32_t c = 0; c < chunks; ++c) {
                if (hipStreamWaitEvent(s, k->ev_chunk[c], 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(k->aux); }
                if (!rc) rc = coder(c);
            }
        } else {
            for (uint32_t c = 0; c < chunks && !rc; ++c) {
                rc = pass(c);
                if (!rc) rc = coder(c);
            }
        }
        if (rc) return rc;
    } else {
    if (snapshot_mode(g)) {  // states replayed ahead of the coder: it reads banks + residuals front to back, no table
        if (int rc = ensure_snapshot_arrays(k)) return rc;
        Timed t(k, s, 0);    // (profile slot 0: the pass takes the place of the state tables whose clear the slot times otherwise)
        HIP_TRY(launch_snapshot(g, static_cast<const uint32_t*>(k->d_sym_or_rec), k->d_lane_order, k->d_snap_sorted,
                                k->d_snap_banks, k->d_snap_res, s));
    }
    {
        Timed t(k, s, 2);
        const bool snap = snapshot_mode(g);
        HIP_TRY(launch_encode_slices(g, snap ? k->d_snap_res : k->d_lane_order, snap ? static_cast<uint64_t*>(k->d_snap_banks) : k->d_states,
                                     k->state_generation, k->d_scratch, static_cast<uint32_t*>(d_slice_len),
                                     k->d_group_off, static_cast<uint32_t*>(d_status), k->d_counters, s));
    }
    }
    {
        Timed t(k, s, 3);
        if (!encoder_writes_group_sums(g)) HIP_TRY(launch_group_sums(g, static_cast<const uint32_t*>(d_slice_len), k->d_group_off, s));
        HIP_TRY(launch_scan_groups(g, k->d_group_off, static_cast<uint64_t*>(d_total), s));
        HIP_TRY(launch_pack_payload(g, k->d_scratch, static_cast<const uint32_t*>(d_slice_len), k->d_group_off,
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
    DoneGuard done_guard{k, s};
    HIP_TRY(hipMemsetAsync(d_status, 0, 4, s));
    {
        Timed t(k, s, 7);
        if (int rc = next_state_generation(k, s)) return rc;
    }
    {
        Timed t(k, s, 4);
        HIP_TRY(launch_group_sums(g, static_cast<const uint32_t*>(d_slice_len), k->d_group_off, s));
        HIP_TRY(launch_scan_groups(g, k->d_group_off, k->d_total_tmp, s));
    }
    {
        Timed t(k, s, 4);
        HIP_TRY(launch_stage_streams(g, static_cast<const uint8_t*>(d_payload), payload_bytes,
                                     static_cast<const uint32_t*>(d_slice_len), k->d_group_off, k->d_scratch,
                                     static_cast<uint32_t*>(d_status), s));
    }
    {
        Timed t(k, s, 5);
        const bool cache = use_bank_cache(k);
        HIP_TRY(launch_decode_slices(g, k->d_scratch, static_cast<const uint32_t*>(d_slice_len), k->d_states, k->state_generation,
                                     static_cast<int16_t*>(k->d_lane_order), static_cast<uint32_t*>(d_status), k->d_counters, cache, s));
        if (cache) queue_feedback(k, s);
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

int llcomp_mi_device_copy_segments(const void* d_src, void* d_dst, const void* d_src_off, const void* d_dst_off, const void* d_len,
                                   uint32_t n_seg, uint64_t max_len, void* stream) {
    if (!n_seg) return LLCOMP_MI_OK;
    if (!d_src || !d_dst || !d_src_off || !d_dst_off || !d_len || n_seg > 65535) return LLCOMP_MI_BAD_ARGS;
    HIP_TRY(launch_copy_segments(static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), static_cast<const uint64_t*>(d_src_off),
                                 static_cast<const uint64_t*>(d_dst_off), static_cast<const uint64_t*>(d_len), n_seg, max_len,
                                 static_cast<hipStream_t>(stream)));
    return LLCOMP_MI_OK;
}

int llcomp_mi_device_range_sums(const void* d_vals, const void* d_start, const void* d_count, void* d_out, uint32_t n, uint32_t cap,
                                void* stream) {
    if (!n) return LLCOMP_MI_OK;
    if (!d_vals || !d_start || !d_count || !d_out) return LLCOMP_MI_BAD_ARGS;
    HIP_TRY(launch_range_sums(static_cast<const uint32_t*>(d_vals), static_cast<const uint64_t*>(d_start), static_cast<const uint64_t*>(d_count),
                              static_cast<uint64_t*>(d_out), n, cap, static_cast<hipStream_t>(stream)));
    return LLCOMP_MI_OK;
}

int llcomp_mi_codec_get_counters(llcomp_mi_codec* k, uint64_t* out, uint32_t n, int reset) {
    if (!k || !out || n > kCtrCount) return LLCOMP_MI_BAD_ARGS;
    DeviceGuard guard(k->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    // the codec's last call has to be done before its counts mean anything: wait for ITS event (not for the device)
    if (k->done && k->done->ev && hipEventSynchronize(k->done->ev) != hipSuccess) (void)hipGetLastError();
    uint64_t dev[kCtrCount];
    HIP_TRY(hipMemcpy(dev, k->d_counters, sizeof(dev), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) out[i] = dev[i] + k->host_counters[i];
    if (reset) {
        HIP_TRY(hipMemset(k->d_counters, 0, sizeof(dev)));
        for (auto& h : k->host_counters) h = 0;
        k->fb_seen[0] = k->fb_seen[1] = 0;
        k->fb_pending = false;  // (a mailbox copy still in flight would carry pre-reset values)
        if (k->h_feedback) k->h_feedback[0] = k->h_feedback[1] = 0;
    }
    return LLCOMP_MI_OK;
}

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

