// stream.hip -- streaming pipeline of libllcomp_mi.so (llcomp_mi_stream_*): many frames of one shape flow
// host -> GPU -> host with several jobs in flight (BASELINE config 5).  The reference codes one image in
// RAM per call (llcompc.cpp:25-41, llcompd.cpp:17-31); this is the same operation as a pipeline.
//
// `depth` slots, each a HostLane (codec object for the frames of one job, private HIP stream, frames + containers in HBM)
// plus a pinned output buffer.  A job is `frames_per_job` frames (1 by default; a few frames per job give the GPU larger
// launches and the link larger copies), every frame its own SLICED container:
//   encode  H2D frame -> kernels -> D2H {payload bytes, status} (16 B, event e1) -> D2H container of the EXACT size (event e2)
//   decode  H2D container -> kernels -> D2H frame + status (event e2)
// The size of a container is known on the GPU only.  Nobody waits for it at submit time: the 16-byte mailbox copy is
// queued behind the kernels and whoever enters the library next (submit, wait or poll) looks at the events of the
// jobs in flight ("pump") and queues the container copies whose size has arrived; wait() blocks on the EVENT that comes
// next -- the size mailbox of the oldest encode job that still lacks one, else the oldest job's last copy -- with the
// object's mutex released, pumps, and blocks again, so the copies of younger jobs start while the caller waits for the
// oldest (round 2 polled with 20 us naps under the mutex).  With two or more slots busy the copies of one job overlap the
// kernels of the others in both PCIe directions.
// Failures: a submit that fails after its first copy was queued drains the lane's stream before it returns (the caller
// may free its buffer as soon as it sees the error), and a HIP error on a job in flight becomes that job's status
// (LLCOMP_MI_HIP_ERROR through wait()) instead of an error of the call that would leave the job at the head of the queue.
// Back-pressure: submit returns LLCOMP_MI_BUSY when every slot is occupied (in flight, or finished and not yet
// released); the caller takes a result (llcomp_mi_stream_wait), uses it and releases the slot.
// Results come back in submission order.  One stream object is driven by one thread at a time (calls are serialised by
// a mutex); use several objects for several producer threads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <deque>
#include <thread>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/llcomp_mi.h"
#include "codec_internal.hpp"
#include "container.hpp"

using namespace llcomp_mi;

namespace {
enum SlotState : int { kFree = 0, kEncSizing, kCopying, kFailed, kHeld };
struct Slot {
    HostLane* lane = nullptr;
    uint8_t* h_out = nullptr;  // pinned: the job's containers back to back (encode) or its frames (decode)
    std::vector<uint64_t> part_off, part_len;  // encode: where container f sits in h_out
    hipEvent_t e1 = nullptr, e2 = nullptr;
    SlotState state = kFree;
    uint32_t kind = 0;
    uint64_t tag = 0;
    uint64_t out_len = 0;
    int status = LLCOMP_MI_OK;
};
}  // namespace

struct llcomp_mi_stream {
    std::mutex mu;
    int device = 0;
    uint32_t w = 0, h = 0, c = 0, tile_w = 0, tile_h = 0, planar = 0, fpj = 1, spf = 0;
    uint64_t raw = 0, out_cap = 0;  // raw: bytes of ONE frame
    uint8_t header[LLCOMP_MI_SLICED_HEADER_BYTES] = {};
    std::vector<Slot> slots;
    std::deque<uint32_t> fifo;  // slots in submission order that have not been handed out by wait() yet
    uint64_t jobs_done = 0;
    // llcomp_mi_stream_create_multi: the object is a dealer in front of one plain pipeline per device -- `subs` non-empty, nothing of
    // the fields above in use except `mu`; `order` = which pipeline every pending job went to, in submission order; a result's slot
    // = pipeline index << 8 | that pipeline's slot
    std::vector<llcomp_mi_stream*> subs;
    std::deque<uint32_t> order;
    uint32_t next = 0;
};

namespace {

// encode job whose size mailbox has arrived: queue the container copies (exact sizes) or fail the job
int start_container_copy(llcomp_mi_stream* s, Slot& sl) {
    HostLane* l = sl.lane;
    const int rc = status_from_bits(uint32_t(l->h_meta[1]));
    if (rc) {
        sl.status = rc;
        sl.state = kFailed;
        return LLCOMP_MI_OK;
    }
    const uint64_t head1 = uint64_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4ull * s->spf;  // header + table of ONE container
    sl.out_len = head1 * s->fpj + l->h_meta[0];
    if (sl.out_len > s->out_cap) {  // cannot happen (payload capacity <= out_cap), but never write past a buffer
        sl.status = LLCOMP_MI_OUTPUT_OVERFLOW;
        sl.state = kFailed;
        return LLCOMP_MI_OK;
    }
    if (s->fpj == 1) {  // the frame's container sits in HBM exactly as on the wire: one copy
        sl.part_off[0] = 0;
        sl.part_len[0] = sl.out_len;
        LLMI_HIP_TRY(hipMemcpyAsync(sl.h_out, l->d_container, sl.out_len, hipMemcpyDeviceToHost, l->stream));
    } else {  // [header][table f][payload f] per frame, back to back: two copies per frame, the header from the host
        uint64_t at = 0, before = 0;
        for (uint32_t f = 0; f < s->fpj; ++f) {
            const uint64_t bytes = l->h_meta[2 + f];
            sl.part_off[f] = at;
            sl.part_len[f] = head1 + bytes;
            std::memcpy(sl.h_out + at, s->header, LLCOMP_MI_SLICED_HEADER_BYTES);
            LLMI_HIP_TRY(hipMemcpyAsync(sl.h_out + at + LLCOMP_MI_SLICED_HEADER_BYTES, l->d_len() + size_t(f) * s->spf, 4ull * s->spf,
                                        hipMemcpyDeviceToHost, l->stream));
            if (bytes)
                LLMI_HIP_TRY(hipMemcpyAsync(sl.h_out + at + head1, l->d_payload() + before, bytes, hipMemcpyDeviceToHost, l->stream));
            at += head1 + bytes;
            before += bytes;
        }
    }
    LLMI_HIP_TRY(hipEventRecord(sl.e2, l->stream));
    sl.state = kCopying;
    return LLCOMP_MI_OK;
}

// a HIP error on a job in flight: the job fails (reported through wait()), the pipeline goes on
void fail_job(Slot& sl, int status) {
    (void)hipGetLastError();
    if (sl.lane && sl.lane->stream) (void)hipStreamSynchronize(sl.lane->stream);  // nothing of it may still be in flight when the slot is reused
    sl.status = status;
    sl.state = kFailed;
}

void pump(llcomp_mi_stream* s) {
    for (uint32_t i : s->fifo) {
        Slot& sl = s->slots[i];
        if (sl.state != kEncSizing) continue;
        const hipError_t q = hipEventQuery(sl.e1);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) { fail_job(sl, LLCOMP_MI_HIP_ERROR); continue; }
        if (int rc = start_container_copy(s, sl)) fail_job(sl, rc);
    }
}

// a submit that fails after work was queued on the lane's stream: drain it, so that "not accepted" means "not touching the
// caller's buffer any more"
int drained(HostLane* l, int rc) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(l->stream);
    return rc;
}

// multi-device object: the job goes to the next pipeline in turn that takes it (one whose slots are all occupied is skipped)
template <typename Submit>
int deal(llcomp_mi_stream* s, Submit submit) {
    std::lock_guard<std::mutex> lock(s->mu);
    const uint32_t n = uint32_t(s->subs.size());
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = (s->next + k) % n;
        const int rc = submit(s->subs[i]);
        if (rc == LLCOMP_MI_BUSY) continue;
        if (rc) return rc;
        s->order.push_back(i);
        s->next = (i + 1) % n;
        return LLCOMP_MI_OK;
    }
    return LLCOMP_MI_BUSY;
}

int free_slot(llcomp_mi_stream* s) {
    for (size_t i = 0; i < s->slots.size(); ++i)
        if (s->slots[i].state == kFree) return int(i);
    return -1;
}

}  // namespace

extern "C" {

int llcomp_mi_stream_create(llcomp_mi_stream** out, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                            uint32_t tile_h, uint32_t planar, uint32_t depth) {
    return llcomp_mi_stream_create_ex(out, device, w, h, c, tile_w, tile_h, planar, depth, 1);
}

int llcomp_mi_stream_create_ex(llcomp_mi_stream** out, int32_t device, uint32_t w, uint32_t h, uint32_t c, uint32_t tile_w,
                               uint32_t tile_h, uint32_t planar, uint32_t depth, uint32_t frames_per_job) {
    if (!out) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    if (depth < 1 || depth > 16 || frames_per_job < 1 || frames_per_job > 64) return LLCOMP_MI_BAD_ARGS;
    if (int rc = check_shape(w, h, c, false)) return rc;
    int dev = 0;
    if (int rc = resolve_device(device, &dev)) return rc;
    llcomp_mi_stream* s = new (std::nothrow) llcomp_mi_stream;
    if (!s) return LLCOMP_MI_NOMEM;
    s->device = dev;
    s->w = w; s->h = h; s->c = c;
    s->tile_w = tile_w == 0 || tile_w > w ? w : tile_w;
    s->tile_h = tile_h == 0 || tile_h > h ? h : tile_h;
    s->planar = planar ? 1 : 0;
    s->raw = uint64_t(w) * h * c;
    s->fpj = frames_per_job;
    s->spf = llcomp_mi_slice_count(w, h, c, s->tile_w, s->tile_h, s->planar);
    s->slots.resize(depth);
    DeviceGuard guard(dev);
    int rc = guard.ok ? LLCOMP_MI_OK : LLCOMP_MI_HIP_ERROR;
    for (uint32_t i = 0; i < depth && !rc; ++i) {
        Slot& sl = s->slots[i];
        // room for 2x raw (noise needs ~1.25x); a frame that needs more fails with OUTPUT_OVERFLOW and can go through
        // llcomp_mi_encode, which retries with the proven worst case of 13 bytes per sample
        const uint64_t cap = (2 * s->raw + 64ull * s->spf + 4096) * s->fpj;
        rc = lane_create(&sl.lane, dev, w, h, c, s->tile_w, s->tile_h, s->planar, false, cap, false, s->fpj);
        if (rc) break;
        if (i == 0) write_sliced_header(s->header, sl.lane->k->g);
        sl.part_off.assign(s->fpj, 0);
        sl.part_len.assign(s->fpj, 0);
        s->out_cap = std::max<uint64_t>(s->raw * s->fpj, uint64_t(LLCOMP_MI_SLICED_HEADER_BYTES) * s->fpj + 4ull * s->spf * s->fpj + sl.lane->payload_cap);
        // (portable: behind a device list an encode result of THIS device's pipeline is handed to the decode job of another's)
        if (hipHostMalloc(reinterpret_cast<void**>(&sl.h_out), s->out_cap, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) rc = LLCOMP_MI_NOMEM;
        else if (hipEventCreateWithFlags(&sl.e1, hipEventDisableTiming) != hipSuccess ||
                 hipEventCreateWithFlags(&sl.e2, hipEventDisableTiming) != hipSuccess)
            rc = LLCOMP_MI_HIP_ERROR;
    }
    if (rc) {
        llcomp_mi_stream_destroy(s);
        return rc;
    }
    *out = s;
    return LLCOMP_MI_OK;
}

int llcomp_mi_stream_create_multi(llcomp_mi_stream** out, const int32_t* devices, uint32_t n_devices, uint32_t w, uint32_t h, uint32_t c,
                                  uint32_t tile_w, uint32_t tile_h, uint32_t planar, uint32_t depth, uint32_t frames_per_job) {
    if (!out) return LLCOMP_MI_BAD_ARGS;
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > LLCOMP_MI_MAX_DEVICES) return LLCOMP_MI_BAD_ARGS;
    clear_device_error();
    for (uint32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0) return LLCOMP_MI_BAD_ARGS;  // a list names its devices
    if (depth < 1 || depth > 16 || frames_per_job < 1 || frames_per_job > 64) return LLCOMP_MI_BAD_ARGS;
    if (int rc = check_shape(w, h, c, false)) return rc;
    llcomp_mi_stream* s = new (std::nothrow) llcomp_mi_stream;
    if (!s) return LLCOMP_MI_NOMEM;
    for (uint32_t i = 0; i < n_devices; ++i) {
        llcomp_mi_stream* sub = nullptr;
        if (int rc = llcomp_mi_stream_create_ex(&sub, devices[i], w, h, c, tile_w, tile_h, planar, depth, frames_per_job)) {
            llcomp_mi_stream_destroy(s);
            return rc == LLCOMP_MI_NO_DEVICE ? rc : device_failed(devices[i], i, rc);  // (no HIP device at all: nobody of the list to blame)
        }
        s->subs.push_back(sub);
    }
    s->fpj = frames_per_job;
    *out = s;
    return LLCOMP_MI_OK;
}

uint32_t llcomp_mi_stream_devices(const llcomp_mi_stream* s) { return !s ? 0 : s->subs.empty() ? 1 : uint32_t(s->subs.size()); }

void llcomp_mi_stream_destroy(llcomp_mi_stream* s) {
    if (!s) return;
    if (!s->subs.empty()) {
        for (auto* sub : s->subs) llcomp_mi_stream_destroy(sub);
        delete s;
        return;
    }
    DeviceGuard guard(s->device);
    for (Slot& sl : s->slots) {
        if (sl.lane && sl.lane->stream) (void)hipStreamSynchronize(sl.lane->stream);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.e1) (void)hipEventDestroy(sl.e1);
        if (sl.e2) (void)hipEventDestroy(sl.e2);
        lane_destroy(sl.lane);
    }
    delete s;
}

uint64_t llcomp_mi_stream_container_capacity(const llcomp_mi_stream* s) {
    return !s ? 0 : s->subs.empty() ? s->out_cap : s->subs[0]->out_cap;
}

int llcomp_mi_stream_submit_encode(llcomp_mi_stream* s, const uint8_t* px, uint64_t tag) {
    if (!s || !px) return LLCOMP_MI_BAD_ARGS;
    if (!s->subs.empty()) return deal(s, [&](llcomp_mi_stream* sub) { return llcomp_mi_stream_submit_encode(sub, px, tag); });
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    pump(s);
    const int i = free_slot(s);
    if (i < 0) return LLCOMP_MI_BUSY;
    Slot& sl = s->slots[size_t(i)];
    HostLane* l = sl.lane;
    LLMI_HIP_TRY(hipMemcpyAsync(l->d_px, px, s->raw * s->fpj, hipMemcpyHostToDevice, l->stream));  // the job's frames, back to back
    if (int rc = lane_enqueue_encode(l)) return drained(l, rc);
    if (hipEventRecord(sl.e1, l->stream) != hipSuccess) return drained(l, LLCOMP_MI_HIP_ERROR);
    sl.state = kEncSizing;
    sl.kind = LLCOMP_MI_JOB_ENCODE;
    sl.tag = tag;
    sl.status = LLCOMP_MI_OK;
    sl.out_len = 0;
    s->fifo.push_back(uint32_t(i));
    return LLCOMP_MI_OK;
}

int llcomp_mi_stream_submit_decode(llcomp_mi_stream* s, const uint8_t* data, size_t len, uint64_t tag) {
    if (!s || s->fpj != 1) return LLCOMP_MI_BAD_ARGS;  // jobs of several frames: llcomp_mi_stream_submit_decode_batch
    return llcomp_mi_stream_submit_decode_batch(s, &data, &len, tag);
}

int llcomp_mi_stream_submit_decode_batch(llcomp_mi_stream* s, const uint8_t* const* data, const size_t* lens, uint64_t tag) {
    if (!s || !data || !lens) return LLCOMP_MI_BAD_ARGS;
    if (!s->subs.empty()) return deal(s, [&](llcomp_mi_stream* sub) { return llcomp_mi_stream_submit_decode_batch(sub, data, lens, tag); });
    const uint64_t head1 = uint64_t(LLCOMP_MI_SLICED_HEADER_BYTES) + 4ull * s->spf;
    std::vector<uint64_t> pay(s->fpj);  // payload bytes every frame's slice table promises
    for (uint32_t f = 0; f < s->fpj; ++f) {
        if (!data[f]) return LLCOMP_MI_BAD_ARGS;
        llcomp_mi_info info;
        if (int rc = llcomp_mi_probe(data[f], lens[f], &info)) return rc;
        if (info.format != LLCOMP_MI_FORMAT_SLICED || info.width != s->w || info.height != s->h || info.channels != s->c ||
            info.tile_w != s->tile_w || info.tile_h != s->tile_h || info.planar != s->planar || info.small_model)
            return LLCOMP_MI_BAD_ARGS;  // a stream object codes ONE geometry
        if (s->fpj > 1) {  // the frames' payloads are laid side by side in HBM at the offsets their tables imply
            uint64_t sum = 0;
            for (uint32_t i = 0; i < s->spf; ++i) sum += get_u32le(data[f] + LLCOMP_MI_SLICED_HEADER_BYTES + 4ull * i);
            if (head1 + sum > lens[f]) return LLCOMP_MI_TRUNCATED;  // (a lone frame is bounds-checked on the GPU instead)
            pay[f] = sum;
        }
    }
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    pump(s);
    const int i = free_slot(s);
    if (i < 0) return LLCOMP_MI_BUSY;
    Slot& sl = s->slots[size_t(i)];
    HostLane* l = sl.lane;
    uint64_t payload_bytes = 0;
    if (s->fpj > 1) {  // (checked before anything is queued)
        uint64_t all = 0;
        for (uint32_t f = 0; f < s->fpj; ++f) all += pay[f];
        if (all > l->payload_cap) return LLCOMP_MI_OUTPUT_OVERFLOW;
    }
    if (s->fpj == 1) {
        // a container longer than the slot's buffer carries bytes no slice can use (the table is bounds-checked on the GPU)
        const uint64_t n = std::min<uint64_t>(lens[0], uint64_t(l->head_bytes) + l->payload_cap);
        LLMI_HIP_TRY(hipMemcpyAsync(l->d_container, data[0], n, hipMemcpyHostToDevice, l->stream));
        payload_bytes = n - l->head_bytes;
    } else {
        for (uint32_t f = 0; f < s->fpj; ++f) {
            if (hipMemcpyAsync(l->d_len() + size_t(f) * s->spf, data[f] + LLCOMP_MI_SLICED_HEADER_BYTES, 4ull * s->spf,
                               hipMemcpyHostToDevice, l->stream) != hipSuccess)
                return drained(l, LLCOMP_MI_HIP_ERROR);
            if (pay[f] && hipMemcpyAsync(l->d_payload() + payload_bytes, data[f] + head1, pay[f], hipMemcpyHostToDevice, l->stream) != hipSuccess)
                return drained(l, LLCOMP_MI_HIP_ERROR);
            payload_bytes += pay[f];
        }
    }
    if (int rc = lane_enqueue_decode(l, payload_bytes)) return drained(l, rc);
    if (hipMemcpyAsync(sl.h_out, l->d_px, s->raw * s->fpj, hipMemcpyDeviceToHost, l->stream) != hipSuccess ||
        hipEventRecord(sl.e2, l->stream) != hipSuccess)
        return drained(l, LLCOMP_MI_HIP_ERROR);
    sl.state = kCopying;
    sl.kind = LLCOMP_MI_JOB_DECODE;
    sl.tag = tag;
    sl.status = LLCOMP_MI_OK;
    sl.out_len = s->raw * s->fpj;
    s->fifo.push_back(uint32_t(i));
    return LLCOMP_MI_OK;
}

uint32_t llcomp_mi_stream_frames_per_job(const llcomp_mi_stream* s) { return s ? s->fpj : 0; }

int llcomp_mi_stream_result_part(llcomp_mi_stream* s, uint32_t slot, uint32_t frame, const uint8_t** data, uint64_t* len) {
    if (!s || !data || !len) return LLCOMP_MI_BAD_ARGS;
    if (!s->subs.empty()) return (slot >> 8) < s->subs.size() ? llcomp_mi_stream_result_part(s->subs[slot >> 8], slot & 0xFF, frame, data, len) : LLCOMP_MI_BAD_ARGS;
    std::lock_guard<std::mutex> lock(s->mu);
    if (slot >= s->slots.size() || frame >= s->fpj) return LLCOMP_MI_BAD_ARGS;
    const Slot& sl = s->slots[slot];
    if (sl.state != kHeld || sl.status != LLCOMP_MI_OK) return LLCOMP_MI_BAD_ARGS;
    if (sl.kind == LLCOMP_MI_JOB_ENCODE) {
        *data = sl.h_out + sl.part_off[frame];
        *len = sl.part_len[frame];
    } else {
        *data = sl.h_out + s->raw * frame;
        *len = s->raw;
    }
    return LLCOMP_MI_OK;
}

int llcomp_mi_stream_pending(llcomp_mi_stream* s) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lock(s->mu);
    return s->subs.empty() ? int(s->fifo.size()) : int(s->order.size());
}

int llcomp_mi_stream_poll(llcomp_mi_stream* s) {
    if (!s) return LLCOMP_MI_BAD_ARGS;
    if (!s->subs.empty()) {
        // every pipeline gets to look at its events (the container copies of younger jobs on OTHER devices are queued by whoever
        // enters the library next: here); the answer is the oldest job's
        llcomp_mi_stream* oldest = nullptr;
        {
            std::lock_guard<std::mutex> lock(s->mu);
            if (s->order.empty()) return LLCOMP_MI_OK;
            oldest = s->subs[s->order.front()];
        }
        int rc = LLCOMP_MI_OK;
        for (auto* sub : s->subs) {
            const int r = llcomp_mi_stream_poll(sub);
            if (sub == oldest) rc = r;
        }
        return rc;
    }
    std::lock_guard<std::mutex> lock(s->mu);
    DeviceGuard guard(s->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    pump(s);
    if (s->fifo.empty()) return LLCOMP_MI_OK;
    Slot& sl = s->slots[s->fifo.front()];
    if (sl.state == kFailed) return LLCOMP_MI_OK;
    if (sl.state == kCopying) {
        const hipError_t q = hipEventQuery(sl.e2);
        if (q == hipSuccess) return LLCOMP_MI_OK;
        if (q != hipErrorNotReady) { fail_job(sl, LLCOMP_MI_HIP_ERROR); return LLCOMP_MI_OK; }  // wait() reports it
    }
    return LLCOMP_MI_BUSY;  // the oldest job is still in flight
}

int llcomp_mi_stream_wait(llcomp_mi_stream* s, llcomp_mi_stream_result* r) {
    if (!s || !r) return LLCOMP_MI_BAD_ARGS;
    std::memset(r, 0, sizeof(*r));
    if (!s->subs.empty()) {
        uint32_t i = 0;
        {
            std::lock_guard<std::mutex> lock(s->mu);
            if (s->order.empty()) return LLCOMP_MI_BAD_ARGS;
            i = s->order.front();  // (single consumer: the oldest job cannot change while this thread waits for it)
        }
        // HIP has no "wait for any of these events", and a pipeline queues the container copy of a job only when somebody enters it
        // (pump): blocking on the oldest job's pipeline alone would leave the size mailboxes of the OTHER devices' younger jobs
        // unanswered for as long as the oldest job takes (measured: 4.9 instead of 6.3 GPix/s through {0,0}).  So the dealer looks at
        // every pipeline's events in turn, 50 us apart, until the oldest job is ready; the blocking wait below then returns at once.
        for (;;) {
            int oldest_ready = LLCOMP_MI_BUSY;
            for (uint32_t j = 0; j < s->subs.size(); ++j) {
                const int q = llcomp_mi_stream_poll(s->subs[j]);
                if (j == i) oldest_ready = q;
            }
            if (oldest_ready != LLCOMP_MI_BUSY) break;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (int rc = llcomp_mi_stream_wait(s->subs[i], r)) return rc;
        r->slot |= i << 8;
        std::lock_guard<std::mutex> lock(s->mu);
        s->order.pop_front();
        return LLCOMP_MI_OK;
    }
    std::unique_lock<std::mutex> lock(s->mu);
    if (s->fifo.empty()) return LLCOMP_MI_BAD_ARGS;  // nothing was submitted
    DeviceGuard guard(s->device);
    if (!guard.ok) return LLCOMP_MI_HIP_ERROR;
    const uint32_t i = s->fifo.front();
    Slot& sl = s->slots[i];
    // Block on the event that comes NEXT, not on the oldest job's last one: while this thread waits for the oldest job,
    // the size mailboxes of younger encode jobs keep arriving, and their container copies have to be queued when they do
    // or the D2H link idles.  Jobs run in submission order on lanes of equal speed, so the next event is the mailbox of
    // the oldest job that still lacks one, else the oldest job's final copy.  The mutex is released while blocked.
    // One consumer: wait / release / destroy of one pipeline object come from ONE thread (include/llcomp_mi.h) -- the oldest
    // job cannot change under this loop while the mutex is released.
    for (;;) {
        pump(s);
        if (sl.state == kFailed) break;
        hipEvent_t next = nullptr;
        Slot* owner = &sl;  // the job whose event `next` is: a failed wait is charged to IT
        if (sl.state == kCopying) {  // the oldest job's last copy is queued: a finished result goes out at once
            const hipError_t q = hipEventQuery(sl.e2);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { fail_job(sl, LLCOMP_MI_HIP_ERROR); break; }
        }
        for (uint32_t j : s->fifo)
            if (s->slots[j].state == kEncSizing) { next = s->slots[j].e1; owner = &s->slots[j]; break; }
        if (!next) next = sl.e2;
        lock.unlock();
        const hipError_t e = hipEventSynchronize(next);
        lock.lock();
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail_job(*owner, LLCOMP_MI_HIP_ERROR);
            if (owner == &sl) break;
        }
    }
    if (sl.state == kCopying && sl.kind == LLCOMP_MI_JOB_DECODE) sl.status = status_from_bits(uint32_t(sl.lane->h_meta[1]));
    s->fifo.pop_front();
    sl.state = kHeld;
    ++s->jobs_done;
    r->slot = i;
    r->kind = sl.kind;
    r->status = sl.status;
    r->tag = sl.tag;
    r->data = sl.status == LLCOMP_MI_OK ? sl.h_out : nullptr;
    r->len = sl.status == LLCOMP_MI_OK ? sl.out_len : 0;
    return LLCOMP_MI_OK;
}

int llcomp_mi_stream_release(llcomp_mi_stream* s, uint32_t slot) {
    if (!s) return LLCOMP_MI_BAD_ARGS;
    if (!s->subs.empty()) return (slot >> 8) < s->subs.size() ? llcomp_mi_stream_release(s->subs[slot >> 8], slot & 0xFF) : LLCOMP_MI_BAD_ARGS;
    std::lock_guard<std::mutex> lock(s->mu);
    if (slot >= s->slots.size() || s->slots[slot].state != kHeld) return LLCOMP_MI_BAD_ARGS;
    s->slots[slot].state = kFree;
    return LLCOMP_MI_OK;
}

}  // extern "C"
