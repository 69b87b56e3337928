// Device-memory cache under every workspace and lane buffer of the library.
//
// Why it exists.  Codec objects and host-API lanes come and go with the shapes a caller uses; round 2's randomised
// parity run (tests/test_gpu_stress.py, ~400 shapes in one process) returned each dropped lane's buffers to the driver
// with hipFree and took the next lane's with hipMalloc right afterwards.  On this stack (PyTorch's bundled HIP 7.0
// runtime over the MI355X box's driver) a buffer obtained that way occasionally showed whole 128-byte lines of ZEROS
// where a kernel had just written data -- seen in the lane-order symbol array between the transposition kernel and the
// slice encoder that reads it ~0.1 s later, 7 runs of 8; with the hipFree calls disabled (everything else identical)
// 0 runs of 16.  The zeros are what the driver writes when it wipes released VRAM.  The library therefore does not hand
// device memory back while it is in use: freed blocks are parked here and reused (best fit, at most 1/8 larger than
// asked), up to a budget PER DEVICE (16 GiB by default; LLCOMP_MI_POOL_MAX_BYTES or llcomp_mi_set_pool_limit, 0 = park
// nothing); llcomp_mi_trim() or a failed allocation of the library's own releases them.  An embedder that shares the GPU
// with another allocator (PyTorch's, say) calls llcomp_mi_trim() when THAT allocator runs out of memory.  Parking is
// stream-ordered: a block parked by a codec object carries the event of the codec's last call and is handed out again
// only after that event.  The cache also takes hipMalloc/hipFree (0.1-1 ms each, plus an implicit device
// synchronisation) off the path of every new shape.  Library-free reproducer of the failing sequence:
// tools/ubench/free_wipe.hip (verdict in DESIGN.md section 5).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <utility>

#include "codec_internal.hpp"

namespace llcomp_mi {

// An event that says "the work that used these blocks is done".  A codec object records one on the caller's stream at the
// end of every encode / decode; when the codec is destroyed its blocks are parked WITH that event, and whoever takes such
// a block out of the cache waits for the event first.  So destroying a codec neither drains the device (round 2 did a
// hipDeviceSynchronize there) nor depends on the caller having synchronised.  Shared by all blocks of one codec.
DoneEvent::~DoneEvent() {
    if (ev) (void)hipEventDestroy(ev);
}
std::shared_ptr<DoneEvent> make_done_event() {
    auto d = std::make_shared<DoneEvent>();
    if (hipEventCreateWithFlags(&d->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        d->ev = nullptr;
        return nullptr;
    }
    return d;
}

namespace {

struct DevPool {
    std::mutex mu;
    struct Block { int dev; uint64_t size; std::shared_ptr<DoneEvent> done; };
    std::unordered_map<void*, Block> owned;                      // every block handed out or parked
    std::multimap<std::pair<int, uint64_t>, void*> idle;         // parked blocks by (device, size)
    std::map<int, uint64_t> idle_bytes;                          // parked bytes per device
    // Parked bytes allowed PER DEVICE; beyond it the largest parked blocks of that device go back to the driver first.
    // LLCOMP_MI_POOL_MAX_BYTES (read once) or llcomp_mi_set_pool_limit(); 0 = park nothing (every free is a hipFree).
    uint64_t max_idle = 16ull << 30;
    bool limit_loaded = false;
};
DevPool& pool() {
    static DevPool* p = new DevPool;  // leaked deliberately: the HIP runtime may already be gone at exit
    return *p;
}
void load_limit_locked(DevPool& dp) {
    if (dp.limit_loaded) return;
    dp.limit_loaded = true;
    if (const char* e = std::getenv("LLCOMP_MI_POOL_MAX_BYTES")) {
        char* end = nullptr;
        const unsigned long long v = std::strtoull(e, &end, 10);
        if (end && end != e) dp.max_idle = v;
    }
}

uint64_t rounded(uint64_t bytes) {
    const uint64_t unit = bytes >= (64ull << 20) ? (2ull << 20) : bytes >= (1ull << 20) ? (64ull << 10) : 4096;
    return (bytes + unit - 1) / unit * unit;
}

// Workspace buffers are tens of GB and every wavefront walks its own region of them: with HBM handed out in small physical
// fragments (after many allocate / free cycles) the same kernels ran up to 2x slower (TLB reach).  Large blocks ask for
// physically contiguous memory first; any refusal falls back to a plain hipMalloc.
hipError_t raw_alloc(void** p, uint64_t bytes) {
    if (bytes >= (64ull << 20) && hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    return hipMalloc(p, bytes);
}

// one parked block back to the driver (hipFree waits for the device itself); the caller holds the lock
void release_block_locked(DevPool& dp, std::multimap<std::pair<int, uint64_t>, void*>::iterator it) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != it->first.first) (void)hipSetDevice(it->first.first);
    (void)hipFree(it->second);
    if (cur != it->first.first) (void)hipSetDevice(cur);
    dp.idle_bytes[it->first.first] -= it->first.second;
    dp.owned.erase(it->second);
    dp.idle.erase(it);
}
// parked blocks of `dev` (or of every device: dev < 0) back to the driver
void release_idle_locked(DevPool& dp, int dev) {
    for (auto it = dp.idle.begin(); it != dp.idle.end();) {
        auto next = std::next(it);
        if (dev < 0 || it->first.first == dev) release_block_locked(dp, it);
        it = next;
    }
}
void enforce_limit_locked(DevPool& dp, int dev) {
    load_limit_locked(dp);
    while (dp.idle_bytes[dev] > dp.max_idle) {  // largest first: few, large releases
        auto big = dp.idle.end();
        for (auto j = dp.idle.lower_bound({dev, 0}); j != dp.idle.end() && j->first.first == dev; ++j) big = j;  // (sorted by size)
        if (big == dp.idle.end()) break;
        release_block_locked(dp, big);
    }
}

}  // namespace

hipError_t dev_alloc(void** p, uint64_t bytes) {
    *p = nullptr;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    const uint64_t need = rounded(bytes ? bytes : 1);
    DevPool& dp = pool();
    std::shared_ptr<DoneEvent> wait_for;
    {
        std::lock_guard<std::mutex> lock(dp.mu);
        auto it = dp.idle.lower_bound({dev, need});
        if (it != dp.idle.end() && it->first.first == dev && it->first.second <= need + need / 8) {
            *p = it->second;
            dp.idle_bytes[dev] -= it->first.second;
            dp.idle.erase(it);
            auto o = dp.owned.find(*p);
            if (o != dp.owned.end()) wait_for = std::move(o->second.done);
        } else {
            hipError_t e = raw_alloc(p, need);
            if (e != hipSuccess) {  // memory pressure: give this device's parked blocks back and try once more
                (void)hipGetLastError();
                release_idle_locked(dp, dev);
                e = raw_alloc(p, need);
            }
            if (e != hipSuccess) return e;
            dp.owned[*p] = {dev, need, nullptr};
        }
    }
    // the block's previous user may still have work in flight on some stream: wait for its completion event (outside
    // the lock; normally the event completed long ago and this returns at once)
    // If that wait FAILS -- the caller destroyed the stream the event was recorded on while work was pending (legal HIP),
    // and the runtime answers with a sticky hipErrorCapturedEvent / invalid handle -- the event can say nothing any more:
    // the error is cleared and the whole device is drained instead, so the block is never handed out on an unproven wait.
    if (wait_for && wait_for->ev && hipEventSynchronize(wait_for->ev) != hipSuccess) {
        (void)hipGetLastError();
        if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    }
    return hipSuccess;
}

void dev_free(void* p, const std::shared_ptr<DoneEvent>& done) {
    if (!p) return;
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    auto it = dp.owned.find(p);
    if (it == dp.owned.end()) return;  // not ours (never happens: every buffer of the library comes from dev_alloc)
    it->second.done = done;
    const int dev = it->second.dev;
    dp.idle.insert({{dev, it->second.size}, p});
    dp.idle_bytes[dev] += it->second.size;
    enforce_limit_locked(dp, dev);
}

void dev_release_idle() {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    release_idle_locked(dp, -1);
}

uint64_t dev_idle_bytes() {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    uint64_t n = 0;
    for (auto& kv : dp.idle_bytes) n += kv.second;
    return n;
}

void dev_set_limit(uint64_t bytes_per_device) {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    dp.limit_loaded = true;
    dp.max_idle = bytes_per_device;
    for (auto& kv : dp.idle_bytes) enforce_limit_locked(dp, kv.first);
}

uint64_t dev_limit() {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    load_limit_locked(dp);
    return dp.max_idle;
}

}  // namespace llcomp_mi
