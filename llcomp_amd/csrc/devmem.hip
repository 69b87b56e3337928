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
// asked), llcomp_mi_trim() or memory pressure releases them.  It also takes hipMalloc/hipFree (0.1-1 ms each, plus an
// implicit device synchronisation) off the path of every new shape.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

#include "codec_internal.hpp"

namespace llcomp_mi {
namespace {

struct DevPool {
    std::mutex mu;
    struct Block { int dev; uint64_t size; };
    std::unordered_map<void*, Block> owned;                      // every block handed out or parked
    std::multimap<std::pair<int, uint64_t>, void*> idle;         // parked blocks by (device, size)
    uint64_t idle_bytes = 0;
    static constexpr uint64_t kMaxIdleBytes = 64ull << 30;       // beyond this, the largest parked blocks go back first
};
DevPool& pool() {
    static DevPool* p = new DevPool;  // leaked deliberately: the HIP runtime may already be gone at exit
    return *p;
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

// parked blocks of `dev` (or of every device: dev < 0) back to the driver; the caller holds the lock
void release_idle_locked(DevPool& dp, int dev) {
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto it = dp.idle.begin(); it != dp.idle.end();) {
        if (dev >= 0 && it->first.first != dev) { ++it; continue; }
        (void)hipSetDevice(it->first.first);
        (void)hipFree(it->second);
        dp.idle_bytes -= it->first.second;
        dp.owned.erase(it->second);
        it = dp.idle.erase(it);
    }
    (void)hipSetDevice(cur);
}

}  // namespace

hipError_t dev_alloc(void** p, uint64_t bytes) {
    *p = nullptr;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    const uint64_t need = rounded(bytes ? bytes : 1);
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    auto it = dp.idle.lower_bound({dev, need});
    if (it != dp.idle.end() && it->first.first == dev && it->first.second <= need + need / 8) {
        *p = it->second;
        dp.idle_bytes -= it->first.second;
        dp.idle.erase(it);
        return hipSuccess;
    }
    hipError_t e = raw_alloc(p, need);
    if (e != hipSuccess) {  // memory pressure: give the parked blocks back and try once more
        (void)hipGetLastError();
        release_idle_locked(dp, dev);
        e = raw_alloc(p, need);
    }
    if (e == hipSuccess) dp.owned[*p] = {dev, need};
    return e;
}

void dev_free(void* p) {
    if (!p) return;
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    auto it = dp.owned.find(p);
    if (it == dp.owned.end()) return;  // not ours (never happens: every buffer of the library comes from dev_alloc)
    dp.idle.insert({{it->second.dev, it->second.size}, p});
    dp.idle_bytes += it->second.size;
    while (dp.idle_bytes > DevPool::kMaxIdleBytes) {  // largest first: few, large releases
        auto big = dp.idle.begin();
        for (auto j = dp.idle.begin(); j != dp.idle.end(); ++j) if (j->first.second > big->first.second) big = j;
        int cur = 0;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(big->first.first);
        (void)hipFree(big->second);
        (void)hipSetDevice(cur);
        dp.idle_bytes -= big->first.second;
        dp.owned.erase(big->second);
        dp.idle.erase(big);
    }
}

void dev_release_idle() {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    release_idle_locked(dp, -1);
}

uint64_t dev_idle_bytes() {
    DevPool& dp = pool();
    std::lock_guard<std::mutex> lock(dp.mu);
    return dp.idle_bytes;
}

}  // namespace llcomp_mi
