// rand_table.hip -- what does one random 8-byte read-modify-write into a multi-GB table cost on gfx950?
// Models the state-bank access of the 2-D slice kernels (one u64 bank per sample out of a private 63 KB table per
// lane): every lane owns a 63408-byte table, picks a pseudo-random bank that depends on the value
// it has just loaded (serial chain, like the decoder), adds one and writes it back.
//   rand_table <waves> <steps> <alloc: 0 hipMalloc | 1 uncached | 2 fine-grained | 3 physically contiguous> <layout: 0 [group][ctx][lane] | 1 [slice][ctx]>
//              <flavour: 0 plain | 1 nontemporal load+store | 2 plain load, nontemporal store>
// Prints accesses/s and the implied line traffic; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for bytes per access.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr uint32_t kCtx = 7926;

template <int FLAVOUR>
__global__ __launch_bounds__(64) void k_rmw(unsigned long long* __restrict__ tab, uint32_t steps, uint32_t layout,
                                            unsigned long long* __restrict__ sink) {
    const size_t slice = size_t(blockIdx.x) * 64 + threadIdx.x;
    unsigned long long* base = layout ? tab + slice * kCtx : tab + size_t(blockIdx.x) * kCtx * 64 + threadIdx.x;
    const size_t stride = layout ? 1 : 64;
    uint32_t x = uint32_t(slice) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (uint32_t i = 0; i < steps; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t ctx = (x >> 8) % kCtx;
        unsigned long long* p = base + size_t(ctx) * stride;
        unsigned long long v;
        if (FLAVOUR == 1) v = __builtin_nontemporal_load(p); else v = *p;
        x += uint32_t(v);  // the next context depends on what was read
        acc += v;
        if (FLAVOUR >= 1) __builtin_nontemporal_store(v + 1, p); else *p = v + 1;
    }
    if (acc == 0x123456789ull) sink[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t waves = argc > 1 ? atoi(argv[1]) : 1530;
    const uint32_t steps = argc > 2 ? atoi(argv[2]) : 2048;
    const int alloc = argc > 3 ? atoi(argv[3]) : 0;
    const uint32_t layout = argc > 4 ? atoi(argv[4]) : 0;
    const int flavour = argc > 5 ? atoi(argv[5]) : 0;
    const size_t bytes = size_t(waves) * 64 * kCtx * 8;
    unsigned long long *tab = nullptr, *sink = nullptr;
    if (alloc == 0) CK(hipMalloc(&tab, bytes));
    else CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&tab), bytes,
                                  alloc == 1 ? hipDeviceMallocUncached : (alloc == 2 ? hipDeviceMallocFinegrained : hipDeviceMallocContiguous)));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(tab, 0, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        if (flavour == 0) k_rmw<0><<<waves, 64>>>(tab, steps, layout, sink);
        else if (flavour == 1) k_rmw<1><<<waves, 64>>>(tab, steps, layout, sink);
        else k_rmw<2><<<waves, 64>>>(tab, steps, layout, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double acc = double(waves) * 64 * steps;
    printf("waves %u steps %u alloc %d layout %u flavour %d table %.2f GB : %.3f ms  %.2f G rmw/s  (%.0f ns per step of a wave)\n", waves,
           steps, alloc, layout, flavour, bytes / 1e9, best, acc / best / 1e6, best * 1e6 / steps);
    return 0;
}
