// free_wipe.hip -- library-free reproducer for the corruption round 2 saw behind hipFree (devmem.hip).
//
// What failed then (tests/test_gpu_stress.py, ~400 shapes in one process): a coding lane is a private non-blocking HIP
// stream plus a handful of device buffers; when the lane cache is full the oldest lane is destroyed (stream drained,
// buffers hipFree'd, stream destroyed) and a new lane takes its buffers with hipMalloc right afterwards.  In a buffer
// obtained that way, whole 128-byte lines that a kernel had just written read back as ZEROS ~0.1 s later from another
// kernel on the same stream.  This program does exactly that and nothing else -- no library, no torch:
//
//   loop: [cache full -> drain + hipFree + hipStreamDestroy the oldest lane]  ->  new lane: non-blocking stream + hipMalloc
//         -> k_fill writes a never-zero pattern into the lane's big buffer  -> k_scan (ONE wavefront, deliberately slow:
//         tens of milliseconds) reads it back line by line and counts lines that differ / are all zero
//         -> count to pinned host memory, stream drained, verdict per iteration.
//
//   free_wipe [iterations=300] [mode: 0 churn with hipFree (the failing sequence) | 1 never free (control)] [seed]
//
// Prints the number of iterations in which damaged lines were seen.  Build: hipcc -O2 --offload-arch=gfx950 -o free_wipe free_wipe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                         \
        }                                                                                         \
    } while (0)

__device__ __forceinline__ uint32_t pattern(uint64_t i, uint32_t salt) {
    uint32_t x = uint32_t(i) * 2654435761u + salt;
    x ^= x >> 15;
    x *= 0x85EBCA6Bu;
    return x | 1u;  // never zero
}

// many workgroups, coalesced: what the transposition kernel did to the lane-order array
__global__ void k_fill(uint32_t* __restrict__ p, uint64_t n_words, uint32_t salt) {
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_words; i += uint64_t(gridDim.x) * blockDim.x)
        p[i] = pattern(i, salt);
}

// ONE wavefront walking the buffer, 256 bytes per step (what the slice encoder did, one lane group per wavefront): slow
// on purpose, so that the buffer is read long after it was written.  bad[0] = words that differ, bad[1] = 128-byte lines
// that are entirely zero, bad[2] = index of the first such line.
__global__ __launch_bounds__(64) void k_scan(const uint32_t* __restrict__ p, uint64_t n_words, uint32_t salt, uint32_t spin,
                                             unsigned long long* __restrict__ bad) {
    unsigned long long diff = 0, zero_lines = 0, first = ~0ull;
    for (uint64_t base = 0; base < n_words; base += 64) {
        const uint64_t i = base + threadIdx.x;
        const uint32_t v = i < n_words ? p[i] : 1u;
        if (i < n_words && v != pattern(i, salt)) ++diff;
        // lanes 0..31 and 32..63 each cover one 128-byte line
        const unsigned long long z = __ballot(v == 0);
        if (threadIdx.x == 0) {
            if ((z & 0xFFFFFFFFull) == 0xFFFFFFFFull) { ++zero_lines; if (first == ~0ull) first = base / 32; }
            if ((z >> 32) == 0xFFFFFFFFull) { ++zero_lines; if (first == ~0ull) first = base / 32 + 1; }
        }
        for (uint32_t s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
    }
    for (int o = 32; o > 0; o >>= 1) diff += __shfl_down(diff, o);
    if (threadIdx.x == 0) {
        bad[0] = diff;
        bad[1] = zero_lines;
        bad[2] = first;
    }
}

struct Lane {
    hipStream_t stream = nullptr;
    std::vector<void*> bufs;
    std::vector<size_t> sizes;
};

static uint32_t rng_state = 12345;
static uint32_t rnd() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 300;
    const int mode = argc > 2 ? std::atoi(argv[2]) : 0;
    if (argc > 3) rng_state = uint32_t(std::atoi(argv[3]));
    CHECK(hipSetDevice(0));
    unsigned long long* h_bad = nullptr;
    CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_bad), 3 * sizeof(unsigned long long), 0));
    std::deque<Lane> lanes;
    std::vector<Lane> never_freed;
    int bad_iters = 0;
    unsigned long long total_zero_lines = 0, total_diff = 0;
    for (int it = 0; it < iters; ++it) {
        if (lanes.size() >= 4) {  // the lane cache is full: the oldest lane goes
            Lane old = lanes.front();
            lanes.pop_front();
            CHECK(hipStreamSynchronize(old.stream));
            if (mode == 0) {
                for (void* b : old.bufs) CHECK(hipFree(b));
                CHECK(hipStreamDestroy(old.stream));
            } else {
                never_freed.push_back(old);  // control: nothing goes back to the driver (~150 MB per iteration stay allocated)
            }
        }
        // a new lane for a new "shape": six buffers like a codec workspace (symbols, lane order, scratch, small tables)
        Lane l;
        CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        const size_t big = (size_t(1 + rnd() % 40) << 20) + (size_t(rnd() % 4096) << 4);  // 1..41 MB, odd sizes
        const size_t szs[6] = {big, big * 2, big / 2 + 4096, (size_t(rnd() % 64) + 1) << 12, 4096, 64};
        for (size_t s : szs) {
            void* p = nullptr;
            CHECK(hipMalloc(&p, s));
            l.bufs.push_back(p);
            l.sizes.push_back(s);
        }
        const uint32_t salt = 0x9E3779B9u * uint32_t(it + 1);
        const uint64_t n_words = l.sizes[0] / 4;
        CHECK(hipMemsetAsync(l.bufs[1], 0, l.sizes[1], l.stream));  // like the per-call state-table clear
        hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, l.stream, static_cast<uint32_t*>(l.bufs[0]), n_words, salt);
        // ~40..120 ms of single-wavefront reading, depending on the size
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), 0, l.stream, static_cast<const uint32_t*>(l.bufs[0]), n_words, salt, 2u,
                           reinterpret_cast<unsigned long long*>(l.bufs[4]));
        CHECK(hipGetLastError());
        CHECK(hipMemcpyAsync(h_bad, l.bufs[4], 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, l.stream));
        CHECK(hipStreamSynchronize(l.stream));
        if (h_bad[0] || h_bad[1]) {
            ++bad_iters;
            total_zero_lines += h_bad[1];
            total_diff += h_bad[0];
            std::printf("iteration %d: %llu words differ, %llu all-zero 128-byte lines (first at line %llu) in a %zu-byte buffer\n", it, h_bad[0], h_bad[1],
                        h_bad[2], l.sizes[0]);
        }
        lanes.push_back(l);
        if ((it + 1) % 50 == 0) {
            std::printf("... %d iterations, %d with damage\n", it + 1, bad_iters);
            std::fflush(stdout);
        }
    }
    std::printf("free_wipe: mode %d (%s), %d iterations: %d with damaged lines (%llu zero lines, %llu differing words)\n", mode,
                mode == 0 ? "hipFree + hipMalloc churn" : "never free (control)", iters, bad_iters, total_zero_lines, total_diff);
    std::printf("verdict: %s\n", bad_iters ? "REPRODUCED without the library" : "not reproduced");
    return 0;
}
