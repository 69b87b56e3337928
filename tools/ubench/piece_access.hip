// piece_access.hip -- what do the hand-overs between "one workgroup per slice" and "one lane per slice" kernels cost in HBM terms?
// (round 4: layout of the snapshot pass, snapshot_kernels.hip)   hipcc -O3 --offload-arch=gfx950 -o piece_access piece_access.hip
//   G lane groups of 64 slices, N elements of 8 bytes per slice.
//   layouts: PIECE  [group][element / 4][lane][4 x 8 B]  (32-byte pieces, a lane's pieces 2 KB apart)
//            SLICE  [slice][element]                      (a slice's 32 KB contiguous)
//   accesses: wg_*    one workgroup of 256 threads per slice moves the slice's N elements (16-byte vector accesses)
//             lane_*  one lane per slice, 64 slices per wavefront, front to back; piece layout: 8 B per lane and step;
//                     slice layout: bursts of 128 B per lane (8 x 16 B back to back)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int N = 4096;

__device__ __forceinline__ void block_slice(int xcd_aware, int& group, int& lane) {
    if (xcd_aware) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; group = ((slot >> 6) << 3) + xcd; lane = slot & 63; }
    else { group = blockIdx.x >> 6; lane = blockIdx.x & 63; }
}
__global__ __launch_bounds__(256) void wg_piece_write(uint4* a, int xcd_aware) {
    int group, lane; block_slice(xcd_aware, group, lane);
    uint4* base = a + (size_t(group) * N * 64 * 8 + lane * 32) / 16;
    for (int q = threadIdx.x; q < N / 4; q += 256) { uint4* o = base + size_t(q) * 128; o[0] = make_uint4(q, 1, 2, 3); o[1] = make_uint4(q, 5, 6, 7); }
}
__global__ __launch_bounds__(256) void wg_piece_read(const uint4* a, uint32_t* sink, int xcd_aware) {
    int group, lane; block_slice(xcd_aware, group, lane);
    const uint4* base = a + (size_t(group) * N * 64 * 8 + lane * 32) / 16;
    uint32_t acc = 0;
    for (int q = threadIdx.x; q < N / 4; q += 256) { const uint4* o = base + size_t(q) * 128; const uint4 x = o[0], y = o[1]; acc += x.x + y.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void wg_slice_write(uint4* a) {
    uint4* base = a + size_t(blockIdx.x) * N * 8 / 16;
    for (int q = threadIdx.x; q < N / 2; q += 256) base[q] = make_uint4(q, 1, 2, 3);
}
__global__ __launch_bounds__(256) void wg_slice_read(const uint4* a, uint32_t* sink) {
    const uint4* base = a + size_t(blockIdx.x) * N * 8 / 16;
    uint32_t acc = 0;
    for (int q = threadIdx.x; q < N / 2; q += 256) acc += base[q].x;
    if (acc == 0x12345678u) sink[0] = acc;
}
// one lane per slice; `work` dependent VALU steps per element stand for the serial kernel's arithmetic
__global__ __launch_bounds__(64) void lane_piece_read(const uint2* a, uint32_t* sink, int work) {
    const uint2* base = a + (size_t(blockIdx.x) * N * 64 * 8 + threadIdx.x * 32) / 8;
    uint32_t acc = 1;
    uint2 v0 = base[0], v1 = base[1];
    for (int i = 0; i < N; ++i) {
        const uint2 v = v0; v0 = v1;
        const int j = min(i + 2, N - 1);
        v1 = base[size_t(j >> 2) * 256 + (j & 3)];
        acc += v.x;
        for (int w = 0; w < work; ++w) acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void lane_piece_write(uint2* a, int work) {
    uint2* base = a + (size_t(blockIdx.x) * N * 64 * 8 + threadIdx.x * 32) / 8;
    uint32_t acc = threadIdx.x;
    for (int i = 0; i < N; ++i) {
        for (int w = 0; w < work; ++w) acc = acc * 1664525u + 1013904223u;
        base[size_t(i >> 2) * 256 + (i & 3)] = make_uint2(acc, i);
    }
}
__global__ __launch_bounds__(64) void lane_burst_read(const uint4* a, uint32_t* sink, int work) {
    const uint4* base = a + (size_t(blockIdx.x) * 64 + threadIdx.x) * N * 8 / 16;
    uint32_t acc = 1;
    uint4 buf[8], nxt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) nxt[k] = base[k];
    for (int i = 0; i < N; i += 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) buf[k] = nxt[k];
        const int j = min(i + 16, N - 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) nxt[k] = base[j / 2 + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc += buf[k].x;
            for (int w = 0; w < work; ++w) acc = acc * 1664525u + 1013904223u;
            acc += buf[k].z;
            for (int w = 0; w < work; ++w) acc = acc * 1664525u + 1013904223u;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(64) void lane_burst_write(uint4* a, int work) {
    uint4* base = a + (size_t(blockIdx.x) * 64 + threadIdx.x) * N * 8 / 16;
    uint32_t acc = threadIdx.x;
    for (int i = 0; i < N; i += 16) {
        uint4 buf[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            for (int w = 0; w < 2 * work; ++w) acc = acc * 1664525u + 1013904223u;
            buf[k] = make_uint4(acc, i, k, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) base[i / 2 + k] = buf[k];
    }
}
template <typename F> float timed(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 1536;  // lane groups (1530 = 16 frames 4K in 64x64 planes)
    const int work = argc > 2 ? atoi(argv[2]) : 8;
    const size_t bytes = size_t(G) * 64 * N * 8;
    uint4* a; uint32_t* sink; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(a, 1, bytes));
    const double gb = bytes / 1e9;
    printf("%d lane groups x 64 slices x %d elements x 8 B = %.2f GB; %d dependent VALU steps per element in the lane kernels\n", G, N, gb, work);
    for (int x = 0; x < 2; ++x) {
        float ms = timed([&] { wg_piece_write<<<G * 64, 256>>>(a, x); });
        printf("wg_piece_write  xcd_aware=%d  %.3f ms  %.2f TB/s\n", x, ms, gb / ms);
        ms = timed([&] { wg_piece_read<<<G * 64, 256>>>(a, sink, x); });
        printf("wg_piece_read   xcd_aware=%d  %.3f ms  %.2f TB/s\n", x, ms, gb / ms);
    }
    float ms = timed([&] { wg_slice_write<<<G * 64, 256>>>(a); });
    printf("wg_slice_write               %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timed([&] { wg_slice_read<<<G * 64, 256>>>(a, sink); });
    printf("wg_slice_read                %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timed([&] { lane_piece_read<<<G, 64>>>(reinterpret_cast<const uint2*>(a), sink, work); });
    printf("lane_piece_read              %.3f ms  %.2f TB/s  %.0f ns per element\n", ms, gb / ms, ms * 1e6 / N);
    ms = timed([&] { lane_piece_write<<<G, 64>>>(reinterpret_cast<uint2*>(a), work); });
    printf("lane_piece_write             %.3f ms  %.2f TB/s  %.0f ns per element\n", ms, gb / ms, ms * 1e6 / N);
    ms = timed([&] { lane_burst_read<<<G, 64>>>(a, sink, work); });
    printf("lane_burst_read              %.3f ms  %.2f TB/s  %.0f ns per element\n", ms, gb / ms, ms * 1e6 / N);
    ms = timed([&] { lane_burst_write<<<G, 64>>>(a, work); });
    printf("lane_burst_write             %.3f ms  %.2f TB/s  %.0f ns per element\n", ms, gb / ms, ms * 1e6 / N);
    return 0;
}
