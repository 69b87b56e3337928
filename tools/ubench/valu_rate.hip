// valu_rate.hip -- how many cycles does one wave64 integer VALU instruction occupy a gfx950 SIMD?
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip ; run on the GPU box.
// Each kernel runs N iterations of 8 independent chains of one instruction kind; with W waves per SIMD on every SIMD
// the time gives wave-instructions per SIMD-cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ void k(unsigned* out, int iters, unsigned seed) {
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7u + i + seed;
    unsigned m = seed | 1u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : );
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(i) asm volatile("v_lshl_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(m) : "vcc");
                REP8(X)
#undef X
            } else if (KIND == 8) {  // dependent chain on ONE register
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r[0]) : "v"(m));
            } else if (KIND == 9) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(r[i]) : "v"(m) : );
                REP8(X)
#undef X
            } else if (KIND == 10) {
#define X(i) asm volatile("s_and_b64 s[12:13], s[10:11], exec" ::: "s12", "s13");
                REP8(X)
#undef X
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int KIND>
double run(int waves_per_simd, int iters) {
    unsigned* d;
    hipMalloc(&d, 4096);
    const int blocks = 256 * 4 * waves_per_simd;  // 64-thread blocks: one wave each
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 64>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<blocks, 64>>>(d, iters, 1);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipFree(d);
    const double insts_per_wave = double(iters) * 16 * 8;
    const double simd_cycles = ms * 1e-3 * 2.4e9;  // nominal 2.4 GHz
    return simd_cycles / (insts_per_wave * waves_per_simd);  // SIMD cycles per wave-instruction at nominal clock
}

int main() {
    const char* names[] = {"v_add_u32", "v_mul_u32_u24", "v_cndmask(vcc)", "v_perm_b32", "v_alignbit_b32", "v_lshl_or_b32", "v_mul_u24_sdwa", "v_sub_co_u32",
                           "v_add dependent chain", "v_cndmask_e64(sgpr mask)", "s_and_b64"};
    for (int w : {1, 2, 4, 8}) {
        printf("waves/SIMD=%d :", w);
        printf(" %s=%.2f", names[0], run<0>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[1], run<1>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[2], run<2>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[3], run<3>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[4], run<4>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[5], run<5>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[6], run<6>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[7], run<7>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[8], run<8>(w, 2000)); fflush(stdout);
        printf(" %s=%.2f", names[9], run<9>(w, 2000)); fflush(stdout);
        printf("\n");
    }
    return 0;
}
