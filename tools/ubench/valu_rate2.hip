// valu_rate2.hip -- issue cost of the integer VALU forms the slice kernels use (gfx950).  See valu_rate.hip.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KCASE(ID, ASM, CLOB) \
    else if (KIND == ID) { _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(ASM) } }

template <int KIND>
__global__ void k(unsigned* out, int iters, unsigned seed) {
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7u + i + seed;
    unsigned m = seed | 1u, c8 = 8;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r[0]), "v"(m) : "vcc");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#define A(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 1) {
#define A(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 2) {
#define A(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r[i]) : "v"(c8));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 3) {
#define A(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 4) {
#define A(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 5) {
#define A(i) asm volatile("v_bfe_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c8));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 6) {
#define A(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 7) {
#define A(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 8) {
#define A(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(m) : "vcc");
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 9) {
#define A(i) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1" : : "v"(r[i]), "v"(m) : "s20", "s21");
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 10) {
#define A(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : "vcc");
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 11) {
#define A(i) asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(m) : "s20", "s21");
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 12) {
#define A(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 13) {
#define A(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(r[i]) : "v"(c8));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 14) {
#define A(i) asm volatile("v_ashrrev_i32 %0, 8, %0" : "+v"(r[i]));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 15) {
#define A(i) asm volatile("v_subbrev_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r[i]) : : "vcc");
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        } else if (KIND == 16) {
#define A(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(m));
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(A) }
#undef A
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= r[i];
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int KIND>
double run(int waves_per_simd, int iters, int per_asm) {
    unsigned* d;
    hipMalloc(&d, 4096);
    const int blocks = 256 * 4 * waves_per_simd;
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
    return ms * 1e-3 * 2.4e9 / (double(iters) * 16 * 8 * per_asm * waves_per_simd);
}

int main() {
    for (int w : {1, 4, 8}) {
        printf("waves/SIMD=%d :", w);
#define R(ID, NAME, PER) printf(" %s=%.2f", NAME, run<ID>(w, 1500, PER)); fflush(stdout);
        R(0, "cndmask_vcc", 1) R(1, "and", 1) R(2, "lshlrev", 1) R(3, "sub", 1) R(4, "min_u32", 1) R(5, "bfe_u32", 1) R(6, "and_or", 1) R(7, "bfi", 1)
        R(8, "cmp_vcc", 1) R(9, "cmp_e64", 1) R(10, "cmp+cndmask_vcc(pair)", 2) R(11, "cmp+cndmask_e64(pair)", 2) R(12, "add3", 1) R(13, "lshl_sdwa", 1) R(14, "ashr_imm", 1)
        R(15, "subbrev_vcc", 1) R(16, "mov", 1)
        printf("\n");
    }
    return 0;
}
