// valu_rate3.hip -- issue cost (cycles per wavefront-instruction per SIMD at 1 / 4 / 8 wavefronts per SIMD, nominal 2.4 GHz)
// of the instruction forms the hand-written encoder / decoder blocks choose between (gfx950).  Same method as
// valu_rate2.hip: 128 independent instructions per loop iteration over 8 registers, whole GPU filled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(TXT, ...) { _Pragma("unroll") for (int u = 0; u < 16; ++u) { REP8(TXT) } }

template <int KIND>
__global__ void k(unsigned* out, int iters, unsigned seed) {
    unsigned r[8];
    unsigned long long q[4];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 7u + i + seed;
    for (int i = 0; i < 4; ++i) q[i] = (unsigned long long)(threadIdx.x * 11u + i + seed) << 20;
    unsigned m = seed | 1u, c8 = 8, zero = seed >> 31;
    unsigned long long sq = __builtin_amdgcn_readfirstlane(seed) * 0x100000001ull;
    unsigned sm = __builtin_amdgcn_readfirstlane(seed | 0x100u);
    __shared__ unsigned char lds[64 * 64];
    unsigned la = threadIdx.x * 36u;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(r[0]), "v"(m) : "vcc");
    for (int it = 0; it < iters; ++it) {
#define A0(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A1(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[i]) : "s"(sm));
#define A2(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(r[i]));
#define A3(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(m) : "vcc");
#define A4(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(m) : "vcc");
#define A5(i) asm volatile("v_lshlrev_b32 %0, 8, %0" : "+v"(r[i]));
#define A6(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(r[i]) : "s"(sm));
#define A7(i) asm volatile("v_lshrrev_b64 %0, 8, %0" : "+v"(q[i & 3]));
#define A8(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r[i]) : "v"(m));
#define A9(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(m), "s"(sm));
#define A10(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A11(i) asm volatile("v_mul_u32_u24_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(r[i]) : "v"(m));
#define A12(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(r[i]) : "v"(c8));
#define A13(i) asm volatile("v_lshlrev_b32_sdwa %0, 8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(r[i]));
#define A14(i) asm volatile("v_cmp_gt_u32 vcc, %1, %0" : : "v"(r[i]), "s"(sm) : "vcc");
#define A15(i) asm volatile("v_cmpx_le_u32 vcc, 0, %0" : : "v"(r[i]) : "vcc");
#define A16(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A17(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A18(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A19(i) asm volatile("v_ffbh_u32 %0, %0" : "+v"(r[i]));
#define A20(i) asm volatile("v_lshl_or_b32 %0, %0, 1, 1" : "+v"(r[i]));
#define A21(i) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(r[i]) : "v"(m));
#define A22(i) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(r[i]) : "v"(m));
#define A23(i) asm volatile("v_and_b32 %0, 0xffffff00, %0" : "+v"(r[i]));
#define A24(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(r[i]));
#define A25(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(r[i]) : "v"(m));
#define A26(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[i]) : "v"(m));
#define A27(i) asm volatile("v_lshrrev_b32 %0, 24, %0" : "+v"(r[i]));
#define A28(i) asm volatile("s_nop 0");
#define A29(i) asm volatile("s_mov_b64 s[20:21], exec" ::: "s20", "s21");
#define A30(i) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n s_mov_b64 exec, s[20:21]" ::: "s20", "s21");
#define A31(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));  /* under exec = 0, see below */
#define A32(i) asm volatile("ds_write_b8 %0, %1" : : "v"(la), "v"(r[i]) : "memory");
#define A33(i) asm volatile("ds_write_b8 %0, %1" : : "v"(la), "v"(r[i]) : "memory");  /* under exec = 0 */
#define A34(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(m));  /* under exec = 0 */
#define A35(i) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n s_cbranch_vccz 1f\n s_nop 0\n1:" : : "v"(r[i]), "s"(sm) : "vcc");
#define A36(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(m) : "vcc");
#define A37(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A38(i) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(r[i]));
#define A39(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "s"(sm));
#define A40(i) asm volatile("v_cndmask_b32_e64 %0, 0, 1, vcc" : "=v"(r[i]));
#define A41(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[i]) : "v"(zero));
#define A42(i) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(r[i]) : "v"(c8));
#define A43(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(q[i & 3]) : "v"(c8));
#define A44(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(m));
#define A45(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i & 3]) : "s"(sq));
#define A46(i) asm volatile("v_cmp_gt_u32_e64 s[20:21], %1, %0" : : "v"(r[i]), "s"(sm) : "s20", "s21");
#define A47(i) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(r[i]));
#define A48(i) asm volatile("s_mov_b64 exec, exec");
#define A49(i) asm volatile("s_or_b64 s[20:21], s[20:21], vcc" ::: "s20", "s21");
#define A50(i) asm volatile("v_cmp_gt_u64 vcc, %1, %0" : : "v"(q[i & 3]), "s"(sq) : "vcc");
#define A51(i) asm volatile("v_cmp_eq_u64 vcc, 0, %0" : : "v"(q[i & 3]) : "vcc");
#define A52(i) asm volatile("v_lshlrev_b64 %0, 8, %0" : "+v"(q[i & 3]));
#define CASE(N) else if (KIND == N) BODY(A##N)
        if (KIND == 31 || KIND == 33 || KIND == 34) {
            asm volatile("s_mov_b64 s[22:23], exec\n s_mov_b64 exec, 0" ::: "s22", "s23");
            if (KIND == 31) BODY(A31) else if (KIND == 33) BODY(A33) else BODY(A34)
            asm volatile("s_mov_b64 exec, s[22:23]");
        }
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
        CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23) CASE(24) CASE(25) CASE(26) CASE(27) CASE(28)
        CASE(29) CASE(30) CASE(32) CASE(35) CASE(36) CASE(37) CASE(38) CASE(39) CASE(40) CASE(41) CASE(42) CASE(43) CASE(44) CASE(45) CASE(46)
        CASE(47) CASE(48) CASE(49) CASE(50) CASE(51) CASE(52)
    }
    unsigned s = lds[threadIdx.x];
    for (int i = 0; i < 8; ++i) s ^= r[i];
    for (int i = 0; i < 4; ++i) s ^= unsigned(q[i]) ^ unsigned(q[i] >> 32);
    if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int KIND>
double run(int waves_per_simd, int iters) {
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
    return ms * 1e-3 * 2.4e9 / (double(iters) * 16 * 8 * waves_per_simd);
}

#define NAMES {"v_add_u32 vgpr", "v_add_u32 sgpr", "v_add_u32 literal", "v_add_co_u32", "v_addc_co_u32", "v_lshlrev imm", "v_lshlrev sgpr", \
        "v_lshrrev_b64", "v_alignbyte", "v_perm", "v_mul_u32_u24", "v_mul_u24_sdwa", "lshl_sdwa vgpr-amount", "lshl_sdwa inline-amount", "v_cmp sgpr", "v_cmpx", \
        "v_xor", "v_or", "v_max_i32", "v_ffbh", "v_lshl_or", "v_mov_sdwa", "v_sub_sdwa", "v_and literal", "v_cndmask 0,v,vcc", "v_mad_u32_u24", "v_lshl_add", \
        "v_lshrrev 24", "s_nop 0", "s_mov_b64 s,exec", "saveexec+restore (pair)", "v_add exec=0", "ds_write_b8", "ds_write_b8 exec=0", "v_mul_u24 exec=0", \
        "v_cmp+cbranch_vccz+nop (3)", "v_sub_co_u32", "v_subrev_u32", "v_bfe_u32 imm", "v_and sgpr", \
        "v_cndmask_e64 0,1,vcc", "v_cndmask vzero,v,vcc", "v_lshrrev vgpr-amount", "v_lshlrev_b64 vgpr-amount", "v_min_u32", "v_lshl_add_u64 sgpr", "v_cmp_e64 sgpr", \
        "v_ashrrev 31", "s_mov_b64 exec,exec", "s_or_b64 s,s,vcc", "v_cmp_gt_u64 sgpr", "v_cmp_eq_u64 0", "v_lshlrev_b64 imm"}
template <int K0>
void run_one(int kind, int w, const char* const* names) {
    if constexpr (K0 < 53) {
        if (kind == K0) { printf("  %-28s %.2f\n", names[K0], run<K0>(w, 300)); fflush(stdout); return; }
        run_one<K0 + 1>(kind, w, names);
    }
}
int main(int argc, char** argv) {
    static const char* names[] = NAMES;
    const int w = 8;
    printf("waves/SIMD=%d\n", w);
    for (int a = 1; a < argc; ++a) run_one<0>(atoi(argv[a]), w, names);
    return 0;
}
