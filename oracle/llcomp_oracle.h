/* oracle/llcomp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-thread CPU restatement of llcomp's predictive-coding path (reference:
 * /root/reference/llcomp.hpp), plus the sliced container this project adds.  It exists to CHECK the
 * HIP product path and to be timed as bench.py's `cpu_baseline` ("port").  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under llcomp_amd/ may.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks it against tests/golden/ (vectors
 * produced by the real reference compiled from /root/reference by oracle/Makefile -> oracle/_ref,
 * generator script oracle/gen_golden.py) and, when oracle/_ref is present, against the real reference
 * live on random inputs (tests/test_oracle_vs_ref.py).
 */
#ifndef LLCOMP_ORACLE_H
#define LLCOMP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORC_OK = 0,
    ORC_BAD_MAGIC = 1,     /* reference: "Invalid magic number" llcomp.hpp:465-467 */
    ORC_BAD_EXPONENT = 2,  /* reference: "Invalid exponent"     llcomp.hpp:232-234 */
    ORC_TRUNCATED = 3,     /* header / slice table longer than the data (shim-level check) */
    ORC_BAD_ARGS = 4,
    ORC_NOMEM = 5
};

#define ORC_MAGIC_LEGACY 0x79 /* llcomp.hpp:19-20: 0x77 + revision(2) */
#define ORC_MAGIC_SLICED 0x9C /* this project's sliced container (DESIGN.md "Container") */
#define ORC_N_CTX 7926        /* reachable contexts 0..7925 */

/* primitives (llcomp.hpp:335-356, 283-293) */
int orc_quant11(int x);
int orc_quant5(int x);
int orc_median(int a, int b, int c);
int orc_state_p(int s);
int orc_state_next(int s, int bit);

/* colour transform, llcomp.hpp:396-414 (forward) / 532-543 (inverse, with the c<3 intent fix D2) */
void orc_forward_rct(const uint8_t* px, long npix, int c, int16_t* out);
void orc_inverse_rct(const int16_t* s, long npix, int c, uint8_t* px);

/* Stage-A intermediate: folded context (>=0) and folded residual per sample of a rect, in coding
 * order (row -> pixel -> channel).  sample(x,y,k) = base[y*row_stride + x*pix_stride + k]. */
void orc_model_rect(const int16_t* base, long row_stride, int pix_stride, int nch, int tw, int th,
                    uint16_t* ctx_out, int16_t* res_out);

/* One bare range-coder stream (no header) for a rect with fresh state and rect-local borders
 * (llcomp.hpp:390-449).  *out is malloc'd; returns length or -1. */
long orc_encode_rect(const int16_t* base, long row_stride, int pix_stride, int nch, int tw, int th,
                     uint8_t** out);
/* Inverse (llcomp.hpp:486-530): writes reconstructed int16 samples into the rect. */
int orc_decode_rect(const uint8_t* data, size_t len, int16_t* base, long row_stride, int pix_stride,
                    int nch, int tw, int th);

/* Legacy whole-image format, byte-identical to llcomp::compressImage wherever that is defined. */
long orc_compress_image(const uint8_t* px, int w, int h, int c, uint8_t** out);
/* Sliced container.  tile_w/tile_h: 0 = full extent.  planar: one slice per channel plane. */
long orc_compress_sliced(const uint8_t* px, int w, int h, int c, int tile_w, int tile_h, int planar,
                         uint8_t** out);
/* Decodes either format (dispatch on magic).  *px malloc'd. */
int orc_decompress(const uint8_t* data, size_t len, uint8_t** px, int* w, int* h, int* c);

/* Bitstream variant of a reference built with LargeModel = false (llcomp.hpp:21, 427-429).  Process-wide, tests only. */
void orc_set_small_model(int on);

/* slice bookkeeping shared by tests */
long orc_slice_count(int w, int h, int c, int tile_w, int tile_h, int planar);
uint64_t orc_fnv1a64(const uint8_t* p, size_t n);
/* coverage counters of the encoder since the last reset: carries that travelled through a run of undecided 0xFF
   bytes (hpp:49-50 with outstanding_count > 0) and the longest such run.  Not thread-safe; tests only. */
void orc_carry_stats(long* runs, long* longest, int reset);
void orc_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
