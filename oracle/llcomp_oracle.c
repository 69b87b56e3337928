/* oracle/llcomp_oracle.c -- TEST INFRASTRUCTURE ONLY (see llcomp_oracle.h header comment).
 *
 * CPU restatement, in plain C, of the algorithm in /root/reference/llcomp.hpp.  Every function cites
 * the reference lines it follows.  Parity: PINNED by tests/golden (made by the real reference).
 *
 * The adaptive-state tables are kept in "pair" form: state s = 2*k + mps, k = confidence level 0..63.
 *   P(bit==1)*256 = mps ? 254 - kLpsProb[k] : kLpsProb[k]          (== stateProbability, hpp:270-281)
 *   bit == mps : k -> min(k+1, 63)                                 (== nextStateMps,     hpp:252-259)
 *   bit != mps : k == 0 ? flip mps : k -> kLpsFall[k]              (== nextStateLps,     hpp:261-268)
 * tests/test_oracle_vs_ref.py compares all 128 states x 2 bits with the reference's own State class.
 */
#include "llcomp_oracle.h"

#include <stdlib.h>
#include <string.h>

static const uint8_t kLpsProb[64] = {
    123, 117, 111, 106, 101, 96, 91, 87, 83, 79, 75, 72, 68, 66, 63, 60, 57, 54, 52, 49, 48, 45,
    43,  41,  40,  38,  36,  35, 33, 32, 30, 30, 28, 27, 26, 25, 24, 23, 22, 21, 21, 20, 19, 18,
    18,  17,  17,  16,  16,  15, 15, 14, 14, 13, 13, 13, 12, 12, 12, 11, 11, 11, 11, 7};
static const uint8_t kLpsFall[64] = {
    0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16,
    18, 18, 19, 19, 21, 21, 22, 22, 23, 24, 24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30,
    31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 38, 38, 38, 38, 39};

/* ---- primitives ----------------------------------------------------------------------------- */

/* hpp:316-337: 256-entry table indexed by clamp(x,-128,127)&0xFF; here in closed form. */
int orc_quant11(int x) {
    int a = x < 0 ? -x : x, q;
    q = a == 0 ? 0 : a == 1 ? 1 : a <= 4 ? 2 : a <= 11 ? 3 : a <= 34 ? 4 : 5;
    return x < 0 ? -q : q;
}
/* hpp:297-314, 339-341 */
int orc_quant5(int x) {
    int a = x < 0 ? -x : x, q;
    q = a == 0 ? 0 : a <= 3 ? 1 : 2;
    return x < 0 ? -q : q;
}
/* hpp:343-356: median of three */
int orc_median(int a, int b, int c) {
    int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
/* hpp:286-289 */
int orc_state_p(int s) {
    int k = s >> 1;
    return (s & 1) ? 254 - kLpsProb[k] : kLpsProb[k];
}
/* hpp:290-292 */
int orc_state_next(int s, int bit) {
    int k = s >> 1, mps = s & 1;
    if ((bit != 0) == mps) return 2 * (k < 63 ? k + 1 : 63) + mps;
    if (k == 0) return mps ^ 1;
    return 2 * kLpsFall[k] + mps;
}

/* flat tables for the coder loops */
static uint8_t g_p[128], g_next[128][2];
static int g_tables_ready;
static void tables_init(void) {
    if (g_tables_ready) return;
    for (int s = 0; s < 128; ++s) {
        g_p[s] = (uint8_t)orc_state_p(s);
        g_next[s][0] = (uint8_t)orc_state_next(s, 0);
        g_next[s][1] = (uint8_t)orc_state_next(s, 1);
    }
    g_tables_ready = 1;
}

/* ---- byte sink -------------------------------------------------------------------------------- */
typedef struct {
    uint8_t* p;
    size_t n, cap;
    int oom;
} sink_t;

static void sink_put(sink_t* s, uint8_t b) {
    if (s->n == s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 4096;
        uint8_t* q = (uint8_t*)realloc(s->p, nc);
        if (!q) { s->oom = 1; return; }
        s->p = q; s->cap = nc;
    }
    s->p[s->n++] = b;
}

/* ---- range encoder (hpp:33-89) ---------------------------------------------------------------- */
typedef struct {
    int low, range, held, pend;
    sink_t* out;
} renc_t;

static void renc_init(renc_t* e, sink_t* out) {  /* hpp:35 */
    e->low = 0; e->range = 0xFF00; e->held = -1; e->pend = 0; e->out = out;
}
/* test coverage only: how often a carry had to travel through a run of undecided 0xFF bytes, and the longest run */
static long g_carry_runs, g_carry_longest;
void orc_carry_stats(long* runs, long* longest, int reset) {
    if (runs) *runs = g_carry_runs;
    if (longest) *longest = g_carry_longest;
    if (reset) g_carry_runs = g_carry_longest = 0;
}
static void renc_renorm(renc_t* e) {  /* hpp:38-58 */
    while (e->range < 0x100) {
        if (e->held < 0) {
            e->held = e->low >> 8;
        } else if (e->low <= 0xFF00) {
            sink_put(e->out, (uint8_t)e->held);
            for (; e->pend; e->pend--) sink_put(e->out, 0xFF);
            e->held = e->low >> 8;
        } else if (e->low >= 0x10000) {
            if (e->pend) {
                g_carry_runs++;
                if (e->pend > g_carry_longest) g_carry_longest = e->pend;
            }
            sink_put(e->out, (uint8_t)(e->held + 1));
            for (; e->pend; e->pend--) sink_put(e->out, 0x00);
            e->held = (e->low >> 8) & 0xFF;
        } else {
            e->pend++;
        }
        e->low = (e->low & 0xFF) << 8;
        e->range <<= 8;
    }
}
static inline void renc_put(renc_t* e, int bit, int p) {  /* hpp:60-73 */
    int r1 = (e->range * p) >> 8;
    if (!bit) {
        e->range -= r1;
    } else {
        e->low += e->range - r1;
        e->range = r1;
    }
    if (e->range < 0x100) renc_renorm(e);
}
static void renc_finish(renc_t* e) {  /* hpp:75-81 */
    e->range = 0xFF; e->low += 0xFF; renc_renorm(e);
    e->range = 0xFF; renc_renorm(e);
}

/* ---- range decoder (hpp:91-127) --------------------------------------------------------------- */
typedef struct {
    int low, range;
    const uint8_t* p;
    size_t pos, len;
} rdec_t;
static inline int rdec_byte(rdec_t* d) {  /* hpp:475-479: reads past the end give 0 */
    return d->pos < d->len ? d->p[d->pos++] : 0;
}
static void rdec_init(rdec_t* d, const uint8_t* p, size_t len) {  /* hpp:93-96 */
    d->p = p; d->pos = 0; d->len = len; d->range = 0xFF00;
    d->low = rdec_byte(d) << 8;
    d->low |= rdec_byte(d);
}
static inline int rdec_get(rdec_t* d, int p) {  /* hpp:98-121 */
    int r1 = (d->range * p) >> 8, bit;
    d->range -= r1;
    if (d->low < d->range) {
        bit = 0;
    } else {
        d->low -= d->range; d->range = r1; bit = 1;
    }
    if (d->range < 0x100) {
        d->range <<= 8;
        d->low = (d->low << 8) + rdec_byte(d);
    }
    return bit;
}

/* ---- binarisation + adaptive states (hpp:166-206, 219-247, 439-444, 517-523) ------------------ */
static inline void code_bin(renc_t* e, uint8_t* bank, int slot, int bit) {
    uint8_t s = bank[slot];
    renc_put(e, bit, g_p[s]);
    bank[slot] = g_next[s][bit];
}
static inline int read_bin(rdec_t* d, uint8_t* bank, int slot) {
    uint8_t s = bank[slot];
    int bit = rdec_get(d, g_p[s]);
    bank[slot] = g_next[s][bit];
    return bit;
}
/* putSymbol<true,4,6,7>: zero flag slot 0; unary exponent slots 1..4 (saturating); mantissa below the
 * leading one MSB-first slots 5..6 (saturating); sign slot 7. */
static inline void put_residual(renc_t* e, uint8_t* bank, int v) {
    if (v == 0) { code_bin(e, bank, 0, 1); return; }
    unsigned a = (unsigned)(v < 0 ? -v : v);
    int ex = 31 - __builtin_clz(a);
    code_bin(e, bank, 0, 0);
    for (int i = 0; i < ex; ++i) code_bin(e, bank, 1 + i < 4 ? 1 + i : 4, 1);
    code_bin(e, bank, 1 + ex < 4 ? 1 + ex : 4, 0);
    for (int i = ex - 1, j = 0; i >= 0; --i, ++j) code_bin(e, bank, 5 + j < 6 ? 5 + j : 6, (a >> i) & 1);
    code_bin(e, bank, 7, v < 0);
}
/* getSymbol<true,4,6,7>; returns ORC_BAD_EXPONENT when the unary run exceeds 31 (hpp:230-235).
 * Arithmetic is done modulo 2^32 so damaged streams behave identically on CPU and GPU. */
static inline int get_residual(rdec_t* d, uint8_t* bank, int* out) {
    if (read_bin(d, bank, 0)) { *out = 0; return ORC_OK; }
    int ex = 0;
    while (read_bin(d, bank, 1 + ex < 4 ? 1 + ex : 4)) {
        if (++ex > 31) return ORC_BAD_EXPONENT;
    }
    uint32_t v = 1;
    for (int j = 0; j < ex; ++j) v += v + (uint32_t)read_bin(d, bank, 5 + j < 6 ? 5 + j : 6);
    if (read_bin(d, bank, 7)) v = 0u - v;
    *out = (int)v;
    return ORC_OK;
}

/* ---- colour transform --------------------------------------------------------------------------- */
void orc_forward_rct(const uint8_t* px, long npix, int c, int16_t* out) {  /* hpp:396-414 */
    for (long i = 0; i < npix; ++i, px += c, out += c) {
        if (c >= 3) {
            int g = px[1], cb = px[2] - g, cr = px[0] - g;
            out[0] = (int16_t)cr;
            out[1] = (int16_t)(g + (cb + cr) / 4); /* C division truncates toward zero, as hpp:402 */
            out[2] = (int16_t)cb;
            for (int k = 3; k < c; ++k) out[k] = px[k];
        } else {
            for (int k = 0; k < c; ++k) out[k] = px[k];
        }
    }
}
static inline uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
void orc_inverse_rct(const int16_t* s, long npix, int c, uint8_t* px) {  /* hpp:532-543 */
    for (long i = 0; i < npix; ++i, px += c, s += c) {
        if (c >= 3) {
            int r = s[0], g = s[1], b = s[2];
            g -= (r + b) / 4; r += g; b += g;
            px[0] = clamp255(r); px[1] = clamp255(g); px[2] = clamp255(b);
            for (int k = 3; k < c; ++k) px[k] = (uint8_t)s[k]; /* hpp:541-543: plain narrowing */
        } else {
            /* reference is broken here; evident intent = mirror of hpp:410-414 */
            for (int k = 0; k < c; ++k) px[k] = (uint8_t)s[k];
        }
    }
}

/* ---- neighbourhood, context, prediction (hpp:417-436) ------------------------------------------ */
typedef struct { int l, t, L, tl, tr, T; } hood_t;

static inline hood_t hood(const int16_t* base, long rs, int ps, int tw, int x, int y, int k) {
    const int16_t* p = base + (long)y * rs + (long)x * ps + k;
    hood_t n;
    n.l = x > 0 ? p[-ps] : (y > 0 ? p[-rs] : 128);
    n.t = y > 0 ? p[-rs] : n.l;
    n.L = x > 1 ? p[-2 * ps] : n.l;
    n.tl = (y > 0 && x > 0) ? p[-rs - ps] : n.t;
    n.tr = (y > 0 && x < tw - 1) ? p[-rs + ps] : n.t;
    n.T = y > 1 ? p[-2 * rs] : n.t;
    return n;
}
/* hpp:21 `LargeModel`: a build-time constant of the reference (true as shipped).  With false the two quant5 terms are
 * left out of the context (hpp:427-429).  Process-wide switch for the tests; the sliced container records it in bit 1
 * of its flags byte, the legacy header cannot. */
static int g_small_model;
void orc_set_small_model(int on) { g_small_model = on != 0; }
static inline int context_of(const hood_t* n) {  /* hpp:424-429, multipliers 1,11,121,605,3025 (D3) */
    const int h3 = orc_quant11(n->l - n->tl) + 11 * orc_quant11(n->tl - n->t) + 121 * orc_quant11(n->t - n->tr);
    if (g_small_model) return h3;
    return h3 + 605 * orc_quant5(n->L - n->l) + 3025 * orc_quant5(n->T - n->t);
}

void orc_model_rect(const int16_t* base, long rs, int ps, int nch, int tw, int th, uint16_t* ctx_out,
                    int16_t* res_out) {
    size_t i = 0;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x)
            for (int k = 0; k < nch; ++k, ++i) {
                hood_t n = hood(base, rs, ps, tw, x, y, k);
                int ctx = context_of(&n);
                int res = base[(long)y * rs + (long)x * ps + k] - orc_median(n.l, n.l + n.t - n.tl, n.t);
                if (ctx < 0) { ctx = -ctx; res = -res; } /* hpp:433-436 */
                ctx_out[i] = (uint16_t)ctx;
                res_out[i] = (int16_t)res;
            }
}

long orc_encode_rect(const int16_t* base, long rs, int ps, int nch, int tw, int th, uint8_t** out) {
    tables_init();
    sink_t sk = {0, 0, 0, 0};
    renc_t e;
    renc_init(&e, &sk);
    uint8_t* table = (uint8_t*)calloc(ORC_N_CTX, 8); /* hpp:385, all states 0; 8 slots per context */
    if (!table) return -1;
    for (int y = 0; y < th; ++y)
        for (int x = 0; x < tw; ++x)
            for (int k = 0; k < nch; ++k) {
                hood_t n = hood(base, rs, ps, tw, x, y, k);
                int ctx = context_of(&n);
                int res = base[(long)y * rs + (long)x * ps + k] - orc_median(n.l, n.l + n.t - n.tl, n.t);
                if (ctx < 0) { ctx = -ctx; res = -res; }
                put_residual(&e, table + (size_t)ctx * 8, res);
            }
    renc_finish(&e);
    free(table);
    if (sk.oom) { free(sk.p); return -1; }
    if (!sk.p) sk.p = (uint8_t*)malloc(1);
    *out = sk.p;
    return (long)sk.n;
}

int orc_decode_rect(const uint8_t* data, size_t len, int16_t* base, long rs, int ps, int nch, int tw,
                    int th) {
    tables_init();
    rdec_t d;
    rdec_init(&d, data, len);
    uint8_t* table = (uint8_t*)calloc(ORC_N_CTX, 8);
    if (!table) return ORC_NOMEM;
    int rc = ORC_OK;
    for (int y = 0; y < th && rc == ORC_OK; ++y)
        for (int x = 0; x < tw && rc == ORC_OK; ++x)
            for (int k = 0; k < nch; ++k) {
                hood_t n = hood(base, rs, ps, tw, x, y, k);
                int ctx = context_of(&n), res, neg = 0;
                if (ctx < 0) { ctx = -ctx; neg = 1; } /* hpp:511-515 */
                rc = get_residual(&d, table + (size_t)ctx * 8, &res);
                if (rc != ORC_OK) break;
                if (neg) res = (int)(0u - (uint32_t)res);
                base[(long)y * rs + (long)x * ps + k] =
                    (int16_t)(uint32_t)((uint32_t)orc_median(n.l, n.l + n.t - n.tl, n.t) + (uint32_t)res);
            }
    free(table);
    return rc;
}

/* ---- legacy whole-image format (hpp:358-452, 461-547) ------------------------------------------ */
long orc_compress_image(const uint8_t* px, int w, int h, int c, uint8_t** out) {
    if (w <= 0 || h <= 0 || c <= 0 || c > 255 || w > 65535 || h > 65535) return -1;
    long npix = (long)w * h;
    int16_t* s = (int16_t*)malloc((size_t)npix * c * sizeof(int16_t));
    if (!s) return -1;
    orc_forward_rct(px, npix, c, s);
    uint8_t* body = 0;
    long n = orc_encode_rect(s, (long)w * c, c, c, w, h, &body);
    free(s);
    if (n < 0) return -1;
    uint8_t* o = (uint8_t*)malloc((size_t)n + 6);
    if (!o) { free(body); return -1; }
    o[0] = ORC_MAGIC_LEGACY; o[1] = (uint8_t)c; /* hpp:375-378 */
    o[2] = (uint8_t)(w & 0xFF); o[3] = (uint8_t)(w >> 8);
    o[4] = (uint8_t)(h & 0xFF); o[5] = (uint8_t)(h >> 8);
    memcpy(o + 6, body, (size_t)n);
    free(body);
    *out = o;
    return n + 6;
}

/* ---- sliced container (this project; DESIGN.md "Container") ------------------------------------ */
static void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

long orc_slice_count(int w, int h, int c, int tile_w, int tile_h, int planar) {
    if (tile_w <= 0 || tile_w > w) tile_w = w;
    if (tile_h <= 0 || tile_h > h) tile_h = h;
    long ntx = (w + tile_w - 1) / tile_w, nty = (h + tile_h - 1) / tile_h;
    return ntx * nty * (planar ? c : 1);
}

long orc_compress_sliced(const uint8_t* px, int w, int h, int c, int tile_w, int tile_h, int planar,
                         uint8_t** out) {
    if (w <= 0 || h <= 0 || c <= 0 || c > 255) return -1;
    if (tile_w <= 0 || tile_w > w) tile_w = w;
    if (tile_h <= 0 || tile_h > h) tile_h = h;
    planar = planar ? 1 : 0;
    long ntx = (w + tile_w - 1) / tile_w, nty = (h + tile_h - 1) / tile_h;
    long ns = ntx * nty * (planar ? c : 1), npix = (long)w * h;
    int16_t* s = (int16_t*)malloc((size_t)npix * c * sizeof(int16_t));
    uint8_t** body = (uint8_t**)calloc((size_t)ns, sizeof(uint8_t*));
    long* blen = (long*)calloc((size_t)ns, sizeof(long));
    if (!s || !body || !blen) { free(s); free(body); free(blen); return -1; }
    orc_forward_rct(px, npix, c, s);
    long total = 0, si = 0, fail = 0;
    for (long ty = 0; ty < nty; ++ty)
        for (long tx = 0; tx < ntx; ++tx) {
            int x0 = (int)(tx * tile_w), y0 = (int)(ty * tile_h);
            int tw = w - x0 < tile_w ? w - x0 : tile_w, th = h - y0 < tile_h ? h - y0 : tile_h;
            const int16_t* origin = s + ((long)y0 * w + x0) * c;
            if (planar) {
                for (int k = 0; k < c; ++k, ++si) {
                    blen[si] = orc_encode_rect(origin + k, (long)w * c, c, 1, tw, th, &body[si]);
                    if (blen[si] < 0) fail = 1; else total += blen[si];
                }
            } else {
                blen[si] = orc_encode_rect(origin, (long)w * c, c, c, tw, th, &body[si]);
                if (blen[si] < 0) fail = 1; else total += blen[si];
                ++si;
            }
        }
    free(s);
    uint8_t* o = fail ? 0 : (uint8_t*)malloc((size_t)(24 + 4 * ns + total));
    long ret = -1;
    if (o) {
        o[0] = ORC_MAGIC_SLICED; o[1] = 1; o[2] = (uint8_t)c; o[3] = (uint8_t)(planar | (g_small_model ? 2 : 0));
        put32(o + 4, (uint32_t)w); put32(o + 8, (uint32_t)h);
        put32(o + 12, (uint32_t)tile_w); put32(o + 16, (uint32_t)tile_h);
        put32(o + 20, (uint32_t)ns);
        uint8_t* q = o + 24 + 4 * ns;
        for (long i = 0; i < ns; ++i) {
            put32(o + 24 + 4 * i, (uint32_t)blen[i]);
            memcpy(q, body[i], (size_t)blen[i]);
            q += blen[i];
        }
        *out = o;
        ret = 24 + 4 * ns + total;
    }
    for (long i = 0; i < ns; ++i) free(body[i]);
    free(body); free(blen);
    return ret;
}

static int decompress_impl(const uint8_t* data, size_t len, uint8_t** px, int* pw, int* ph, int* pc);
int orc_decompress(const uint8_t* data, size_t len, uint8_t** px, int* pw, int* ph, int* pc) {
    /* a sliced container says in bit 1 of its flags byte which model wrote it; a legacy stream follows the switch */
    const int saved = g_small_model;
    if (len >= 4 && data[0] == ORC_MAGIC_SLICED) g_small_model = (data[3] >> 1) & 1;
    const int rc = decompress_impl(data, len, px, pw, ph, pc);
    g_small_model = saved;
    return rc;
}
static int decompress_impl(const uint8_t* data, size_t len, uint8_t** px, int* pw, int* ph, int* pc) {
    if (len < 1) return ORC_TRUNCATED;
    int w, h, c, tile_w, tile_h, planar;
    long ns;
    const uint8_t* lens = 0;
    const uint8_t* payload;
    size_t payload_len;
    if (data[0] == ORC_MAGIC_LEGACY) {  /* hpp:463-470 */
        if (len < 6) return ORC_TRUNCATED;
        c = data[1]; w = data[2] | (data[3] << 8); h = data[4] | (data[5] << 8);
        tile_w = w; tile_h = h; planar = 0; ns = 1;
        payload = data + 6; payload_len = len - 6;
    } else if (data[0] == ORC_MAGIC_SLICED) {
        if (len < 24) return ORC_TRUNCATED;
        if (data[1] != 1) return ORC_BAD_ARGS;
        c = data[2]; planar = data[3] & 1;
        w = (int)get32(data + 4); h = (int)get32(data + 8);
        tile_w = (int)get32(data + 12); tile_h = (int)get32(data + 16);
        ns = (long)get32(data + 20);
        if (w <= 0 || h <= 0 || c <= 0 || tile_w <= 0 || tile_h <= 0 || tile_w > w || tile_h > h) return ORC_BAD_ARGS;
        if (ns != orc_slice_count(w, h, c, tile_w, tile_h, planar)) return ORC_BAD_ARGS;
        if (len < (size_t)(24 + 4 * ns)) return ORC_TRUNCATED;
        lens = data + 24;
        payload = data + 24 + 4 * ns; payload_len = len - (size_t)(24 + 4 * ns);
    } else {
        return ORC_BAD_MAGIC;
    }
    if (w <= 0 || h <= 0 || c <= 0) {
        /* degenerate legacy header: nothing to decode */
        *px = (uint8_t*)malloc(1); *pw = w; *ph = h; *pc = c;
        return *px ? ORC_OK : ORC_NOMEM;
    }
    long npix = (long)w * h;
    int16_t* s = (int16_t*)calloc((size_t)npix * c, sizeof(int16_t));
    if (!s) return ORC_NOMEM;
    long ntx = (w + tile_w - 1) / tile_w, nty = (h + tile_h - 1) / tile_h, si = 0;
    size_t off = 0;
    int rc = ORC_OK;
    for (long ty = 0; ty < nty && rc == ORC_OK; ++ty)
        for (long tx = 0; tx < ntx && rc == ORC_OK; ++tx) {
            int x0 = (int)(tx * tile_w), y0 = (int)(ty * tile_h);
            int tw = w - x0 < tile_w ? w - x0 : tile_w, th = h - y0 < tile_h ? h - y0 : tile_h;
            int16_t* origin = s + ((long)y0 * w + x0) * c;
            int nsub = planar ? c : 1;
            for (int k = 0; k < nsub && rc == ORC_OK; ++k, ++si) {
                size_t sl = lens ? get32(lens + 4 * si) : payload_len;
                if (off + sl > payload_len) { rc = ORC_TRUNCATED; break; }
                rc = orc_decode_rect(payload + off, sl, origin + (planar ? k : 0), (long)w * c, c,
                                     planar ? 1 : c, tw, th);
                off += sl;
            }
        }
    if (rc == ORC_OK) {
        uint8_t* o = (uint8_t*)malloc((size_t)npix * c);
        if (!o) rc = ORC_NOMEM;
        else {
            orc_inverse_rct(s, npix, c, o);
            *px = o; *pw = w; *ph = h; *pc = c;
        }
    }
    free(s);
    return rc;
}

uint64_t orc_fnv1a64(const uint8_t* p, size_t n) {
    uint64_t hsh = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { hsh ^= p[i]; hsh *= 1099511628211ull; }
    return hsh;
}
void orc_free(void* p) { free(p); }
