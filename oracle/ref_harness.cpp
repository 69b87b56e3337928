// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" shell around the *real* reference header, compiled where it lies
// (-I/root/reference, see oracle/Makefile, target _ref/libllcomp_ref.so).  Nothing of the
// reference is copied into this repository; this file only calls it.
//
//   O1  = the reference's own entry points, unmodified:
//           llcomp::compressImage   (llcomp.hpp:358)   llcomp::decompressImage (llcomp.hpp:461)
//   O2  = "component driver": the reference's own RangeEncoder (llcomp.hpp:33), cabac::State
//           (llcomp.hpp:283), binarization::putSymbol (llcomp.hpp:166), quant11/quant5
//           (llcomp.hpp:335/339) and median (llcomp.hpp:343) driven over a plane of already
//           colour-transformed int16 samples with a growable sink.  Needed because O1 overflows its
//           fixed w*h*c output buffer when the stream is larger than the raw image (llcomp.hpp:362,
//           SURVEY.md D1) and because per-channel slices code int16 planes that no uint8 call can
//           express (parity level P1).
//
// The GPU box never sees /root/reference; only the prebuilt .so travels (oracle/_ref is gitignored).
#include "llcomp.hpp"

#include <cstring>
#include <cstdlib>

namespace {

// Neighbourhood of sample (col,row,ch) in a w x h x c interleaved int16 image, following the border
// rules of llcomp.hpp:417-422 (slice-local: the image handed in *is* the slice).
struct Hood { int l, t, L, tl, tr, T; };

inline Hood hood_at(const int16_t* s, int w, int c, int col, int row, int ch) {
    auto at = [&](int x, int y) -> int { return s[(size_t(y) * w + x) * c + ch]; };
    Hood n;
    n.l  = col > 0 ? at(col - 1, row) : (row > 0 ? at(col, row - 1) : 128);
    n.t  = row > 0 ? at(col, row - 1) : n.l;
    n.L  = col > 1 ? at(col - 2, row) : n.l;
    n.tl = (row > 0 && col > 0) ? at(col - 1, row - 1) : n.t;
    n.tr = (row > 0 && col < w - 1) ? at(col + 1, row - 1) : n.t;
    n.T  = row > 1 ? at(col, row - 2) : n.t;
    return n;
}

inline int ctx_of(const Hood& n) {  // llcomp.hpp:424-429, following the header's own LargeModel constant
    int ctx = llcomp::quant11(n.l - n.tl) + 11 * llcomp::quant11(n.tl - n.t) + 121 * llcomp::quant11(n.t - n.tr);
    if (llcomp::LargeModel) ctx += 605 * llcomp::quant5(n.L - n.l) + 3025 * llcomp::quant5(n.T - n.t);
    return ctx;
}

}  // namespace

extern "C" {

// ---- O1: unmodified reference entry points --------------------------------------------------
// Returns stream length, or -1 if `cap` is too small.  CALLER must make sure the reference's own
// buffer cannot overflow (stream <= w*h*c bytes): use ref_o2_* first to learn the length.
long ref_compress_image(const uint8_t* px, int w, int h, int c, uint8_t* out, long cap) {
    std::vector<uint8_t> in(px, px + size_t(w) * h * c);
    std::vector<uint8_t> s = llcomp::compressImage(in, w, h, c);
    if (long(s.size()) > cap) return -1;
    std::memcpy(out, s.data(), s.size());
    return long(s.size());
}

// Returns 0 ok, 1 "Invalid magic number", 2 "Invalid exponent", 3 other exception, -1 cap too small.
// Only defined for channels >= 3.
int ref_decompress_image(const uint8_t* data, long len, uint8_t* out, long cap, int* w, int* h, int* c) {
    try {
        std::vector<uint8_t> in(data, data + len);
        llcomp::RawImage img = llcomp::decompressImage(in);
        *w = img.width; *h = img.height; *c = img.channels;
        if (long(img.pixels.size()) > cap) return -1;
        std::memcpy(out, img.pixels.data(), img.pixels.size());
        return 0;
    } catch (const std::runtime_error& e) {
        if (std::strcmp(e.what(), "Invalid magic number") == 0) return 1;
        if (std::strcmp(e.what(), "Invalid exponent") == 0) return 2;
        return 3;
    } catch (...) {
        return 3;
    }
}

// ---- O2: reference components over int16 samples, growable sink -------------------------------
// Forward colour transform exactly as the encoder's inner block does it (llcomp.hpp:396-414):
// channels>=3: [r-g, g+((b-g)+(r-g))/4, b-g, extras...]; otherwise raw copy.
void ref_o2_forward_rct(const uint8_t* px, long npix, int c, int16_t* out) {
    for (long p = 0; p < npix; ++p) {
        const uint8_t* q = px + p * c;
        int16_t* o = out + p * c;
        if (c >= 3) {
            int g = q[1], b = q[2] - g, r = q[0] - g;
            g += (b + r) / 4;
            o[0] = int16_t(r); o[1] = int16_t(g); o[2] = int16_t(b);
            for (int k = 3; k < c; ++k) o[k] = q[k];
        } else {
            for (int k = 0; k < c; ++k) o[k] = q[k];
        }
    }
}

// Codes a w x h x c interleaved int16 image as ONE bare range-coder stream (no 6-byte header),
// sample order row -> pixel -> channel with one shared state table (llcomp.hpp:390-447).
// Returns the stream length or -1 when cap is too small.
long ref_o2_encode_samples(const int16_t* s, int w, int h, int c, uint8_t* out, long cap) {
    std::vector<uint8_t> sink;
    sink.reserve(size_t(w) * h * c / 2 + 64);
    llcomp::RangeEncoder enc([&](uint8_t b) { sink.push_back(b); });
    std::vector<llcomp::cabac::State> table(llcomp::getStatesNb());
    for (int row = 0; row < h; ++row)
        for (int col = 0; col < w; ++col)
            for (int ch = 0; ch < c; ++ch) {
                const Hood n = hood_at(s, w, c, col, row, ch);
                int ctx = ctx_of(n);
                int res = int(s[(size_t(row) * w + col) * c + ch]) - llcomp::median(n.l, n.l + n.t - n.tl, n.t);
                if (ctx < 0) { ctx = -ctx; res = -res; }
                llcomp::cabac::State* bank = &table[size_t(ctx) * llcomp::substates_nb];
                llcomp::binarization::putSymbol<true, llcomp::param_e_lim, llcomp::param_r_lim, llcomp::param_s_bit>(
                    res, [&](int slot, bool bit) {
                        enc.put(bit, bank[slot].P());
                        bank[slot].update(bit);
                    });
            }
    enc.finish();
    if (long(sink.size()) > cap) return -1;
    std::memcpy(out, sink.data(), sink.size());
    return long(sink.size());
}

// Whole-image O2: 6-byte header + stream, from uint8 pixels (what compressImage would emit if its
// buffer could grow).
long ref_o2_compress_image(const uint8_t* px, int w, int h, int c, uint8_t* out, long cap) {
    if (cap < 6) return -1;
    std::vector<int16_t> s(size_t(w) * h * c);
    ref_o2_forward_rct(px, long(w) * h, c, s.data());
    out[0] = llcomp::magic_revision; out[1] = uint8_t(c);
    out[2] = uint8_t(w & 0xFF); out[3] = uint8_t((w >> 8) & 0xFF);
    out[4] = uint8_t(h & 0xFF); out[5] = uint8_t((h >> 8) & 0xFF);
    long n = ref_o2_encode_samples(s.data(), w, h, c, out + 6, cap - 6);
    return n < 0 ? -1 : n + 6;
}

// Table/primitive probes so the restatement's tables can be compared entry by entry.
int ref_quant11(int x) { return llcomp::quant11(x); }
int ref_quant5(int x) { return llcomp::quant5(x); }
int ref_median(int a, int b, int c) { return llcomp::median(a, b, c); }
int ref_state_p(int s) { llcomp::cabac::State st; st.state = uint8_t(s); return st.P(); }
int ref_state_next(int s, int bit) { llcomp::cabac::State st; st.state = uint8_t(s); st.update(bit != 0); return st.state; }
int ref_states_nb(void) { return int(llcomp::getStatesNb()); }
int ref_magic(void) { return llcomp::magic_revision; }
int ref_large_model(void) { return llcomp::LargeModel ? 1 : 0; }

}  // extern "C"
