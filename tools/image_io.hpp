// image_io.hpp -- the image file I/O the reference delegates to stb (not vendored, not installed, no network):
// readers for binary PNM/PAM (P5 grey, P6 RGB, P7 with 1..4 channels, maxval 255), for PNG (every colour type and bit
// depth, Adam7 interlacing; own inflate), for uncompressed BMP (8-bit palette, 24-bit, 32-bit with masks) and for TGA (true
// colour / grey, raw or run-length), and a writer
// for PNG (adaptive row filters, LZ77 + fixed-Huffman deflate).  Plays the role of stbi_load (llcompc.cpp:25) / stbi_write_png (llcompd.cpp:29); it is host glue of the
// CLIs, not part of the coding path.  Channel counts follow stb: grey 1, grey+alpha 2, RGB / palette 3, RGBA 4, and
// a tRNS chunk adds the alpha channel.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

namespace image_io {

inline bool read_token(std::istream& in, std::string& tok) {
    tok.clear();
    int ch;
    while ((ch = in.get()) != EOF) {
        if (ch == '#') {
            while ((ch = in.get()) != EOF && ch != '\n') {}
            continue;
        }
        if (!isspace(ch)) break;
    }
    if (ch == EOF) return false;
    do {
        tok.push_back(char(ch));
        ch = in.get();
    } while (ch != EOF && !isspace(ch));
    return true;
}

// Returns an empty string on success, else the failure reason (like stbi_failure_reason()).
inline std::string load_pnm(const std::string& path, std::vector<uint8_t>& px, int& w, int& h, int& c) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "can't fopen";
    std::string magic;
    if (!read_token(in, magic)) return "empty file";
    int maxval = 0;
    if (magic == "P5" || magic == "P6") {
        std::string a, b, m;
        if (!read_token(in, a) || !read_token(in, b) || !read_token(in, m)) return "bad PNM header";
        w = atoi(a.c_str()); h = atoi(b.c_str()); maxval = atoi(m.c_str());
        c = magic == "P5" ? 1 : 3;
    } else if (magic == "P7") {
        w = h = c = 0;
        std::string key;
        while (read_token(in, key) && key != "ENDHDR") {
            std::string val;
            if (!read_token(in, val)) return "bad PAM header";
            if (key == "WIDTH") w = atoi(val.c_str());
            else if (key == "HEIGHT") h = atoi(val.c_str());
            else if (key == "DEPTH") c = atoi(val.c_str());
            else if (key == "MAXVAL") maxval = atoi(val.c_str());
        }
        // read_token consumed exactly one whitespace after ENDHDR
    } else {
        return "unknown image type (PNG or binary PGM/PPM/PAM: stb_image is not available)";
    }
    if (w <= 0 || h <= 0 || c < 1 || c > 4 || maxval != 255) return "unsupported PNM geometry (8-bit, 1..4 channels)";
    px.resize(size_t(w) * h * c);
    in.read(reinterpret_cast<char*>(px.data()), std::streamsize(px.size()));
    if (size_t(in.gcount()) != px.size()) return "truncated pixel data";
    return "";
}

// ---- inflate (RFC 1951) + zlib wrapper (RFC 1950), for the PNG reader ----------------------------------------------
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t acc = 0;
    int have = 0;
    bool fail = false;
    uint32_t bits(int k) {  // k <= 16, LSB first
        while (have < k) {
            if (pos >= n) { fail = true; return 0; }
            acc |= uint32_t(p[pos++]) << have;
            have += 8;
        }
        const uint32_t v = acc & ((1u << k) - 1);
        acc >>= k;
        have -= k;
        return v;
    }
    void align() { acc = 0; have = 0; }
};
struct Huffman {  // canonical code, decoded bit by bit (RFC 1951 3.2.2)
    uint16_t count[16] = {0}, symbol[320] = {0};
    bool build(const uint8_t* len, int n) {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) count[len[i]]++;
        int left = 1;
        for (int l = 1; l < 16; ++l) {
            left = (left << 1) - count[l];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = uint16_t(offs[l] + count[l]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = uint16_t(i);
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            code |= int(br.bits(1));
            if (br.fail) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};
inline bool inflate_zlib(const std::vector<uint8_t>& z, std::vector<uint8_t>& out, size_t expect) {
    if (z.size() < 6 || (z[0] & 0x0F) != 8 || ((unsigned(z[0]) << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return false;
    BitReader br{z.data() + 2, z.size() - 2};
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.clear();
    out.reserve(expect);
    for (bool last = false; !last;) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (uint32_t(br.p[br.pos + 1]) << 8);
            const uint32_t nlen = br.p[br.pos + 2] | (uint32_t(br.p[br.pos + 3]) << 8);
            br.pos += 4;
            if ((len ^ 0xFFFF) != nlen || br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        uint8_t len[320];
        if (type == 1) {
            for (int i = 0; i < 288; ++i) len[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            lit.build(len, 288);
            for (int i = 0; i < 30; ++i) len[i] = 5;
            dist.build(len, 30);
        } else {
            const int nlen = int(br.bits(5)) + 257, ndist = int(br.bits(5)) + 1, ncode = int(br.bits(4)) + 4;
            if (br.fail || nlen > 286 || ndist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = uint8_t(br.bits(3));
            Huffman code;
            if (br.fail || !code.build(cl, 19)) return false;
            for (int i = 0; i < nlen + ndist;) {
                const int sym = code.decode(br);
                if (sym < 0) return false;
                if (sym < 16) { len[i++] = uint8_t(sym); continue; }
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return false;
                    val = len[i - 1];
                    rep = 3 + int(br.bits(2));
                } else if (sym == 17) rep = 3 + int(br.bits(3));
                else rep = 11 + int(br.bits(7));
                if (br.fail || i + rep > nlen + ndist) return false;
                while (rep--) len[i++] = uint8_t(val);
            }
            if (len[256] == 0 || !lit.build(len, nlen) || !dist.build(len + nlen, ndist)) return false;
        }
        for (;;) {
            int sym = lit.decode(br);
            if (sym < 0) return false;
            if (sym < 256) { out.push_back(uint8_t(sym)); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return false;
            const size_t n = lbase[sym] + br.bits(lext[sym]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds >= 30) return false;
            const size_t d = dbase[ds] + br.bits(dext[ds]);
            if (br.fail || d > out.size()) return false;
            for (size_t k = 0; k < n; ++k) out.push_back(out[out.size() - d]);
        }
        if (out.size() > expect + 64) return false;  // more than the picture can hold: damaged
    }
    return true;
}

// PNG: every colour type at every bit depth the format allows (1/2/4/8/16), Adam7-interlaced or not.  Samples come out
// as 8 bits per channel the way stbi_load (8-bit interface) hands them over: low bit depths of grey are scaled to
// 0..255 (x255, x85, x17), 16-bit samples keep their high byte, palette indices are looked up.  Returns an empty string on
// success, else the failure reason.
inline std::string load_png(const std::string& path, std::vector<uint8_t>& px, int& w, int& h, int& c) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "can't fopen";
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || std::memcmp(f.data(), sig, 8) != 0) return "bad png sig";
    auto be32 = [&](size_t o) { return (uint32_t(f[o]) << 24) | (uint32_t(f[o + 1]) << 16) | (uint32_t(f[o + 2]) << 8) | f[o + 3]; };
    std::vector<uint8_t> idat, plte, trns;
    int depth = 0, ctype = -1, interlace = 0;
    w = h = 0;
    for (size_t o = 8; o + 12 <= f.size();) {
        const uint32_t n = be32(o);
        if (o + 12 + size_t(n) > f.size()) return "corrupt PNG chunk";
        const std::string type(reinterpret_cast<const char*>(&f[o + 4]), 4);
        const uint8_t* body = &f[o + 8];
        if (type == "IHDR" && n == 13) {
            w = int(be32(o + 8)); h = int(be32(o + 12));
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (type == "PLTE") plte.assign(body, body + n);
        else if (type == "tRNS") trns.assign(body, body + n);
        else if (type == "IDAT") idat.insert(idat.end(), body, body + n);
        else if (type == "IEND") break;
        o += 12 + size_t(n);
    }
    if (w <= 0 || h <= 0 || ctype < 0) return "no IHDR";
    int fc;  // channels in the file
    switch (ctype) {
        case 0: fc = 1; break;
        case 2: fc = 3; break;
        case 3: fc = 1; break;
        case 4: fc = 2; break;
        case 6: fc = 4; break;
        default: return "bad PNG colour type";
    }
    // PNG spec table 11.1: grey 1/2/4/8/16, palette 1/2/4/8, everything else 8/16
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) return "bad PNG bit depth";
    if (interlace > 1) return "bad PNG interlace method";
    if (ctype == 3 && plte.size() < 3) return "missing PLTE";
    if (uint64_t(w) * uint64_t(h) * 4 >= (1ull << 31)) return "too large";
    const int bpp = depth * fc;                       // bits per pixel in the file
    const size_t fdist = size_t(bpp >= 8 ? bpp / 8 : 1);  // filter distance in bytes (PNG spec 9.2)
    auto row_bytes = [&](int pw) { return (size_t(pw) * size_t(bpp) + 7) / 8; };
    // the (sub)images in the stream: one, or the seven Adam7 passes
    struct Pass { int x0, y0, dx, dy, pw, ph; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, w, h});
    else {
        static const int A[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        for (auto& a : A) {
            const int pw = (w - a[0] + a[2] - 1) / a[2], ph = (h - a[1] + a[3] - 1) / a[3];
            if (pw > 0 && ph > 0) passes.push_back({a[0], a[1], a[2], a[3], pw, ph});
        }
    }
    size_t need = 0;
    for (auto& ps : passes) need += (row_bytes(ps.pw) + 1) * size_t(ps.ph);
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat, raw, need) || raw.size() < need) return "bad zlib stream";
    // samples at file depth, one uint16 per sample, [y][x][fc]
    std::vector<uint16_t> smp(size_t(w) * h * fc);
    size_t at = 0;
    std::vector<uint8_t> cur_row, up_row;
    for (auto& ps : passes) {
        const size_t row = row_bytes(ps.pw);
        cur_row.assign(row, 0);
        up_row.assign(row, 0);
        for (int y = 0; y < ps.ph; ++y) {  // undo the row filters (PNG spec 9.2), then unpack the samples
            const uint8_t* src = &raw[at];
            at += row + 1;
            const int ft = src[0];
            if (ft > 4) return "bad PNG filter";
            for (size_t i = 0; i < row; ++i) {
                const int a = i >= fdist ? cur_row[i - fdist] : 0, b = up_row[i], cc = i >= fdist ? up_row[i - fdist] : 0;
                int pred = 0;
                if (ft == 1) pred = a;
                else if (ft == 2) pred = b;
                else if (ft == 3) pred = (a + b) >> 1;
                else if (ft == 4) {
                    const int pp = a + b - cc, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - cc);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : cc);
                }
                cur_row[i] = uint8_t(src[1 + i] + pred);
            }
            const int oy = ps.y0 + y * ps.dy;
            for (int x = 0; x < ps.pw; ++x) {
                uint16_t* o = &smp[(size_t(oy) * w + size_t(ps.x0 + x * ps.dx)) * fc];
                for (int k = 0; k < fc; ++k) {
                    const size_t si = size_t(x) * fc + k;  // sample index in the row
                    if (depth == 16) o[k] = uint16_t((cur_row[2 * si] << 8) | cur_row[2 * si + 1]);
                    else if (depth == 8) o[k] = cur_row[si];
                    else {  // 1, 2, 4 bits: leftmost sample in the high-order bits
                        const size_t bit = si * size_t(depth);
                        o[k] = uint16_t((cur_row[bit >> 3] >> (8 - depth - int(bit & 7))) & ((1 << depth) - 1));
                    }
                }
            }
            cur_row.swap(up_row);  // (the row just finished becomes "up"; cur is overwritten next)
        }
    }
    // to the channel layout and sample range stb would hand over
    const bool pal = ctype == 3;
    const bool key = !pal && (ctype == 0 || ctype == 2) && trns.size() >= size_t(2 * fc);
    c = pal ? (trns.empty() ? 3 : 4) : fc + (key ? 1 : 0);
    px.resize(size_t(w) * h * c);
    const size_t npx = size_t(w) * h;
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;  // grey below 8 bits
    auto to8 = [&](uint16_t v) { return uint8_t(depth == 16 ? v >> 8 : depth == 8 ? v : v * scale); };
    for (size_t i = 0; i < npx; ++i) {
        uint8_t* o = &px[i * c];
        const uint16_t* sp = &smp[i * fc];
        if (pal) {
            const size_t idx = sp[0];
            for (int k = 0; k < 3; ++k) o[k] = idx * 3 + k < plte.size() ? plte[idx * 3 + k] : 0;
            if (c == 4) o[3] = idx < trns.size() ? trns[idx] : 255;
        } else {
            for (int k = 0; k < fc; ++k) o[k] = to8(sp[k]);
            if (key) {  // colour key: 16-bit big-endian values at file depth
                bool same = true;
                for (int k = 0; same && k < fc; ++k) {
                    const uint16_t t = uint16_t((trns[2 * k] << 8) | trns[2 * k + 1]);
                    same = depth == 16 ? t == sp[k] : (t & 0xFF) == sp[k];
                }
                o[fc] = same ? 0 : 255;
            }
        }
    }
    return "";
}

// BMP (what stb_image reads of it most often): BITMAPINFOHEADER and later, uncompressed, 8-bit palette / 24-bit / 32-bit,
// bottom-up or top-down; 32-bit files with BI_BITFIELDS masks (any 8-bit-aligned byte positions) carry alpha when the
// header names an alpha mask.  Channels: 3, or 4 with alpha -- like stb.
inline std::string load_bmp(const std::string& path, std::vector<uint8_t>& px, int& w, int& h, int& c) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "can't fopen";
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    auto le16 = [&](size_t o) { return uint32_t(f[o]) | (uint32_t(f[o + 1]) << 8); };
    auto le32 = [&](size_t o) { return le16(o) | (le16(o + 2) << 16); };
    if (f.size() < 54 || f[0] != 'B' || f[1] != 'M') return "not BMP";
    const uint32_t off = le32(10), hsz = le32(14);
    if (hsz < 40 || 14 + size_t(hsz) > f.size()) return "unsupported BMP header";
    const int32_t bw = int32_t(le32(18)), bh = int32_t(le32(22));
    const uint32_t planes = le16(26), bits = le16(28), comp = le32(30);
    if (planes != 1 || bw <= 0 || bh == 0) return "bad BMP";
    if (!(bits == 8 || bits == 24 || bits == 32)) return "unsupported BMP bit count";
    if (!(comp == 0 || (comp == 3 && bits == 32))) return "compressed BMP is not supported";
    w = bw;
    h = bh < 0 ? -bh : bh;
    if (uint64_t(w) * uint64_t(h) * 4 >= (1ull << 31)) return "too large";
    uint32_t mask[4] = {0x00FF0000u, 0x0000FF00u, 0x000000FFu, 0};  // r, g, b, a of an uncompressed 32-bit pixel
    if (comp == 3) {
        const size_t mo = 14 + 40;  // masks follow the 40-byte header (they are part of the V4/V5 headers)
        if (mo + 12 > f.size()) return "bad BMP";
        for (int k = 0; k < 3; ++k) mask[k] = le32(mo + 4 * size_t(k));
        if (hsz >= 56 && mo + 16 <= f.size()) mask[3] = le32(mo + 12);
    } else if (bits == 32 && hsz >= 56) {
        mask[3] = le32(14 + 40 + 12);
    }
    int shift[4];
    for (int k = 0; k < 4; ++k) {
        shift[k] = -1;
        for (int sft = 0; sft < 32; sft += 8)
            if (mask[k] == (0xFFu << sft)) shift[k] = sft;
        if (k < 3 && bits == 32 && shift[k] < 0) return "unsupported BMP channel masks";
    }
    const bool alpha = bits == 32 && mask[3] != 0 && shift[3] >= 0;
    c = alpha ? 4 : 3;
    const size_t stride = ((size_t(w) * bits + 31) / 32) * 4;
    if (size_t(off) + stride * size_t(h) > f.size()) return "truncated BMP";
    const size_t pal_at = 14 + size_t(hsz);
    uint32_t ncol = le32(46);
    if (bits == 8) {
        if (ncol == 0 || ncol > 256) ncol = 256;
        if (pal_at + 4 * size_t(ncol) > f.size()) return "bad BMP palette";
    }
    px.resize(size_t(w) * h * c);
    for (int y = 0; y < h; ++y) {
        const uint8_t* src = &f[off + stride * size_t(bh < 0 ? y : h - 1 - y)];
        uint8_t* o = &px[size_t(y) * w * c];
        for (int x = 0; x < w; ++x, o += c) {
            if (bits == 8) {
                const uint32_t idx = src[x];
                const uint8_t* q = &f[pal_at + 4 * size_t(idx < ncol ? idx : 0)];
                o[0] = q[2]; o[1] = q[1]; o[2] = q[0];
            } else if (bits == 24) {
                o[0] = src[3 * x + 2]; o[1] = src[3 * x + 1]; o[2] = src[3 * x];
            } else {
                const uint32_t v = uint32_t(src[4 * x]) | (uint32_t(src[4 * x + 1]) << 8) | (uint32_t(src[4 * x + 2]) << 16) | (uint32_t(src[4 * x + 3]) << 24);
                for (int k = 0; k < c; ++k) o[k] = uint8_t(v >> shift[k]);
            }
        }
    }
    return "";
}

// TGA (no signature: recognised by extension and header sanity, like stb does last): true-colour and grey, 8 / 24 / 32
// bits, raw or run-length packets, either vertical origin.  Channels: 1, 3 or 4 -- like stb.
inline std::string load_tga(const std::string& path, std::vector<uint8_t>& px, int& w, int& h, int& c) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "can't fopen";
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (f.size() < 18) return "not TGA";
    const int id_len = f[0], cmap = f[1], type = f[2], bits = f[16], desc = f[17];
    w = f[12] | (f[13] << 8);
    h = f[14] | (f[15] << 8);
    const bool rle = type == 10 || type == 11, grey = type == 3 || type == 11;
    if (cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11)) return "unsupported TGA type";
    if (w <= 0 || h <= 0) return "bad TGA";
    if (!(grey ? bits == 8 : (bits == 24 || bits == 32))) return "unsupported TGA bit count";
    c = bits / 8;
    const size_t npx = size_t(w) * h;
    px.resize(npx * c);
    size_t at = 18 + size_t(id_len);
    auto put = [&](size_t i, const uint8_t* q) {  // file order is B, G, R(, A)
        uint8_t* o = &px[i * c];
        if (c == 1) o[0] = q[0];
        else { o[0] = q[2]; o[1] = q[1]; o[2] = q[0]; if (c == 4) o[3] = q[3]; }
    };
    std::vector<uint8_t> lin(npx * c);  // pixels in file order first
    if (!rle) {
        if (at + npx * c > f.size()) return "truncated TGA";
        lin.assign(f.begin() + long(at), f.begin() + long(at + npx * c));
    } else {
        size_t i = 0;
        while (i < npx) {
            if (at >= f.size()) return "truncated TGA";
            const int head = f[at++], n = (head & 127) + 1;
            if (i + size_t(n) > npx) return "bad TGA packet";
            if (head & 128) {
                if (at + size_t(c) > f.size()) return "truncated TGA";
                for (int k = 0; k < n; ++k) std::memcpy(&lin[(i + size_t(k)) * c], &f[at], size_t(c));
                at += size_t(c);
            } else {
                if (at + size_t(n) * c > f.size()) return "truncated TGA";
                std::memcpy(&lin[i * c], &f[at], size_t(n) * c);
                at += size_t(n) * c;
            }
            i += size_t(n);
        }
    }
    const bool top_down = (desc & 0x20) != 0, right_left = (desc & 0x10) != 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t src = size_t(top_down ? y : h - 1 - y) * w + size_t(right_left ? w - 1 - x : x);
            put(size_t(y) * w + x, &lin[src * c]);
        }
    return "";
}

// stbi_load's role: PNG, BMP or binary PNM/PAM by signature; TGA (which has none) by its extension.
inline std::string load_image(const std::string& path, std::vector<uint8_t>& px, int& w, int& h, int& c) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "can't fopen";
    const int first = in.get(), second = in.get();
    in.close();
    if (first == 0x89) return load_png(path, px, w, h, c);
    if (first == 'B' && second == 'M') return load_bmp(path, px, w, h, c);
    if (path.size() >= 4) {
        std::string ext = path.substr(path.size() - 4);
        for (auto& ch : ext) ch = char(std::tolower(static_cast<unsigned char>(ch)));
        if (ext == ".tga") return load_tga(path, px, w, h, c);
    }
    return load_pnm(path, px, w, h, c);
}

inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

inline void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}
inline void put_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body) {
    put_be32(out, uint32_t(body.size()));
    std::vector<uint8_t> t(type, type + 4);
    t.insert(t.end(), body.begin(), body.end());
    out.insert(out.end(), t.begin(), t.end());
    put_be32(out, crc32(t.data(), t.size()));
}

// ---- deflate for the PNG writer: LZ77 (hash chains, 32 KiB window) + the fixed Huffman code of RFC 1951 3.2.6 -------------
struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int n = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t v, int bits) {  // LSB first
        acc |= v << n;
        n += bits;
        while (n >= 8) { out.push_back(uint8_t(acc)); acc >>= 8; n -= 8; }
    }
    void put_code(uint32_t code, int bits) {  // Huffman codes go in MSB first
        uint32_t r = 0;
        for (int i = 0; i < bits; ++i) r |= ((code >> i) & 1u) << (bits - 1 - i);
        put(r, bits);
    }
    void flush() { if (n) { out.push_back(uint8_t(acc)); acc = 0; n = 0; } }
};
inline void put_fixed_symbol(BitWriter& bw, int sym) {  // literal / length alphabet
    if (sym < 144) bw.put_code(0x30 + uint32_t(sym), 8);
    else if (sym < 256) bw.put_code(0x190 + uint32_t(sym - 144), 9);
    else if (sym < 280) bw.put_code(uint32_t(sym - 256), 7);
    else bw.put_code(0xC0 + uint32_t(sym - 280), 8);
}
// zlib stream (RFC 1950) around one fixed-Huffman deflate block
inline std::vector<uint8_t> deflate_zlib(const std::vector<uint8_t>& in) {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    std::vector<uint8_t> z = {0x78, 0x5E};
    BitWriter bw(z);
    bw.put(1, 1);  // BFINAL
    bw.put(1, 2);  // BTYPE = 01: fixed Huffman
    const size_t n = in.size();
    constexpr int kHashBits = 15, kWindow = 32768, kMaxChain = 24;
    std::vector<int32_t> head(size_t(1) << kHashBits, -1), prev(n ? n : 1, -1);
    auto hash3 = [&](size_t i) { return ((uint32_t(in[i]) << 16 | uint32_t(in[i + 1]) << 8 | in[i + 2]) * 0x9E3779B1u) >> (32 - kHashBits); };
    size_t i = 0;
    while (i < n) {
        int best_len = 0, best_dist = 0;
        if (i + 3 <= n) {
            const uint32_t hh = hash3(i);
            int chain = 0;
            for (int32_t c = head[hh]; c >= 0 && int(i - size_t(c)) <= kWindow && chain < kMaxChain; c = prev[size_t(c)], ++chain) {
                const size_t maxl = n - i < 258 ? n - i : 258;
                size_t l = 0;
                while (l < maxl && in[size_t(c) + l] == in[i + l]) ++l;
                if (int(l) > best_len) { best_len = int(l); best_dist = int(i - size_t(c)); if (l == maxl) break; }
            }
        }
        const size_t step = best_len >= 3 ? size_t(best_len) : 1;
        if (best_len >= 3) {
            int lc = 28;
            while (len_base[lc] > best_len) --lc;
            put_fixed_symbol(bw, 257 + lc);
            if (len_extra[lc]) bw.put(uint32_t(best_len - len_base[lc]), len_extra[lc]);
            int dc = 29;
            while (dist_base[dc] > best_dist) --dc;
            bw.put_code(uint32_t(dc), 5);
            if (dist_extra[dc]) bw.put(uint32_t(best_dist - dist_base[dc]), dist_extra[dc]);
        } else {
            put_fixed_symbol(bw, in[i]);
        }
        for (size_t k = 0; k < step; ++k, ++i)  // every position enters the hash chains
            if (i + 3 <= n) { const uint32_t hh = hash3(i); prev[i] = head[hh]; head[hh] = int32_t(i); }
    }
    put_fixed_symbol(bw, 256);  // end of block
    bw.flush();
    uint32_t a = 1, b = 0;
    for (uint8_t v : in) { a = (a + v) % 65521; b = (b + a) % 65521; }
    put_be32(z, (b << 16) | a);
    return z;
}

// 1 on success, 0 on failure (the convention of stbi_write_png).  Per row the filter with the smallest sum of absolute
// (signed) residuals is taken, like stb_image_write; the rows are deflated with the coder above.
inline int write_png(const std::string& path, int w, int h, int c, const uint8_t* px, int stride) {
    if (w <= 0 || h <= 0 || c < 1 || c > 4) return 0;
    static const uint8_t colour_type[5] = {0, 0, 4, 2, 6};
    const size_t row = size_t(w) * c;
    std::vector<uint8_t> raw;
    raw.reserve((row + 1) * h);
    std::vector<uint8_t> cand(row), best(row);
    for (int y = 0; y < h; ++y) {
        const uint8_t* cur = px + size_t(y) * stride;
        const uint8_t* up = y ? px + size_t(y - 1) * stride : nullptr;
        long best_sum = -1;
        int best_ft = 0;
        for (int ft = 0; ft < 5; ++ft) {
            long sum = 0;
            for (size_t i = 0; i < row; ++i) {
                const int a = i >= size_t(c) ? cur[i - c] : 0, b = up ? up[i] : 0, cc = (up && i >= size_t(c)) ? up[i - c] : 0;
                int pred = 0;
                if (ft == 1) pred = a;
                else if (ft == 2) pred = b;
                else if (ft == 3) pred = (a + b) >> 1;
                else if (ft == 4) {
                    const int pp = a + b - cc, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - cc);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : cc);
                }
                cand[i] = uint8_t(cur[i] - pred);
                sum += std::abs(int(int8_t(cand[i])));
            }
            if (best_sum < 0 || sum < best_sum) { best_sum = sum; best_ft = ft; best.swap(cand); }
        }
        raw.push_back(uint8_t(best_ft));
        raw.insert(raw.end(), best.begin(), best.end());
    }
    const std::vector<uint8_t> z = deflate_zlib(raw);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, uint32_t(w)); put_be32(ihdr, uint32_t(h));
    ihdr.push_back(8); ihdr.push_back(colour_type[c]); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) return 0;
    f.write(reinterpret_cast<const char*>(out.data()), std::streamsize(out.size()));
    return f.good() ? 1 : 0;
}

}  // namespace image_io
