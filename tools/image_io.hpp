// image_io.hpp -- the image file I/O the reference delegates to stb (not vendored, not installed, no network):
// a reader for binary PNM/PAM (P5 grey, P6 RGB, P7 with 1..4 channels, maxval 255) and a writer for PNG with
// stored (uncompressed) deflate blocks.  Plays the role of stbi_load (llcompc.cpp:25) / stbi_write_png
// (llcompd.cpp:29); it is host glue of the CLIs, not part of the coding path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
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
        return "unknown image type (binary PGM/PPM/PAM only: stb_image is not available)";
    }
    if (w <= 0 || h <= 0 || c < 1 || c > 4 || maxval != 255) return "unsupported PNM geometry (8-bit, 1..4 channels)";
    px.resize(size_t(w) * h * c);
    in.read(reinterpret_cast<char*>(px.data()), std::streamsize(px.size()));
    if (size_t(in.gcount()) != px.size()) return "truncated pixel data";
    return "";
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

// 1 on success, 0 on failure (the convention of stbi_write_png).
inline int write_png(const std::string& path, int w, int h, int c, const uint8_t* px, int stride) {
    if (w <= 0 || h <= 0 || c < 1 || c > 4) return 0;
    static const uint8_t colour_type[5] = {0, 0, 4, 2, 6};
    std::vector<uint8_t> raw;
    raw.reserve((size_t(w) * c + 1) * h);
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), px + size_t(y) * stride, px + size_t(y) * stride + size_t(w) * c);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521; b = (b + a) % 65521; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(uint8_t(n)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n)); z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    put_be32(z, (b << 16) | a);
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
