// cli_common.hpp -- small file helpers shared by the two command-line tools.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace cli {

inline bool slurp(const std::string& path, std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    bytes.clear();
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
    std::fclose(f);
    return true;
}

inline bool spill(const std::string& path, const std::vector<uint8_t>& bytes) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes.empty() || std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    return std::fclose(f) == 0 && ok;
}

// "0,1,2" -> {0,1,2}; false for anything that is not a comma-separated list of non-negative ordinals
inline bool parse_device_list(const char* text, std::vector<int>& out) {
    out.clear();
    const char* p = text;
    while (*p) {
        if (*p < '0' || *p > '9') return false;
        long v = 0;
        while (*p >= '0' && *p <= '9' && v < 100000) v = v * 10 + (*p++ - '0');
        out.push_back(int(v));
        if (*p == ',') { ++p; if (!*p) return false; }
        else if (*p) return false;
    }
    return !out.empty();
}

// exit codes of the reference tools: 0 done, 1 usage / I/O / codec error (llcompc.cpp:20-38, llcompd.cpp:13-34),
// 2 non-standard exception (llcompd.cpp:35-37)
enum Exit { kDone = 0, kFailed = 1, kUnknown = 2 };

}  // namespace cli
