// llcomp_mi.hpp -- C++ drop-in for the reference's header /root/reference/llcomp.hpp, backed by libllcomp_mi.so
// (hand-written HIP kernels for MI355X behind the C ABI of llcomp_mi.h).
//
// Same namespace, names, argument meaning and error behaviour as the reference, so a caller such as llcompc.cpp:33 /
// llcompd.cpp:26 only has to change its #include and link -lllcomp_mi -lamdhip64:
//
//   reference (llcomp.hpp)                                              here
//   ---------------------------------------------------------------    ------------------------------------------------
//   llcomp::ext = ".llcomp"                                    :18      llcomp::ext
//   std::vector<uint8_t> llcomp::compressImage(rgb,w,h,c)      :358     same signature (+ optional llcomp::Options)
//   struct llcomp::RawImage{pixels,width,height,channels}      :454     same members (width/height widened to 32 bit)
//   llcomp::RawImage llcomp::decompressImage(data)             :461     same signature (+ overload with a device list)
//   throw std::runtime_error("Invalid magic number")           :466     same text
//   throw std::runtime_error("Invalid exponent")               :233     same text
//
// Without options the output is the reference's own whole-image format (magic 0x79), byte-identical to the
// reference's stream.  Options{.sliced = true, ...} selects the parallel container (magic 0x9C); decompressImage
// reads both.  Conditions the reference leaves undefined (buffer overflow D1, channels < 3 decode D2, truncated
// input D5, size mismatch D6) raise std::runtime_error / std::invalid_argument here instead.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "llcomp_mi.h"

namespace llcomp {

constexpr inline auto ext = ".llcomp";

struct Options {
    bool sliced = false;    // false: reference format, one serial stream; true: independent slices
    uint32_t tile_w = 0;    // slice width  (0 = full width)
    uint32_t tile_h = 0;    // slice height (0 = full height)
    bool planar = true;     // one slice per colour-transformed channel plane
    int device = -1;        // HIP device ordinal, -1 = current
    bool small_model = false;  // bitstream of a reference built with LargeModel = false (llcomp.hpp:21)
    // One image over several GPUs inside this process (sliced only): the tile rows are dealt over these HIP ordinals, every GPU gets
    // only its rows over its own PCIe link and copies its payload straight into the container -- byte-identical to one device's.
    // Empty = `device`.  An ordinal may repeat (two lanes on one GPU).
    std::vector<int> devices;
    uint32_t chunks_per_device = 0;  // 0 = 4
};

struct RawImage {
    std::vector<uint8_t> pixels;
    uint32_t width;
    uint32_t height;
    uint8_t channels;
};

namespace detail {
[[noreturn]] inline void raise(int status) {
    const char* msg = llcomp_mi_strerror(status);
    if (status == LLCOMP_MI_BAD_ARGS || status == LLCOMP_MI_OUT_OF_RANGE) throw std::invalid_argument(msg);
    if (status == LLCOMP_MI_DEVICE_FAILED) {  // which device of the list, and what it said
        int32_t dev = -1;
        int why = 0;
        if (llcomp_mi_last_device_error(&dev, nullptr, &why))
            throw std::runtime_error(std::string(msg) + " [device " + std::to_string(dev) + ": " + llcomp_mi_strerror(why) + "]");
    }
    throw std::runtime_error(msg);
}
// this header and the library it is linked against must come from the same ABI version (llcomp_mi.h: one struct layout per version)
inline void check_abi() {
    static const bool ok = llcomp_mi_abi_version() == LLCOMP_MI_ABI_VERSION;
    if (!ok) throw std::runtime_error("libllcomp_mi.so was built from another llcomp_mi.h (ABI version mismatch): rebuild");
}
}  // namespace detail

inline std::vector<uint8_t> compressImage(const std::vector<uint8_t>& rgb, int width, int height, int channels,
                                          const Options& opt = {}) {
    if (width <= 0 || height <= 0 || channels <= 0 ||
        rgb.size() != size_t(width) * size_t(height) * size_t(channels))  // the reference only asserts this (:361)
        detail::raise(LLCOMP_MI_BAD_ARGS);
    detail::check_abi();
    llcomp_mi_opts o{};
    o.struct_size = sizeof(o);
    o.format = opt.sliced ? LLCOMP_MI_FORMAT_SLICED : LLCOMP_MI_FORMAT_LEGACY;
    o.tile_w = opt.tile_w;
    o.tile_h = opt.tile_h;
    o.planar = opt.planar ? 1u : 0u;
    o.device = opt.device;
    o.small_model = opt.small_model ? 1u : 0u;
    std::vector<int32_t> devs(opt.devices.begin(), opt.devices.end());
    o.n_devices = uint32_t(devs.size());
    o.devices = devs.empty() ? nullptr : devs.data();
    o.chunks_per_device = opt.chunks_per_device;
    uint8_t* out = nullptr;
    size_t n = 0;
    if (int rc = llcomp_mi_encode(rgb.data(), uint32_t(width), uint32_t(height), uint32_t(channels), &o, &out, &n))
        detail::raise(rc);
    std::vector<uint8_t> v(out, out + n);
    llcomp_mi_free(out);
    return v;
}

inline RawImage decompressImage(const std::vector<uint8_t>& data, int device = -1, bool legacy_small_model = false) {
    detail::check_abi();
    uint8_t* px = nullptr;
    uint32_t w = 0, h = 0, c = 0;
    if (int rc = llcomp_mi_decode_flags(data.data(), data.size(), device, legacy_small_model ? LLCOMP_MI_FLAG_SMALL_MODEL : 0u, &px, &w, &h, &c))
        detail::raise(rc);
    RawImage img{std::vector<uint8_t>(px, px + size_t(w) * h * c), w, h, uint8_t(c)};
    llcomp_mi_free(px);
    return img;
}

// ... decoded over a list of GPUs inside this process (llcomp_mi_decode_devices); how the stream was encoded does not matter
inline RawImage decompressImage(const std::vector<uint8_t>& data, const std::vector<int>& devices, bool legacy_small_model = false) {
    if (devices.empty()) return decompressImage(data, -1, legacy_small_model);
    detail::check_abi();
    std::vector<int32_t> devs(devices.begin(), devices.end());
    uint8_t* px = nullptr;
    uint32_t w = 0, h = 0, c = 0;
    if (int rc = llcomp_mi_decode_devices(data.data(), data.size(), devs.data(), uint32_t(devs.size()), 0,
                                          legacy_small_model ? LLCOMP_MI_FLAG_SMALL_MODEL : 0u, &px, &w, &h, &c))
        detail::raise(rc);
    RawImage img{std::vector<uint8_t>(px, px + size_t(w) * h * c), w, h, uint8_t(c)};
    llcomp_mi_free(px);
    return img;
}

}  // namespace llcomp
