// llcompc <image> [--sliced TWxTH | --sliced auto] [--interleaved] [--legacy] [--small-model] [--devices a,b,...]
//
// Compressor front end on libllcomp_mi.so with the observable behaviour of the reference tool
// (/root/reference/llcompc.cpp:14-43): exactly one required positional argument, output written next to the input as
// "<image>.llcomp", exit status 1 for usage, unreadable image or unwritable output, 0 otherwise.  stb_image is neither
// vendored nor installed, so the image reader is tools/image_io.hpp (PNG, binary PGM / PPM / PAM).  Without options the
// stream is the reference's own format (magic 0x79), as a drop-in must: that is ONE serial range-coder chain, which a
// GPU runs on a single lane -- many times slower than the CPU reference on a large image.  The
// tool says so on stderr for images above one megapixel; --sliced TWxTH selects the parallel container (e.g. 480x1),
// --sliced auto takes one-row slices of the width llcomp_mi_suggest_tile_w returns for ONE image per call (a lone 4K
// frame: 80 pixels, 0.76 ms instead of 2.2 ms at 480), --legacy states the default explicitly and silences the note.
// --devices 0,1,2,...: the image's tile rows are dealt over these GPUs inside this process (llcomp::Options::devices; sliced
// containers only, byte-identical to the one-GPU container).
#include <cstdio>
#include <exception>
#include <string>
#include <vector>

#include "../include/llcomp_mi.hpp"
#include "cli_common.hpp"
#include "image_io.hpp"

namespace {

bool parse_flags(int argc, char** argv, llcomp::Options& opt, bool& explicit_legacy, bool& auto_width) {
    for (int i = 2; i < argc; ++i) {
        const std::string flag = argv[i];
        if (flag == "--interleaved") {
            opt.planar = false;
        } else if (flag == "--legacy") {
            explicit_legacy = true;
        } else if (flag == "--small-model") {  // the bitstream of a reference built with LargeModel = false (llcomp.hpp:21)
            opt.small_model = true;
        } else if (flag == "--devices" && i + 1 < argc) {
            if (!cli::parse_device_list(argv[++i], opt.devices)) return false;
        } else if (flag == "--sliced" && i + 1 < argc) {
            unsigned tw = 0, th = 0;
            if (std::string(argv[i + 1]) == "auto") {  // width chosen once the image size is known (compress_file)
                ++i;
                auto_width = true;
            } else if (std::sscanf(argv[++i], "%ux%u", &tw, &th) != 2) {
                return false;
            }
            opt.sliced = true;
            opt.tile_w = tw;
            opt.tile_h = th;
        }  // anything else is ignored, as the reference ignores everything after its first argument
    }
    return true;
}

int compress_file(const std::string& image_path, llcomp::Options opt, bool explicit_legacy, bool auto_width) {
    std::vector<uint8_t> pixels;
    int w = 0, h = 0, c = 0;
    if (const std::string reason = image_io::load_image(image_path, pixels, w, h, c); !reason.empty()) {
        std::fprintf(stderr, "Error loading image: %s\n", reason.c_str());
        return cli::kFailed;
    }
    if (auto_width) {
        opt.tile_w = llcomp_mi_suggest_tile_w(1, uint32_t(w), uint32_t(h), uint32_t(c), opt.planar ? 1u : 0u);
        opt.tile_h = 1;
    }
    if (!opt.sliced && !explicit_legacy && size_t(w) * size_t(h) > (1u << 20))
        std::fprintf(stderr,
                     "note: writing the reference's single-stream format (%dx%d): one serial chain = one GPU lane, slower than the CPU "
                     "reference; use --sliced auto (or e.g. 480x1, 64x64) for the parallel container, --legacy to silence this note\n", w, h);
    std::vector<uint8_t> stream;
    try {
        stream = llcomp::compressImage(pixels, w, h, c, opt);
    } catch (const std::exception& e) {  // no counterpart in the reference (its encoder cannot fail): no GPU, ...
        std::fprintf(stderr, "Error compressing image: %s\n", e.what());
        return cli::kFailed;
    }
    const std::string target = image_path + llcomp::ext;
    if (!cli::spill(target, stream)) {
        std::fprintf(stderr, "Error opening output file: %s\n", target.c_str());
        return cli::kFailed;
    }
    return cli::kDone;
}

}  // namespace

int main(int argc, char** argv) {
    llcomp::Options opt;
    bool explicit_legacy = false, auto_width = false;
    if (argc < 2 || !parse_flags(argc, argv, opt, explicit_legacy, auto_width)) {
        std::fprintf(stderr, "Usage: %s <image_path> [--sliced TWxTH|auto] [--interleaved] [--legacy] [--small-model] [--devices a,b,...]\n", argc ? argv[0] : "llcompc");
        return cli::kFailed;
    }
    return compress_file(argv[1], opt, explicit_legacy, auto_width);
}
