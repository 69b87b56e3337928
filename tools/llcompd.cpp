// llcompd <file.llcomp> [--small-model] [--devices a,b,...]
//
// Decompressor front end on libllcomp_mi.so with the observable behaviour of the reference tool
// (/root/reference/llcompd.cpp:11-41): one positional argument, the picture is written as "<file>.png", exit status 1
// when the input cannot be read or the stream is rejected with a std::exception (message printed), 2 for any other
// exception, and -- faithfully -- still 0 when only writing the PNG failed (llcompd.cpp:29-31).  stb_image_write is not
// available; tools/image_io.hpp writes the PNG (adaptive row filters, own deflate).  Reads both wire formats.  --small-model: the file is a
// reference-format stream written by a reference built with `LargeModel = false` (llcomp.hpp:21) -- that header does not
// record the variant (a sliced container does).  --devices 0,1,...: a sliced container is decoded over these GPUs inside this
// process (llcomp_mi_decode_devices).
#include <cstdio>
#include <exception>
#include <string>
#include <vector>

#include "../include/llcomp_mi.hpp"
#include "cli_common.hpp"
#include "image_io.hpp"

namespace {

int expand_file(const std::string& stream_path, bool legacy_small_model, const std::vector<int>& devices) {
    std::vector<uint8_t> stream;
    if (!cli::slurp(stream_path, stream)) {
        std::fprintf(stderr, "Error opening input file: %s\n", stream_path.c_str());
        return cli::kFailed;
    }
    llcomp::RawImage picture;
    try {
        picture = devices.empty() ? llcomp::decompressImage(stream, -1, legacy_small_model) : llcomp::decompressImage(stream, devices, legacy_small_model);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error decompressing image: %s\n", e.what());
        return cli::kFailed;
    } catch (...) {
        std::fprintf(stderr, "Unknown error occurred\n");
        return cli::kUnknown;
    }
    const std::string target = stream_path + ".png";
    const int row_bytes = int(picture.width) * picture.channels;
    if (!image_io::write_png(target, int(picture.width), int(picture.height), picture.channels, picture.pixels.data(), row_bytes))
        std::fprintf(stderr, "Error writing output file: %s\n", target.c_str());  // not an error status in the reference
    return cli::kDone;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "Usage: %s <image_path> [--small-model] [--devices a,b,...]\n", argc ? argv[0] : "llcompd");
        return cli::kFailed;
    }
    bool small = false;
    std::vector<int> devices;
    for (int i = 2; i < argc; ++i) {
        const std::string flag = argv[i];
        if (flag == "--small-model") small = true;
        else if (flag == "--devices" && i + 1 < argc && !cli::parse_device_list(argv[++i], devices)) {
            std::fprintf(stderr, "Usage: %s <image_path> [--small-model] [--devices a,b,...]\n", argv[0]);
            return cli::kFailed;
        }
    }
    return expand_file(argv[1], small, devices);
}
