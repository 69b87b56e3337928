// llcompc -- compressor CLI with the reference's behaviour (/root/reference/llcompc.cpp:14-43): one positional image
// path, writes <path>.llcomp, exit 1 on usage / load / open errors.  The image decoder is image_io.hpp (binary
// PGM/PPM/PAM) because stb_image is neither vendored nor installed.  Default output is the reference's own format;
// optional flags after the path select the parallel container:  --sliced TWxTH  [--interleaved]
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../include/llcomp_mi.hpp"
#include "image_io.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <image_path> [--sliced TWxTH] [--interleaved]" << std::endl;
        return 1;
    }
    const char* filename = argv[1];
    llcomp::Options opt;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--sliced" && i + 1 < argc) {
            opt.sliced = true;
            unsigned tw = 0, th = 0;
            if (sscanf(argv[++i], "%ux%u", &tw, &th) != 2) {
                std::cerr << "Usage: --sliced TWxTH" << std::endl;
                return 1;
            }
            opt.tile_w = tw;
            opt.tile_h = th;
        } else if (a == "--interleaved") {
            opt.planar = false;
        }
    }
    int width, height, channels;
    std::vector<uint8_t> rgb;
    const std::string why = image_io::load_pnm(filename, rgb, width, height, channels);
    if (!why.empty()) {
        std::cerr << "Error loading image: " << why << std::endl;
        return 1;
    }
    std::vector<uint8_t> compressed;
    try {
        compressed = llcomp::compressImage(rgb, width, height, channels, opt);
    } catch (const std::exception& e) {  // the reference has no failure mode here; the GPU path can (no device, ...)
        std::cerr << "Error compressing image: " << e.what() << std::endl;
        return 1;
    }
    std::string outputFile = std::string(filename) + llcomp::ext;
    std::ofstream outFile(outputFile, std::ios::binary);
    if (!outFile) {
        std::cerr << "Error opening output file: " << outputFile << std::endl;
        return 1;
    }
    outFile.write(reinterpret_cast<const char*>(compressed.data()), std::streamsize(compressed.size()));
    outFile.close();
    return 0;
}
