// llcompd -- decompressor CLI with the reference's behaviour (/root/reference/llcompd.cpp:11-41): one positional
// path, writes <path>.png, exit 1 on std::exception, 2 on anything else, and -- like the reference -- still 0 when
// only the PNG write fails (llcompd.cpp:29-31).  PNG writing is image_io.hpp (stb_image_write is not available).
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "../include/llcomp_mi.hpp"
#include "image_io.hpp"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <image_path>" << std::endl;
        return 1;
    }
    const char* filename = argv[1];
    std::ifstream inFile(filename, std::ios::binary);
    if (!inFile) {
        std::cerr << "Error opening input file: " << filename << std::endl;
        return 1;
    }
    std::vector<uint8_t> compressed((std::istreambuf_iterator<char>(inFile)), std::istreambuf_iterator<char>());
    inFile.close();
    try {
        auto [pixels, width, height, channels] = llcomp::decompressImage(compressed);
        std::string outputFile = std::string(filename) + ".png";
        if (!image_io::write_png(outputFile, int(width), int(height), channels, pixels.data(), int(width) * channels)) {
            std::cerr << "Error writing output file: " << outputFile << std::endl;
        }
    } catch (const std::exception& e) {
        std::cerr << "Error decompressing image: " << e.what() << std::endl;
        return 1;
    } catch (...) {
        std::cerr << "Unknown error occurred" << std::endl;
        return 2;
    }
    return 0;
}
