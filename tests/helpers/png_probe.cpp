// Test helper: loads an image with the CLIs' reader (tools/image_io.hpp) and writes "w h c\n" + raw pixels to stdout;
// with a second argument it writes the image back as PNG with the CLIs' writer instead.
#include <cstdio>
#include <string>
#include <vector>

#include "../../tools/image_io.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::vector<uint8_t> px;
    int w = 0, h = 0, c = 0;
    const std::string why = image_io::load_image(argv[1], px, w, h, c);
    if (!why.empty()) {
        std::fprintf(stderr, "%s\n", why.c_str());
        return 1;
    }
    if (argc > 2) return image_io::write_png(argv[2], w, h, c, px.data(), w * c) ? 0 : 3;  // re-encode with the CLIs' writer
    std::printf("%d %d %d\n", w, h, c);
    std::fwrite(px.data(), 1, px.size(), stdout);
    return 0;
}
