// Host-side check of llcomp_amd/csrc/geometry.hpp: which kernel family a geometry selects (printed as one line per case:
// "name flags lane_shift lpw n_slices slice_samples").   g++ -std=c++17 -I llcomp_amd/csrc tests/helpers/geometry_check.cpp
#include <cstdio>

#include "geometry.hpp"

using namespace llcomp_mi;

static void show(const char* name, uint32_t frames, uint32_t w, uint32_t h, uint32_t c, uint32_t tw, uint32_t th, uint32_t planar, Tuning t = Tuning{}) {
    Geometry g{};
    if (!make_geometry(g, frames, w, h, c, tw, th, planar, t)) { std::printf("%s rejected\n", name); return; }
    std::printf("%s %u %u %u %u %u\n", name, g.flags, g.lane_shift, g.lpw, g.n_slices, g.slice_samples);
}

int main() {
    show("rows_4k_480x1", 32, 3840, 2160, 3, 480, 1, 1);
    show("tiles_4k_64x64", 16, 3840, 2160, 3, 64, 64, 1);
    show("tiles_4k_32x32_interleaved", 16, 3840, 2160, 3, 32, 32, 0);
    show("tiles_4k_128x128", 16, 3840, 2160, 3, 128, 128, 1);
    show("tiles_4k_64x64_interleaved", 16, 3840, 2160, 3, 64, 64, 0);
    show("tiles_4k_256x256", 16, 3840, 2160, 3, 256, 256, 1);
    show("tiles_4k_128x129", 16, 3840, 2160, 3, 128, 129, 1);
    show("one_frame_256x256", 1, 3840, 2160, 3, 256, 256, 1);
    show("legacy_bulk_512", 512, 256, 256, 3, 256, 256, 0);
    show("lone_legacy", 1, 1920, 1080, 3, 0, 0, 0);
    Tuning nosnap; nosnap.nosnap = true;
    show("tiles_4k_64x64_nosnap", 16, 3840, 2160, 3, 64, 64, 1, nosnap);
    Tuning norows; norows.norows = true;
    show("rows_forced_general", 4, 960, 64, 3, 480, 1, 1, norows);
    show("nine_channels_one_row", 2, 400, 20, 9, 100, 1, 0);
    Tuning nocache; nocache.nocache = true;
    show("tiles_4k_64x64_nocache", 16, 3840, 2160, 3, 64, 64, 1, nocache);
    show("nine_channels_tiles", 16, 1920, 1080, 9, 32, 8, 0);
    return 0;
}
