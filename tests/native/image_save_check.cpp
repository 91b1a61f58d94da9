// Host-only check of tinyrt::Image (include/tinyrt.hpp): fill the linear sums with a known ramp, save PNG and PPM.
//   image_save_check <width> <height> <out.png> <out.ppm>
#include <cstdlib>

#include "tinyrt.hpp"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const uint32_t w = (uint32_t)std::atoi(argv[1]), h = (uint32_t)std::atoi(argv[2]);
    tinyrt::Image img(w, h);
    float* p = img.linear();
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            float* c = p + ((size_t)y * w + x) * 3;
            c[0] = (float)x / (float)w;                    // ramp
            c[1] = (float)y / (float)h * 1.5f;             // beyond 1: clamps to 254
            c[2] = ((x + y) % 7 == 0) ? -0.25f : 0.18f;    // negative -> NaN after powf -> 0
        }
    img.save(argv[3]);
    img.save(argv[4]);
    return 0;
}
