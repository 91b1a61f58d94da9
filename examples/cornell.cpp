// cornell.cpp — the reference binary (src/main.rs:5-87) written against tinyrt.hpp: same scene,
// same camera, same Renderer::new(300, 8, 20, true, Some(0.001)) arguments, on an MI355X.
//
//   g++ -std=c++17 -Iinclude examples/cornell.cpp -Ltiny-raytracer_amd -ltinyrt -Wl,-rpath,$PWD/tiny-raytracer_amd -o build/cornell
//   ./build/cornell [width height spp] -> output.png
#include <cstdio>
#include <cstdlib>

#include "tinyrt.hpp"

using namespace tinyrt;

static void build_materials(World& world) {                       // src/main.rs:80-87
    world.add_material("red", Lambertian(Vec3(0.65f, 0.05f, 0.05f)));
    world.add_material("white", Lambertian(Vec3(0.73f, 0.73f, 0.73f)));
    world.add_material("green", Lambertian(Vec3(0.12f, 0.45f, 0.15f)));
    world.add_material("light", Light(Vec3::new_diagonal(15.0f)));
}

static void build_objects(World& world) {                         // src/main.rs:29-78
    auto mat = [&](const char* n) { return world.get_material(n).value(); };
    world.add_geometry(Quad{Vec3(100, 0, 0), Vec3(0, 100, 0), Vec3(0, 0, 100), mat("green")});
    world.add_geometry(Quad{Vec3(0, 0, 0), Vec3(0, 100, 0), Vec3(0, 0, 100), mat("red")});
    world.add_geometry(Quad{Vec3(65, 100, 60), Vec3(-30, 0, 0), Vec3(0, 0, -20), mat("light")});
    world.add_geometry(Quad{Vec3(0, 0, 0), Vec3(100, 0, 0), Vec3(0, 0, 100), mat("white")});
    world.add_geometry(Quad{Vec3(100, 100, 100), Vec3(-100, 0, 0), Vec3(0, 0, -100), mat("white")});
    world.add_geometry(Quad{Vec3(0, 0, 100), Vec3(100, 0, 0), Vec3(0, 100, 0), mat("white")});
    world.add_box(Vec3(25, 0, 50), Vec3(55, 60, 80), mat("white"));
    world.add_box(Vec3(45, 0, 10), Vec3(75, 30, 40), mat("white"));
}

int main(int argc, char** argv) {
    const uint32_t w = argc > 2 ? (uint32_t)std::atoi(argv[1]) : 300, h = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 300;
    const uint32_t spp = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 300;
    const int ngpu = argc > 4 ? std::atoi(argv[4]) : 1;                 // 0 = every GPU of the node (trt_render_multi)
    try {
        World world;
        build_materials(world);
        build_objects(world);
        Camera camera(140.0f, 0.6f, Vec3(50, 50, -140), Vec3(50, 50, 0), Vec3(0, 1, 0), 40.0f, w, h);
        Renderer instance(spp, 8, 20, true, Vec3::new_diagonal(0.001f));
        trt_stats st{};
        Image image = ngpu == 1 ? instance.render(camera, world, &st)
                                : instance.render_multi(camera, world, ngpu > 0 ? std::vector<int>((size_t)ngpu, 0) : std::vector<int>{}, &st);
        image.save("output.png");                                      // src/main.rs:20
        std::printf("%ux%u, %u spp: %llu rays in %.2f ms (%.1f Mray/s) -> output.png\n", w, h, spp, (unsigned long long)st.rays,
                    st.kernel_ms, st.rays / st.kernel_ms / 1e3);
    } catch (const Error& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
