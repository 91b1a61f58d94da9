// tinyrt.hpp — C++17 host-side mirror of the reference crate's World / Camera / Renderer surface,
// header-only, over the C ABI in tinyrt.h.
//
// The reference is compiled code (Rust); no Rust toolchain exists in the build image, so the host
// side above the C ABI is C++ and keeps the crate's names, argument order and meaning (paths
// relative to raytracer/src):
//
//   World::add_material / add_geometry / get_material / get_bvh     hittable/world.rs:16-45
//   Sphere(center, radius, material), Quad(corner, u, v, material)  hittable/sphere.rs:16-26, quad.rs:20-29
//   Lambertian / Metal / Dielectric / Light                          material/*.rs
//   Camera(focus_distance, defocus_angle, position, look_at, up, vertical_fov, width, height)  camera.rs:17-26
//   Renderer(samples_per_pixel, num_sampler_threads, max_bounces, progressbar, background)     renderer/renderer.rs:21-35
//   Renderer::render(camera, world) -> Image                         renderer/renderer.rs:37-79
//   Image::get_pixel / save                                          utils/image.rs:46-48,66-69
//
// Error behaviour: where the reference panics (duplicate material name world.rs:29-31, failed
// joins renderer.rs:75-77) this throws tinyrt::Error carrying the C ABI's status and message.
// A Rust `-sys` binding of the same ABI is sketched in INTEGRATION.md.
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "tinyrt.h"

namespace tinyrt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc) {
    if (rc != TRT_OK) throw Error(rc, std::string("tinyrt error ") + std::to_string(rc) + ": " + trt_last_error());
}

struct Vec3 : trt_vec3 {
    Vec3() : trt_vec3{0.0f, 0.0f, 0.0f} {}
    Vec3(float x_, float y_, float z_) : trt_vec3{x_, y_, z_} {}
    static Vec3 new_diagonal(float v) { return Vec3(v, v, v); }        // math/vec3.rs:23-25
    static Vec3 zero() { return Vec3(); }                              // math/vec3.rs:27-29
};

// ---- materials: values handed to World::add_material (material/*.rs) ----
struct Material { trt_material pod; };
inline Material Lambertian(Vec3 albedo) { return Material{{TRT_LAMBERTIAN, albedo, 0.0f}}; }
inline Material Metal(Vec3 albedo, float fuzz) { return Material{{TRT_METAL, albedo, fuzz}}; }
inline Material Dielectric(Vec3 albedo, float refraction_index) { return Material{{TRT_DIELECTRIC, albedo, refraction_index}}; }
inline Material Light(Vec3 color) { return Material{{TRT_LIGHT, color, 0.0f}}; }
using MaterialHandle = uint32_t;     // stands in for Arc<Box<dyn Material>>

// ---- geometry values handed to World::add_geometry ----
struct Sphere { Vec3 center; float radius; MaterialHandle material; };
struct Quad { Vec3 corner, u, v; MaterialHandle material; };

class World {
public:
    World() { check(trt_world_create(&w_)); }
    ~World() { if (scene_) trt_scene_destroy(scene_); trt_world_destroy(w_); }
    World(const World&) = delete;
    World& operator=(const World&) = delete;

    void add_material(const std::string& name, const Material& m) { check(trt_world_add_material(w_, name.c_str(), &m.pod)); }
    std::optional<MaterialHandle> get_material(const std::string& name) const {
        uint32_t idx = 0;
        int rc = trt_world_get_material(w_, name.c_str(), &idx);
        if (rc == TRT_ERR_NOT_FOUND) return std::nullopt;              // world.rs:35-41 returns Option
        check(rc);
        return idx;
    }
    void add_geometry(const Sphere& s) { invalidate(); check(trt_world_add_sphere(w_, s.center, s.radius, s.material)); }
    // n spheres in array order: the loop of add_geometry(Sphere) in one call (center_radius: 4n floats x, y, z, r)
    void add_spheres(uint32_t n, const float* center_radius, const uint32_t* material) { invalidate(); check(trt_world_add_spheres(w_, n, center_radius, material)); }
    void add_geometry(const Quad& q) { invalidate(); check(trt_world_add_quad(w_, q.corner, q.u, q.v, q.material)); }
    // new_box(a, b, material) of the reference binary (src/main.rs:89-125): six quads, same push order
    void add_box(Vec3 a, Vec3 b, MaterialHandle m) {
        Vec3 mn(a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z);
        Vec3 mx(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z);
        Vec3 dx(mx.x - mn.x, 0, 0), dy(0, mx.y - mn.y, 0), dz(0, 0, mx.z - mn.z);
        auto neg = [](Vec3 v) { return Vec3(-v.x, -v.y, -v.z); };
        add_geometry(Quad{Vec3(mn.x, mn.y, mx.z), dx, dy, m});
        add_geometry(Quad{Vec3(mx.x, mn.y, mx.z), neg(dz), dy, m});
        add_geometry(Quad{Vec3(mx.x, mn.y, mn.z), neg(dx), dy, m});
        add_geometry(Quad{Vec3(mn.x, mn.y, mn.z), dz, dy, m});
        add_geometry(Quad{Vec3(mn.x, mx.y, mx.z), dx, neg(dz), m});
        add_geometry(Quad{Vec3(mn.x, mn.y, mn.z), dx, dz, m});
    }
    // World::get_bvh (world.rs:43-45): the reference-order BVH, packed for the GPU; cached until the world changes
    trt_scene* get_bvh() {
        if (!scene_) check(trt_scene_create_ex(w_, has_options_ ? &options_ : nullptr, &scene_));
        return scene_;
    }
    // How the scene is compiled (trt_scene_options: placement only, every value renders the same frames).  Starts from the library
    // defaults; takes effect at the next get_bvh().
    trt_scene_options& scene_options() {
        if (!has_options_) { trt_scene_options_default(&options_); has_options_ = true; }
        invalidate();
        return options_;
    }
    int num_geometries() const { return trt_world_num_geometries(w_); }
    // Frees the device scratch (render workspaces, frame buffers) the compiled scene caches between renders (trt_scene_trim).
    void trim() { if (scene_) check(trt_scene_trim(scene_)); }

private:
    void invalidate() { if (scene_) { trt_scene_destroy(scene_); scene_ = nullptr; } }
    trt_world* w_ = nullptr;
    trt_scene* scene_ = nullptr;
    trt_scene_options options_{};
    bool has_options_ = false;
};

class Camera {
public:
    Camera(float focus_distance, float defocus_angle, Vec3 position, Vec3 look_at, Vec3 up, float vertical_fov, uint32_t width,
           uint32_t height) {
        check(trt_camera_init(&pod, focus_distance, defocus_angle, position, look_at, up, vertical_fov, width, height));
    }
    std::pair<uint32_t, uint32_t> get_image_size() const { return {pod.width, pod.height}; }   // camera.rs:68-70
    trt_camera pod;
};

// Minimal PNG writer (8-bit RGB, zlib "stored" blocks: valid for every decoder, no compression) so that
// Image::save("x.png") yields what the reference's `image` crate call yields: a PNG of the quantised frame.
namespace detail {
inline uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc) {
    crc = ~crc;
    for (size_t i = 0; i < n; i++) {
        crc ^= p[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}
inline void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
inline void png_chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& data) {
    put_be32(out, (uint32_t)data.size());
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc32(out.data() + at, out.size() - at, 0u));
}
inline std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h) {
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w); put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});                       // 8 bits, colour type 2 (RGB), deflate, no filter, no interlace
    png_chunk(out, "IHDR", ihdr);
    std::vector<uint8_t> raw;                                        // scanlines, each with filter byte 0
    raw.reserve((size_t)h * ((size_t)w * 3 + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};                           // zlib header, then stored deflate blocks of <= 65535 bytes
    uint32_t a = 1, b = 0;                                           // Adler-32 of the raw data
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    size_t pos = 0;
    do {
        const size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    } while (pos < raw.size());
    put_be32(z, b << 16 | a);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    return out;
}
}  // namespace detail

// utils/image.rs: Image with gamma 2.2 as the Imager builds it (imager.rs:37-41); holds the linear sums
class Image {
public:
    Image(uint32_t w, uint32_t h, float gamma = 2.2f) : width_(w), height_(h), gamma_(gamma), data_((size_t)w * h * 3, 0.0f) {}
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    float* linear() { return data_.data(); }
    const float* linear() const { return data_.data(); }
    std::vector<uint8_t> to_rgb8() const {
        std::vector<uint8_t> rgb(data_.size());
        check(trt_tonemap_u8(data_.data(), width_ * height_, gamma_, rgb.data()));
        return rgb;
    }
    // Image::save (image.rs:66-69; the reference writes PNG through the `image` crate): PNG for "*.png", binary PPM otherwise
    void save(const std::string& filename) const {
        auto rgb = to_rgb8();
        FILE* f = std::fopen(filename.c_str(), "wb");
        if (!f) throw Error(TRT_ERR_INVALID_ARG, "cannot open " + filename);
        const bool png = filename.size() >= 4 && filename.compare(filename.size() - 4, 4, ".png") == 0;
        if (png) {
            auto bytes = detail::encode_png_rgb8(rgb.data(), width_, height_);
            std::fwrite(bytes.data(), 1, bytes.size(), f);
        } else {
            std::fprintf(f, "P6\n%u %u\n255\n", width_, height_);
            std::fwrite(rgb.data(), 1, rgb.size(), f);
        }
        std::fclose(f);
    }

private:
    uint32_t width_, height_;
    float gamma_;
    std::vector<float> data_;
};

class Renderer {
public:
    Renderer(uint32_t samples_per_pixel, uint32_t num_sampler_threads, uint32_t max_bounces, bool progressbar,
             std::optional<Vec3> background_color, uint32_t seed = 1, uint32_t backend = TRT_BACKEND_AUTO) {
        (void)num_sampler_threads;     // the GPU needs no sampler-task count; kept for signature parity
        (void)progressbar;
        params_ = trt_render_params{};
        params_.samples_per_pixel = samples_per_pixel;
        params_.max_bounces = max_bounces;
        params_.background = background_color.value_or(Vec3::zero());  // renderer.rs:33
        params_.seed = seed;
        params_.backend = backend;
    }
    // Renderer::render (renderer.rs:37-79); synchronous (the reference returns a JoinHandle<Image> to await)
    Image render(const Camera& camera, World& world, trt_stats* stats = nullptr) const {
        Image img(camera.pod.width, camera.pod.height);
        check(trt_render(world.get_bvh(), &camera.pod, &params_, img.linear(), stats));
        return img;
    }
    // The same call over several GPUs of the node (trt_render_multi): `devices` empty = every visible device.
    Image render_multi(const Camera& camera, World& world, const std::vector<int>& devices = {}, trt_stats* stats = nullptr) const {
        Image img(camera.pod.width, camera.pod.height);
        check(trt_render_multi(world.get_bvh(), &camera.pod, &params_, devices.empty() ? nullptr : devices.data(),
                               (uint32_t)devices.size(), img.linear(), stats));
        return img;
    }
    trt_render_params& params() { return params_; }
    // Scheduling knobs of this renderer (trt_tuning: every value renders the same frame).  Starts from the library defaults.
    trt_tuning& tuning() {
        if (!params_.tuning) { trt_tuning_default(&tuning_); params_.tuning = &tuning_; }
        return tuning_;
    }
    Renderer(const Renderer& o) : params_(o.params_), tuning_(o.tuning_) { if (o.params_.tuning) params_.tuning = &tuning_; }
    Renderer& operator=(const Renderer& o) {
        params_ = o.params_; tuning_ = o.tuning_;
        if (o.params_.tuning) params_.tuning = &tuning_;
        return *this;
    }

private:
    trt_render_params params_;
    trt_tuning tuning_{};
};

}  // namespace tinyrt
